// mmap-backed GGUF v2/v3 reader: header, metadata key/values, tensor directory.
// What the reference gets from candle_core::quantized::gguf_file::Content::read
// (crane-core/src/models/qwen3/model.rs:138-147; hunyuan_dense/modeling.rs:14-95 `Gguf` helper).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace cmgguf {

enum GgmlType { F32 = 0, F16 = 1, Q4_0 = 2, Q5_0 = 6, Q8_0 = 8, Q3_K = 11, Q4_K = 12, Q6_K = 14, BF16 = 30 };

inline bool type_layout(uint32_t t, size_t& block_elems, size_t& block_bytes) {
    switch (t) {
        case F32: block_elems = 1; block_bytes = 4; return true;
        case F16: case BF16: block_elems = 1; block_bytes = 2; return true;
        case Q8_0: block_elems = 32; block_bytes = 34; return true;
        case Q4_0: block_elems = 32; block_bytes = 18; return true;
        case Q5_0: block_elems = 32; block_bytes = 22; return true;
        case Q3_K: block_elems = 256; block_bytes = 110; return true;
        case Q4_K: block_elems = 256; block_bytes = 144; return true;
        case Q6_K: block_elems = 256; block_bytes = 210; return true;
        default: return false;
    }
}

struct Value {
    uint32_t type = 0;                // GGUF value type id
    uint64_t u = 0;                   // unsigned / bool
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    std::vector<Value> arr;
    bool is_int() const { return type <= 5 || type == 7 || type == 10 || type == 11; }
    double as_f64() const { return (type == 6 || type == 12) ? f : (is_signed() ? (double)i : (double)u); }
    bool is_signed() const { return type == 1 || type == 3 || type == 5 || type == 11; }
    uint64_t as_u64() const { return is_signed() ? (uint64_t)i : u; }
};

struct TensorInfo {
    std::string name;
    std::vector<uint64_t> shape;      // outermost first (GGUF stores innermost first)
    uint32_t type = 0;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    // element count; UINT64_MAX when the product overflows
    uint64_t numel() const {
        uint64_t n = 1;
        for (uint64_t d : shape) {
            if (d != 0 && n > UINT64_MAX / d) return UINT64_MAX;
            n *= d;
        }
        return n;
    }
};

class File {
public:
    explicit File(const std::string& path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { ::close(fd_); throw std::runtime_error("cannot stat " + path); }
        size_ = (size_t)st.st_size;
        base_ = (const uint8_t*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (base_ == MAP_FAILED) { ::close(fd_); throw std::runtime_error("cannot mmap " + path); }
        try { parse(path); } catch (...) { munmap((void*)base_, size_); ::close(fd_); throw; }
    }
    ~File() { if (base_) munmap((void*)base_, size_); if (fd_ >= 0) ::close(fd_); }
    File(const File&) = delete;
    File& operator=(const File&) = delete;

    const Value* meta(const std::string& k) const { auto it = md_.find(k); return it == md_.end() ? nullptr : &it->second; }
    bool has(const std::string& n) const { return tensors_.count(n) != 0; }
    const TensorInfo& tensor(const std::string& n) const {
        auto it = tensors_.find(n);
        if (it == tensors_.end()) throw std::runtime_error("GGUF missing tensor " + n);
        return it->second;
    }
    const std::map<std::string, TensorInfo>& tensors() const { return tensors_; }

private:
    int fd_ = -1;
    const uint8_t* base_ = nullptr;
    size_t size_ = 0, pos_ = 0;
    std::map<std::string, Value> md_;
    std::map<std::string, TensorInfo> tensors_;

    template <typename T> T rd() {
        if (pos_ + sizeof(T) > size_) throw std::runtime_error("truncated GGUF");
        T v; memcpy(&v, base_ + pos_, sizeof(T)); pos_ += sizeof(T); return v;
    }
    std::string rstr() {
        const uint64_t n = rd<uint64_t>();
        if (n > size_ - pos_) throw std::runtime_error("truncated GGUF string");
        std::string s((const char*)base_ + pos_, (size_t)n); pos_ += (size_t)n; return s;
    }
    Value rval(uint32_t t, int depth = 0) {
        Value v; v.type = t;
        switch (t) {
            case 0: v.u = rd<uint8_t>(); break;
            case 1: v.i = rd<int8_t>(); break;
            case 2: v.u = rd<uint16_t>(); break;
            case 3: v.i = rd<int16_t>(); break;
            case 4: v.u = rd<uint32_t>(); break;
            case 5: v.i = rd<int32_t>(); break;
            case 6: v.f = rd<float>(); break;
            case 7: v.u = rd<uint8_t>(); break;
            case 8: v.s = rstr(); break;
            case 9: {
                if (depth > 2) throw std::runtime_error("GGUF array nesting too deep");
                const uint32_t et = rd<uint32_t>();
                const uint64_t n = rd<uint64_t>();
                // every element takes at least one byte of the file; 2^26 bounds the memory a hostile length can claim
                // (real tokenizer arrays hold ~10^5 .. 10^6 entries)
                if (n > size_ || n > (1ull << 26)) throw std::runtime_error("bad GGUF array length");
                v.arr.reserve((size_t)std::min<uint64_t>(n, 1 << 20));
                for (uint64_t i = 0; i < n; ++i) v.arr.push_back(rval(et, depth + 1));
                break;
            }
            case 10: v.u = rd<uint64_t>(); break;
            case 11: v.i = rd<int64_t>(); break;
            case 12: v.f = rd<double>(); break;
            default: throw std::runtime_error("unknown GGUF value type " + std::to_string(t));
        }
        return v;
    }
    void parse(const std::string& path) {
        if (rd<uint32_t>() != 0x46554747u) throw std::runtime_error(path + " is not a GGUF file");
        const uint32_t ver = rd<uint32_t>();
        if (ver != 2 && ver != 3) throw std::runtime_error("unsupported GGUF version " + std::to_string(ver));
        const uint64_t nt = rd<uint64_t>(), nkv = rd<uint64_t>();
        for (uint64_t i = 0; i < nkv; ++i) {
            std::string k = rstr();
            const uint32_t t = rd<uint32_t>();
            md_[k] = rval(t);
        }
        std::vector<std::pair<TensorInfo, uint64_t>> infos;
        for (uint64_t i = 0; i < nt; ++i) {
            TensorInfo ti;
            ti.name = rstr();
            const uint32_t nd = rd<uint32_t>();
            if (nd > 8) throw std::runtime_error("bad GGUF tensor rank");
            std::vector<uint64_t> dims(nd);
            for (uint32_t d = 0; d < nd; ++d) dims[d] = rd<uint64_t>();
            ti.shape.assign(dims.rbegin(), dims.rend());
            ti.type = rd<uint32_t>();
            const uint64_t off = rd<uint64_t>();
            infos.emplace_back(std::move(ti), off);
        }
        uint64_t align = 32;
        if (const Value* a = meta("general.alignment")) align = a->as_u64() ? a->as_u64() : 32;
        if (align > (1u << 20) || (align & (align - 1))) throw std::runtime_error("bad general.alignment");
        const size_t data0 = (pos_ + align - 1) / align * align;
        if (data0 > size_) throw std::runtime_error("truncated GGUF (no tensor data)");
        for (auto& p : infos) {
            TensorInfo& ti = p.first;
            size_t be = 0, bb = 0;
            if (type_layout(ti.type, be, bb)) {
                const uint64_t n = ti.numel();
                if (n == UINT64_MAX) throw std::runtime_error("tensor " + ti.name + ": shape overflows");
                if (n % be) throw std::runtime_error("tensor " + ti.name + ": element count is not a multiple of the block size");
                if (n / be > UINT64_MAX / bb) throw std::runtime_error("tensor " + ti.name + ": size overflows");
                ti.nbytes = (size_t)(n / be * bb);
                const size_t room = size_ - data0;                      // overflow-safe: offset and size against what is left
                if (p.second > room || ti.nbytes > room - p.second) throw std::runtime_error("tensor " + ti.name + " runs past the end of the file");
            } else if (p.second > size_ - data0) {
                throw std::runtime_error("tensor " + ti.name + " starts past the end of the file");
            }
            ti.data = base_ + data0 + p.second;
            tensors_[ti.name] = ti;
        }
    }
};

}  // namespace cmgguf
