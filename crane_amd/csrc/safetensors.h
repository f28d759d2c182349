// mmap-backed safetensors reader with sharded-index discovery.
// Mirrors what the reference gets from VarBuilder::from_mmaped_safetensors +
// get_safetensors_files (crane-core/src/models/qwen3/model.rs:91-92,
// crane-core/src/utils/utils.rs:16-57): "model.safetensors" or every shard named
// in "model.safetensors.index.json", else every *.safetensors in the directory.
#pragma once
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <cstdint>
#include <string>
#include <vector>

#include "json_min.h"

namespace cmst {

struct TensorView {
    std::string dtype;            // "BF16", "F16", "F32"
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    // element count with overflow / negative-dimension checks (-1: invalid)
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) {
            if (d < 0) return -1;
            if (d != 0 && n > INT64_MAX / d) return -1;
            n *= d;
        }
        return n;
    }
    static size_t elem_size(const std::string& dt) {
        if (dt == "BF16" || dt == "F16" || dt == "I16" || dt == "U16") return 2;
        if (dt == "F32" || dt == "I32" || dt == "U32") return 4;
        if (dt == "F64" || dt == "I64" || dt == "U64") return 8;
        if (dt == "I8" || dt == "U8" || dt == "BOOL" || dt == "F8_E4M3" || dt == "F8_E5M2") return 1;
        return 0;                 // unknown dtype: the loader rejects the tensor when it is asked for
    }
};

class MappedFile {
  public:
    explicit MappedFile(const std::string& path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { ::close(fd_); throw std::runtime_error("cannot stat " + path); }
        size_ = (size_t)st.st_size;
        base_ = (uint8_t*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (base_ == MAP_FAILED) { ::close(fd_); throw std::runtime_error("cannot mmap " + path); }
    }
    ~MappedFile() { if (base_ && base_ != MAP_FAILED) munmap(base_, size_); if (fd_ >= 0) ::close(fd_); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    const uint8_t* base() const { return base_; }
    size_t size() const { return size_; }

  private:
    int fd_ = -1;
    uint8_t* base_ = nullptr;
    size_t size_ = 0;
};

inline std::string read_text(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot read " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

inline bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

class Checkpoint {
  public:
    explicit Checkpoint(const std::string& dir) {
        std::vector<std::string> files;
        const std::string single = dir + "/model.safetensors";
        const std::string index = dir + "/model.safetensors.index.json";
        if (file_exists(index)) {
            auto j = cmjson::parse(read_text(index));
            const cmjson::Value* wm = j->get("weight_map");
            if (!wm) throw std::runtime_error("index.json without weight_map");
            std::set<std::string> uniq;
            for (auto& kv : wm->obj) uniq.insert(kv.second->str);
            for (auto& f : uniq) files.push_back(dir + "/" + f);
        } else if (file_exists(single)) {
            files.push_back(single);
        } else {
            DIR* d = opendir(dir.c_str());
            if (!d) throw std::runtime_error("cannot open model dir " + dir);
            while (dirent* e = readdir(d)) {
                std::string n = e->d_name;
                if (n.size() > 12 && n.substr(n.size() - 12) == ".safetensors") files.push_back(dir + "/" + n);
            }
            closedir(d);
            std::sort(files.begin(), files.end());
        }
        if (files.empty()) throw std::runtime_error("no safetensors files in " + dir);
        for (auto& f : files) add_file(f);
    }

    bool has(const std::string& name) const { return tensors_.count(name) != 0; }
    const TensorView& get(const std::string& name) const {
        auto it = tensors_.find(name);
        if (it == tensors_.end()) throw std::runtime_error("missing tensor " + name);
        return it->second;
    }
    std::vector<std::string> names() const {
        std::vector<std::string> n;
        for (auto& kv : tensors_) n.push_back(kv.first);
        return n;
    }

  private:
    std::vector<std::unique_ptr<MappedFile>> files_;
    std::map<std::string, TensorView> tensors_;

    void add_file(const std::string& path) {
        files_.emplace_back(new MappedFile(path));
        const MappedFile& mf = *files_.back();
        if (mf.size() < 8) throw std::runtime_error("truncated safetensors " + path);
        uint64_t hl = 0;
        memcpy(&hl, mf.base(), 8);
        if (hl > mf.size() - 8) throw std::runtime_error("bad safetensors header length in " + path);    // (8 + hl would wrap)
        auto j = cmjson::Parser((const char*)mf.base() + 8, (size_t)hl).parse();
        const uint8_t* data0 = mf.base() + 8 + hl;
        const size_t data_len = mf.size() - 8 - hl;
        for (auto& kv : j->obj) {
            if (kv.first == "__metadata__") continue;
            const cmjson::Value& t = *kv.second;
            TensorView tv;
            tv.dtype = t.string("dtype", "");
            const cmjson::Value* sh = t.get("shape");
            const cmjson::Value* off = t.get("data_offsets");
            if (!sh || !off || off->arr.size() != 2) throw std::runtime_error("bad tensor entry " + kv.first);
            for (auto& d : sh->arr) {
                if (!(d->num >= 0.0 && d->num <= 9.0e15)) throw std::runtime_error("bad dimension in tensor " + kv.first);
                tv.shape.push_back((int64_t)d->num);
            }
            const double bo = off->arr[0]->num, eo = off->arr[1]->num;
            if (!(bo >= 0.0 && eo >= bo && eo <= (double)data_len)) throw std::runtime_error("tensor out of file bounds: " + kv.first);
            const size_t b = (size_t)bo, e = (size_t)eo;
            tv.data = data0 + b;
            tv.nbytes = e - b;
            // every dtype: the byte range must be exactly numel * element size (checked multiplication)
            const int64_t ne = tv.numel();
            const size_t es = TensorView::elem_size(tv.dtype);
            if (ne < 0) throw std::runtime_error("bad shape in tensor " + kv.first);
            if (es != 0 && ((uint64_t)ne > UINT64_MAX / es || (uint64_t)ne * es != (uint64_t)tv.nbytes))
                throw std::runtime_error("tensor " + kv.first + ": data_offsets do not match shape x dtype");
            tensors_[kv.first] = tv;
        }
    }
};

}  // namespace cmst
