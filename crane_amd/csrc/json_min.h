// Minimal JSON reader for config.json / safetensors headers / index files.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace cmjson {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<ValuePtr> arr;
    std::vector<std::pair<std::string, ValuePtr>> obj;   // insertion order kept

    const Value* get(const std::string& k) const {
        for (auto& kv : obj) if (kv.first == k) return kv.second.get();
        return nullptr;
    }
    bool has(const std::string& k) const { const Value* v = get(k); return v && v->kind != Null; }
    double number(const std::string& k, double def) const {
        const Value* v = get(k);
        return (v && v->kind == Num) ? v->num : def;
    }
    long long integer(const std::string& k, long long def) const {
        const Value* v = get(k);
        return (v && v->kind == Num) ? (long long)v->num : def;
    }
    bool boolean(const std::string& k, bool def) const {
        const Value* v = get(k);
        return (v && v->kind == Bool) ? v->b : def;
    }
    std::string string(const std::string& k, const std::string& def) const {
        const Value* v = get(k);
        return (v && v->kind == Str) ? v->str : def;
    }
};

class Parser {
  public:
    Parser(const char* p, size_t n) : p_(p), e_(p + n) {}
    ValuePtr parse() {
        ValuePtr v = value();
        ws();
        if (p_ != e_) fail("trailing characters");
        return v;
    }

  private:
    const char* p_;
    const char* e_;
    [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m); }
    void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
    bool lit(const char* s) {
        size_t n = strlen(s);
        if ((size_t)(e_ - p_) >= n && memcmp(p_, s, n) == 0) { p_ += n; return true; }
        return false;
    }
    std::string str() {
        if (p_ >= e_ || *p_ != '"') fail("expected string");
        ++p_;
        std::string out;
        while (p_ < e_ && *p_ != '"') {
            if (*p_ == '\\') {
                if (++p_ >= e_) fail("bad escape");
                switch (*p_) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (e_ - p_ < 5) fail("bad \\u");
                        unsigned cp = (unsigned)strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p_;
                }
                ++p_;
            } else {
                out += *p_++;
            }
        }
        if (p_ >= e_) fail("unterminated string");
        ++p_;
        return out;
    }
    // nesting is bounded: the parser recurses per level, and a header is attacker-controlled input
    static constexpr int kMaxDepth = 64;
    int depth_ = 0;
    struct DepthGuard { int& d; explicit DepthGuard(int& x) : d(x) { ++d; } ~DepthGuard() { --d; } };
    ValuePtr value() {
        DepthGuard guard(depth_);
        if (depth_ > kMaxDepth) fail("nesting too deep");
        ws();
        if (p_ >= e_) fail("unexpected end");
        auto v = std::make_shared<Value>();
        if (*p_ == '{') {
            v->kind = Value::Obj;
            ++p_; ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                std::string k = str();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v->obj.emplace_back(k, value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (*p_ == '[') {
            v->kind = Value::Arr;
            ++p_; ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return v; }
            for (;;) {
                v->arr.push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (*p_ == '"') {
            v->kind = Value::Str; v->str = str();
        } else if (lit("true")) { v->kind = Value::Bool; v->b = true; }
        else if (lit("false")) { v->kind = Value::Bool; v->b = false; }
        else if (lit("null")) { v->kind = Value::Null; }
        else {
            char* end = nullptr;
            std::string tmp(p_, (size_t)std::min<ptrdiff_t>(e_ - p_, 64));
            double d = strtod(tmp.c_str(), &end);
            if (end == tmp.c_str()) fail("bad token");
            p_ += (end - tmp.c_str());
            v->kind = Value::Num; v->num = d;
        }
        return v;
    }
};

inline ValuePtr parse(const std::string& s) { return Parser(s.data(), s.size()).parse(); }

}  // namespace cmjson
