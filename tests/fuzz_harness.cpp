// AddressSanitizer harness for the two container parsers of the loader (crane_amd/csrc/gguf.h, safetensors.h + json_min.h).
// Test infrastructure: built and run by tests/test_loader_host.py with
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I crane_amd/csrc tests/fuzz_harness.cpp
// The parsers mmap their file; here mmap / munmap are replaced by an exact-size heap copy, so that a read of even ONE byte
// past the end of the file lands in an ASan red zone instead of the slack of the last page.
//     fuzz_harness <dir>      every *.gguf file and every sub-directory holding a model.safetensors under <dir>
// A parser that throws is fine (the C ABI turns that into a status code); the process only dies on a sanitizer report.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>

static void* fz_mmap(void*, size_t n, int, int, int fd, off_t) {
    unsigned char* p = (unsigned char*)malloc(n ? n : 1);
    size_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, p + got, n - got, (off_t)got);
        if (r <= 0) break;
        got += (size_t)r;
    }
    return p;
}
static int fz_munmap(void* p, size_t) { free(p); return 0; }
#define mmap fz_mmap
#define munmap fz_munmap
#include "gguf.h"
#include "safetensors.h"

static unsigned long long g_sink = 0;

static void one_gguf(const std::string& path) {
    try {
        cmgguf::File f(path);
        for (auto& kv : f.tensors()) {                                  // touch what the loader would read
            const cmgguf::TensorInfo& t = kv.second;
            for (size_t i = 0; i < t.nbytes; i += 61) g_sink += t.data[i];
            if (t.nbytes) g_sink += t.data[t.nbytes - 1];
        }
        if (const cmgguf::Value* a = f.meta("general.architecture")) g_sink += a->s.size();
    } catch (const std::exception&) {
    }
}

static void one_checkpoint(const std::string& dir) {
    try {
        cmst::Checkpoint ck(dir);
        for (const std::string& n : ck.names()) {
            const cmst::TensorView& t = ck.get(n);
            for (size_t i = 0; i < t.nbytes; ++i) g_sink += t.data[i];  // cm_checkpoint_inspect hashes every byte
        }
    } catch (const std::exception&) {
    }
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string root = argv[1];
    DIR* d = opendir(root.c_str());
    if (!d) return 2;
    int n = 0;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        const std::string p = root + "/" + name;
        if (name.size() > 5 && name.substr(name.size() - 5) == ".gguf") { one_gguf(p); ++n; }
        else {
            struct stat st;
            if (stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) { one_checkpoint(p); ++n; }
        }
    }
    closedir(d);
    printf("fuzz_harness: %d inputs, no sanitizer report (%llu)\n", n, g_sink);
    return 0;
}
