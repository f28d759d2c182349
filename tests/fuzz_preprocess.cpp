// AddressSanitizer / UBSan harness for the host-only image preprocessor (crane_amd/csrc/image_preprocess.cpp): random
// configurations (patch 1..32, merge 1..4, temporal 1..3, min / max pixel counts that force shrinking AND enlarging) on random
// images of 1..300 pixels a side, output buffer of EXACTLY n_patches * row floats.  Built and run by tests/test_preprocess.py:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I include tests/fuzz_preprocess.cpp \
//         crane_amd/csrc/image_preprocess.cpp
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "crane_mi355.h"
int main(int argc, char** argv) {
    std::mt19937_64 r(argc > 1 ? atoi(argv[1]) : 1);
    int iters = argc > 2 ? atoi(argv[2]) : 2000, ok = 0, err = 0;
    for (int it = 0; it < iters; ++it) {
        cm_preproc_config c{};
        c.patch_size = 1 + r() % 32; c.temporal_patch_size = 1 + r() % 3; c.merge_size = 1 + r() % 4;
        c.min_pixels = (r() % 4 == 0) ? r() % 200000 : r() % 4096; c.max_pixels = 1 + r() % 300000;
        for (int i = 0; i < 3; ++i) { c.image_mean[i] = 0.5f; c.image_std[i] = 0.25f; }
        uint32_t h = 1 + r() % 300, w = 1 + r() % 300;
        if (r() % 50 == 0) { h = 1 + r() % 3; }
        if (r() % 50 == 0) { w = 1 + r() % 3; }
        std::vector<uint8_t> img((size_t)h * w * 3);
        for (auto& b : img) b = (uint8_t)r();
        uint32_t grid[3]; size_t n = 0;
        int rc = cm_image_preprocess(&c, img.data(), h, w, nullptr, 0, grid, &n);
        if (rc != 0) { ++err; continue; }
        size_t in_dim = (size_t)c.temporal_patch_size * 3 * c.patch_size * c.patch_size;
        std::vector<float> out(n * in_dim);                      // exact size: ASan red zone right behind
        rc = cm_image_preprocess(&c, img.data(), h, w, out.data(), out.size(), grid, &n);
        if (rc == 0) ++ok; else ++err;
    }
    printf("preprocess fuzz: %d ok, %d rejected\n", ok, err);
    return 0;
}
