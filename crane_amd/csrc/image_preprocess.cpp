// Image preprocessor of the vision-language path, host only (no device work):
//   PreprocessorConfig::process          crane-core/src/models/qwen3_5/processor.rs:114-210
//   smart_resize                         processor.rs:64-88   (HF Qwen2VLImageProcessor.smart_resize)
//   resize_exact(.., CatmullRom)         processor.rs:96-112  ("HF's preprocessor_config sets resample: 3 = PIL BICUBIC,
//                                                              whose kernel (a = -0.5) is Catmull-Rom")
// The resampling restated here is PIL's (libImaging/Resample.c: separable, support scaled by the ratio when shrinking,
// 22-bit fixed-point coefficients, 8-bit intermediate, horizontal pass first) because that is the arithmetic the reference
// says it mirrors and the one that can be pinned bit for bit in this image (tests/test_preprocess.py compares with PIL).
// The `image` crate the reference links is absent from /root/reference: parity with ITS rounding is unpinned.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../../include/crane_mi355.h"

namespace {

thread_local std::string g_pre_err;

// processor.rs:64-88
void smart_resize(uint32_t h, uint32_t w, uint32_t factor, uint64_t min_pixels, uint64_t max_pixels, uint32_t* ho, uint32_t* wo) {
    const double hf = (double)h, wf = (double)w;
    // multiples of `factor` as a count q, clamped so that q * factor stays inside uint32 whatever the configuration says
    // (the caller rejects a result it cannot hold; a hostile min_pixels must not become an out-of-range float -> int cast)
    const double qmax = std::floor(4294967295.0 / (double)factor);
    auto count = [&](double q) { return (uint32_t)(q < 0.0 || q != q ? 0.0 : (q > qmax ? qmax : q)); };
    auto round_by_factor = [&](double x) {
        // Rust's f64::round: half away from zero
        const double r = std::floor(x / (double)factor + 0.5);
        return count(r < 1.0 ? 1.0 : r) * factor;
    };
    uint32_t hb = round_by_factor(hf), wb = round_by_factor(wf);
    const uint64_t area = (uint64_t)hb * (uint64_t)wb;
    if (area > max_pixels) {
        const double beta = std::sqrt(hf * wf / (double)max_pixels);
        hb = count(std::floor(hf / beta / (double)factor)) * factor;
        wb = count(std::floor(wf / beta / (double)factor)) * factor;
    } else if (area < min_pixels) {
        const double beta = std::sqrt((double)min_pixels / (hf * wf));
        hb = count(std::ceil(hf * beta / (double)factor)) * factor;
        wb = count(std::ceil(wf * beta / (double)factor)) * factor;
    }
    *ho = hb < factor ? factor : hb;
    *wo = wb < factor ? factor : wb;
}

inline double bicubic(double x) {          // Catmull-Rom, a = -0.5, support 2
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Coeffs { int ksize; std::vector<int> bounds; std::vector<int32_t> kk; };

Coeffs precompute(int in_size, int out_size) {
    Coeffs c;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    c.ksize = (int)std::ceil(support) * 2 + 1;
    c.bounds.resize((size_t)out_size * 2);
    c.kk.assign((size_t)out_size * c.ksize, 0);
    std::vector<double> k((size_t)c.ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[(size_t)x] = bicubic((x + xmin - center + 0.5) * ss); ww += k[(size_t)x]; }
        for (int x = 0; x < xmax; ++x) {
            double v = ww != 0.0 ? k[(size_t)x] / ww : k[(size_t)x];
            c.kk[(size_t)xx * c.ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (double)(1 << PRECISION_BITS));
        }
        c.bounds[(size_t)xx * 2] = xmin;
        c.bounds[(size_t)xx * 2 + 1] = xmax;
    }
    return c;
}

inline uint8_t clip8(int32_t v) {
    const int32_t s = v >> PRECISION_BITS;
    return (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
}

// PIL Image.resize((w_out, h_out), BICUBIC) on an RGB8 image
std::vector<uint8_t> resize_bicubic(const uint8_t* src, int h, int w, int ho, int wo) {
    std::vector<uint8_t> cur(src, src + (size_t)h * w * 3);
    int ch = h, cw = w;
    if (wo != cw) {                                             // horizontal pass
        const Coeffs c = precompute(cw, wo);
        std::vector<uint8_t> out((size_t)ch * wo * 3);
        for (int y = 0; y < ch; ++y)
            for (int xx = 0; xx < wo; ++xx) {
                const int xmin = c.bounds[(size_t)xx * 2], xmax = c.bounds[(size_t)xx * 2 + 1];
                const int32_t* k = &c.kk[(size_t)xx * c.ksize];
                for (int b = 0; b < 3; ++b) {
                    int32_t ss = 1 << (PRECISION_BITS - 1);
                    for (int x = 0; x < xmax; ++x) ss += (int32_t)cur[((size_t)y * cw + x + xmin) * 3 + b] * k[x];
                    out[((size_t)y * wo + xx) * 3 + b] = clip8(ss);
                }
            }
        cur.swap(out);
        cw = wo;
    }
    if (ho != ch) {                                             // vertical pass
        const Coeffs c = precompute(ch, ho);
        std::vector<uint8_t> out((size_t)ho * cw * 3);
        for (int yy = 0; yy < ho; ++yy) {
            const int ymin = c.bounds[(size_t)yy * 2], ymax = c.bounds[(size_t)yy * 2 + 1];
            const int32_t* k = &c.kk[(size_t)yy * c.ksize];
            for (int x = 0; x < cw; ++x)
                for (int b = 0; b < 3; ++b) {
                    int32_t ss = 1 << (PRECISION_BITS - 1);
                    for (int y = 0; y < ymax; ++y) ss += (int32_t)cur[((size_t)(y + ymin) * cw + x) * 3 + b] * k[y];
                    out[((size_t)yy * cw + x) * 3 + b] = clip8(ss);
                }
        }
        cur.swap(out);
        ch = ho;
    }
    return cur;
}

bool check_cfg(const cm_preproc_config* c) {
    if (!c || c->patch_size == 0 || c->temporal_patch_size == 0 || c->merge_size == 0 || c->max_pixels == 0) { g_pre_err = "bad preprocessor config"; return false; }
    // bounds far above any real configuration (patch 14 / 16, temporal 2, merge 2): they keep patch_size * merge_size and the
    // row length T * 3 * P^2 inside 32 bits
    if (c->patch_size > 1024 || c->temporal_patch_size > 64 || c->merge_size > 64) { g_pre_err = "preprocessor config out of range"; return false; }
    for (int i = 0; i < 3; ++i) if (!(c->image_std[i] > 0.f)) { g_pre_err = "image_std must be positive"; return false; }
    return true;
}

}  // namespace

extern "C" {

const char* cm_preprocess_last_error(void) { return g_pre_err.c_str(); }

int cm_image_smart_resize(const cm_preproc_config* cfg, uint32_t height, uint32_t width, uint32_t* h_out, uint32_t* w_out) {
    if (!check_cfg(cfg) || !h_out || !w_out || height == 0 || width == 0) { if (g_pre_err.empty()) g_pre_err = "bad argument"; return CM_ERR_INVALID; }
    smart_resize(height, width, cfg->patch_size * cfg->merge_size, cfg->min_pixels, cfg->max_pixels, h_out, w_out);
    return CM_OK;
}

int cm_image_preprocess(const cm_preproc_config* cfg, const uint8_t* rgb, uint32_t height, uint32_t width, float* pixel_values_out,
                        size_t cap_floats, uint32_t grid_thw_out[3], size_t* n_patches_out) {
    if (!check_cfg(cfg)) return CM_ERR_INVALID;
    if (!rgb || !grid_thw_out || !n_patches_out || height == 0 || width == 0 || height > 65535 || width > 65535) { g_pre_err = "bad argument"; return CM_ERR_INVALID; }
    const uint32_t P = cfg->patch_size, T = cfg->temporal_patch_size, M = cfg->merge_size;
    uint32_t hn = 0, wn = 0;
    smart_resize(height, width, P * M, cfg->min_pixels, cfg->max_pixels, &hn, &wn);
    // 2^28 pixels (16384 x 16384) bounds what a configuration can make this call allocate
    if ((uint64_t)hn * (uint64_t)wn > (1ull << 28)) { g_pre_err = "resized image too large (min_pixels / max_pixels out of range)"; return CM_ERR_RANGE; }
    const uint32_t hp = hn / P, wp = wn / P;
    const size_t n_patches = (size_t)hp * wp, in_dim = (size_t)T * 3 * P * P;
    grid_thw_out[0] = 1; grid_thw_out[1] = hp; grid_thw_out[2] = wp;
    *n_patches_out = n_patches;
    if (pixel_values_out == nullptr || cap_floats == 0) return CM_OK;                 // size query
    if (cap_floats < n_patches * in_dim) { g_pre_err = "pixel_values buffer too small"; return CM_ERR_RANGE; }
    if (hp % M || wp % M) { g_pre_err = "resized grid is not a multiple of the merge size"; return CM_ERR_INVALID; }
    std::vector<uint8_t> img;
    try {
        img = (hn == height && wn == width) ? std::vector<uint8_t>(rgb, rgb + (size_t)height * width * 3)
                                            : resize_bicubic(rgb, (int)height, (int)width, (int)hn, (int)wn);
    } catch (const std::exception& e) {                       // (bad_alloc must not cross the C ABI)
        g_pre_err = std::string("image resize failed: ") + e.what();
        return CM_ERR_OOM;
    }
    // rows ordered (h block, w block, row in block, col in block): the 2x2 patches of a merge block are contiguous;
    // each row laid out (channel, temporal copy, y, x) -- processor.rs:150-196
    const size_t pp = (size_t)P * P;
    size_t patch_idx = 0;
    for (uint32_t hb = 0; hb < hp / M; ++hb)
        for (uint32_t wb = 0; wb < wp / M; ++wb)
            for (uint32_t mr = 0; mr < M; ++mr)
                for (uint32_t mc = 0; mc < M; ++mc) {
                    const uint32_t py = hb * M + mr, px = wb * M + mc;
                    float* row = pixel_values_out + patch_idx * in_dim;
                    for (uint32_t c = 0; c < 3; ++c) {
                        const float mean = cfg->image_mean[c], sd = cfg->image_std[c];
                        for (uint32_t j = 0; j < P; ++j)
                            for (uint32_t i = 0; i < P; ++i) {
                                const size_t y = (size_t)py * P + j, x = (size_t)px * P + i;
                                const float v = (float)img[(y * wn + x) * 3 + c] / 255.0f;
                                const float nv = (v - mean) / sd;
                                for (uint32_t t = 0; t < T; ++t) row[c * (T * pp) + t * pp + j * P + i] = nv;
                            }
                    }
                    ++patch_idx;
                }
    return CM_OK;
}

}  // extern "C"
