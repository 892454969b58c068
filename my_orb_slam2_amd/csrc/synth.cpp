// Seeded, integer-only synthetic frame generator (host code).
//
// The EuRoC images the reference's examples read (orb_slam2_mono_example/mono_euroc.cc:86) are not
// available offline, so benchmarks and parity tests run on frames every machine can regenerate
// bit-for-bit from an index: value-noise background, painted rectangles and checker patches (many
// FAST corners, some empty cells), plus +-3 sensor noise.  A "right view" paints every object
// shifted left by its own disparity in [0,47] px so stereo band matching has true matches.
#include "orbx.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform integer in [lo, hi]
    int range(int lo, int hi) { return lo + (int)((next() >> 11) % (uint64_t)(hi - lo + 1)); }
};

inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void fill_rect(uint8_t* out, size_t stride, int w, int h, int x0, int y0, int rw, int rh, int gray) {
    int x1 = x0 + rw, y1 = y0 + rh;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > w) x1 = w;
    if (y1 > h) y1 = h;
    for (int y = y0; y < y1; y++)
        if (x1 > x0) std::memset(out + (size_t)y * stride + x0, gray, (size_t)(x1 - x0));
}

}  // namespace

extern "C" int orbx_synth_frame(uint64_t index, int w, int h, int kind, int nrect, int ncheck,
                                int right_view, uint8_t* out, size_t stride) {
    if (!out || w <= 0 || h <= 0 || stride < (size_t)w) return ORBX_ERR_INVALID;
    if (kind == ORBX_SYNTH_ZERO || kind == ORBX_SYNTH_CONST128) {
        for (int y = 0; y < h; y++)
            std::memset(out + (size_t)y * stride, kind == ORBX_SYNTH_ZERO ? 0 : 128, (size_t)w);
        return ORBX_OK;
    }
    SplitMix64 rng(0x9E3779B97F4A7C15ull * (index + 1));
    if (kind == ORBX_SYNTH_NOISE) {
        for (int y = 0; y < h; y++) {
            uint8_t* row = out + (size_t)y * stride;
            for (int x = 0; x < w; x += 8) {
                uint64_t r = rng.next();
                for (int k = 0; k < 8 && x + k < w; k++) row[x + k] = (uint8_t)(r >> (8 * k));
            }
        }
        return ORBX_OK;
    }
    if (kind != ORBX_SYNTH_SCENE) return ORBX_ERR_INVALID;

    // background: value noise on an 8-px lattice (amplitude 64..191), bilinear in integers
    const int gw = w / 8 + 2, gh = h / 8 + 2;
    std::vector<uint8_t> lat((size_t)gw * gh);
    for (size_t i = 0; i < lat.size(); i++) lat[i] = (uint8_t)(64 + (rng.next() >> 11) % 128);
    for (int y = 0; y < h; y++) {
        const int gy = y >> 3, fy = y & 7;
        uint8_t* row = out + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            const int gx = x >> 3, fx = x & 7;
            const int g00 = lat[(size_t)gy * gw + gx], g10 = lat[(size_t)gy * gw + gx + 1];
            const int g01 = lat[(size_t)(gy + 1) * gw + gx], g11 = lat[(size_t)(gy + 1) * gw + gx + 1];
            row[x] = (uint8_t)(((8 - fx) * (8 - fy) * g00 + fx * (8 - fy) * g10 +
                                (8 - fx) * fy * g01 + fx * fy * g11 + 32) >> 6);
        }
    }
    // rectangles, painted in order; each has its own disparity for the right view
    for (int i = 0; i < nrect; i++) {
        int x0 = rng.range(-60, w - 1), y0 = rng.range(-60, h - 1);
        int rw = rng.range(4, 120), rh = rng.range(4, 120);
        int gray = rng.range(0, 255), disp = rng.range(0, 47);
        fill_rect(out, stride, w, h, x0 - (right_view ? disp : 0), y0, rw, rh, gray);
    }
    // checker patches
    for (int i = 0; i < ncheck; i++) {
        int cell = rng.range(6, 16), nx = rng.range(3, 8), ny = rng.range(3, 8);
        int x0 = rng.range(0, w - 1), y0 = rng.range(0, h - 1);
        int g0 = rng.range(0, 255), g1 = rng.range(0, 255), disp = rng.range(0, 47);
        if (right_view) x0 -= disp;
        for (int cy = 0; cy < ny; cy++)
            for (int cx = 0; cx < nx; cx++)
                fill_rect(out, stride, w, h, x0 + cx * cell, y0 + cy * cell, cell, cell,
                          ((cx + cy) & 1) ? g1 : g0);
    }
    // sensor noise, uniform in [-3, 3]; the two views draw from different streams
    SplitMix64 nz(0xD1B54A32D192ED03ull * (index + 1) + (right_view ? 0x632BE59BD9B4E019ull : 0));
    for (int y = 0; y < h; y++) {
        uint8_t* row = out + (size_t)y * stride;
        for (int x = 0; x < w; x += 8) {
            uint64_t r = nz.next();
            for (int k = 0; k < 8 && x + k < w; k++)
                row[x + k] = clamp_u8((int)row[x + k] + (int)((r >> (8 * k)) & 0xFF) % 7 - 3);
        }
    }
    return ORBX_OK;
}
