// Walks the host-array entry points of include/orbx.h with odd sizes, strides and empty inputs. Meant to be built
// with the host side of the library under AddressSanitizer (tests/cpp/run_host_asan.sh): what it checks is that
// the host code of the ABI (staging buffers, packing, validation, the greedy passes) stays inside its buffers.
// Results are only sanity-checked here; parity is the business of the Python tests.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbx.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != ORBX_OK) {                                                                \
            std::printf("FAILED %s -> %d (%s) at line %d\n", #call, rc_, orbx_last_error(), __LINE__); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:268-283) on the host, for the matcher calls below
static void make_grid(const std::vector<orbx_keypoint>& k, float w, float h, std::vector<int32_t>& start,
                      std::vector<int32_t>& items) {
    const int C = ORBX_GRID_COLS, R = ORBX_GRID_ROWS;
    std::vector<std::vector<int32_t>> cells((size_t)C * R);
    for (size_t i = 0; i < k.size(); i++) {
        const int px = (int)std::round(k[i].x * ((float)C / w)), py = (int)std::round(k[i].y * ((float)R / h));
        if (px < 0 || px >= C || py < 0 || py >= R) continue;
        cells[(size_t)px * R + py].push_back((int32_t)i);
    }
    start.assign((size_t)C * R + 1, 0);
    items.clear();
    for (int c = 0; c < C * R; c++) {
        start[c] = (int32_t)items.size();
        items.insert(items.end(), cells[c].begin(), cells[c].end());
    }
    start[(size_t)C * R] = (int32_t)items.size();
    if (items.empty()) items.push_back(0);       // never hand over a null pointer
}

int main() {
    // ---- extraction: several sizes through one context, padded rows, batch with gaps, too small a capacity ----
    orbx_config cfg = {700, 1.2f, 8, 20, 7, 0, 800, 600, 3, ORBX_BLUR_CV_4_0_0};
    orbx_ctx* ctx = nullptr;
    CHECK(orbx_create(&cfg, &ctx));
    const int cap = orbx_keypoint_capacity(ctx);
    std::vector<orbx_keypoint> kps((size_t)3 * cap);
    std::vector<uint8_t> desc((size_t)3 * cap * 32);
    const int sizes[][2] = {{752, 480}, {751, 479}, {333, 222}, {800, 600}, {97, 131}, {40, 40}, {1, 1}};
    std::vector<orbx_keypoint> k752;
    std::vector<uint8_t> d752;
    for (const auto& wh : sizes) {
        const int w = wh[0], h = wh[1];
        const size_t stride = (size_t)w + (rnd() % 9);
        std::vector<uint8_t> img(stride * h);
        CHECK(orbx_synth_frame(rnd() % 1000, w, h, ORBX_SYNTH_SCENE, 300, 40, 0, img.data(), stride));
        int n = -1;
        CHECK(orbx_extract(ctx, img.data(), w, h, stride, kps.data(), desc.data(), cap, &n));
        std::printf("%dx%d stride %zu: %d keypoints\n", w, h, stride, n);
        if (n < 0 || n > cap) return 2;
        if (w == 752) {
            k752.assign(kps.begin(), kps.begin() + n);
            d752.assign(desc.begin(), desc.begin() + (size_t)n * 32);
            int tiny = 0;
            std::vector<orbx_keypoint> k5(5);
            std::vector<uint8_t> d5(5 * 32);
            const int rc = orbx_extract(ctx, img.data(), w, h, stride, k5.data(), d5.data(), 5, &tiny);
            if (rc != ORBX_ERR_CAPACITY || tiny != n) return 3;
        }
        std::vector<uint8_t> level((size_t)(w + 38) * (h + 38));
        int lw = 0, lh = 0;
        if (orbx_level_size(ctx, 1, &lw, &lh) == ORBX_OK && lw > 0 && lh > 0)
            CHECK(orbx_get_level(ctx, 0, 1, 1, level.data(), (size_t)lw + 38));
    }
    {
        const int w = 321, h = 243, B = 3;
        const size_t rs = 330, fs = rs * (h + 2) + 7;
        std::vector<uint8_t> frames(fs * B + 64);
        for (int f = 0; f < B; f++) CHECK(orbx_synth_frame(50 + f, w, h, ORBX_SYNTH_SCENE, 100, 10, 0, frames.data() + 3 + f * fs, rs));
        int32_t counts[3] = {-1, -1, -1};
        CHECK(orbx_extract_batch(ctx, frames.data() + 3, B, w, h, rs, fs, kps.data(), desc.data(), cap, counts));
        std::printf("batch of 3 padded frames: %d %d %d\n", counts[0], counts[1], counts[2]);
    }
    int dummy = 7;
    CHECK(orbx_extract(ctx, nullptr, 0, 0, 0, kps.data(), desc.data(), cap, &dummy));
    if (dummy != 7) return 4;
    orbx_destroy(ctx);

    // ---- matcher entry points on the 752x480 result against a shifted copy of itself ----
    orbx_matcher* m = nullptr;
    CHECK(orbx_matcher_create(0, &m));
    const int n1 = (int)k752.size();
    std::vector<orbx_keypoint> k2 = k752;
    std::vector<uint8_t> d2 = d752;
    for (int i = 0; i < n1; i++) {
        k2[i].x += (float)(rnd() % 7) - 3.f;
        d2[(size_t)i * 32 + rnd() % 32] ^= (uint8_t)(1u << (rnd() % 8));
    }
    std::vector<int32_t> start, items;
    make_grid(k2, 752.f, 480.f, start, items);
    {
        std::vector<int64_t> idx(n1);
        std::vector<int32_t> b1(n1), b2(n1);
        CHECK(orbx_hamming_knn2(m, d752.data(), n1, d2.data(), n1, 256, idx.data(), b1.data(), b2.data()));
        CHECK(orbx_hamming_knn2(m, d752.data(), 1, d2.data(), 0, 256, idx.data(), b1.data(), b2.data()));
        CHECK(orbx_hamming_knn2(m, d752.data(), 0, d2.data(), 5, 256, idx.data(), b1.data(), b2.data()));
    }
    {
        std::vector<float> prev((size_t)2 * n1);
        for (int i = 0; i < n1; i++) { prev[2 * i] = k752[i].x; prev[2 * i + 1] = k752[i].y; }
        std::vector<int32_t> m12(n1);
        int nm = -1;
        for (int window : {1, 30, 100, 2000})
            CHECK(orbx_search_for_initialization(m, k752.data(), d752.data(), n1, k2.data(), d2.data(), n1, start.data(),
                                                 items.data(), 0.f, 0.f, 752.f, 480.f, prev.data(), window, 0.9f, 1,
                                                 m12.data(), &nm));
        std::printf("SearchForInitialization: %d matches\n", nm);
        CHECK(orbx_search_for_initialization(m, k752.data(), d752.data(), 0, k2.data(), d2.data(), n1, start.data(),
                                             items.data(), 0.f, 0.f, 752.f, 480.f, prev.data(), 100, 0.9f, 1, m12.data(), &nm));
    }
    {
        std::vector<float> uv((size_t)2 * n1), iz(n1, 0.5f), ang(n1), xr(n1), vc(n1, 0.999f), ur(n1, -1.f);
        std::vector<int32_t> oct(n1), who(n1);
        std::vector<uint8_t> obs(n1, 1), occ(n1, 0);
        float sf[8];
        sf[0] = 1.f;
        for (int l = 1; l < 8; l++) sf[l] = sf[l - 1] * 1.2f;
        for (int i = 0; i < n1; i++) {
            uv[2 * i] = k752[i].x + (float)(rnd() % 5) - 2.f;
            uv[2 * i + 1] = k752[i].y;
            ang[i] = k752[i].angle;
            oct[i] = k752[i].octave;
            xr[i] = uv[2 * i] - 10.f;
            if (rnd() % 3 == 0) ur[i] = k2[i].x - 10.f;
            if (rnd() % 20 == 0) iz[i] = -1.f;
            if (rnd() % 20 == 0) oct[i] = 99;
            occ[i] = rnd() % 10 == 0;
        }
        int nm = -1;
        for (int dir = 0; dir < 3; dir++)
            CHECK(orbx_search_by_projection_frame(m, uv.data(), iz.data(), oct.data(), ang.data(), d752.data(), obs.data(), n1,
                                                  k2.data(), d2.data(), dir ? ur.data() : nullptr, dir == 1 ? occ.data() : nullptr,
                                                  n1, start.data(), items.data(), 0.f, 0.f, 752.f, 480.f, sf, 8, 15.f, 40.f,
                                                  dir, 100, 1, who.data(), &nm));
        std::printf("SearchByProjection(frame): %d matches\n", nm);
        for (int i = 0; i < n1; i++) if (oct[i] == 99) oct[i] = 7;
        CHECK(orbx_search_by_projection_points(m, uv.data(), xr.data(), vc.data(), oct.data(), d752.data(), obs.data(), n1,
                                               k2.data(), d2.data(), ur.data(), occ.data(), n1, start.data(), items.data(),
                                               0.f, 0.f, 752.f, 480.f, sf, 8, 3.f, 0.8f, who.data(), &nm));
        std::printf("SearchByProjection(points): %d matches\n", nm);
        CHECK(orbx_search_by_projection_points(m, uv.data(), nullptr, vc.data(), oct.data(), d752.data(), obs.data(), 0,
                                               k2.data(), d2.data(), nullptr, nullptr, n1, start.data(), items.data(),
                                               0.f, 0.f, 752.f, 480.f, sf, 8, 1.f, 0.8f, who.data(), &nm));
    }
    {
        // feature vectors: node = index / 9 on side 1, (index / 9) with a few strays on side 2
        std::vector<uint32_t> nodes1, items1, nodes2, items2;
        std::vector<int32_t> start1{0}, start2{0};
        for (int i = 0; i < n1; i++) {
            if (i % 9 == 0 && i) { start1.push_back((int32_t)items1.size()); start2.push_back((int32_t)items2.size()); }
            if (i % 9 == 0) { nodes1.push_back((uint32_t)(i / 9) * 3u); nodes2.push_back((uint32_t)(i / 9) * 3u); }
            items1.push_back((uint32_t)i);
            { const int j = (i / 9) * 9 + (8 - i % 9); items2.push_back((uint32_t)(j < n1 ? j : i)); }   // reversed inside a node
        }
        start1.push_back((int32_t)items1.size());
        start2.push_back((int32_t)items2.size());
        std::vector<uint8_t> v1(n1, 1), v2(n1, 1);
        for (int i = 0; i < n1; i += 5) v1[i] = 0;
        std::vector<int32_t> out(n1);
        int nm = -1;
        for (int mode : {ORBX_BOW_KEYFRAME_FRAME, ORBX_BOW_KEYFRAME_KEYFRAME})
            CHECK(orbx_search_by_bow(m, mode, k752.data(), d752.data(), v1.data(), n1, nodes1.data(), start1.data(), items1.data(),
                                     (int)nodes1.size(), k2.data(), d2.data(), v2.data(), n1, nodes2.data(), start2.data(),
                                     items2.data(), (int)nodes2.size(), 0.7f, 1, out.data(), &nm));
        std::printf("SearchByBoW: %d matches\n", nm);
        // refusals must not touch anything
        items2[3] = (uint32_t)n1 + 5;
        if (orbx_search_by_bow(m, 0, k752.data(), d752.data(), v1.data(), n1, nodes1.data(), start1.data(), items1.data(),
                               (int)nodes1.size(), k2.data(), d2.data(), v2.data(), n1, nodes2.data(), start2.data(),
                               items2.data(), (int)nodes2.size(), 0.7f, 1, out.data(), &nm) == ORBX_OK) return 6;
        items[0] = n1 + 1;
        std::vector<float> prev((size_t)2 * n1, 10.f);
        if (orbx_search_for_initialization(m, k752.data(), d752.data(), n1, k2.data(), d2.data(), n1, start.data(), items.data(),
                                           0.f, 0.f, 752.f, 480.f, prev.data(), 100, 0.9f, 1, out.data(), &nm) == ORBX_OK) return 7;
    }
    orbx_matcher_destroy(m);
    std::printf("api walk ok\n");
    return 0;
}
