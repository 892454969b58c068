// Latency of the drop-in call from plain C++: orbx_extract on one 752x480 frame in (pageable) host memory, results
// into caller-owned host arrays, as Frame::ExtractORB calls ORBextractor::operator() once per frame on the tracking
// thread (src/Frame.cc:285-291). No Python, no argument marshalling: what a C++ caller of the ABI sees.
//   g++ -std=c++17 -O2 -Iinclude tools/latency_single.cpp -o tools/latency_single -Lmy_orb_slam2_amd -lorbx \
//       -Wl,-rpath,'$ORIGIN/../my_orb_slam2_amd' -Wl,-rpath-link,/opt/rocm/lib
//   tools/latency_single [calls]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx.h"

int main(int argc, char** argv) {
    const int calls = argc > 1 ? std::atoi(argv[1]) : 4000;
    const int W = 752, H = 480;
    orbx_config cfg{};
    cfg.nfeatures = 1000;
    cfg.scale_factor = 1.2f;
    cfg.nlevels = 8;
    cfg.ini_th_fast = 20;
    cfg.min_th_fast = 7;
    cfg.device = 0;
    cfg.max_width = W;
    cfg.max_height = H;
    cfg.max_batch = 1;
    cfg.blur_taps = ORBX_BLUR_CV_4_0_0;
    orbx_ctx* ctx = nullptr;
    if (orbx_create(&cfg, &ctx) != ORBX_OK) {
        std::printf("orbx_create: %s\n", orbx_last_error());
        return 1;
    }
    const int cap = orbx_keypoint_capacity(ctx);
    std::vector<std::vector<uint8_t>> frames(8, std::vector<uint8_t>((size_t)W * H));
    for (int i = 0; i < 8; i++)
        if (orbx_synth_frame((uint64_t)i, W, H, ORBX_SYNTH_SCENE, 300, 40, 0, frames[i].data(), W) != ORBX_OK) return 1;
    std::vector<orbx_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    long total = 0;
    for (int i = 0; i < 100; i++)
        if (orbx_extract(ctx, frames[i % 8].data(), W, H, W, kps.data(), desc.data(), cap, &n) != ORBX_OK) {
            std::printf("orbx_extract: %s\n", orbx_last_error());
            return 1;
        }
    std::vector<double> us(calls);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < calls; i++) {
        const auto a = std::chrono::steady_clock::now();
        if (orbx_extract(ctx, frames[i % 8].data(), W, H, W, kps.data(), desc.data(), cap, &n) != ORBX_OK) return 1;
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
        total += n;
    }
    const double mean = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
    std::sort(us.begin(), us.end());
    std::printf("{\"workload\": \"orbx_extract, one 752x480 frame per call, host memory in and out\", \"calls\": %d, "
                "\"mean_keypoints\": %.1f, \"mean_us\": %.1f, \"median_us\": %.1f, \"p10_us\": %.1f, \"p90_us\": %.1f, "
                "\"p99_us\": %.1f}\n",
                calls, (double)total / calls, mean, us[calls / 2], us[calls / 10], us[calls * 9 / 10], us[calls * 99 / 100]);
    orbx_destroy(ctx);
    return 0;
}
