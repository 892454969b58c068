// Exercises the C++ shim the way Frame::ExtractORB does and prints a digest that the Python test
// compares with the oracle's result for the same synthetic frame.
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "ORBextractorGpu.h"

int main() {
    const int W = 752, H = 480;
    std::vector<uint8_t> img(static_cast<size_t>(W) * H);
    if (orbx_synth_frame(0, W, H, ORBX_SYNTH_SCENE, 300, 40, 0, img.data(), W) != ORBX_OK) return 2;
    ORB_SLAM2::ORBextractorGpu ex(1000, 1.2f, 8, 20, 7, 0, W, H);
    std::vector<orbx_keypoint> k;
    std::vector<uint8_t> d;
    ex(img.data(), W, H, W, k, d, true);
    unsigned long long h = 1469598103934665603ull;   // FNV-1a over descriptors
    for (uint8_t b : d) { h ^= b; h *= 1099511628211ull; }
    std::printf("n=%zu levels=%d scale1=%.9g desc_fnv=%016llx kp0=(%.1f,%.1f,%.6f) pyr1=%dx%d px=%d dist=%d\n",
                k.size(), ex.GetLevels(), ex.GetScaleFactors()[1], h, k[0].x, k[0].y, k[0].angle,
                ex.mvImagePyramid[1].cols, ex.mvImagePyramid[1].rows, (int)ex.mvImagePyramid[1].ptr()[0],
                ORB_SLAM2::DescriptorDistance(d.data(), d.data() + 32));
    // map-file block: count + raw 28-byte records, {rows, cols, type} + bytes; round trip
    std::stringstream ss;
    ORB_SLAM2::WriteKeyPoints(ss, k);
    ORB_SLAM2::WriteDescriptors(ss, d);
    if (ss.str().size() != 8 + 28 * k.size() + 12 + d.size()) return 4;
    std::vector<orbx_keypoint> kb;
    std::vector<uint8_t> db;
    if (!ORB_SLAM2::ReadKeyPoints(ss, kb) || !ORB_SLAM2::ReadDescriptors(ss, db)) return 5;
    if (kb.size() != k.size() || std::memcmp(kb.data(), k.data(), 28 * k.size()) != 0 || db != d) return 5;
    {
        // corrupt blocks are refused without a huge allocation: a count the file cannot hold, a truncated
        // block, a descriptor header that is not rows x 32 of CV_8UC1
        const std::string good = ss.str();
        std::string bad = good;
        bad[6] = 0x7f;                                            // count = 2^55 or so
        std::stringstream s1(bad);
        if (ORB_SLAM2::ReadKeyPoints(s1, kb) || !kb.empty()) return 6;
        std::stringstream s2(good.substr(0, 8 + 28 * k.size() - 5));
        if (ORB_SLAM2::ReadKeyPoints(s2, kb) || !kb.empty()) return 6;
        for (int field = 1; field <= 2; field++) {
            bad = good;
            bad[8 + 28 * k.size() + 4 * field] ^= 1;              // cols 33, then type 1
            std::stringstream s3(bad);
            if (!ORB_SLAM2::ReadKeyPoints(s3, kb) || ORB_SLAM2::ReadDescriptors(s3, db) || !db.empty()) return 6;
        }
        std::stringstream s4(good.substr(0, good.size() - 1));
        if (!ORB_SLAM2::ReadKeyPoints(s4, kb) || ORB_SLAM2::ReadDescriptors(s4, db)) return 6;
    }
    {
        // a keyframe without keypoints: the reference writes an empty cv::Mat as {0, 0, 0}; it must read back as
        // "no descriptors", and so must {0, 32, 0}
        const uint32_t empties[2][3] = {{0u, 0u, 0u}, {0u, 32u, 0u}};
        for (const auto& hdr : empties) {
            std::stringstream s5(std::string(reinterpret_cast<const char*>(hdr), sizeof(hdr)));
            db.assign(7, 1);
            if (!ORB_SLAM2::ReadDescriptors(s5, db) || !db.empty()) return 7;
        }
    }
    {
        // the levels come to the host only when asked for: a call without fetchPyramid leaves mvImagePyramid as it
        // was, ImagePyramid() fills it once
        ORB_SLAM2::ORBextractorGpu lazy(500, 1.2f, 8, 20, 7, 0, W, H);
        std::vector<orbx_keypoint> k3;
        std::vector<uint8_t> d3;
        lazy(img.data(), W, H, W, k3, d3);
        if (lazy.mvImagePyramid[1].cols != 0) return 8;
        const auto& pyr = lazy.ImagePyramid();
        if (pyr[1].cols != ex.mvImagePyramid[1].cols || pyr[1].data != ex.mvImagePyramid[1].data) return 8;
        lazy.SetFetchPyramid(true);
        lazy(img.data(), W, H, W, k3, d3);
        if (lazy.mvImagePyramid[2].data != ex.mvImagePyramid[2].data) return 8;
    }
    // empty image: outputs untouched
    std::vector<orbx_keypoint> k2(3);
    std::vector<uint8_t> d2(5);
    ex(nullptr, 0, 0, 0, k2, d2);
    return (k2.size() == 3 && d2.size() == 5) ? 0 : 3;
}
