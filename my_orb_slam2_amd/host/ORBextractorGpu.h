// ORBextractorGpu.h — C++ host shim over the C ABI of include/orbx.h.
//
// Mirrors ORB_SLAM2::ORBextractor (my_orb_slam2/include/ORBextractor.h:46-126): same constructor
// arguments, same getters, same call operator, a public mvImagePyramid.  Frame/Tracking code keeps
// its shape; only the class name in Tracking.cc:116-122 and Frame.h changes.  Header-only, needs
// nothing but liborbx.so.  Compile with -DORBX_WITH_OPENCV to get the cv::InputArray overload the
// reference's Frame::ExtractORB (src/Frame.cc:285-291) calls.
#pragma once

#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"

#ifdef ORBX_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace ORB_SLAM2 {

// One pyramid level as plain memory (stands in for the cv::Mat views of mvImagePyramid).
struct PyramidLevel {
    int cols = 0, rows = 0;
    size_t step = 0;                 // bytes per row of `data`
    std::vector<uint8_t> data;       // (rows + 38) x (cols + 38) with the 19-px reflect-101 border
    // pointer to pixel (0,0) of the level itself, like mvImagePyramid[l].data
    const uint8_t* ptr() const { return data.data() + 19 * step + 19; }
};

class ORBextractorGpu {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    ORBextractorGpu(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                    int device = 0, int maxWidth = 1920, int maxHeight = 1080,
                    int blurTaps = ORBX_BLUR_CV_4_0_0) {
        orbx_config cfg{};
        cfg.nfeatures = nfeatures;
        cfg.scale_factor = scaleFactor;
        cfg.nlevels = nlevels;
        cfg.ini_th_fast = iniThFAST;
        cfg.min_th_fast = minThFAST;
        cfg.device = device;
        cfg.max_width = maxWidth;
        cfg.max_height = maxHeight;
        cfg.max_batch = 1;
        cfg.blur_taps = blurTaps;
        if (orbx_create(&cfg, &ctx_) != ORBX_OK)
            throw std::runtime_error(std::string("orbx_create: ") + orbx_last_error());
        nlevels_ = nlevels;
        cap_ = orbx_keypoint_capacity(ctx_);
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractorGpu() { orbx_destroy(ctx_); }
    ORBextractorGpu(const ORBextractorGpu&) = delete;
    ORBextractorGpu& operator=(const ORBextractorGpu&) = delete;

    // operator()(image, mask, keypoints, descriptors) on plain memory; mask is ignored as in the
    // reference. descriptors receives keypoints.size() rows of 32 bytes. An empty image leaves both
    // outputs untouched (src/ORBextractor.cc:1038-1039). fetchPyramid fills mvImagePyramid at once (stereo
    // matching on the host reads it, src/Frame.cc:510,600-617); without it the eight levels stay on the device
    // until ImagePyramid() is called, which mono frames (src/Frame.cc:214-240) never do.
    void operator()(const uint8_t* image, int cols, int rows, size_t step,
                    std::vector<orbx_keypoint>& keypoints, std::vector<uint8_t>& descriptors,
                    bool fetchPyramid = false) {
        if (!image || cols <= 0 || rows <= 0) return;
        pyramidFresh_ = false;
        keypoints.resize(cap_);
        descriptors.resize(static_cast<size_t>(cap_) * ORBX_DESC_BYTES);
        int n = 0;
        const int rc = orbx_extract(ctx_, image, cols, rows, step, keypoints.data(), descriptors.data(), cap_, &n);
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error());
        keypoints.resize(n);
        descriptors.resize(static_cast<size_t>(n) * ORBX_DESC_BYTES);
        if (fetchPyramid || fetchAlways_) FetchPyramid();
    }

    // Stereo on the host (the reference's own Frame::ComputeStereoMatches) reads mvImagePyramid after every call:
    // SetFetchPyramid(true) on both extractors of a stereo system restores that behaviour for both overloads.
    // Callers of orbx_compute_stereo_matches_device never need the levels on the host.
    void SetFetchPyramid(bool always) { fetchAlways_ = always; }

    // mvImagePyramid of the last call, copied from the device on first use.
    const std::vector<PyramidLevel>& ImagePyramid() {
        if (!pyramidFresh_) FetchPyramid();
        return mvImagePyramid;
    }

#ifdef ORBX_WITH_OPENCV
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "orbx_keypoint must match cv::KeyPoint");
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                    cv::OutputArray _descriptors) {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        std::vector<orbx_keypoint> k;
        std::vector<uint8_t> d;
        (*this)(image.data, image.cols, image.rows, image.step, k, d, false);   // levels on demand: ImagePyramid()
        _keypoints.resize(k.size());
        if (!k.empty()) std::memcpy(static_cast<void*>(_keypoints.data()), k.data(), k.size() * sizeof(orbx_keypoint));
        if (k.empty()) { _descriptors.release(); return; }
        _descriptors.create(static_cast<int>(k.size()), 32, CV_8U);
        std::memcpy(_descriptors.getMat().data, d.data(), d.size());
    }
#endif

    void FetchPyramid() {
        for (int l = 0; l < nlevels_; l++) {
            PyramidLevel& p = mvImagePyramid[l];
            if (orbx_level_size(ctx_, l, &p.cols, &p.rows) != ORBX_OK || p.cols < 1) continue;
            p.step = static_cast<size_t>(p.cols) + 38;
            p.data.resize(p.step * (static_cast<size_t>(p.rows) + 38));
            if (orbx_get_level(ctx_, 0, l, 1, p.data.data(), p.step) != ORBX_OK)
                throw std::runtime_error(std::string("orbx_get_level: ") + orbx_last_error());
        }
        pyramidFresh_ = true;
    }

    int GetLevels() { return orbx_get_levels(ctx_); }
    float GetScaleFactor() { return orbx_get_scale_factor(ctx_); }
    std::vector<float> GetScaleFactors() { return floats(orbx_get_scale_factors); }
    std::vector<float> GetInverseScaleFactors() { return floats(orbx_get_inverse_scale_factors); }
    std::vector<float> GetScaleSigmaSquares() { return floats(orbx_get_scale_sigma_squares); }
    std::vector<float> GetInverseScaleSigmaSquares() { return floats(orbx_get_inverse_scale_sigma_squares); }

    std::vector<PyramidLevel> mvImagePyramid;
    orbx_ctx* context() { return ctx_; }

private:
    std::vector<float> floats(int (*fn)(const orbx_ctx*, float*)) {
        std::vector<float> v(nlevels_);
        fn(ctx_, v.data());
        return v;
    }
    orbx_ctx* ctx_ = nullptr;
    int nlevels_ = 0, cap_ = 0;
    bool fetchAlways_ = false, pyramidFresh_ = false;
};

// ORBmatcher::DescriptorDistance on two 32-byte descriptor rows (src/ORBmatcher.cc:1698-1714).
inline int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return orbx_hamming256(a, b); }

// Keypoint / descriptor block of the reference's binary map file (src/KeyFrame.cc:934-942 written through
// DUtils/IoUtils.cpp:4-33): a uint64 count followed by raw 28-byte cv::KeyPoint records, and a matrix as
// {rows, cols, type} uint32 header (type 0 = CV_8UC1) followed by rows*cols bytes. orbx_keypoint has the
// cv::KeyPoint layout, so maps written by either side load on the other.
inline void WriteKeyPoints(std::ostream& f, const std::vector<orbx_keypoint>& kps) {
    const uint64_t size = kps.size();
    f.write(reinterpret_cast<const char*>(&size), sizeof(size));
    f.write(reinterpret_cast<const char*>(kps.data()), static_cast<std::streamsize>(size * sizeof(orbx_keypoint)));
}
// Bytes left between the read position and the end of a seekable stream; "unknown" (max) for others.
inline uint64_t StreamBytesLeft(std::istream& f) {
    const std::istream::pos_type here = f.tellg();
    if (here == std::istream::pos_type(-1)) return UINT64_MAX;
    f.seekg(0, std::ios::end);
    const std::istream::pos_type end = f.tellg();
    f.seekg(here);
    return end >= here ? static_cast<uint64_t>(end - here) : 0;
}
// Returns false (and leaves kps empty) on a short read or a count the rest of the file cannot hold.
inline bool ReadKeyPoints(std::istream& f, std::vector<orbx_keypoint>& kps) {
    kps.clear();
    uint64_t size = 0;
    f.read(reinterpret_cast<char*>(&size), sizeof(size));
    if (!f.good() || size > StreamBytesLeft(f) / sizeof(orbx_keypoint)) return false;
    kps.resize(size);
    f.read(reinterpret_cast<char*>(kps.data()), static_cast<std::streamsize>(size * sizeof(orbx_keypoint)));
    if (!f.good()) kps.clear();
    return f.good();
}
inline void WriteDescriptors(std::ostream& f, const std::vector<uint8_t>& desc) {
    const uint32_t header[3] = {static_cast<uint32_t>(desc.size() / ORBX_DESC_BYTES), ORBX_DESC_BYTES, 0u};
    f.write(reinterpret_cast<const char*>(header), sizeof(header));
    f.write(reinterpret_cast<const char*>(desc.data()), static_cast<std::streamsize>(desc.size()));
}
// Returns false (and leaves desc empty) on a short read or a header that is not rows x 32 of CV_8UC1. An empty
// matrix is written as {0, 0, 0} by the reference (a released cv::Mat has no columns either) and reads back as no
// descriptors.
inline bool ReadDescriptors(std::istream& f, std::vector<uint8_t>& desc) {
    desc.clear();
    uint32_t header[3] = {0, 0, 0};
    f.read(reinterpret_cast<char*>(header), sizeof(header));
    if (f.good() && header[0] == 0u && header[2] == 0u) return true;
    if (!f.good() || header[1] != ORBX_DESC_BYTES || header[2] != 0u ||
        header[0] > StreamBytesLeft(f) / ORBX_DESC_BYTES)
        return false;
    desc.resize(static_cast<size_t>(header[0]) * ORBX_DESC_BYTES);
    f.read(reinterpret_cast<char*>(desc.data()), static_cast<std::streamsize>(desc.size()));
    if (!f.good()) desc.clear();
    return f.good();
}

}  // namespace ORB_SLAM2
