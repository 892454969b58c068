// Orientation, blur and steered BRIEF for the selected keypoints, one 64-lane wave per keypoint
// (IC_Angle / computeOrientation src/ORBextractor.cc:69-96,464-471; the per-level
// GaussianBlur :1077-1078; computeOrbDescriptor :99-139; final scaling and concatenation
// :1067-1096).
//
// The reference blurs every whole level and then samples 512 pixels per keypoint. Here the wave
// stages the 43x43 raw patch around its keypoint in LDS (radius 15 for the moments, 18 for the
// rotated pattern, +3 for the 7x7 taps) and evaluates the fixed-point Gaussian only at the 512
// sampled positions; the separable 8.8 fixed-point filter has no intermediate rounding
// (u16 row sums, 16.16 column sums), so a direct 7x7 evaluation gives the same byte. No blurred
// level is ever written to HBM. orbx_debug_blur_level() runs the same arithmetic over a full
// level for parity tests.
#include "orbx_internal.h"

namespace orbx {

namespace {

__device__ const int8_t c_pattern[1024] = {
#include "../../include/orbx_pattern31.inc"
};

__constant__ int c_umax[kHalfPatch + 1] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

constexpr int kPR = 21;               // patch radius
constexpr int kPW = 2 * kPR + 1;      // 43
constexpr int kPitch = 48;            // LDS row pitch of the patch (bytes)
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ int reflect101(int p, int len) {
    p = p < 0 ? -p : p;
    return p >= len ? 2 * (len - 1) - p : p;
}

// cv::fastAtan2 (core mathfuncs: atan_f32), float32 with every product and sum rounded separately.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float k = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// 7x7 fixed-point Gaussian at patch position (px, py): cv fixedSmoothInvoker<uint8_t, ufixedpoint16>
__device__ __forceinline__ int blur_at(const uint8_t* patch, int px, int py, const int* K) {
    const uint8_t* p = patch + (py - 3) * kPitch + (px - 3);
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        unsigned h = 0;
#pragma unroll
        for (int c = 0; c < 7; c++) h += (unsigned)K[c] * p[r * kPitch + c];
        acc += (unsigned)K[r] * min(h, 0xFFFFu);
    }
    return (int)min((acc + 32768u) >> 16, 255u);
}

}  // namespace

__global__ __launch_bounds__(64 * kWavesPerBlock) void orient_desc_kernel(FrameSet fs, PyramidGeom g,
                                                                           int blurTaps,
                                                                           orbx_keypoint* __restrict__ kpsOut,
                                                                           uint8_t* __restrict__ descOut, int cap,
                                                                           int32_t* __restrict__ countsOut) {
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[kWavesPerBlock][kPW * kPitch];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = blockIdx.y;
    const int slot = blockIdx.x * kWavesPerBlock + wv;    // index into the frame's selected regions
    // No wave leaves before the barrier below; inactive waves just skip the work.
    bool active = slot < g.totalSel;
    int l = 0, k = 0, outIdx = 0, kx = 0, ky = 0;
    uint32_t packed = 0;
    const uint8_t* img = nullptr;
    long long rowStride = 0;
    if (active) {
        while (l + 1 < g.nlevels && slot >= g.lv[l + 1].selBase) l++;
        k = slot - g.lv[l].selBase;
        const int32_t* selCount = fs.selCount + (long long)f * g.nlevels;
        // position in the frame's output: levels are concatenated in order (:1067-1096)
        int before = 0, total = 0;
        for (int i = 0; i < g.nlevels; i++) {
            const int c = selCount[i];
            before += i < l ? c : 0;
            total += c;
        }
        if (slot == 0 && lane == 0) countsOut[f] = total;
        outIdx = before + k;
        active = k < selCount[l] && outIdx < cap;
    }
    const LevelGeom& L = g.lv[l];
    uint8_t* patch = s_patch[wv];
    if (active) {
        packed = fs.sel[(long long)f * g.totalSel + slot];
        kx = cand_x(packed) + kBorder;   // level coordinates (:835-836)
        ky = cand_y(packed) + kBorder;
        if (l == 0) {
            img = fs.img + (long long)f * fs.imgFrameStride;
            rowStride = fs.imgRowStride;
        } else {
            img = fs.pyr + (long long)f * g.pyrBytes + L.offset;
            rowStride = L.stride;
        }
        // stage the raw patch; coordinates outside the level are reflected (BORDER_REFLECT_101)
        for (int i = lane; i < kPW * kPW; i += 64) {
            const int r = i / kPW, c = i - r * kPW;
            const int sy = reflect101(ky - kPR + r, L.h), sx = reflect101(kx - kPR + c, L.w);
            patch[r * kPitch + c] = img[(long long)sy * rowStride + sx];
        }
    }
    __syncthreads();
    if (!active) return;

    // ---- intensity centroid on the unblurred patch (:69-96) ----
    int m10 = 0, m01 = 0;
    for (int i = lane; i < kPatchSize * kPatchSize; i += 64) {
        const int v = i / kPatchSize - kHalfPatch, u = i - (v + kHalfPatch) * kPatchSize - kHalfPatch;
        if (abs(u) <= c_umax[abs(v)]) {
            const int val = patch[(kPR + v) * kPitch + kPR + u];
            m10 += u * val;
            m01 += v * val;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off);
        m01 += __shfl_xor(m01, off);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- steered BRIEF on the blurred patch (:99-139) ----
    const int K0[7] = {18, 34, 49, 55, 49, 34, 18};
    const int K1[7] = {18, 34, 48, 56, 48, 34, 18};
    int K[7];
#pragma unroll
    for (int i = 0; i < 7; i++) K[i] = blurTaps == ORBX_BLUR_CV_4_0_0 ? K0[i] : K1[i];
    const float rad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.f));
    const float a = (float)cos((double)rad), b = (float)sin((double)rad);
    unsigned long long* desc64 = reinterpret_cast<unsigned long long*>(descOut + ((long long)f * cap + outIdx) * 32);
#pragma unroll 1
    for (int round = 0; round < 4; round++) {
        const int pair = round * 64 + lane;
        const int8_t* pt = c_pattern + 4 * pair;
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = blur_at(patch, kPR + c0, kPR + r0, K);
        const int t1 = blur_at(patch, kPR + c1, kPR + r1, K);
        const unsigned long long bits = __ballot(t0 < t1);
        if (lane == 0) desc64[round] = bits;   // pair 64*round + j is bit j%8 of byte 8*round + j/8
    }
    if (lane == 0) {
        orbx_keypoint kp;
        // coordinates are scaled after the descriptors, and only above level 0 (:1087-1093)
        kp.x = l ? __fmul_rn((float)kx, L.scale) : (float)kx;
        kp.y = l ? __fmul_rn((float)ky, L.scale) : (float)ky;
        kp.size = L.kpSize;
        kp.angle = angle;
        kp.response = (float)cand_score(packed);
        kp.octave = l;
        kp.class_id = -1;
        kpsOut[(long long)f * cap + outIdx] = kp;
    }
}

// Full-level blur with the same arithmetic (parity probe for src/ORBextractor.cc:1078).
__global__ __launch_bounds__(256) void blur_level_kernel(const uint8_t* __restrict__ src, int w, int h,
                                                         long long stride, uint8_t* __restrict__ dst,
                                                         long long dstStride, int blurTaps) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int K0[7] = {18, 34, 49, 55, 49, 34, 18};
    const int K1[7] = {18, 34, 48, 56, 48, 34, 18};
    unsigned acc = 0;
    for (int r = 0; r < 7; r++) {
        const uint8_t* row = src + (long long)reflect101(y + r - 3, h) * stride;
        unsigned hsum = 0;
        for (int c = 0; c < 7; c++)
            hsum += (unsigned)(blurTaps == ORBX_BLUR_CV_4_0_0 ? K0[c] : K1[c]) * row[reflect101(x + c - 3, w)];
        acc += (unsigned)(blurTaps == ORBX_BLUR_CV_4_0_0 ? K0[r] : K1[r]) * min(hsum, 0xFFFFu);
    }
    dst[(long long)y * dstStride + x] = (uint8_t)min((acc + 32768u) >> 16, 255u);
}

hipError_t launch_orient_desc(const FrameSet& fs, const PyramidGeom& g, int nframes, int blurTaps,
                              orbx_keypoint* kps, uint8_t* desc, int cap, int32_t* counts, hipStream_t s) {
    dim3 grid((g.totalSel + kWavesPerBlock - 1) / kWavesPerBlock, nframes, 1), block(64 * kWavesPerBlock, 1, 1);
    hipLaunchKernelGGL(orient_desc_kernel, grid, block, 0, s, fs, g, blurTaps, kps, desc, cap, counts);
    return hipGetLastError();
}

hipError_t launch_blur_level(const uint8_t* src, int w, int h, long long stride, uint8_t* dst,
                             long long dstStride, int blurTaps, hipStream_t s) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(blur_level_kernel, grid, block, 0, s, src, w, h, stride, dst, dstStride, blurTaps);
    return hipGetLastError();
}

}  // namespace orbx
