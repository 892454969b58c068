// The step right after extraction in every Frame constructor: Frame::UndistortKeyPoints
// (src/Frame.cc:442-472) and Frame::AssignFeaturesToGrid (:268-283, PosInGrid :420-430).
//
// cv::undistortPoints(src, dst, K, D, Mat(), K) as of OpenCV 4.0.0 (imgproc undistort.cpp,
// cvUndistortPointsInternal): all in double, normalise with 1/fx, 1/fy, five fixed-point iterations of
// the radial-tangential model (TermCriteria(MAX_ITER, 5, 0.01): the count alone stops it), re-project
// with K, store as float. One workgroup per frame: a thread undistorts its keypoints and computes their
// grid cell; LDS atomics build the per-cell histogram, a scan turns it into cell starts, and every
// keypoint finds its place inside its cell by counting the earlier keypoints of the same cell, so cell
// contents come out in ascending keypoint order like the reference's push_back loop.
#include "orbx_internal.h"

namespace orbx {

namespace {
constexpr int kCells = ORBX_GRID_COLS * ORBX_GRID_ROWS;

template <typename T>
__host__ __device__ inline void undistort_point(const orbx_camera& c, T u, T v, T* xo, T* yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2, k4 = c.k3;   // cv order: k1 k2 p1 p2 k3
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        // k4..k6 of the rational model and the thin-prism terms are zero here: numerator 1, no s1..s4
        const double icdist = 1. / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // RR = P * R with R = I and P = K
    *xo = (T)(fx * x + cx);
    *yo = (T)(fy * y + cy);
}
}  // namespace

void undistort_point_host(const orbx_camera& cam, float u, float v, float* xo, float* yo) {
    undistort_point<float>(cam, u, v, xo, yo);
}

__global__ __launch_bounds__(256) void undistort_grid_kernel(int cap, const orbx_keypoint* __restrict__ kps,
                                                             const int32_t* __restrict__ counts, orbx_camera cam,
                                                             float minX, float minY, float invW, float invH,
                                                             orbx_keypoint* __restrict__ kpsUn,
                                                             int32_t* __restrict__ cellStart,
                                                             int32_t* __restrict__ cellItems) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int* hist = reinterpret_cast<int*>(smem);                     // kCells + 1
    int* part = hist + kCells + 1;                                // 256
    uint16_t* cellOf = reinterpret_cast<uint16_t*>(part + 256);   // cap entries, 0xFFFF = outside the grid
    const int f = blockIdx.x, t = threadIdx.x;
    const int n = min(counts[f], cap);
    const long long o = (long long)f * cap;
    for (int i = t; i <= kCells; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        orbx_keypoint kp = kps[o + i];
        if (cam.k1 != 0.0f) undistort_point<float>(cam, kp.x, kp.y, &kp.x, &kp.y);   // (:444-448)
        kpsUn[o + i] = kp;
        // PosInGrid: round() of a float expression, half away from zero (:422-423)
        const int posX = (int)roundf(__fmul_rn(__fsub_rn(kp.x, minX), invW));
        const int posY = (int)roundf(__fmul_rn(__fsub_rn(kp.y, minY), invH));
        int cell = 0xFFFF;
        if (!(posX < 0 || posX >= ORBX_GRID_COLS || posY < 0 || posY >= ORBX_GRID_ROWS)) {
            cell = posX * ORBX_GRID_ROWS + posY;
            atomicAdd(&hist[cell], 1);
        }
        cellOf[i] = (uint16_t)cell;
    }
    __syncthreads();
    // exclusive scan of the histogram: each thread owns 12 consecutive cells
    constexpr int kPer = kCells / 256;
    int local[kPer], sum = 0;
    for (int k = 0; k < kPer; k++) { local[k] = hist[t * kPer + k]; sum += local[k]; }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int k = 0; k < kPer; k++) { hist[t * kPer + k] = run; run += local[k]; }
    if (t == 255) hist[kCells] = run;
    __syncthreads();
    int32_t* cs = cellStart + (long long)f * (kCells + 1);
    for (int i = t; i <= kCells; i += 256) cs[i] = hist[i];
    for (int i = t; i < n; i += 256) {
        const int cell = cellOf[i];
        if (cell == 0xFFFF) continue;
        int before = 0;
        for (int j = 0; j < i; j++) before += cellOf[j] == cell;
        cellItems[o + hist[cell] + before] = i;
    }
}

hipError_t launch_undistort_grid(int nframes, int cap, const orbx_keypoint* kps, const int32_t* counts,
                                 orbx_camera cam, float minX, float minY, float invW, float invH,
                                 orbx_keypoint* kpsUn, int32_t* cellStart, int32_t* cellItems, hipStream_t s) {
    if (nframes <= 0) return hipSuccess;
    const size_t lds = (size_t)(kCells + 1 + 256) * 4 + (size_t)cap * 2 + 16;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(undistort_grid_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(undistort_grid_kernel, dim3(nframes), dim3(256), lds, s, cap, kps, counts, cam, minX, minY,
                       invW, invH, kpsUn, cellStart, cellItems);
    return hipGetLastError();
}

}  // namespace orbx
