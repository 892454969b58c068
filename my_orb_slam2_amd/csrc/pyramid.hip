// Scale pyramid: level k = bilinear resize of level k-1 (ORBextractor::ComputePyramid,
// src/ORBextractor.cc:1099-1124), in the 11-bit fixed-point arithmetic of OpenCV's 8-bit
// INTER_LINEAR path so every byte equals cv::resize's.
//
// Layout: level 0 is the caller's frame, read in place; levels >= 1 of frame f live in
// fs.pyr + f*pyrBytes + lv[k].offset with a 64-byte-multiple row pitch. The reference's 19-px
// BORDER_REFLECT_101 frame around each level is never read by the extractor (FAST stays 16 px
// inside, the orientation disc has radius 15, rotated pattern points reach 18 px and keypoints
// sit >= 19 px from the edge), so it is not stored; orbx_get_level() rebuilds it on request.
#include "orbx_internal.h"

namespace orbx {

// One thread produces four horizontally adjacent destination pixels and stores them as one dword.
__global__ __launch_bounds__(256) void pyr_resize_kernel(const uint8_t* __restrict__ srcBase,
                                                         long long srcRowStride, long long srcFrameStride,
                                                         int sw, int sh, uint8_t* __restrict__ dstBase,
                                                         long long dstRowStride, long long dstFrameStride,
                                                         int dw, int dh, ResizeTab tab) {
    const int f = blockIdx.x;
    const int x4 = (blockIdx.y * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.z * 4 + threadIdx.y;
    if (x4 >= dw || dy >= dh) return;
    const uint8_t* src = srcBase + (long long)f * srcFrameStride;
    uint8_t* dst = dstBase + (long long)f * dstFrameStride;
    int sy0 = tab.yofs[dy], sy1 = sy0 + 1;
    sy0 = min(max(sy0, 0), sh - 1);
    sy1 = min(max(sy1, 0), sh - 1);
    const int b0 = tab.ibeta[2 * dy], b1 = tab.ibeta[2 * dy + 1];
    const uint8_t* S0 = src + (long long)sy0 * srcRowStride;
    const uint8_t* S1 = src + (long long)sy1 * srcRowStride;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = min(x4 + k, dw - 1);
        const int sx = tab.xofs[dx];
        const int sx1 = min(sx + 1, sw - 1);  // weight is 0 whenever sx+1 would leave the row
        const int a0 = tab.ialpha[2 * dx], a1 = tab.ialpha[2 * dx + 1];
        const int h0 = S0[sx] * a0 + S0[sx1] * a1;
        const int h1 = S1[sx] * a0 + S1[sx1] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)min(max(v, 0), 255) << (8 * k);
    }
    // rows are padded to a multiple of 64 bytes, so the dword store never leaves the row
    *reinterpret_cast<uint32_t*>(dst + (long long)dy * dstRowStride + x4) = packed;
}

// Fast path: same arithmetic, far fewer memory instructions. The four source spans a thread needs
// lie within 12 bytes of (sx[first] & ~3), so each source row costs three aligned dword loads and the
// neighbouring byte pairs are cut out with v_alignbyte; the per-column table entry {sx, a0, a1} is
// one 8-byte record, 32 contiguous bytes per thread. Needs 4-byte aligned rows and a
// source/destination ratio small enough for the 12-byte window.
__global__ __launch_bounds__(256) void pyr_resize_dword_kernel(const uint8_t* __restrict__ srcBase,
                                                               long long srcRowStride, long long srcFrameStride,
                                                               int sw, int sh, uint8_t* __restrict__ dstBase,
                                                               long long dstRowStride, long long dstFrameStride,
                                                               int dw, int dh, ResizeTab tab) {
    const int f = blockIdx.x;                               // frame-fastest grid: one frame per XCD
    const int x4 = (blockIdx.y * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.z * 4 + threadIdx.y;
    if (x4 >= dw || dy >= dh) return;
    const uint8_t* src = srcBase + (long long)f * srcFrameStride;
    uint8_t* dst = dstBase + (long long)f * dstFrameStride;
    int sy0 = tab.yofs[dy], sy1 = sy0 + 1;
    sy0 = min(max(sy0, 0), sh - 1);
    sy1 = min(max(sy1, 0), sh - 1);
    const int b0 = tab.ibeta[2 * dy], b1 = tab.ibeta[2 * dy + 1];
    // the table is padded to a multiple of four entries
    const uint4* xe = reinterpret_cast<const uint4*>(tab.xentry + 2 * x4);
    const uint4 e01 = xe[0], e23 = xe[1];
    const int sx[4] = {(int)e01.x, (int)e01.z, (int)e23.x, (int)e23.z};
    const uint32_t aa[4] = {e01.y, e01.w, e23.y, e23.w};
    const int c0 = sx[0] & ~3;
    const int lastDw = ((sw + 3) >> 2) - 1;
    const int q0 = c0 >> 2;
    const uint32_t* R0 = reinterpret_cast<const uint32_t*>(src + (long long)sy0 * srcRowStride);
    const uint32_t* R1 = reinterpret_cast<const uint32_t*>(src + (long long)sy1 * srcRowStride);
    const int i0 = min(q0, lastDw), i1 = min(q0 + 1, lastDw), i2 = min(q0 + 2, lastDw);
    const uint32_t u0 = R0[i0], u1 = R0[i1], u2 = R0[i2];
    const uint32_t v0 = R1[i0], v1 = R1[i1], v2 = R1[i2];
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int o = sx[k] - c0;          // 0..10
        const int q = o >> 2;
        const uint32_t ulo = q == 0 ? u0 : (q == 1 ? u1 : u2), uhi = q == 0 ? u1 : u2;
        const uint32_t vlo = q == 0 ? v0 : (q == 1 ? v1 : v2), vhi = q == 0 ? v1 : v2;
        const uint32_t pu = __builtin_amdgcn_alignbyte(uhi, ulo, (uint32_t)(o & 3));   // S0[sx], S0[sx+1] in bytes 0,1
        const uint32_t pv = __builtin_amdgcn_alignbyte(vhi, vlo, (uint32_t)(o & 3));
        const int a0 = (int)(int16_t)(aa[k] & 0xFFFFu), a1 = (int)(int16_t)(aa[k] >> 16);
        const int h0 = (int)(pu & 0xFFu) * a0 + (int)((pu >> 8) & 0xFFu) * a1;
        const int h1 = (int)(pv & 0xFFu) * a0 + (int)((pv >> 8) & 0xFFu) * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)min(max(v, 0), 255) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + (long long)dy * dstRowStride + x4) = packed;
}

hipError_t launch_pyramid(const FrameSet& fs, const PyramidGeom& g, const ResizeTab* tabs, int nframes,
                          hipStream_t s) {
    for (int l = 1; l < g.nlevels; l++) {
        const LevelGeom& D = g.lv[l];
        const LevelGeom& S = g.lv[l - 1];
        if (D.w < 1 || D.h < 1 || S.w < 1 || S.h < 1) continue;
        const uint8_t* src;
        long long srs, sfs;
        if (l == 1) {
            src = fs.img;
            srs = fs.imgRowStride;
            sfs = fs.imgFrameStride;
        } else {
            src = fs.pyr + S.offset;
            srs = S.stride;
            sfs = g.pyrBytes;
        }
        dim3 block(64, 4, 1);
        dim3 grid(nframes, (D.w + 255) / 256, (D.h + 3) / 4);
        const bool aligned = ((uintptr_t)src % 4 == 0) && (srs % 4 == 0) && (sfs % 4 == 0);
        // 12-byte window: 3 (alignment) + largest step of three columns + 1 neighbour <= 11
        const bool narrow = 3.0 * S.w / D.w + 5.0 <= 11.0;
        if (aligned && narrow)
            hipLaunchKernelGGL(pyr_resize_dword_kernel, grid, block, 0, s, src, srs, sfs, S.w, S.h,
                               fs.pyr + D.offset, (long long)D.stride, g.pyrBytes, D.w, D.h, tabs[l]);
        else
            hipLaunchKernelGGL(pyr_resize_kernel, grid, block, 0, s, src, srs, sfs, S.w, S.h,
                               fs.pyr + D.offset, (long long)D.stride, g.pyrBytes, D.w, D.h, tabs[l]);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace orbx
