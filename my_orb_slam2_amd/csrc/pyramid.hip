// Scale pyramid: level k = bilinear resize of level k-1 (ORBextractor::ComputePyramid,
// src/ORBextractor.cc:1099-1124), in the 11-bit fixed-point arithmetic of OpenCV's 8-bit
// INTER_LINEAR path so every byte equals cv::resize's.
//
// Layout: level 0 is the caller's frame, read in place; levels >= 1 of frame f live in
// fs.pyr + f*pyrBytes + lv[k].offset with a 64-byte-multiple row pitch. The reference's 19-px
// BORDER_REFLECT_101 frame around each level is never read by the extractor (FAST stays 16 px
// inside, the orientation disc has radius 15, rotated pattern points reach 18 px and keypoints
// sit >= 19 px from the edge), so it is not stored; orbx_get_level() rebuilds it on request.
#include <cstdio>

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

// Fast path: same arithmetic, far fewer instructions. A thread makes four destination pixels per row for
// kRowsPerThread consecutive rows; a wave (threadIdx.y) owns whole rows, so everything that depends only
// on the row is scalar. Per source row: three dword loads through a buffer resource (the 12 bytes around the
// first source column: scalar row offset + per-lane constant offsets), two v_alignbyte to bring that column to
// byte 0 of an 8-byte window, and per pixel one v_perm_b32 that puts S[sx], S[sx+1] into the two 16-bit halves
// plus one v_dot2_u32_u16 with the packed weights (a0, a1). The vertical pass is two v_mul_hi_u32 per pixel with
// the row weights pre-shifted by 16. Consecutive destination rows share a source row most of the time (the lower
// row of one is the upper row of the next); its horizontal sums are kept in registers. The per-column
// table record is 8 bytes, {sx for the first column of a group of four, else the v_perm selector;
// a0 | a1 << 16}, built on the host and loaded once per thread. Needs 4-byte aligned rows and a
// source/destination ratio <= 2.
constexpr int kRowsPerThread = 8;
// A horizontal sum S[sx]*a0 + S[sx+1]*a1 is at most 255 * 2048 < 2^19; its low four bits are dropped.
constexpr uint32_t kHsumMask = 0x0007FFF0u;

// Bits 32..47 of the product of two values below 2^24 (v_mul_hi_u32_u24, full rate; v_mul_hi_u32 is quarter rate).
__device__ __forceinline__ uint32_t mulhi24(uint32_t a, uint32_t b) {
    return __umulhi(a & 0x00FFFFFFu, b & 0x00FFFFFFu);      // the masks tell the compiler which instruction fits
}

__global__ __launch_bounds__(256) void pyr_resize_dword_kernel(const uint8_t* __restrict__ srcBase,
                                                               long long srcRowStride, long long srcFrameStride,
                                                               int sw, int sh, uint8_t* __restrict__ dstBase,
                                                               long long dstRowStride, long long dstFrameStride,
                                                               int dw, int dh, ResizeTab tab) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const int f = blockIdx.x;                               // frame-fastest grid: one frame per XCD
    const int x4 = (blockIdx.y * 64 + threadIdx.x) * 4;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int dyBegin = (blockIdx.z * 4 + wy) * kRowsPerThread;
    if (x4 >= dw || dyBegin >= dh) return;
    const uint8_t* src = srcBase + (long long)f * srcFrameStride;
    uint8_t* dst = dstBase + (long long)f * dstFrameStride + x4;
    // the table is padded to a multiple of four columns
    const uint4* xe = reinterpret_cast<const uint4*>(tab.xentry + 2 * x4);
    const uint4 e01 = xe[0], e23 = xe[1];
    const int sx0 = (int)e01.x;
    const uint32_t sel[4] = {0x0c010c00u, e01.z, e23.x, e23.z};
    const uint32_t aa[4] = {e01.y, e01.w, e23.y, e23.w};
    const int lastDw = ((sw + 3) >> 2) - 1;
    const int q0 = sx0 >> 2;
    const uint32_t sh8 = (uint32_t)(sx0 & 3);
    // Source rows are read through a buffer resource: the row offset is scalar, the three dword offsets inside
    // the row are this lane's constants, so a load needs no address arithmetic at all.
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (int)((long long)sh * srcRowStride), 0x00027000);
    const int o0 = 4 * min(q0, lastDw), o1 = 4 * min(q0 + 1, lastDw), o2 = 4 * min(q0 + 2, lastDw);

    // horizontal pass of one source row for this thread's four columns: (S[sx]*a0 + S[sx+1]*a1) >> 4, kept
    // shifted left by 4 (the low four bits cleared) for the 24-bit multiply of the vertical pass.
    // The weights are never negative (they are (1-f)*2048 and f*2048), so the unsigned dot product is exact.
    auto hrow = [&](int sy, uint32_t (&h)[4]) {
        const int rowOff = sy * (int)srcRowStride;                          // wave-uniform; a level is < 2^31 bytes
        const uint32_t u0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o0, rowOff, 0),
                       u1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o1, rowOff, 0),
                       u2 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o2, rowOff, 0);
        // 8-byte window starting at source column sx0 (bytes past the row end only meet zero weights)
        const uint32_t w0 = __builtin_amdgcn_alignbyte(u1, u0, sh8), w1 = __builtin_amdgcn_alignbyte(u2, u1, sh8);
#pragma unroll
        for (int k = 0; k < 4; k++)
            h[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(w1, w0, sel[k])),
                                          __builtin_bit_cast(us2, aa[k]), 0u, false) & kHsumMask;
    };
    // vertical pass + store of one destination row from the sums of its two source rows
    auto vrow = [&](int dy, uint32_t bb, const uint32_t (&a)[4], const uint32_t (&b)[4]) {
        const uint32_t b0 = (bb << 12) & 0x00FFF000u, b1 = (bb >> 4) & 0x00FFF000u;   // weights << 12 (both <= 2048)
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            // ((b0 * h0) >> 16) + ((b1 * h1) >> 16) + 2 with h = sum >> 4: (b << 12) * (h << 4) is b * h << 16 and
            // both factors stay below 2^24, so bits 32.. of the full-rate 24-bit product are that shifted product.
            // The sum >> 2 never leaves [0, 255] (a0 + a1, b0 + b1 <= 2049, weights >= 0), so v < 1024.
            v[k] = mulhi24(b0, a[k]) + mulhi24(b1, b[k]) + 2u;
        // two sums per dword, one shift for both, then bytes 0 and 2 of each (byte 1 holds the shifted-out bits)
        const uint32_t p01 = (v[0] | (v[1] << 16)) >> 2, p23 = (v[2] | (v[3] << 16)) >> 2;
        *reinterpret_cast<uint32_t*>(dst + (long long)dy * dstRowStride) = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
    };

    // Consecutive destination rows usually share a source row (the lower row of one is the upper row of the
    // next). The two sets of sums swap roles from row to row (the loop is unrolled by two with the names
    // exchanged), so a shared row is never copied between registers.
    int cachedSy = -1;       // source row whose sums sit in the set that is `up` next
    uint32_t hA[4] = {0, 0, 0, 0}, hB[4] = {0, 0, 0, 0};
    const int dyEnd = min(dyBegin + kRowsPerThread, dh);
    const uint32_t* beta2 = reinterpret_cast<const uint32_t*>(tab.ibeta);   // (b0, b1) of a row as one dword
    auto row = [&](int dy, uint32_t (&up)[4], uint32_t (&lo)[4]) {
        int sy0 = tab.yofs[dy], sy1 = sy0 + 1;
        sy0 = min(max(sy0, 0), sh - 1);
        sy1 = min(max(sy1, 0), sh - 1);
        const uint32_t bb = beta2[dy];
        if (sy0 != cachedSy) hrow(sy0, up);
        hrow(sy1, lo);       // (sy1 == sy0 only on rows clamped at the bottom edge: read again rather than copied)
        cachedSy = sy1;
        vrow(dy, bb, up, lo);
    };
    for (int dy = dyBegin; dy < dyEnd; dy += 2) {
        row(dy, hA, hB);
        if (dy + 1 < dyEnd) row(dy + 1, hB, hA);
    }
}

// Copies frames of any base address and strides into the 4-byte-aligned layout every other kernel's fast path
// wants (dword loads): one thread moves four pixels. Bytes past the row's end repeat its last pixel.
__global__ __launch_bounds__(256) void repack_frames_kernel(const uint8_t* __restrict__ src, long long srcRowStride,
                                                            long long srcFrameStride, uint8_t* __restrict__ dst,
                                                            long long dstRowStride, long long dstFrameStride, int w) {
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= w) return;
    const uint8_t* S = src + (long long)blockIdx.z * srcFrameStride + (long long)blockIdx.y * srcRowStride;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (uint32_t)S[min(x4 + k, w - 1)] << (8 * k);
    *reinterpret_cast<uint32_t*>(dst + (long long)blockIdx.z * dstFrameStride + (long long)blockIdx.y * dstRowStride + x4) = v;
}

hipError_t launch_repack_frames(const uint8_t* src, long long srcRowStride, long long srcFrameStride, uint8_t* dst,
                                long long dstRowStride, long long dstFrameStride, int w, int h, int nframes,
                                hipStream_t s) {
    dim3 grid((w + 1023) / 1024, h, nframes), block(256, 1, 1);
    hipLaunchKernelGGL(repack_frames_kernel, grid, block, 0, s, src, srcRowStride, srcFrameStride, dst, dstRowStride,
                       dstFrameStride, w);
    return hipGetLastError();
}

// ---- several levels per launch ----
// The arithmetic of one strip of destination rows for four adjacent columns, exactly as in
// pyr_resize_dword_kernel; only where a source row comes from differs: FROM_LDS = false reads the group's source
// level through a buffer resource, true reads the tile of the level above that this workgroup has just made.
template <bool FROM_LDS, typename Store>
__device__ __forceinline__ void resize_strip(const FusedLevel& Lv, int x4, int dyBegin, int dyEnd,
                                             __amdgpu_buffer_rsrc_t rsrc, int srcRowStride, const uint32_t* in32,
                                             int inPitch4, int inC0, int inR0, Store store) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint4* xe = reinterpret_cast<const uint4*>(Lv.xentry + 2 * x4);
    const uint4 e01 = xe[0], e23 = xe[1];
    const int sx0 = (int)e01.x;
    const uint32_t sel[4] = {0x0c010c00u, e01.z, e23.x, e23.z};
    const uint32_t aa[4] = {e01.y, e01.w, e23.y, e23.w};
    const uint32_t sh8 = (uint32_t)(sx0 & 3);
    const int q0 = sx0 >> 2;
    const int lastDw = ((Lv.sw + 3) >> 2) - 1;
    const int o0 = 4 * min(q0, lastDw), o1 = 4 * min(q0 + 1, lastDw), o2 = 4 * min(q0 + 2, lastDw);
    const int qrel = q0 - (inC0 >> 2);
    auto hrow = [&](int sy, uint32_t (&h)[4]) {
        uint32_t u0, u1, u2;
        if (FROM_LDS) {
            const uint32_t* r = in32 + (sy - inR0) * inPitch4 + qrel;
            u0 = r[0]; u1 = r[1]; u2 = r[2];
        } else {
            const int rowOff = sy * srcRowStride;
            u0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o0 + rowOff, 0, 0);
            u1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o1 + rowOff, 0, 0);
            u2 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o2 + rowOff, 0, 0);
        }
        const uint32_t w0 = __builtin_amdgcn_alignbyte(u1, u0, sh8), w1 = __builtin_amdgcn_alignbyte(u2, u1, sh8);
#pragma unroll
        for (int k = 0; k < 4; k++)
            h[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(w1, w0, sel[k])),
                                          __builtin_bit_cast(us2, aa[k]), 0u, false) & kHsumMask;
    };
    int cachedSy = -1;
    uint32_t hA[4] = {0, 0, 0, 0}, hB[4] = {0, 0, 0, 0};
    auto row = [&](int dy, uint32_t (&up)[4], uint32_t (&lo)[4]) {
        int sy0 = Lv.yofs[dy], sy1 = sy0 + 1;
        sy0 = min(max(sy0, 0), Lv.sh - 1);
        sy1 = min(max(sy1, 0), Lv.sh - 1);
        const uint32_t bb = Lv.beta2[dy];
        if (sy0 != cachedSy) hrow(sy0, up);
        hrow(sy1, lo);
        cachedSy = sy1;
        const uint32_t b0 = (bb << 12) & 0x00FFF000u, b1 = (bb >> 4) & 0x00FFF000u;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = mulhi24(b0, up[k]) + mulhi24(b1, lo[k]) + 2u;
        const uint32_t p01 = (v[0] | (v[1] << 16)) >> 2, p23 = (v[2] | (v[3] << 16)) >> 2;
        store(dy, __builtin_amdgcn_perm(p23, p01, 0x06040200u));
    };
    for (int dy = dyBegin; dy < dyEnd; dy += 2) {
        row(dy, hA, hB);
        if (dy + 1 < dyEnd) row(dy + 1, hB, hA);
    }
}

__global__ __launch_bounds__(256) void pyr_fused_kernel(FusedGroup G, uint8_t* __restrict__ pyr, long long pyrBytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int f = blockIdx.x;                                  // frame-fastest grid: one frame per XCD
    const int ty = blockIdx.y / G.ntx, tx = blockIdx.y - ty * G.ntx;
    const int t = threadIdx.x;
    uint8_t* frameDst = pyr + (long long)f * pyrBytes;
    const uint8_t* src = G.src + (long long)f * G.srcFrameStride;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(src), 0, (int)((long long)G.lv[0].sh * G.srcRowStride), 0x00027000);
    int inC0 = 0, inR0 = 0, inPitch4 = 0;
    for (int j = 0; j < G.nl; j++) {
        const FusedLevel& Lv = G.lv[j];
        const FusedRange cr = G.cols[j * G.ntx + tx], rr = G.rows[j * G.nty + ty];
        const int ncg = (cr.hi - cr.lo) >> 2, nr = rr.hi - rr.lo;
        // a power-of-two number of lanes per row of column groups; the other lanes' bits pick the strip of rows
        int lg = 0;
        while ((1 << lg) < ncg) lg++;
        const int cg = t & ((1 << lg) - 1), strip = t >> lg;
        const int rps = (nr + (256 >> lg) - 1) >> (8 - lg);
        const int dyBegin = rr.lo + strip * rps, dyEnd = min(dyBegin + rps, rr.hi);
        const int outPitch4 = ncg + 2;                          // two spare dwords: the 12-byte window of the next level
        uint32_t* out32 = reinterpret_cast<uint32_t*>(lds + (j & 1) * G.bufBytes);
        const uint32_t* in32 = reinterpret_cast<const uint32_t*>(lds + ((j + 1) & 1) * G.bufBytes);
        if (cg < ncg && dyBegin < dyEnd) {
            const int x4 = cr.lo + 4 * cg;
            const bool keep = j + 1 < G.nl, ownCol = x4 >= cr.ownLo && x4 < cr.ownHi;
            uint8_t* dst = frameDst + Lv.dstOffset + x4;
            auto store = [&](int dy, uint32_t px) {
                if (keep) out32[(dy - rr.lo) * outPitch4 + cg] = px;
                if (ownCol && dy >= rr.ownLo && dy < rr.ownHi)
                    *reinterpret_cast<uint32_t*>(dst + (long long)dy * Lv.dstStride) = px;
            };
            if (j == 0)
                resize_strip<false>(Lv, x4, dyBegin, dyEnd, rsrc, (int)G.srcRowStride, nullptr, 0, 0, 0, store);
            else
                resize_strip<true>(Lv, x4, dyBegin, dyEnd, rsrc, 0, in32, inPitch4, inC0, inR0, store);
        }
        inC0 = cr.lo;
        inR0 = rr.lo;
        inPitch4 = outPitch4;
        __syncthreads();
    }
}

namespace {
// Ranges of one dimension of a group: tiles of `tile` units on the last level, and for every level above it what
// each tile has to compute (the sources of what it computes one level down) and which part of that it owns.
// ofs[j][i] = source index (in level j-1 of the group, or the group's source for j = 0) of index i of level j,
// already clamped into the source. n[j] = size of level j. align = 4 for columns (dword granularity), 1 for rows.
void plan_dimension(const std::vector<const int32_t*>& ofs, const std::vector<int>& n, const std::vector<int>& nsrc,
                    int tile, int align, std::vector<std::vector<FusedRange>>& out) {
    const int nl = (int)n.size();
    auto up = [&](int v) { return (v + align - 1) / align * align; };
    auto down = [&](int v) { return v / align * align; };
    const int last = nl - 1;
    const int nt = (n[last] + tile - 1) / tile;
    out.assign(nl, std::vector<FusedRange>(nt));
    for (int i = 0; i < nt; i++) {
        const int lo = i * tile, hi = std::min((i + 1) * tile, up(n[last]));
        out[last][i] = FusedRange{lo, i + 1 == nt ? up(n[last]) : hi, lo, i + 1 == nt ? up(n[last]) : hi};
    }
    for (int j = last - 1; j >= 0; j--) {
        std::vector<int> start(nt + 1), needHi(nt);
        for (int i = 0; i < nt; i++) {
            const FusedRange& c = out[j + 1][i];
            const int first = std::min(c.lo, n[j + 1] - 1), lastIdx = std::min(c.hi, n[j + 1]) - 1;
            auto srcOf = [&](int idx) { return std::min(std::max(ofs[j + 1][idx], 0), nsrc[j + 1] - 1); };
            start[i] = i == 0 ? 0 : down(srcOf(first));
            needHi[i] = std::min(srcOf(lastIdx) + 1, nsrc[j + 1] - 1) + 1;
        }
        start[nt] = up(n[j]);
        for (int i = 0; i < nt; i++) start[i] = std::min(start[i], start[nt]);
        for (int i = nt - 1; i > 0; i--) start[i] = std::min(start[i], start[i + 1]);   // (already monotone; kept safe)
        for (int i = 0; i < nt; i++) {
            FusedRange r;
            r.lo = r.ownLo = start[i];
            r.ownHi = start[i + 1];
            r.hi = std::min(std::max(up(needHi[i]), r.ownHi), up(n[j]));
            r.hi = std::max(r.hi, r.lo);
            out[j][i] = r;
        }
    }
}
}  // namespace

void plan_fused_pyramid(const PyramidGeom& g, const std::vector<HostResizeTab>& ht, const ResizeTab* devTabs,
                        int tileW, int tileH, int groupLevels, FusedPlan& plan) {
    plan = FusedPlan();
    const int nd = g.nlevels - 1;                      // derived levels 1 .. nlevels-1
    if (nd < 1) return;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].w < 1 || g.lv[l].h < 1) return;
    for (int l = 1; l < g.nlevels; l++)
        if (devTabs[l].maxSpan > 6) return;            // the 8-byte window of the fast path
    FusedPlan p;
    int first = 0;
    while (first < nd) {
        const int nl = std::min(std::min(groupLevels, kFusedMaxLevels), nd - first);
        FusedGroup G{};
        G.nl = nl;
        std::vector<const int32_t*> xo(nl), yo(nl);
        std::vector<int> nw(nl), nh(nl), sw(nl), sh(nl);
        for (int j = 0; j < nl; j++) {
            const int l = first + 1 + j;               // destination level
            FusedLevel& F = G.lv[j];
            F.dw = g.lv[l].w; F.dh = g.lv[l].h; F.sw = g.lv[l - 1].w; F.sh = g.lv[l - 1].h;
            F.dstOffset = g.lv[l].offset; F.dstStride = g.lv[l].stride;
            F.xentry = devTabs[l].xentry; F.yofs = devTabs[l].yofs;
            F.beta2 = reinterpret_cast<const uint32_t*>(devTabs[l].ibeta);
            xo[j] = ht[l].xofs.data(); yo[j] = ht[l].yofs.data();
            nw[j] = F.dw; nh[j] = F.dh; sw[j] = F.sw; sh[j] = F.sh;
        }
        // the widest tile of the group must fit the 32 lanes a row of column groups gets: narrow the tiles until it does
        std::vector<std::vector<FusedRange>> cols, rows;
        plan_dimension(yo, nh, sh, tileH, 1, rows);
        int tw = std::max(tileW & ~3, 8), buf = 0;
        for (;; tw -= 8) {
            if (tw < 8) return;
            plan_dimension(xo, nw, sw, tw, 4, cols);
            int widest = 0;
            buf = 0;
            for (int j = 0; j < nl; j++) {
                int wmax = 0, hmax = 0;
                for (const FusedRange& r : cols[j]) wmax = std::max(wmax, r.hi - r.lo);
                for (const FusedRange& r : rows[j]) hmax = std::max(hmax, r.hi - r.lo);
                widest = std::max(widest, wmax);
                if (j + 1 < nl) buf = std::max(buf, (wmax + 8) * hmax);
            }
            if (widest <= 128) break;
        }
        G.ntx = (int)cols[0].size();
        G.nty = (int)rows[0].size();
        G.bufBytes = (buf + 15) & ~15;
        if (2 * G.bufBytes > 64 * 1024) return;
        G.cols = reinterpret_cast<const FusedRange*>(p.ranges.size());      // offsets until uploaded
        for (int j = 0; j < nl; j++) p.ranges.insert(p.ranges.end(), cols[j].begin(), cols[j].end());
        G.rows = reinterpret_cast<const FusedRange*>(p.ranges.size());
        for (int j = 0; j < nl; j++) p.ranges.insert(p.ranges.end(), rows[j].begin(), rows[j].end());
        p.groups.push_back(G);
        p.firstLevel.push_back(first);
        first += nl;
    }
    plan = p;
}

hipError_t launch_pyramid(const FrameSet& fs, const PyramidGeom& g, const ResizeTab* tabs, int nframes,
                          hipStream_t s, const FusedPlan* fused) {
    const bool level0Aligned = ((uintptr_t)fs.img % 4 == 0) && (fs.imgRowStride % 4 == 0) && (fs.imgFrameStride % 4 == 0);
    if (fused && !fused->groups.empty() && level0Aligned) {
        for (size_t k = 0; k < fused->groups.size(); k++) {
            FusedGroup G = fused->groups[k];
            const int l0 = fused->firstLevel[k];
            if (l0 == 0) {
                G.src = fs.img;
                G.srcRowStride = fs.imgRowStride;
                G.srcFrameStride = fs.imgFrameStride;
            } else {
                G.src = fs.pyr + g.lv[l0].offset;
                G.srcRowStride = g.lv[l0].stride;
                G.srcFrameStride = g.pyrBytes;
            }
            hipLaunchKernelGGL(pyr_fused_kernel, dim3(nframes, G.ntx * G.nty, 1), dim3(256, 1, 1), 2 * (size_t)G.bufBytes, s,
                               G, fs.pyr, g.pyrBytes);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
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
        // 8-byte window from the first source column: three column steps + 1 neighbour must stay <= 7
        const bool narrow = tabs[l].maxSpan <= 6;
        if (aligned && narrow)
            hipLaunchKernelGGL(pyr_resize_dword_kernel,
                               dim3(nframes, (D.w + 255) / 256, (D.h + 4 * kRowsPerThread - 1) / (4 * kRowsPerThread)),
                               block, 0, s, src, srs, sfs, S.w, S.h,
                               fs.pyr + D.offset, (long long)D.stride, g.pyrBytes, D.w, D.h, tabs[l]);
        else
            hipLaunchKernelGGL(pyr_resize_kernel, grid, block, 0, s, src, srs, sfs, S.w, S.h,
                               fs.pyr + D.offset, (long long)D.stride, g.pyrBytes, D.w, D.h, tabs[l]);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Checks the invariants the one-launch kernel relies on: on every level the owned ranges of the tiles partition
// the level (so each pixel is written exactly once), a tile owns only what it computes, and everything it
// computes finds both of its source columns / rows inside what the same tile computed one level up.
bool check_fused_plan(const PyramidGeom& g, const std::vector<HostResizeTab>& ht, const FusedPlan& plan,
                      std::string& err) {
    auto fail = [&](const char* what, size_t k, int j, int i) {
        char buf[160];
        snprintf(buf, sizeof(buf), "%s (group %zu, level %d of it, tile %d)", what, k, j, i);
        err = buf;
        return false;
    };
    int covered = 0;
    for (size_t k = 0; k < plan.groups.size(); k++) {
        const FusedGroup& G = plan.groups[k];
        if (plan.firstLevel[k] != covered) return fail("groups do not follow one another", k, 0, 0);
        covered += G.nl;
        for (int dim = 0; dim < 2; dim++) {
            const int nt = dim ? G.nty : G.ntx, align = dim ? 1 : 4;
            const FusedRange* R = plan.ranges.data() + reinterpret_cast<size_t>(dim ? G.rows : G.cols);
            for (int j = 0; j < G.nl; j++) {
                const int l = plan.firstLevel[k] + 1 + j;
                const int n = dim ? g.lv[l].h : g.lv[l].w, nsrc = dim ? g.lv[l - 1].h : g.lv[l - 1].w;
                const int nup = (n + align - 1) / align * align;
                const std::vector<int32_t>& ofs = dim ? ht[l].yofs : ht[l].xofs;
                int next = 0;
                for (int i = 0; i < nt; i++) {
                    const FusedRange& r = R[j * nt + i];
                    if (r.ownLo != next || r.ownHi < r.ownLo) return fail("owned ranges do not tile the level", k, j, i);
                    next = r.ownHi;
                    if (r.lo > r.ownLo || r.hi < r.ownHi || r.lo % align || r.hi % align || r.hi > nup || r.lo < 0)
                        return fail("computed range does not hold the owned one", k, j, i);
                    if (!dim && r.hi - r.lo > 128) return fail("tile wider than 128 columns", k, j, i);
                    if (j == 0) continue;
                    const FusedRange& up = R[(j - 1) * nt + i];
                    for (int x = r.lo; x < std::min(r.hi, n); x++) {
                        const int s0 = std::min(std::max(ofs[x], 0), nsrc - 1), s1 = std::min(s0 + 1, nsrc - 1);
                        if (s0 < up.lo || s1 >= up.hi) return fail("a source lies outside the tile of the level above", k, j, i);
                    }
                }
                if (next != nup) return fail("owned ranges stop short of the level's end", k, j, nt - 1);
            }
        }
    }
    if (!plan.groups.empty() && covered != g.nlevels - 1) return fail("groups do not cover all levels", 0, 0, 0);
    return true;
}

}  // namespace orbx
