// Orientation, blur and steered BRIEF for the selected keypoints, one 64-lane wave per keypoint
// (IC_Angle / computeOrientation src/ORBextractor.cc:69-96,464-471; the per-level
// GaussianBlur :1077-1078; computeOrbDescriptor :99-139; final scaling and concatenation
// :1067-1096).
//
// The reference blurs every whole level and then samples 512 pixels per keypoint. Here a wave
// stages the 43-row patch around its keypoint in LDS (radius 15 for the moments, 18 for the
// rotated pattern, +3 for the 7x7 taps), blurs it there with the same separable 8.8 fixed-point
// filter (u16 row sums, 16.16 column sums, one rounding at the end: cv fixedSmoothInvoker), and
// samples the blurred patch. No blurred level is ever written to HBM.
//   R  raw bytes, 43 rows x 48 B, columns aligned to a dword of the image row (coalesced dword loads)
//   T  row sums as u16 (they never exceed 257*255), rows 2j and 2j+1 packed into one dword per column,
//      22 x 44 dwords: four outputs per lane from three raw dwords via v_alignbyte + v_dot4_u32_u8
//   Bl blurred bytes, 37 x 44 (overlays R once the row pass is done): four outputs per lane, each from
//      four v_dot2_u32_u16 over the packed row pairs (the 7 taps straddle 4 pairs either way round)
// orbx_debug_blur_level() runs the same arithmetic over a full level for parity tests.
#include "orbx_internal.h"

namespace orbx {

namespace {

__device__ const int8_t c_pattern[1024] = {
#include "../../include/orbx_pattern31.inc"
};

// End of each row of the radius-15 disc (umax, src/ORBextractor.cc:446-461).
constexpr int kUmax[kHalfPatch + 1] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// Byte weights for the intensity-centroid sums, four columns per dword: row |v| (16) x dword q (8),
// column u = 4q - 15 + byte. Set 0: 1 inside the disc (plain row sum); set 1: u + 15 inside the disc.
struct MomentWeights {
    uint32_t w[2][16][8];
};
constexpr MomentWeights make_moment_weights() {
    MomentWeights m{};
    for (int av = 0; av < 16; av++)
        for (int q = 0; q < 8; q++) {
            uint32_t w0 = 0, w1 = 0;
            for (int b = 0; b < 4; b++) {
                const int u = 4 * q - 15 + b;
                const int au = u < 0 ? -u : u;
                if (au <= kUmax[av]) {
                    w0 |= 1u << (8 * b);
                    w1 |= (uint32_t)(u + 15) << (8 * b);
                }
            }
            m.w[0][av][q] = w0;
            m.w[1][av][q] = w1;
        }
    return m;
}
__device__ const MomentWeights c_momw = make_moment_weights();

constexpr int kPR = 21;                 // patch radius
constexpr int kRows = 2 * kPR + 1;      // 43 patch rows
constexpr int kRPitch = 48;             // bytes per raw row (12 dwords)
constexpr int kTPitch = 44;             // dwords per packed row pair of T, bytes per row of Bl
constexpr int kTPairs = 22;             // row pairs of T (43 rows)
constexpr int kGroups = 10;             // 4-column groups blurred per row (40 columns cover cx-18..cx+18)
constexpr int kBlRows = 37;             // blurred rows: patch rows 3..39
constexpr int kWavesPerBlock = 4;
constexpr int kWaveLds = kRows * kRPitch + kTPairs * kTPitch * 4;  // 2064 + 3872 bytes
constexpr int kMomLds = 2 * 16 * 8 * 4;                             // moment weights, shared by the block

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

}  // namespace

template <bool ALIGNED4>
__global__ __launch_bounds__(64 * kWavesPerBlock) void orient_desc_kernel(FrameSet fs, PyramidGeom g,
                                                                           int blurTaps,
                                                                           orbx_keypoint* __restrict__ kpsOut,
                                                                           uint8_t* __restrict__ descOut, int cap,
                                                                           int32_t* __restrict__ countsOut) {
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[kWavesPerBlock * kWaveLds + kMomLds];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps the set-up scalar
    const int f = blockIdx.x;                     // frame-fastest grid: one frame per XCD
    const int slot = blockIdx.y * kWavesPerBlock + wv;    // index into the frame's selected regions
    uint32_t* momw = reinterpret_cast<uint32_t*>(s_mem + kWavesPerBlock * kWaveLds);
    momw[threadIdx.x] = (&c_momw.w[0][0][0])[threadIdx.x];
    // No wave leaves before the last barrier; inactive waves just skip the work.
    bool active = slot < g.totalSel;
    int l = 0, k = 0, outIdx = 0, kx = 0, ky = 0, cx = kPR;
    uint32_t packed = 0;
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
    uint8_t* R = s_mem + wv * kWaveLds;                                   // raw patch, later Bl
    uint32_t* T = reinterpret_cast<uint32_t*>(R + kRows * kRPitch);      // packed row sums
    if (active) {
        packed = fs.sel[(long long)f * g.totalSel + slot];
        kx = cand_x(packed) + kBorder;   // level coordinates (:835-836)
        ky = cand_y(packed) + kBorder;
        const uint8_t* img;
        long long rowStride;
        if (l == 0) {
            img = fs.img + (long long)f * fs.imgFrameStride;
            rowStride = fs.imgRowStride;
        } else {
            img = fs.pyr + (long long)f * g.pyrBytes + L.offset;
            rowStride = L.stride;
        }
        // ---- stage the raw patch: LDS column j <-> image column xa + j, LDS row r <-> image row ky-21+r ----
        const int xa = (kx - kPR) & ~3;
        const bool interior = kx - kPR >= 0 && kx + kPR < L.w && ky - kPR >= 0 && ky + kPR < L.h &&
                              xa + kRPitch <= rowStride && (ALIGNED4 || l > 0);
        if (interior) {
            cx = kx - xa;
            uint32_t* R32 = reinterpret_cast<uint32_t*>(R);
            // lane + 64q -> (row, dword): 64 = 5*12 + 4, so both advance without a division; all loads
            // are issued before the first LDS store so their latencies overlap
            constexpr int kIter = (kRows * 12 + 63) / 64;
            int r = lane / 12, d = lane - r * 12;
            const uint8_t* src = img + (long long)(ky - kPR + r) * rowStride + xa + d * 4;
            uint32_t v[kIter];
#pragma unroll
            for (int q = 0; q < kIter; q++) {
                v[q] = r < kRows ? *reinterpret_cast<const uint32_t*>(src) : 0u;
                d += 4;
                r += 5;
                src += 5 * rowStride + 16;
                if (d >= 12) {
                    d -= 12;
                    r += 1;
                    src += rowStride - 48;
                }
            }
#pragma unroll
            for (int q = 0; q < kIter; q++)
                if (lane + 64 * q < kRows * 12) R32[lane + 64 * q] = v[q];
        } else {
            // near the level border or unaligned caller memory: bytes, BORDER_REFLECT_101 coordinates
            cx = kPR;
            for (int i = lane; i < kRows * kRPitch; i += 64) {
                const int r = i / kRPitch, c = i - r * kRPitch;
                const int sy = reflect101(ky - kPR + r, L.h), sx = reflect101(kx - kPR + c, L.w);
                R[i] = img[(long long)sy * rowStride + sx];
            }
        }
    }
    __syncthreads();

    // ---- intensity centroid on the unblurred patch (:69-96) ----
    float angle = 0.f;
    const int g0 = (cx - 18) >> 2;     // first blurred 4-column group: columns 4*g0 .. 4*g0+39 cover cx-18 .. cx+18
    if (active) {
        // m01 = sum v*I, m10 = sum u*I over the disc; each lane takes dwords (four columns) of the 31x31
        // window: row sums with 0/1 byte weights, sum (u+15)*I with byte weights, both by v_dot4_u32_u8
        const uint32_t* R32 = reinterpret_cast<const uint32_t*>(R);
        const int c0 = cx - kHalfPatch;            // first column of the window
        const uint32_t sh = (uint32_t)(c0 & 3);
        int s0 = 0, s1 = 0, m01 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = lane + 64 * q;           // 31 rows x 8 dwords = 248 items
            if (i < 31 * 8) {
                const int row = i >> 3, dq = i & 7;
                const int v = row - kHalfPatch, av = abs(v);
                const int di = (kPR + v) * 12 + (c0 >> 2) + dq;
                const uint32_t D = __builtin_amdgcn_alignbyte(R32[di + 1], R32[di], sh);
                const uint32_t rs = __builtin_amdgcn_udot4(D, momw[av * 8 + dq], 0u, false);
                s0 += (int)rs;
                m01 += v * (int)rs;
                s1 = (int)__builtin_amdgcn_udot4(D, momw[128 + av * 8 + dq], (uint32_t)s1, false);
            }
        }
        int m10 = s1 - kHalfPatch * s0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m10 += __shfl_xor(m10, off);
            m01 += __shfl_xor(m01, off);
        }
        angle = fast_atan2_deg((float)m01, (float)m10);

        // ---- row pass: T[r][j] = sum_c K[c] * R[r][j-3+c], four columns per lane ----
        const uint32_t klo = blurTaps == ORBX_BLUR_CV_4_0_0 ? 0x37312212u : 0x38302212u;  // K0..K3
        const uint32_t khi = blurTaps == ORBX_BLUR_CV_4_0_0 ? 0x00122231u : 0x00122230u;  // K4..K6
        int r = lane / kGroups, gq = lane - r * kGroups;      // 64 = 6*10 + 4
#pragma unroll
        for (int q = 0; q < (kRows * kGroups + 63) / 64; q++) {
            if (r < kRows) {
                const int gg = g0 + gq;
                const uint32_t A = gg ? R32[r * 12 + gg - 1] : 0u;   // columns 4g-4 .. 4g-1
                const uint32_t B = R32[r * 12 + gg];                 // columns 4g   .. 4g+3
                const uint32_t C = R32[r * 12 + gg + 1];             // columns 4g+4 .. 4g+7
                uint4 o;
                o.x = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 1), klo,
                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 1), khi, 0u, false), false);
                o.y = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 2), klo,
                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 2), khi, 0u, false), false);
                o.z = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 3), klo,
                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 3), khi, 0u, false), false);
                o.w = __builtin_amdgcn_udot4(B, klo, __builtin_amdgcn_udot4(C, khi, 0u, false), false);
                // row r is the low (even r) or high (odd r) half of pair r/2
                uint16_t* t16 = reinterpret_cast<uint16_t*>(T + (r >> 1) * kTPitch + 4 * gg) + (r & 1);
                t16[0] = (uint16_t)o.x;
                t16[2] = (uint16_t)o.y;
                t16[4] = (uint16_t)o.z;
                t16[6] = (uint16_t)o.w;
            }
            gq += 4;
            r += 6;
            if (gq >= kGroups) {
                gq -= kGroups;
                r += 1;
            }
        }
    }
    __syncthreads();

    // ---- column pass: Bl[y][j] = min(255, (sum_r K[r] * T[y+r][j] + 32768) >> 16), y = patch row - 3 ----
    if (active) {
        // taps as u16 pairs. Blurred row y uses patch rows y..y+6; for even y those are the pairs
        // (y,y+1)(y+2,y+3)(y+4,y+5)(y+6,-), for odd y (-,y)(y+1,y+2)(y+3,y+4)(y+5,y+6).
        const uint32_t k2 = blurTaps == ORBX_BLUR_CV_4_0_0 ? 49u : 48u, k3 = blurTaps == ORBX_BLUR_CV_4_0_0 ? 55u : 56u;
        const uint32_t we[4] = {18u | (34u << 16), k2 | (k3 << 16), k2 | (34u << 16), 18u};
        const uint32_t wo[4] = {18u << 16, 34u | (k2 << 16), k3 | (k2 << 16), 34u | (18u << 16)};
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        uint32_t* Bl32 = reinterpret_cast<uint32_t*>(R);
        int y = lane / kGroups, gq = lane - y * kGroups;
#pragma unroll
        for (int q = 0; q < (kBlRows * kGroups + 63) / 64; q++) {
            if (y < kBlRows) {
                const int gg = g0 + gq;
                uint4 acc = make_uint4(32768u, 32768u, 32768u, 32768u);
                const bool odd = y & 1;
                const uint32_t* tp = T + (y >> 1) * kTPitch + 4 * gg;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint4 t = *reinterpret_cast<const uint4*>(tp + r * kTPitch);
                    const us2 w = __builtin_bit_cast(us2, odd ? wo[r] : we[r]);
                    acc.x = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.x), w, acc.x, false);
                    acc.y = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.y), w, acc.y, false);
                    acc.z = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.z), w, acc.z, false);
                    acc.w = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.w), w, acc.w, false);
                }
                Bl32[y * (kTPitch / 4) + gg] = min(acc.x >> 16, 255u) | (min(acc.y >> 16, 255u) << 8) |
                                               (min(acc.z >> 16, 255u) << 16) | (min(acc.w >> 16, 255u) << 24);
            }
            gq += 4;
            y += 6;
            if (gq >= kGroups) {
                gq -= kGroups;
                y += 1;
            }
        }
    }
    __syncthreads();
    if (!active) return;

    // ---- steered BRIEF on the blurred patch (:99-139) ----
    const uint8_t* Bl = R;
    const float rad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.f));
    double sd, cd;
    sincos((double)rad, &sd, &cd);
    const float a = (float)cd, b = (float)sd;
    unsigned long long* desc64 = reinterpret_cast<unsigned long long*>(descOut + ((long long)f * cap + outIdx) * 32);
    const int base = (kPR - 3) * kTPitch + cx;   // blurred row y <-> patch row y + 3
#pragma unroll
    for (int round = 0; round < 4; round++) {
        const int pair = round * 64 + lane;
        const char4 pt = *reinterpret_cast<const char4*>(c_pattern + 4 * pair);
        const float x0 = (float)pt.x, y0 = (float)pt.y, x1 = (float)pt.z, y1 = (float)pt.w;
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = Bl[base + r0 * kTPitch + c0];
        const int t1 = Bl[base + r1 * kTPitch + c1];
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
    dim3 grid(nframes, (g.totalSel + kWavesPerBlock - 1) / kWavesPerBlock, 1), block(64 * kWavesPerBlock, 1, 1);
    const bool aligned = ((uintptr_t)fs.img % 4 == 0) && (fs.imgRowStride % 4 == 0) && (fs.imgFrameStride % 4 == 0);
    if (aligned)
        hipLaunchKernelGGL(orient_desc_kernel<true>, grid, block, 0, s, fs, g, blurTaps, kps, desc, cap, counts);
    else
        hipLaunchKernelGGL(orient_desc_kernel<false>, grid, block, 0, s, fs, g, blurTaps, kps, desc, cap, counts);
    return hipGetLastError();
}

hipError_t launch_blur_level(const uint8_t* src, int w, int h, long long stride, uint8_t* dst,
                             long long dstStride, int blurTaps, hipStream_t s) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(blur_level_kernel, grid, block, 0, s, src, w, h, stride, dst, dstStride, blurTaps);
    return hipGetLastError();
}

}  // namespace orbx
