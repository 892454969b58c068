// Per-cell FAST-9/16 with threshold fallback and non-maximum suppression
// (ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:781-821, which calls
// cv::FAST(cell, iniThFAST, true) and, when that returns nothing, cv::FAST(cell, minThFAST, true)).
//
// One 64-lane wave owns one cell of one frame. A cell is an independent sub-image for the
// reference: FAST scans it 3 px inside its own border and suppression treats everything outside
// the scanned rectangle as score 0, so no data crosses cells and no score map goes to HBM.
//
//   1. the cell's pixels (cell + 6 px) are staged in LDS as coalesced dwords;
//   then, at iniThFAST and - only if that leaves the cell empty - again at minThFAST:
//   2. every scanned pixel runs a cheap antipodal-pair test (two of the eight pairs) at the round's
//      threshold; survivors are queued in LDS in row-major order (ballot + prefix count);
//   3. queued pixels get the exact corner strength S = max over the 16 arcs of 9 of the smallest
//      |centre - circle| with one sign; a pixel is a corner at threshold t iff S > t and its FAST
//      score (cv cornerScore) is S - 1; corners keep their S in an LDS score map;
//   4. corners that beat their 8 neighbours strictly are written in row-major order to the cell's
//      slot array.
#include "orbx_internal.h"

namespace orbx {

namespace {

// Three-input minimum / maximum of both 16-bit halves at once (gfx950's packed half-float forms). The halves hold
// a pixel value 0..255 as it is: as half-floats these are zero and the 255 smallest subnormals, which compare like
// the integers (kernels run with half-float subnormals kept), so these are exact integer minima and maxima.
// SWAP = 0: the operands as they are; 1: the third with its halves exchanged; 2: the second and the third (the
// operand-select bits of the packed encoding, no extra instruction).
template <int SWAP>
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    if (SWAP == 0) asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SWAP == 1) asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int SWAP>
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    if (SWAP == 0) asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SWAP == 1) asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Smallest (MIN) or largest value of every run of nine consecutive circle positions, all sixteen runs at once.
// w[j] holds positions j (low half) and j + 8 (high half); with x[j] = w[j] for j < 8 and w[j - 8] with its halves
// exchanged for j >= 8, x[j] holds positions j and j + 8 for every j mod 16, and the run that starts at k (low
// half) and the one that starts at k + 8 (high half) are the reduction of x[k] .. x[k+8]: three reductions of
// three neighbours, t[k], t[k+3], t[k+6], where t[j + 8] is again t[j] with its halves exchanged. Sixteen
// three-input instructions for the sixteen runs.
template <bool MIN>
__device__ __forceinline__ void arc_runs(const uint32_t (&w)[8], uint32_t (&run)[8]) {
    auto op0 = [](uint32_t a, uint32_t b, uint32_t c) { return MIN ? pk_min3<0>(a, b, c) : pk_max3<0>(a, b, c); };
    auto op1 = [](uint32_t a, uint32_t b, uint32_t c) { return MIN ? pk_min3<1>(a, b, c) : pk_max3<1>(a, b, c); };
    auto op2 = [](uint32_t a, uint32_t b, uint32_t c) { return MIN ? pk_min3<2>(a, b, c) : pk_max3<2>(a, b, c); };
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 6; j++) t[j] = op0(w[j], w[j + 1], w[j + 2]);
    t[6] = op1(w[6], w[7], w[0]);
    t[7] = op2(w[7], w[0], w[1]);
    run[0] = op0(t[0], t[3], t[6]);
    run[1] = op0(t[1], t[4], t[7]);
    run[2] = op1(t[2], t[5], t[0]);
    run[3] = op1(t[3], t[6], t[1]);
    run[4] = op1(t[4], t[7], t[2]);
    run[5] = op2(t[5], t[0], t[3]);
    run[6] = op2(t[6], t[1], t[4]);
    run[7] = op2(t[7], t[2], t[5]);
}

// Corner strength of the pixel whose scan offset is e; `at` is the LDS byte address of the tile plus e, so the
// pixel sits at at + 3 * pitch + 4 and the circle at constant offsets from there: see step 3 above.
// With d_k = v - c_k: max over runs of min(d) = v - min over runs of max(c), and min over runs of max(d) =
// v - max over runs of min(c), so the sixteen differences are never formed. The sixteen circle bytes come in through
// sixteen byte loads at constant offsets; one v_lshl_or_b32 per pair puts position k + 8 above position k, and the
// packed network above starts from there. (The loads that fill only one half of a register would save nothing:
// on this chip they clear the other half, so two of them cannot share a destination.)
template <int pitch>
__device__ __forceinline__ int corner_strength(uint32_t at) {
    constexpr int c0 = 3 * pitch + 4;      // the pixel itself
    uint32_t w[8], h[8], v;
    asm volatile(
        "ds_read_u8 %0, %17 offset:%18\n\tds_read_u8 %8, %17 offset:%26\n\t"
        "ds_read_u8 %1, %17 offset:%19\n\tds_read_u8 %9, %17 offset:%27\n\t"
        "ds_read_u8 %2, %17 offset:%20\n\tds_read_u8 %10, %17 offset:%28\n\t"
        "ds_read_u8 %3, %17 offset:%21\n\tds_read_u8 %11, %17 offset:%29\n\t"
        "ds_read_u8 %4, %17 offset:%22\n\tds_read_u8 %12, %17 offset:%30\n\t"
        "ds_read_u8 %5, %17 offset:%23\n\tds_read_u8 %13, %17 offset:%31\n\t"
        "ds_read_u8 %6, %17 offset:%24\n\tds_read_u8 %14, %17 offset:%32\n\t"
        "ds_read_u8 %7, %17 offset:%25\n\tds_read_u8 %15, %17 offset:%33\n\t"
        "ds_read_u8 %16, %17 offset:%34\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]),
          "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]), "=&v"(h[6]), "=&v"(h[7]), "=&v"(v)
        : "v"(at),
          // positions 0..7: (0,3) (1,3) (2,2) (3,1) (3,0) (3,-1) (2,-2) (1,-3)
          "i"(c0 + 3 * pitch), "i"(c0 + 3 * pitch + 1), "i"(c0 + 2 * pitch + 2), "i"(c0 + pitch + 3), "i"(c0 + 3),
          "i"(c0 - pitch + 3), "i"(c0 - 2 * pitch + 2), "i"(c0 - 3 * pitch + 1),
          // positions 8..15: (0,-3) (-1,-3) (-2,-2) (-3,-1) (-3,0) (-3,1) (-2,2) (-1,3)
          "i"(c0 - 3 * pitch), "i"(c0 - 3 * pitch - 1), "i"(c0 - 2 * pitch - 2), "i"(c0 - pitch - 3), "i"(c0 - 3),
          "i"(c0 + pitch - 3), "i"(c0 + 2 * pitch - 2), "i"(c0 + 3 * pitch - 1),
          "i"(c0)
        : "memory");
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] |= h[j] << 16;          // one v_lshl_or_b32 per pair
    uint32_t lo[8], hi[8];
    arc_runs<true>(w, lo);        // smallest value of every run
    arc_runs<false>(w, hi);       // largest value of every run
    const uint32_t loBest = pk_max3<0>(pk_max3<0>(lo[0], lo[1], lo[2]), pk_max3<0>(lo[3], lo[4], lo[5]), pk_max3<0>(lo[6], lo[7], lo[7]));
    const uint32_t hiBest = pk_min3<0>(pk_min3<0>(hi[0], hi[1], hi[2]), pk_min3<0>(hi[3], hi[4], hi[5]), pk_min3<0>(hi[6], hi[7], hi[7]));
    const int loMax = (int)max(loBest & 0xFFFFu, loBest >> 16);     // largest run minimum
    const int hiMin = (int)min(hiBest & 0xFFFFu, hiBest >> 16);     // smallest run maximum
    return max((int)v - hiMin, loMax - (int)v);      // darker-run strength, brighter-run strength
}

// Unsigned 16-bit operations on both halves of a dword.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
// Lanes whose low (word0) / high (word1) 16-bit half of a is below that of b, straight into a lane mask.
__device__ __forceinline__ unsigned long long cmp_lt_word0(uint32_t a, uint32_t b) {
    unsigned long long m;
    asm("v_cmp_lt_u16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ unsigned long long cmp_lt_word1(uint32_t a, uint32_t b) {
    unsigned long long m;
    asm("v_cmp_lt_u16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_1" : "=s"(m) : "v"(a), "v"(b));
    return m;
}

// x + (this lane's bit of m): the lane mask goes in as the carry of an add-with-carry.
__device__ __forceinline__ int add_mask_bit(int x, unsigned long long m) {
    int r;
    unsigned long long carryOut;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carryOut) : "v"(x), "s"(m));
    return r;
}

// number of set bits of m below this lane's own
__device__ __forceinline__ int prefix_count(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

}  // namespace

// PITCH is the LDS row pitch of the staged cell (bytes, multiple of 4); a compile-time constant so that
// all sixteen circle offsets fold into the ds_read immediate offsets.
constexpr int kTileRowBatch = 40;    // tile rows staged per round of loads; the tile area is a multiple of it
constexpr int kDefaultCellsPerBlock = 4;   // waves per workgroup; every wave works alone on its own cell and LDS slice

// Orders a wave's own LDS traffic: the hardware runs one wave's DS instructions in order, so only the
// compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int PITCH, int kCellsPerBlock>
__global__ __launch_bounds__(64 * kCellsPerBlock) void fast_cells_kernel(FrameSet fs, PyramidGeom g, int iniTh,
                                                                          int minTh, int ldsPerWave) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* smem = smem_all + wv * ldsPerWave;
    const int f = blockIdx.x;                         // frame-fastest grid: one frame per XCD
    const int cell = blockIdx.y * kCellsPerBlock + wv;
    if (cell >= g.totalCells) return;
    // the cell's record: rectangle in level coordinates (:783-798), its level's place in the pyramid, its slots
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    const u32x8 rec = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(
        reinterpret_cast<uintptr_t>(fs.cells + cell));
    int32_t* countOut = fs.cellCount + (long long)f * g.totalCells + cell;
    if (rec[6] >> 31) {                                // the reference skips this cell
        if (lane == 0) *countOut = 0;
        return;
    }
    const int iniX = (int)(rec[0] & 0xFFFFu), iniY = (int)(rec[0] >> 16);
    const int maxX = (int)(rec[1] & 0xFFFFu), maxY = (int)(rec[1] >> 16);
    const int sw = maxX - iniX - 6, sh = maxY - iniY - 6;  // scanned rectangle (3-px FAST margin)
    const int th = maxY - iniY;
    const int slotCap = (int)(rec[6] & 0xFFFFu);
    const int levelH = (int)(rec[5] >> 8);
    uint32_t* slotOut = fs.slots + (long long)f * g.totalSlots + rec[2];

    const uint8_t* img;
    long long rowStride;
    if ((rec[5] & 0xFFu) == 0) {
        img = fs.img + (long long)f * fs.imgFrameStride;
        rowStride = fs.imgRowStride;
    } else {
        img = fs.pyr + (long long)f * g.pyrBytes + rec[3];
        rowStride = rec[4];
    }

    // ---- 1. stage the cell in LDS ----
    uint8_t* tile = smem;
    // the tile's LDS byte address (xo = 1: scan offset e is tile byte e + 3 * PITCH + 4)
    const uint32_t tileAt = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)tile;
    // Tile column c holds image column iniX - 1 + c, so the first scanned pixel (iniX + 3) sits at tile column 4
    // and groups of four scanned pixels are whole LDS dwords.
    {
        const int xa = (iniX - 1) & ~3;
        const uint32_t shift = (uint32_t)((iniX - 1) & 3);
        const int ndw = (maxX - iniX + 1 + 3) >> 2;
        // A row of the tile is at most PITCH/4 dwords: 16 lanes (four rows per step) or 32 lanes (two rows)
        // take one row each, so stepping to the next rows is a constant on the LDS side. A tile dword is cut
        // from the two aligned dwords around it (v_alignbyte); both stay left of column maxX + 6 <= w - 10.
        constexpr int kRowLanes = PITCH <= 64 ? 16 : 32, kRowsPerStep = 64 / kRowLanes;
        constexpr int kBatch = kTileRowBatch / kRowsPerStep;
        const int td = lane & (kRowLanes - 1), tr = lane / kRowLanes;
        uint32_t* tile32 = reinterpret_cast<uint32_t*>(tile) + tr * (PITCH / 4) + td;
        // Rows come through a buffer resource over the level: a load's address is the scalar row offset plus this
        // lane's constant offset, so a row of loads costs no vector instruction beyond the v_alignbyte (and a
        // read past the level's last byte would return zero instead of faulting). Row groups past the tile's last
        // one re-read that group (its rows may reach up to three rows below the cell, still inside the level:
        // cells end 16 rows above the level's last row) and are stored too (the tile area holds whole batches
        // of rows), which keeps every load and store unconditional.
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const int stride32 = (int)rowStride;                                  // a level is < 2^31 bytes
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, levelH * stride32, 0x00027000);
        const int laneOff = (int)__umul24((uint32_t)tr, (uint32_t)stride32) + xa + td * 4;
        const int lastGroup = ((th - 1) / kRowsPerStep) * kRowsPerStep;
        const int lastOff = (iniY + lastGroup) * stride32, stepOff = kRowsPerStep * stride32;
        if (td < ndw) {
            for (int r0 = 0; r0 < th; r0 += kTileRowBatch) {
                u32x2 v[kBatch];
                int rowOff = (iniY + r0) * stride32;                                           // wave-uniform
#pragma unroll
                for (int q = 0; q < kBatch; q++) {
                    v[q] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, laneOff, min(rowOff, lastOff), 0);
                    rowOff += stepOff;
                    asm("" : "+s"(rowOff));            // keep the running sum (one add per row, not a multiply per row)
                }
#pragma unroll
                for (int q = 0; q < kBatch; q++)
                    tile32[(r0 + kRowsPerStep * q) * (PITCH / 4)] = __builtin_amdgcn_alignbyte(v[q].y, v[q].x, shift);
            }
        }
    }
    // Score map: sh + 2 rows of the tile's own pitch with a zero ring, so that a pixel's offset e = y * PITCH + x
    // (scan coordinates) addresses the tile (+ 3 * PITCH + 4) and the map (+ PITCH + 1) alike, and a queue entry
    // is that offset.
    uint8_t* score = smem + g.maxTileBytes;
    uint16_t* queue = reinterpret_cast<uint16_t*>(score + g.maxScoreBytes);
    {
        // 16 bytes per lane and store; the last store may run a few bytes into the queue, which is filled later
        uint4* s128 = reinterpret_cast<uint4*>(score);
        const int n16 = ((sh + 2) * PITCH + 15) >> 4;
        for (int i = lane; i < n16; i += 64) s128[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    wave_sync();

    // The reference runs cv::FAST at iniThFAST and, only if that finds nothing in the cell, again at minThFAST
    // (:801-808); so do steps 2-4. Most cells stop after the first round, whose higher threshold also lets
    // fewer pixels through to the exact strength. The score map is still all zero when a second round starts.
    int nout = 0;
    for (int round = 0; round < 2; round++) {
    const int tcur = round == 0 ? iniTh : minTh;
    // ---- 2. cheap test at this round's threshold, queue survivors in row-major order ----
    // A lane tests four horizontally adjacent pixels per step from five LDS dwords (the row itself: the group,
    // its left and right neighbours; rows -3 and +3: the group), every byte picked by an SDWA operand select.
    // 8 lanes span a scan of up to 32 px (8 rows per step), 16 lanes a wider one (4 rows per step).
    // A queue entry is y * PITCH + x.
    const bool narrow = sw <= 32;                      // cells are < 60 px wide (width / floor(width / 30))
    const int rowsPerIter = narrow ? 8 : 4;
    const int gx = narrow ? lane & 7 : lane & 15, ly = narrow ? lane >> 3 : lane >> 4;
    int nq = 0;
    {
        constexpr int P4 = PITCH / 4;
        // Thresholds as the packed test wants them, per 16-bit half: t << 8 to take from v << 8, (t << 8) | 0xFF to
        // add to it. A pixel outside the scan (right of the cell's last column) gets 0xFFFF in both, which
        // saturates its two limits to "nothing is darker" and "nothing is brighter": no mask needed later.
        // Index 0: pixels 0 and 2 of the lane's group, index 1: pixels 1 and 3.
        uint32_t tLo[2], tHi[2];
#pragma unroll
        for (int odd = 0; odd < 2; odd++) {
            const uint32_t in0 = 4 * gx + odd < sw ? 0u : 0xFFFFu, in1 = 4 * gx + odd + 2 < sw ? 0u : 0xFFFF0000u;
            tLo[odd] = ((uint32_t)tcur * 0x01000100u) | in0 | in1;
            tHi[odd] = tLo[odd] | 0x00FF00FFu;
        }
        int e0 = ly * PITCH + 4 * gx;
        const int eEnd = sh * PITCH;
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile) + (ly + 3) * P4 + gx;
        for (int y0 = 0; y0 < sh; y0 += rowsPerIter) {
            const uint32_t A = t32[0], B = t32[1], C = t32[2], U = t32[1 - 3 * P4], D = t32[1 + 3 * P4];
            // Two pixels per instruction: a pixel rides in the high byte of a 16-bit half (an unsigned 16-bit
            // min / max is decided by the high bytes, so whatever sits in the low bytes never reaches a result's
            // high byte). Register E holds pixels 0 and 2 of the group, O pixels 1 and 3. Circle pixel k and
            // its antipode k+8: a run of 9 holds at least one of each pair, so all pairs have a pixel < v-t
            // <=> max over pairs of min < v-t; likewise for > v+t. Two of the eight pairs (0/8 above and below,
            // 4/12 right and left) reject ~85% of the pixels; what passes goes on to the exact strength.
            unsigned long long m[4];
            uint32_t mnmax[2], mxmin[2], lo[2], hi[2];
#pragma unroll
            for (int odd = 0; odd < 2; odd++) {
                const uint32_t vc = odd ? (B & 0xFF00FF00u) : __builtin_amdgcn_perm(B, B, 0x020c000cu);  // v << 8
                const uint32_t p8 = odd ? U : U << 8, p0 = odd ? D : D << 8;
                const uint32_t p12 = odd ? __builtin_amdgcn_perm(B, A, 0x040c020cu) : A;
                const uint32_t p4 = odd ? C << 8 : __builtin_amdgcn_perm(C, B, 0x050c030cu);
                mnmax[odd] = pk_max(pk_min(p0, p8), pk_min(p4, p12));
                mxmin[odd] = pk_min(pk_max(p0, p8), pk_max(p4, p12));
                // (v - t) << 8 with low byte 0x00 and (v + t) << 8 with low byte 0xFF, both saturating: the low
                // bytes decide a comparison between equal high bytes the way the pixel comparison wants it
                lo[odd] = pk_sub_sat(vc, tLo[odd]);
                hi[odd] = pk_add_sat(vc, tHi[odd]);
            }
#pragma unroll
            for (int odd = 0; odd < 2; odd++) {
                m[odd] = cmp_lt_word0(mnmax[odd], lo[odd]) | cmp_lt_word0(hi[odd], mxmin[odd]);
                m[odd + 2] = cmp_lt_word1(mnmax[odd], lo[odd]) | cmp_lt_word1(hi[odd], mxmin[odd]);
            }
            if (y0 + rowsPerIter > sh) {               // last step: lanes on rows below the scan drop out
                const unsigned long long ym = __builtin_amdgcn_ballot_w64(e0 < eEnd);
#pragma unroll
                for (int j = 0; j < 4; j++) m[j] &= ym;
            }
            // A quarter of all steps let no pixel through (flat parts of the image): nothing to rank or store then.
            if ((m[0] | m[1] | m[2] | m[3]) != 0) {                // wave-uniform
                // row-major rank: everything queued by lower lanes, then this lane's own earlier pixels (what earlier
                // steps queued is a scalar and goes into the base address)
                int pos = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[j], pos));
                uint16_t* qdst = queue + nq;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool pass = __builtin_amdgcn_inverse_ballot_w64(m[j]);
                    if (pass) qdst[pos] = (uint16_t)(e0 + j);
                    if (j < 3) pos = add_mask_bit(pos, m[j]);          // pos += pass
                    nq += __popcll(m[j]);
                }
            }
            e0 += rowsPerIter * PITCH;
            t32 += rowsPerIter * P4;
        }
    }
    wave_sync();

    // ---- 3. exact corner strength for queued pixels; keep corners (S > tcur) in order ----
    int nc = 0;
    for (int base = 0; base < nq; base += 64) {
        const int qi = base + lane;
        // lanes past the end of the queue repeat its last entry and drop out at the comparison: no branch around
        // the network, and the lane mask of the corners comes straight from two compares
        const int e = queue[min(qi, nq - 1)];
        const int S = corner_strength<PITCH>(tileAt + (uint32_t)e);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(S > tcur) & __builtin_amdgcn_ballot_w64(qi < nq);
        const bool corner = __builtin_amdgcn_inverse_ballot_w64(m);
        if (corner) score[PITCH + 1 + e] = (uint8_t)S;
        // in-place compaction: position nc+prefix <= qi, and every lane has read queue[qi] above
        if (corner) queue[nc + prefix_count(m)] = (uint16_t)e;
        nc += __popcll(m);
    }
    wave_sync();

    // ---- 4. suppression + ordered output ----
    for (int base = 0; base < nc; base += 64) {
        const int qi = base + lane;
        const int e = queue[min(qi, nc - 1)];       // lanes past the end: as above
        constexpr int sp = PITCH;
        const uint8_t* s = score + PITCH + 1 + e;
        const int sc = s[0] - 1;                     // every queued pixel is a corner at this threshold
        const int a0 = s[-sp - 1], a1 = s[-sp], a2 = s[-sp + 1], a3 = s[-1], a4 = s[1], a5 = s[sp - 1], a6 = s[sp],
                  a7 = s[sp + 1];
        const int nb = max(max(max(a0, a1), max(a2, a3)), max(max(a4, a5), max(a6, a7)));
        // the map holds S only for corners (S > threshold); everything else counts as score 0
        const unsigned long long m = __builtin_amdgcn_ballot_w64(sc > max(nb - 1, 0)) & __builtin_amdgcn_ballot_w64(qi < nc);
        const bool keep = __builtin_amdgcn_inverse_ballot_w64(m);
        if (keep) {
            const int pos = nout + prefix_count(m);
            // e = y * PITCH + x with e < 2^13: the quotient by a 20-bit reciprocal is exact (the error of the rounded-up
            // reciprocal, e * 2^-20, stays far below 1 / PITCH)
            constexpr int kRecip = ((1 << 20) + PITCH - 1) / PITCH;
            const int y = (int)(__umul24((uint32_t)e, (uint32_t)kRecip) >> 20), x = e - y * PITCH;
            // FAST reports sub-image coordinates (x+3, y+3); the caller adds j*wCell, i*hCell (:814-815)
            if (pos < slotCap) slotOut[pos] = pack_cand(iniX - kBorder + x + 3, iniY - kBorder + y + 3, sc);
        }
        nout += __popcll(m);
    }
    if (nout > 0 || iniTh == minTh) break;
    wave_sync();
    }
    if (lane == 0) *countOut = min(nout, slotCap);
}

hipError_t launch_fast_cells(const FrameSet& fs, const PyramidGeom& g, int nframes, int iniTh, int minTh,
                             hipStream_t s) {
    if (g.totalCells == 0) return hipSuccess;
    constexpr int waves = kDefaultCellsPerBlock;
    const int ldsPerWave = (g.maxTileBytes + g.maxScoreBytes + 2 * g.maxScanPix + 16 + 15) & ~15;
    const size_t lds = (size_t)ldsPerWave * waves;
    dim3 grid(nframes, (g.totalCells + waves - 1) / waves, 1), block(64 * waves, 1, 1);
    // level 0 is read with dword loads like the other levels: the caller of this launcher has made it 4-byte aligned
    if ((uintptr_t)fs.img % 4 || fs.imgRowStride % 4 || fs.imgFrameStride % 4) return hipErrorInvalidValue;
    const int tp = g.tilePitch;
    // the widest possible cells (59 x 59 px at pitch 80) need a little over the default 64 KB of dynamic LDS
    auto launch = [&](auto kernel) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, grid, block, lds, s, fs, g, iniTh, minTh, ldsPerWave);
        return hipGetLastError();
    };
    return tp == 40 ? launch(fast_cells_kernel<40, waves>) : tp == 48 ? launch(fast_cells_kernel<48, waves>) : launch(fast_cells_kernel<80, waves>);
}

}  // namespace orbx
