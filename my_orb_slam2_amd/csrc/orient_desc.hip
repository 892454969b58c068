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
//   R  raw bytes, 44 rows x 48 B; LDS column j is image column kx - 23 + j for every keypoint (the
//      dword loads are re-aligned with v_alignbyte on the way in), so all later indices are constants
//   T  row sums as u16 (they never exceed 257*255), rows 2j and 2j+1 packed into one dword per column,
//      22 x 44 dwords. One lane makes both rows of a pair for four columns (v_dot4_u32_u8 of the three dwords
//      around the group with the tap vector shifted per column) and stores them as one 16-byte write.
//   Bl blurred bytes, 38 x 44 (overlays R once the row pass is done). One lane makes blurred rows 2p
//      and 2p+1 for four columns from the same four packed row pairs (v_dot2_u32_u16; the 7 taps
//      straddle the pairs as (K0,K1)(K2,K3)(K4,K5)(K6,-) for the even row and (-,K0)(K1,K2)(K3,K4)(K5,K6)
//      for the odd one).
// Only what the rotated pattern can reach is blurred: its points lie within radius sqrt(338) of the
// keypoint, so the work items of both passes come from tables built at compile time for that disc.
// orbx_debug_keypoint_blur() hands the blurred patches of one frame's keypoints to the parity tests.
#include "orbx_internal.h"

namespace orbx {

namespace {

// rBRIEF pattern (x0, y0, x1, y1 per pair); the kernel reads it as floats reordered to x0, x1, y0, y1
constexpr int kPatternInt[1024] = {
#include "../../include/orbx_pattern31.inc"
};
struct PatternF {
    float v[1024];
};
constexpr PatternF make_pattern_f() {
    PatternF p{};
    for (int i = 0; i < 256; i++) {
        p.v[4 * i + 0] = (float)kPatternInt[4 * i + 0];
        p.v[4 * i + 1] = (float)kPatternInt[4 * i + 2];
        p.v[4 * i + 2] = (float)kPatternInt[4 * i + 1];
        p.v[4 * i + 3] = (float)kPatternInt[4 * i + 3];
    }
    return p;
}
__device__ const PatternF c_patternf_tab = make_pattern_f();

// End of each row of the radius-15 disc (umax, src/ORBextractor.cc:446-461).
constexpr int kUmax[kHalfPatch + 1] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

constexpr int kPR = 21;                 // patch radius
constexpr int kRowsAlloc = 44;          // 43 patch rows, staged in fours; Bl row 37 also lands in the spare row
constexpr int kRPitch = 48;             // bytes per raw row (12 dwords)
constexpr int kCx = 23;                 // LDS column of the keypoint
constexpr int kTPitch = 44;             // bytes per row of Bl (columns 0..43 of the patch)
constexpr int kTPairs = 22;             // row pairs of T (43 rows)
constexpr int kTItems = 191;            // (row pair, column group) cells of T that the pattern's disc needs
constexpr int kTOffset = 43 * kRPitch;  // T starts on the raw patch's spare row 43 (read only into the unused half of T's last pair)
constexpr int kBlRows = 37;             // blurred rows: patch rows 3..39
constexpr int kWaveLds = kTOffset + kTItems * 16;   // 2064 + 3056 = 5120 bytes: 32 waves per CU
constexpr int kHIters = 3, kVIters = 3; // 64-item rounds of the row pass and of the column pass
static_assert(kBlRows == kBlurPatchRows && kTPitch == kBlurPatchPitch && kPR - 3 == kBlurPatchCy && kCx == kBlurPatchCx,
              "the probe's patch shape is the kernel's");

// Byte weights for the intensity-centroid sums, four columns per dword, in the order the lanes fetch them: round q
// (4) x lane (64) -> patch row v + 15 = 8q + moment_row(lane) (32 rows, the last one all zero), dword lane & 7,
// column u = 4 (lane & 7) - 15 + byte. .x: 1 inside the disc (plain row sum); .y: u + 15 inside the disc.
// moment_row puts the even rows of a block of eight on the lower half-wave and the odd rows on the upper one: with
// 12 dwords per raw row, four rows two apart start 8 banks apart, so the 32 dwords a half-wave reads cover each
// bank once (rows 0..3 together would put rows 0 and 3 on the same four banks).
__host__ __device__ constexpr int moment_row(int lane) { return 2 * ((lane >> 3) & 3) + (lane >> 5); }
struct MomentWeights {
    uint32_t w[4][64][2];
};
constexpr MomentWeights make_moment_weights() {
    MomentWeights m{};
    for (int qq = 0; qq < 4; qq++)
        for (int lane = 0; lane < 64; lane++) {
            const int row = 8 * qq + moment_row(lane), q = lane & 7;
            if (row >= 31) continue;
            const int av = row < 15 ? 15 - row : row - 15;
            uint32_t w0 = 0, w1 = 0;
            for (int b = 0; b < 4; b++) {
                const int u = 4 * q - 15 + b;
                const int au = u < 0 ? -u : u;
                if (au <= kUmax[av]) {
                    w0 |= 1u << (8 * b);
                    w1 |= (uint32_t)(u + 15) << (8 * b);
                }
            }
            m.w[qq][lane][0] = w0;
            m.w[qq][lane][1] = w1;
        }
    return m;
}

// Can the rounded, rotated pattern land on offset (dx, dy)? Its real position is within radius sqrt(338)
// (the farthest pattern point is (-13,-13)); (dx, dy) is reachable if its unit box touches that disc.
constexpr bool reachable(int dx, int dy) {
    const int nx = dx < 0 ? -2 * dx - 1 : 2 * dx - 1, ny = dy < 0 ? -2 * dy - 1 : 2 * dy - 1;  // twice the box's nearest point
    const int px = nx < 0 ? 0 : nx, py = ny < 0 ? 0 : ny;
    return px * px + py * py <= 4 * 338 + 2;
}
// blurred row y (0..36, dy = y - 18), column group gg (columns 4gg .. 4gg+3, dx = column - 23)
constexpr bool blur_group_needed(int y, int gg) {
    if (y < 0 || y >= kBlRows) return false;
    for (int b = 0; b < 4; b++)
        if (reachable(4 * gg + b - kCx, y - 18)) return true;
    return false;
}
// patch row r (0..42) feeds blurred rows r-6 .. r
constexpr bool rowsum_group_needed(int r, int gg) {
    for (int y = r - 6; y <= r; y++)
        if (blur_group_needed(y, gg)) return true;
    return false;
}
// T holds, for every needed (row pair j, column group gg), one 16-byte cell: the packed sums of rows 2j, 2j+1 in
// the group's four columns. Cells of one group are consecutive in j (the pairs a group needs are a contiguous run,
// the disc being convex), so the four pairs a column-pass item reads are 16 bytes apart. The order of the groups
// (kTGroupOrder) and the lane each work item runs on (kRowItemSlot, kColItemSlot below) are free choices: the
// items of a pass are independent. tools/lds/orient_desc_lds_search.py picks them so that the lanes an LDS
// instruction serves together (half-waves for dword accesses, groups of 8 for 16-byte stores, four interleaved
// groups of 16 for 16-byte loads) fall on different banks: 20 extra LDS cycles per keypoint in both passes,
// where the plain order (items ascending by row pair, then group) takes 136. 18 of the 20 are a floor: the raw
// rows' 48-byte pitch puts up to nine items on one bank residue and a pass has six half-waves.
constexpr int kTGroupOrder[10] = {3, 1, 7, 9, 10, 6, 2, 4, 5, 8};
struct TLayout {
    int lo[11], hi[11], start[11];      // per group: first and last needed pair, first cell
    int cells;
};
constexpr TLayout make_t_layout() {
    TLayout t{};
    for (int gg = 1; gg <= 10; gg++) {
        t.lo[gg] = kTPairs;
        t.hi[gg] = -1;
        for (int j = 0; j < kTPairs; j++)
            if (rowsum_group_needed(2 * j, gg) || rowsum_group_needed(2 * j + 1, gg)) {
                if (j < t.lo[gg]) t.lo[gg] = j;
                t.hi[gg] = j;
            }
    }
    int at = 0;
    for (int k = 0; k < 10; k++) {
        const int gg = kTGroupOrder[k];
        t.start[gg] = at;
        at += t.hi[gg] - t.lo[gg] + 1;
    }
    t.cells = at;
    return t;
}
constexpr TLayout kTLayout = make_t_layout();
static_assert(kTLayout.cells == kTItems, "T layout and its size disagree");
constexpr int t_cell_offset(int j, int gg) { return 16 * (kTLayout.start[gg] + j - kTLayout.lo[gg]); }

// Work items. Row pass: patch row pair j, group gg -> (byte offset in R of the dword left of the group in
// row 2j) | (byte offset in T of the cell) << 16. Column pass: blurred row pair p, group gg ->
// (byte offset in T of pair p's cell; pairs p+1 .. p+3 follow it) | (byte offset in Bl of row 2p's dword) << 16.
// Slot s (round s / 64, lane s % 64) runs the item whose number, counting the needed (pair, group) combinations by
// pair and then by group, is kRowItemSlot[s] / kColItemSlot[s]. Spare slots repeat an item of their own lane
// group (same addresses as a neighbouring lane, same stores).
constexpr unsigned char kRowItemSlot[kHIters * 64] = {
     83, 113,  62, 124,   8, 126,  94, 103, 122, 170,  47,   1,  35, 140,   7, 177,  99, 125,  38, 151,  36,  31, 164,  29,  39, 137,  20, 105,  51, 117, 112,  64,
     90, 157,  88,  19,  45,  89,  84, 134, 155, 188,  54, 143, 154,   9, 147,  79, 178,   2, 127, 181,  33, 114,  26, 161,  71, 171,  30, 180,  52,  11,  22,  61,
     40, 118, 132,  98, 121, 133,  48,  46, 108, 167, 184,  70, 190, 156,  14, 141,  10,  56, 173, 165, 119, 142,  69,  91,  67,  21,  77, 185,  63, 186,  27, 176,
    189,  34,  76,  43,  72,  37,  87, 158,  75, 169, 104, 174, 106,  28,   6,  96, 168,  42, 179, 175,  60,  32, 128, 110, 102, 182,  13,  16, 101, 160,  68,  17,
     57, 129, 135,  81, 163,  74, 183, 116,  55,  80,  92,   5,  58, 115, 111, 159, 139, 145,  44, 172,  86, 153,  53, 149, 152,  82,  12, 146,  85, 131, 130,   0,
    123, 100,  78,   4,  93, 120, 150,  97,   3, 136,  15,  23, 109, 144,  65,  41,  95,  66,  24,  73,  50, 187,  25, 138, 148, 148,  49, 107, 166,  59,  18, 162,
};
constexpr unsigned char kColItemSlot[kVIters * 64] = {
    101, 101,  56,   0, 146,  43, 115, 146,  36, 158,   9,  65, 101,  76,  70,  44, 146,   8, 157,  78,  90, 101, 151,  10, 123, 160,  80,  89,  17,  75,  83, 135,
     18,  18,  14, 122, 105,   7,  81, 106, 120, 159,  60, 138, 113,  61,  71,  16, 105, 136,  22,  86,  18,  47,  73, 125,  51,  26, 143,  93,  62,  69, 130, 103,
      4, 108, 107,  29,  98, 129,  59,  98, 145, 102, 114,  98,  31, 142,  92,   4,  91,  98,  98,  49,   4,  95, 137,  35,   4,  94,  99,  85, 133,  33,   6,  15,
    112,  27,  40,  64,  55, 152,  55, 140,  30,  55,  72,  55,  20, 112, 141, 112,  55,  19,  77,  79,  50, 117, 119, 104,  97, 110,  37,  58,  55,  87, 147,  48,
     41, 153,  52,  41,  23,  54, 156,  74, 148,  53, 131,  39,  84,  68,  63,  11,  23,  66,  24,  42, 128, 155,  41,   1,  41,  45,  88,   2,  13,  57,  23,   3,
    139, 139,  32, 144,  25,  82, 127, 126, 109,  25, 134,  25,  21, 121,  34,  96, 150, 100, 149, 124,  46, 139,  28,  38, 154, 118, 111,   5,  12, 116, 132,  67,
};
struct ItemTables {
    uint32_t h[kHIters * 64];
    uint32_t v[kVIters * 64];
    int nh, nv;
    bool ok;
};
constexpr ItemTables make_item_tables() {
    ItemTables t{};
    uint32_t hn[kHIters * 64] = {}, vn[kVIters * 64] = {};
    int nh = 0, nv = 0;
    bool ok = true;
    for (int j = 0; j < kTPairs; j++)
        for (int gg = 1; gg <= 10; gg++)
            if (rowsum_group_needed(2 * j, gg) || rowsum_group_needed(2 * j + 1, gg)) {
                ok = ok && j >= kTLayout.lo[gg] && j <= kTLayout.hi[gg];
                if (nh < kHIters * 64)
                    hn[nh] = (uint32_t)(2 * j * kRPitch + (gg - 1) * 4) | (uint32_t)t_cell_offset(j, gg) << 16;
                nh++;
            }
    for (int p = 0; 2 * p < kBlRows; p++)
        for (int gg = 1; gg <= 10; gg++)
            if (blur_group_needed(2 * p, gg) || blur_group_needed(2 * p + 1, gg)) {
                ok = ok && p >= kTLayout.lo[gg] && p + 3 <= kTLayout.hi[gg];      // all four pairs are stored
                if (nv < kVIters * 64)
                    vn[nv] = (uint32_t)t_cell_offset(p, gg) | (uint32_t)(2 * p * kTPitch + 4 * gg) << 16;
                nv++;
            }
    t.nh = nh;
    t.nv = nv;
    // every item sits on at least one slot, and no slot names an item that does not exist
    bool hSeen[kHIters * 64] = {}, vSeen[kVIters * 64] = {};
    for (int s = 0; s < kHIters * 64; s++) {
        ok = ok && kRowItemSlot[s] < nh;
        t.h[s] = hn[kRowItemSlot[s]];
        hSeen[kRowItemSlot[s]] = true;
    }
    for (int s = 0; s < kVIters * 64; s++) {
        ok = ok && kColItemSlot[s] < nv;
        t.v[s] = vn[kColItemSlot[s]];
        vSeen[kColItemSlot[s]] = true;
    }
    for (int i = 0; i < nh && i < kHIters * 64; i++) ok = ok && hSeen[i];
    for (int i = 0; i < nv && i < kVIters * 64; i++) ok = ok && vSeen[i];
    t.ok = ok;
    return t;
}
constexpr ItemTables kItems = make_item_tables();
static_assert(kItems.nh <= kHIters * 64 && kItems.nv <= kVIters * 64, "work item tables overflow");
static_assert(kItems.nh == kTItems && kItems.ok, "a work item points outside the stored part of T, or the slot tables miss an item");
// What a lane fetches before its keypoint's pixels arrive, 64 bytes per lane in one place (four 16-byte loads off one
// base address): its work items of both blur passes and its intensity-centroid weights.
struct LaneTable {
    uint32_t w[64][16];     // per lane: h[0..2], v[0..2], two spare words, momw[0..3] as (.x, .y) pairs
};
constexpr LaneTable make_lane_table() {
    LaneTable t{};
    const ItemTables it = make_item_tables();
    const MomentWeights mw = make_moment_weights();
    for (int lane = 0; lane < 64; lane++) {
        for (int q = 0; q < kHIters; q++) t.w[lane][q] = it.h[64 * q + lane];
        for (int q = 0; q < kVIters; q++) t.w[lane][3 + q] = it.v[64 * q + lane];
        for (int q = 0; q < 4; q++) {
            t.w[lane][8 + 2 * q] = mw.w[q][lane][0];
            t.w[lane][9 + 2 * q] = mw.w[q][lane][1];
        }
    }
    return t;
}
static_assert(kHIters == 3 && kVIters == 3, "the lane table has three item words per pass");
__device__ __attribute__((aligned(64))) const LaneTable c_lane = make_lane_table();


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
    // both branches of the original divide the smaller magnitude by (the larger + DBL_EPSILON) and run the same
    // polynomial; only the last step differs, so the quotient and the polynomial are evaluated once
    const float c = __fdiv_rn(fminf(ax, ay), __fadd_rn(fmaxf(ax, ay), (float)2.2204460492503131e-16));
    const float c2 = __fmul_rn(c, c);
    const float p = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    float a = ax >= ay ? p : __fsub_rn(90.f, p);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sin and cos of x in [0, 8] to within an ulp of double; rounded to float afterwards this is the correctly rounded
// float value in all but vanishingly rare cases. The reference calls the float overloads of cos / sin
// (src/ORBextractor.cc:103-104 under `using namespace std`), i.e. cosf / sinf of its C runtime, whose last bit
// differs between runtimes; the correctly rounded value is the runtime-independent choice. Quadrant reduction with a two-part pi/2, then the
// minimax polynomials of fdlibm's __kernel_sin / __kernel_cos on |r| <= pi/4.
// a * b + k with the constant k held in scalar registers (v_fma_f64 takes one SGPR source; left to itself the
// compiler picks v_fmac_f64 and spends two v_mov per coefficient to seed the accumulator)
__device__ __forceinline__ double fma_k(double a, double b, double k) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

__device__ __forceinline__ void sincos_small(double x, double* s, double* c) {
    const double kq = __builtin_rint(x * 6.36619772367581382433e-01);             // x * 2/pi, 0..5
    double r = __builtin_fma(-kq, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-kq, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma_k(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma_k(z, ps, 2.75573137070700676789e-06);
    ps = fma_k(z, ps, -1.98412698298579493134e-04);
    ps = fma_k(z, ps, 8.33333333332248946124e-03);
    ps = fma_k(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(z * r, ps, r);
    double pc = fma_k(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma_k(z, pc, -2.75573143513906633035e-07);
    pc = fma_k(z, pc, 2.48015872894767294178e-05);
    pc = fma_k(z, pc, -1.38888888888741095749e-03);
    pc = fma_k(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + z * z * pc);
    const int n = (int)kq;
    const double so = (n & 1) ? cs : sn, co = (n & 1) ? sn : cs;
    *s = (n & 2) ? -so : so;
    *c = ((n + 1) & 2) ? -co : co;
}

// Sum over the wave; every lane must be active. The total is returned to all lanes.
__device__ __forceinline__ int wave_sum(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);  // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);  // row_mirror: every lane holds its row's sum
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}

// Orders a wave's own LDS traffic (one wave's DS instructions execute in order).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The top bytes of four 32-bit sums as four bytes (three v_perm_b32). The column pass keeps its sums shifted
// left by 8 and accumulates with saturation, so a sum's top byte is min(255, sum >> 16) of the unshifted sum.
__device__ __forceinline__ uint32_t pack4_top(const uint4& v) {
    const uint32_t lo = __builtin_amdgcn_perm(v.y, v.x, 0x0c0c0703u);     // bytes 0, 1
    const uint32_t hi = __builtin_amdgcn_perm(v.w, v.z, 0x0c0c0703u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);   // low halves of lo and hi
}

// The seven taps laid over the twelve bytes A B C (columns 4g-4 .. 4g+7) for the four output columns 4g+j: column
// 4g+j takes bytes j+1 .. j+7, so its weights are the tap vector shifted by j+1 bytes and cut into three dwords.
// The products go through v_dot4_u32_u8 with the weights as scalar operands: no realignment of the pixels.
struct RowTaps {
    uint32_t a0, b0;            // j = 0: bytes 1..7 (A: -,K0,K1,K2; B: K3..K6)
    uint32_t a1, b1, c1;        // j = 1: bytes 2..8
    uint32_t a2, b2, c2;        // j = 2: bytes 3..9
    uint32_t b3, c3;            // j = 3: bytes 4..10 (B: K0..K3; C: K4,K5,K6,-)
};
__device__ __forceinline__ RowTaps make_row_taps(int blurTaps) {
    const uint32_t k0 = 18, k1 = 34, k2 = blurTaps == ORBX_BLUR_CV_4_0_0 ? 49 : 48,
                   k3 = blurTaps == ORBX_BLUR_CV_4_0_0 ? 55 : 56;
    auto w = [](uint32_t x, uint32_t y, uint32_t z, uint32_t u) { return x | (y << 8) | (z << 16) | (u << 24); };
    RowTaps t;
    t.a0 = w(0, k0, k1, k2);  t.b0 = w(k3, k2, k1, k0);
    t.a1 = w(0, 0, k0, k1);   t.b1 = w(k2, k3, k2, k1);  t.c1 = w(k0, 0, 0, 0);
    t.a2 = w(0, 0, 0, k0);    t.b2 = w(k1, k2, k3, k2);  t.c2 = w(k1, k0, 0, 0);
    t.b3 = w(k0, k1, k2, k3); t.c3 = w(k2, k1, k0, 0);
    return t;
}
__device__ __forceinline__ uint32_t dot4(uint32_t x, uint32_t w, uint32_t acc) {
    return __builtin_amdgcn_udot4(x, w, acc, false);
}

}  // namespace

// Exclusive prefix of every frame's per-level keypoint counts, and the frame totals the caller receives
// (batched calls; a call of a few frames lets every wave sum the counts itself and saves the launch).
__global__ __launch_bounds__(64) void sel_prefix_kernel(FrameSet fs, int nlevels, int nframes,
                                                        int32_t* __restrict__ countsOut) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= nframes) return;
    int run = 0;
    for (int l = 0; l < nlevels; l++) {
        fs.selBefore[(long long)f * nlevels + l] = run;
        run += fs.selCount[(long long)f * nlevels + l];
    }
    countsOut[f] = run;
}

// 5120 bytes of LDS per wave let 32 waves share a CU (eight per SIMD, which need <= 64 VGPRs: why the pattern rows
// are fetched only after the row pass). Waves are independent, so the workgroup size (WAVES) is free: see
// launch_orient_desc.
template <bool OWN_PREFIX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void orient_desc_kernel(FrameSet fs, PyramidGeom g,
                                                                           int blurTaps,
                                                                           orbx_keypoint* __restrict__ kpsOut,
                                                                           uint8_t* __restrict__ descOut, int cap,
                                                                           int32_t* __restrict__ countsOut,
                                                                           uint8_t* __restrict__ blDump) {
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[WAVES * kWaveLds];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps the set-up scalar
    const int f = blockIdx.x;                     // frame-fastest grid: one frame per XCD
    const int slot = blockIdx.y * WAVES + wv;             // index into the frame's selected regions
    if (slot >= g.totalSel) return;               // the waves of a block never synchronise with each other
    // fetched together with the counts below (the slot exists even when it holds no keypoint), so the patch
    // loads wait for one round trip instead of two
    const uint32_t packed = fs.sel[(long long)f * g.totalSel + slot];
    // the entry's level, its index there and the level's geometry from the host-made table (one scalar load, issued
    // together with the one above)
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    const u32x8 rec = *reinterpret_cast<const u32x8 __attribute__((address_space(4)))*>(
        reinterpret_cast<uintptr_t>(fs.selRec + slot));
    const int l = (int)(rec[0] & 0xFFu), k = (int)(rec[0] >> 8);
    const int levelW = (int)(rec[3] & 0xFFFFu), levelH = (int)(rec[3] >> 16);
    // Position in the frame's output: levels are concatenated in order (:1067-1096), so a keypoint sits behind
    // all keypoints of the lower levels. Batched calls read that number from sel_prefix_kernel's table (two
    // loads); in a call of a few frames, where one more dependent launch costs more than some scalar work per
    // wave, the frame's short row of per-level counts is fetched whole and summed here without branches.
    int before = 0, mine = 0;
    if (OWN_PREFIX) {
        const int32_t* cnt = fs.selCount + (long long)f * g.nlevels;
        int total = 0, c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = i < g.nlevels ? cnt[i] : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            before += i < l ? c[i] : 0;
            mine = i == l ? c[i] : mine;
            total += c[i];
        }
        for (int i = 8; i < g.nlevels; i++) {            // pyramids deeper than eight levels: rare
            const int ci = cnt[i];
            before += i < l ? ci : 0;
            mine = i == l ? ci : mine;
            total += ci;
        }
        if (slot == 0 && lane == 0) countsOut[f] = total;  // the frame's keypoint count, as the caller receives it
    } else {
        mine = fs.selCount[(long long)f * g.nlevels + l];
        before = fs.selBefore[(long long)f * g.nlevels + l];
    }
    const int outIdx = before + k;
    if (k >= mine || outIdx >= cap) return;
    uint8_t* R = s_mem + wv * kWaveLds;                                        // raw patch, later Bl
    uint8_t* Tb = R + kTOffset;                                                // packed row sums
    uint32_t* R32 = reinterpret_cast<uint32_t*>(R);

    // work items of both blur passes (independent of everything else: their latency hides behind the staging)
    uint32_t hItem[kHIters], vItem[kVIters];
    uint2 momw[4];                                 // this lane's intensity-centroid weights, fetched just as early
    {
        const uint4* lt = reinterpret_cast<const uint4*>(&c_lane.w[lane][0]);
        const uint4 a = lt[0], b = lt[1], c2 = lt[2], d = lt[3];
        hItem[0] = a.x, hItem[1] = a.y, hItem[2] = a.z;
        vItem[0] = a.w, vItem[1] = b.x, vItem[2] = b.y;
        momw[0] = make_uint2(c2.x, c2.y), momw[1] = make_uint2(c2.z, c2.w);
        momw[2] = make_uint2(d.x, d.y), momw[3] = make_uint2(d.z, d.w);
    }

    const int kx = cand_x(packed) + kBorder;   // level coordinates (:835-836)
    const int ky = cand_y(packed) + kBorder;
    const uint8_t* img;
    int rowStride;
    if (l == 0) {
        img = fs.img + (long long)f * fs.imgFrameStride;
        rowStride = (int)fs.imgRowStride;
    } else {
        img = fs.pyr + (long long)f * g.pyrBytes + rec[1];
        rowStride = (int)rec[2];
    }
    // ---- stage the raw patch: LDS column j <-> image column kx-23+j, LDS row r <-> image row ky-21+r ----
    // Columns kx-21 .. kx+21 are used. When they all lie inside the level the rows come in as dwords: lane
    // (row slot, dword d) reads the two aligned dwords around its four bytes; dwords that stick out of the
    // row are clamped into it (they only hold unused columns). Rows outside the level are reflected.
    if (kx - kPR >= 0 && kx + kPR < levelW) {
        const int xa = (kx - kCx) & ~3;
        const uint32_t shift = (uint32_t)((kx - kCx) & 3);
        const int d = lane & 15, rs = lane >> 4;
        const int colLo = min(max(xa + 4 * d, 0), rowStride - 4), colHi = min(max(xa + 4 * d + 4, 0), rowStride - 4);
        const bool rowsInside = ky - kPR >= 0 && ky - kPR + kRowsAlloc <= levelH;
        uint32_t lo[kRowsAlloc / 4], hi[kRowsAlloc / 4];
        // Rows come through a buffer resource over the level: the row offset is scalar, the two dword offsets are
        // this lane's constants, so a load needs no address arithmetic (and cannot fault).
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, levelH * rowStride, 0x00027000);
        if (rowsInside) {
            const int offLo = rs * rowStride + colLo, offHi = rs * rowStride + colHi;
            const int row0 = (ky - kPR) * rowStride;                         // wave-uniform
#pragma unroll
            for (int q = 0; q < kRowsAlloc / 4; q++) {
                lo[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offLo, row0 + 4 * q * rowStride, 0);
                hi[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, offHi, row0 + 4 * q * rowStride, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < kRowsAlloc / 4; q++) {
                const int sy = reflect101(ky - kPR + rs + 4 * q, levelH);
                lo[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, sy * rowStride + colLo, 0, 0);
                hi[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, sy * rowStride + colHi, 0, 0);
            }
        }
        if (d < 12) {
#pragma unroll
            for (int q = 0; q < kRowsAlloc / 4; q++)
                R32[(rs + 4 * q) * 12 + d] = __builtin_amdgcn_alignbyte(hi[q], lo[q], shift);
        }
    } else {
        // within two pixels of the level's left/right border: bytes,
        // BORDER_REFLECT_101 coordinates
        for (int i = lane; i < kRowsAlloc * kRPitch; i += 64) {
            const int r = i / kRPitch, c = i - r * kRPitch;
            const int sy = reflect101(ky - kPR + r, levelH), sx = reflect101(kx - kCx + c, levelW);
            R[i] = img[(long long)sy * rowStride + sx];
        }
    }
    wave_sync();

    // ---- intensity centroid on the unblurred patch (:69-96) ----
    // m01 = sum v*I, m10 = sum u*I over the disc; each lane takes dwords (four columns) of the 31x31
    // window (columns 8..39 of R): row sums with 0/1 byte weights, sum (u+15)*I with byte weights
    int s0 = 0, s1 = 0, m01 = 0;
    {
        const int mrow = moment_row(lane);
        const uint32_t* Rw = R32 + (kPR - kHalfPatch) * 12 + (kCx - kHalfPatch) / 4 + mrow * 12 + (lane & 7);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t D = Rw[q * 8 * 12];
            const uint2 w = momw[q];
            const uint32_t rs = __builtin_amdgcn_udot4(D, w.x, 0u, false);
            s0 += (int)rs;
            m01 += __mul24(mrow + 8 * q - kHalfPatch, (int)rs);
            s1 = (int)__builtin_amdgcn_udot4(D, w.y, (uint32_t)s1, false);
        }
    }
    const int m10 = wave_sum(s1 - kHalfPatch * s0);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- row pass: T[r][j] = sum_c K[c] * R[r][j-3+c]; a lane makes rows 2j, 2j+1 of four columns ----
    const RowTaps kt = make_row_taps(blurTaps);
    // all three items' pixels are requested before the first is used: one LDS round trip for the pass, not three
    uint32_t rowIn[kHIters][2][3];
#pragma unroll
    for (int q = 0; q < kHIters; q++) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(R + (hItem[q] & 0xFFFFu));
#pragma unroll
        for (int h = 0; h < 2; h++) {
            rowIn[q][h][0] = p[12 * h];
            rowIn[q][h][1] = p[12 * h + 1];
            rowIn[q][h][2] = p[12 * h + 2];
        }
    }
#pragma unroll
    for (int q = 0; q < kHIters; q++) {
        uint4 o[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t A = rowIn[q][h][0], B = rowIn[q][h][1], C = rowIn[q][h][2];   // columns 4g-4 .. 4g+7
            o[h].x = dot4(A, kt.a0, dot4(B, kt.b0, 0u));
            o[h].y = dot4(A, kt.a1, dot4(B, kt.b1, dot4(C, kt.c1, 0u)));
            o[h].z = dot4(A, kt.a2, dot4(B, kt.b2, dot4(C, kt.c2, 0u)));
            o[h].w = dot4(B, kt.b3, dot4(C, kt.c3, 0u));
        }
        *reinterpret_cast<uint4*>(Tb + (hItem[q] >> 16)) =
            make_uint4(__builtin_amdgcn_perm(o[1].x, o[0].x, 0x05040100u), __builtin_amdgcn_perm(o[1].y, o[0].y, 0x05040100u),
                       __builtin_amdgcn_perm(o[1].z, o[0].z, 0x05040100u), __builtin_amdgcn_perm(o[1].w, o[0].w, 0x05040100u));
    }
    wave_sync();

    float4 patv[4];                                // this lane's four pattern pairs; they arrive during the column pass
#pragma unroll
    for (int q = 0; q < 4; q++) patv[q] = reinterpret_cast<const float4*>(c_patternf_tab.v)[64 * q + lane];

    // ---- column pass: Bl[y][j] = min(255, (sum_r K[r] * T[y+r][j] + 32768) >> 16), y = patch row - 3 ----
    {
        // Taps and rounding constant times 256: a 16.16 sum of at least 256.0 then no longer fits 32 bits, and the
        // saturating form of v_dot2_u32_u16 stops at 0xFFFFFFFF, so bits 24..31 of the result are min(255, sum >> 16)
        // (the partial sums only grow, so a sum that fits never saturates on the way).
        const uint32_t k0 = 18u << 8, k1 = 34u << 8, k2 = (blurTaps == ORBX_BLUR_CV_4_0_0 ? 49u : 48u) << 8,
                       k3 = (blurTaps == ORBX_BLUR_CV_4_0_0 ? 55u : 56u) << 8;
        const uint32_t we[4] = {k0 | (k1 << 16), k2 | (k3 << 16), k2 | (k1 << 16), k0};
        const uint32_t wo[4] = {k0 << 16, k1 | (k2 << 16), k3 | (k2 << 16), k1 | (k0 << 16)};
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        // an item's four row pairs are requested while the item before it is computed
        uint4 tin[2][4];
#pragma unroll
        for (int r = 0; r < 4; r++) tin[0][r] = *reinterpret_cast<const uint4*>(Tb + (vItem[0] & 0xFFFFu) + r * 16);
#pragma unroll
        for (int q = 0; q < kVIters; q++) {
            if (q + 1 < kVIters) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    tin[(q + 1) & 1][r] = *reinterpret_cast<const uint4*>(Tb + (vItem[q + 1] & 0xFFFFu) + r * 16);
            }
            uint4 e = make_uint4(32768u << 8, 32768u << 8, 32768u << 8, 32768u << 8), o = e;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint4 t = tin[q & 1][r];
                const us2 a = __builtin_bit_cast(us2, we[r]), b = __builtin_bit_cast(us2, wo[r]);
                e.x = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.x), a, e.x, true);
                e.y = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.y), a, e.y, true);
                e.z = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.z), a, e.z, true);
                e.w = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.w), a, e.w, true);
                o.x = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.x), b, o.x, true);
                o.y = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.y), b, o.y, true);
                o.z = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.z), b, o.z, true);
                o.w = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, t.w), b, o.w, true);
            }
            uint32_t* bl = reinterpret_cast<uint32_t*>(R + (vItem[q] >> 16));
            bl[0] = pack4_top(e);
            bl[kTPitch / 4] = pack4_top(o);
        }
    }
    wave_sync();
    if (blDump) {
        // parity probe: this keypoint's blurred patch as the descriptor stage reads it, kBlRows x kTPitch bytes
        // (only the part the rotated pattern can reach has been computed)
        uint8_t* out = blDump + ((long long)f * cap + outIdx) * (kBlRows * kTPitch);
        for (int i = lane; i < kBlRows * kTPitch; i += 64) out[i] = R[i];
    }

    // ---- steered BRIEF on the blurred patch (:99-139) ----
    const float rad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.f));
    double sd, cd;
    sincos_small((double)rad, &sd, &cd);
    const float a = (float)cd, b = (float)sd;
    unsigned long long* desc64 = reinterpret_cast<unsigned long long*>(descOut + ((long long)f * cap + outIdx) * 32);
    const uint8_t* Bl = R + (kPR - 3) * kTPitch + kCx;   // blurred row y <-> patch row y + 3
    // the table holds x0, x1, y0, y1 per pair so that both points go through packed float instructions
    // (IEEE single products and sums, rounded separately, as the reference's scalar code)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 aa = {a, a}, bb = {b, b};
    const f32x2 kMagic = {12582912.f, 12582912.f};                                       // 0x4B400000
    constexpr uint32_t kMagicBias = 0x400000u * (uint32_t)kTPitch + 0x4B400000u;        // what the raw bits add
#pragma unroll
    for (int round = 0; round < 4; round++) {
        const float4 pt = patv[round];
        const f32x2 px = {pt.x, pt.y}, py = {pt.z, pt.w};
        const f32x2 rr = px * bb + py * aa;              // x*b + y*a (:110-112)
        const f32x2 cc = px * aa - py * bb;              // x*a - y*b
        // cvRound (:111): adding 1.5 * 2^23 rounds to the nearest integer, ties to even, and leaves it in the
        // low mantissa bits (|value| < 2^22); v_mad_u32_u24 then forms 44*row + column on the raw bit patterns
        const f32x2 rm = rr + kMagic, cm = cc + kMagic;
        const uint32_t i0 = __umul24(__float_as_uint(rm.x), (uint32_t)kTPitch) + __float_as_uint(cm.x);
        const uint32_t i1 = __umul24(__float_as_uint(rm.y), (uint32_t)kTPitch) + __float_as_uint(cm.y);
        const int t0 = Bl[(int)(i0 - kMagicBias)];
        const int t1 = Bl[(int)(i1 - kMagicBias)];
        const unsigned long long bits = __builtin_amdgcn_ballot_w64(t0 < t1);
        if (OWN_PREFIX) {
            // calls of a few frames may write straight into the caller's (pinned host) memory: the record leaves as
            // two wide stores made by neighbouring lanes instead of eleven small ones from lane 0
            // (staged in the row-sum table, which nobody reads any more; R holds the blurred patch being sampled)
            if (lane == 0) reinterpret_cast<unsigned long long*>(Tb)[round] = bits;
        } else if (lane == 0) {
            desc64[round] = bits;   // pair 64*round + j is bit j%8 of byte 8*round + j/8
        }
    }
    if (OWN_PREFIX) {
        if (lane == 0) {
            float* kf = reinterpret_cast<float*>(Tb + 32);
            kf[0] = l ? __fmul_rn((float)kx, __uint_as_float(rec[4])) : (float)kx;
            kf[1] = l ? __fmul_rn((float)ky, __uint_as_float(rec[4])) : (float)ky;
            kf[2] = __uint_as_float(rec[5]);
            kf[3] = angle;
            kf[4] = (float)cand_score(packed);
            reinterpret_cast<int32_t*>(kf)[5] = l;
            reinterpret_cast<int32_t*>(kf)[6] = -1;
        }
        wave_sync();
        const uint32_t word = reinterpret_cast<const uint32_t*>(Tb)[lane & 15];
        if (lane < 8) reinterpret_cast<uint32_t*>(desc64)[lane] = word;
        else if (lane < 15) reinterpret_cast<uint32_t*>(kpsOut + ((long long)f * cap + outIdx))[lane - 8] = word;
        return;
    }
    if (lane == 0) {
        orbx_keypoint kp;
        // coordinates are scaled after the descriptors, and only above level 0 (:1087-1093)
        kp.x = l ? __fmul_rn((float)kx, __uint_as_float(rec[4])) : (float)kx;
        kp.y = l ? __fmul_rn((float)ky, __uint_as_float(rec[4])) : (float)ky;
        kp.size = __uint_as_float(rec[5]);
        kp.angle = angle;
        kp.response = (float)cand_score(packed);
        kp.octave = l;
        kp.class_id = -1;
        kpsOut[(long long)f * cap + outIdx] = kp;
    }
}

hipError_t launch_orient_desc(const FrameSet& fs, const PyramidGeom& g, int nframes, int blurTaps,
                              orbx_keypoint* kps, uint8_t* desc, int cap, int32_t* counts, hipStream_t s,
                              uint8_t* blDump) {
    // level 0 is read with dword loads like the other levels: the caller of this launcher has made it 4-byte aligned
    if ((uintptr_t)fs.img % 4 || fs.imgRowStride % 4 || fs.imgFrameStride % 4) return hipErrorInvalidValue;
    // Waves per workgroup, from measurements with three contexts in flight: on 752x480 frames two waves are 2-3 %
    // ahead of four (smaller pieces fit sooner into what the other streams' kernels leave free), on 1241x376 the
    // two tie, on 1920x1080 four are 6-13 % ahead (the four keypoints of a workgroup are neighbours in the
    // quadtree's order and share patch rows in L1, which matters once level 0 no longer sits in L2).
    const bool large = (long long)g.lv[0].w * g.lv[0].h > 600000;
    auto launch = [&](auto kernel, int waves) {
        hipLaunchKernelGGL(kernel, dim3(nframes, (g.totalSel + waves - 1) / waves, 1), dim3(64 * waves, 1, 1), 0, s, fs, g,
                           blurTaps, kps, desc, cap, counts, blDump);
    };
    if (nframes <= 8 && g.totalSel > 0) {
        if (large) launch(orient_desc_kernel<true, 4>, 4); else launch(orient_desc_kernel<true, 2>, 2);
    } else {
        hipLaunchKernelGGL(sel_prefix_kernel, dim3((nframes + 63) / 64), dim3(64), 0, s, fs, g.nlevels, nframes, counts);
        if (g.totalSel > 0) {
            if (large) launch(orient_desc_kernel<false, 4>, 4); else launch(orient_desc_kernel<false, 2>, 2);
        }
    }
    return hipGetLastError();
}

}  // namespace orbx
