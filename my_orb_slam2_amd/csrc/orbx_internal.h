// Internal declarations shared by the host side and the gfx950 kernels of liborbx.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "orbx.h"

namespace orbx {

constexpr int kEdge = 19;        // EDGE_THRESHOLD (src/ORBextractor.cc:66)
constexpr int kBorder = 16;      // EDGE_THRESHOLD - 3: first column/row FAST may look at (:765)
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE (:65)
constexpr int kPatchSize = 31;   // PATCH_SIZE (:64)
constexpr float kCellSize = 30;  // W (:761)
constexpr int kSelPad = 4;       // a level keeps at most max(quota + 3, 4 * nIni) nodes; see LevelGeom::selCap

// Geometry of one pyramid level for the current image size. Passed to kernels by value.
struct LevelGeom {
    int w, h;               // level size (:1104)
    int stride;             // row pitch in bytes inside the pyramid buffer (levels >= 1)
    long long offset;       // byte offset of the level inside one frame's pyramid buffer (levels >= 1)
    int nCols, nRows;       // FAST cell grid (:776-777); 0 when the level is too small for one cell
    uint32_t colsMagic;     // floor(2^32 / nCols) + 1 for nCols > 1: cell / nCols = mulhi(cell, colsMagic) for any cell index of a level
    int wCell, hCell;       // (:778-779)
    int cellBase;           // first cell of this level in a frame's cell arrays
    int slotCap;            // candidate capacity of one cell
    long long slotBase;     // first slot entry of this level in a frame's slot / candidate arrays
    int quota;              // mnFeaturesPerLevel[level]
    int selBase;            // first entry of this level's selected-keypoint region (selCap long)
    int selCap;             // most keypoints the quadtree can keep: the first sweep splits all nIni root nodes
                            // whatever the quota (up to 4 * nIni nodes), later splits stop within 3 of the quota
    int nIni;               // root nodes of the quadtree (:535)
    float hX;               // root width (:537)
    float scale;            // mvScaleFactor[level]
    float kpSize;           // (float)(int)(31*scale) (:829)
};

struct PyramidGeom {
    int nlevels;
    int width, height;      // level-0 size
    int totalCells;         // per frame
    long long totalSlots;   // per frame
    int totalSel;           // per frame
    long long pyrBytes;     // per frame, levels >= 1
    int nodeCap;            // quadtree node table capacity
    int tilePitch;          // LDS row pitch of a staged FAST cell: 40, 48 or 80 bytes
    int maxTileBytes;       // LDS bytes of the largest FAST cell tile
    int maxScanPix;         // most scanned pixels in one cell
    int maxScoreBytes;      // LDS bytes of the largest cell score map
    LevelGeom lv[ORBX_MAX_LEVELS];
};

// One FAST cell of a frame as the kernel needs it, made on the host once per image size: a wave fetches its cell's
// record with one scalar 32-byte load instead of walking the level table and redoing the rectangle arithmetic
// (src/ORBextractor.cc:783-798) with scalar instructions.
struct CellRecord {
    uint32_t ini;           // iniX | iniY << 16 (level coordinates of the cell's first column and row)
    uint32_t max;           // maxX | maxY << 16 (one past its last column and row, FAST margins included)
    uint32_t slotOff;       // first entry of the cell inside a frame's slot array
    uint32_t levelOff;      // byte offset of the cell's level inside a frame's pyramid buffer (levels >= 1)
    uint32_t stride;        // row pitch of that level in bytes (levels >= 1; level 0 has the caller's pitch)
    uint32_t levelH;        // level | rows of the level << 8
    uint32_t slotCap;       // candidate capacity of the cell | 1 << 31 when the reference skips the cell
    uint32_t pad;
};
static_assert(sizeof(CellRecord) == 32, "one s_load_dwordx8 per cell");

// One entry of a frame's selected-keypoint regions as the descriptor kernel needs it, made on the host once per image
// size: which level the entry belongs to, its index there, and that level's geometry (one scalar 32-byte load per
// wave instead of a walk over the level table followed by loads of the level's fields).
struct SelRecord {
    uint32_t levelK;        // level | index of the entry inside the level's region << 8
    uint32_t levelOff;      // byte offset of the level inside a frame's pyramid buffer (levels >= 1)
    uint32_t stride;        // row pitch of the level in bytes (levels >= 1)
    uint32_t wh;            // level width | height << 16
    float scale;            // mvScaleFactor[level]
    float kpSize;           // keypoint size at that level
    uint32_t pad[2];
};
static_assert(sizeof(SelRecord) == 32, "one s_load_dwordx8 per keypoint");

// Device pointers of one batched call. Passed to kernels by value.
struct FrameSet {
    const uint8_t* img;     // level 0: caller's frames
    long long imgRowStride, imgFrameStride;
    uint8_t* pyr;           // levels >= 1
    int32_t* cellCount;     // [frame][totalCells]
    const CellRecord* cells;  // [totalCells] of the current image size (the same for every frame)
    const SelRecord* selRec;  // [totalSel] level and level geometry of every entry of a frame's selected-keypoint regions
    uint32_t* slots;        // [frame][totalSlots] packed candidates, grouped per cell
    uint32_t* cand;         // [frame][totalSlots] candidates of a level gathered in reference order
    uint16_t* nid;          // [frame][totalSlots] quadtree node of each candidate
    uint32_t* sel;          // [frame][totalSel] packed selected keypoints
    int32_t* selCount;      // [frame][nlevels]
    int32_t* selBefore;     // [frame][nlevels] keypoints of the frame's lower levels (exclusive prefix of selCount)
};

// Bilinear resize tables of one level (cv::resize INTER_LINEAR, 8U fixed point).
struct ResizeTab {
    const int32_t* xofs;    // [dw]
    const int32_t* xentry;  // [dw rounded up to 4][2]: {xofs (column % 4 == 0) or v_perm selector, ialpha0 | ialpha1 << 16}
    int maxSpan;            // largest xofs[4g + k] - xofs[4g] over all groups of four columns
    const int16_t* ialpha;  // [dw][2]
    const int32_t* yofs;    // [dh]
    const int16_t* ibeta;   // [dh][2]
    int xmax;
};

// ---- several pyramid levels in one launch (pyramid.hip) ----
// A group is a run of consecutive destination levels made by one kernel: a workgroup owns a tile of the group's
// last level and works its way down to it from the group's source level, keeping the tiles of the levels in
// between in LDS (each is also written to the pyramid buffer, every pixel by exactly one workgroup).
constexpr int kFusedMaxLevels = 8;
struct FusedRange {             // per (level of the group, tile column) or (level, tile row)
    int lo, hi;                 // columns (rows) of the level this tile computes; columns in multiples of four
    int ownLo, ownHi;           // the part of them it writes to the pyramid buffer
};
struct FusedLevel {
    int dw, dh;                 // destination level
    int sw, sh;                 // its source level
    long long dstOffset;        // of the destination level inside a frame's pyramid buffer
    int dstStride;
    const int32_t* xentry;      // as ResizeTab
    const int32_t* yofs;
    const uint32_t* beta2;      // ibeta read as one dword per row
};
struct FusedGroup {
    int nl, ntx, nty;
    int bufBytes;               // bytes of each of the two LDS tile buffers
    const uint8_t* src;         // the group's source level of frame 0
    long long srcRowStride, srcFrameStride;
    const FusedRange* cols;     // [nl][ntx]
    const FusedRange* rows;     // [nl][nty]
    FusedLevel lv[kFusedMaxLevels];
};
struct FusedPlan {
    std::vector<FusedGroup> groups;         // cols / rows hold offsets into `ranges` until the plan is uploaded
    std::vector<int> firstLevel;            // source level of each group
    std::vector<FusedRange> ranges;         // all range tables, one after the other
};

// candidate packing: x (12 bits) | y (12 bits) << 12 | score << 24; x, y relative to the 16-px border
__host__ __device__ inline uint32_t pack_cand(int x, int y, int score) {
    return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)score << 24);
}
__host__ __device__ inline int cand_x(uint32_t c) { return (int)(c & 0xFFFu); }
__host__ __device__ inline int cand_y(uint32_t c) { return (int)((c >> 12) & 0xFFFu); }
__host__ __device__ inline int cand_score(uint32_t c) { return (int)(c >> 24); }

// Grid convention of the batched kernels: blockIdx.x = frame, the tile / cell / keypoint index goes in
// blockIdx.y (and .z). Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in
// launch order, x fastest, so with a batch that is a multiple of 8 every block of frame f lands on
// XCD f % 8: neighbouring cells, patches and source rows of one frame then share one L2 instead of
// being fetched by up to eight. Pure performance hint; any placement is correct.

// ---- host-side tables (tables.cpp) ----
struct ScaleTables {
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> perLevel;
    int umax[kHalfPatch + 1];
};
void build_scale_tables(int nfeatures, float scaleFactor, int nlevels, ScaleTables& t);
bool build_geometry(int width, int height, const ScaleTables& t, PyramidGeom& g, std::string& err);
// The FAST cells of a geometry in the order of a frame's cell arrays; false if a field does not fit its record.
bool build_cell_table(const PyramidGeom& g, std::vector<CellRecord>& cells);
struct HostResizeTab {
    std::vector<int32_t> xofs, yofs;
    std::vector<int16_t> ialpha, ibeta;
    int xmax;
};
void build_resize_table(int sw, int sh, int dw, int dh, HostResizeTab& t);

// Where the levels of the frames of a context's last batched call live (for kernels that read the
// pyramids of two contexts, e.g. the stereo refinement).
struct LevelView {
    const uint8_t* base;            // level pixel (0,0) of frame 0
    long long rowStride, frameStride;
    int w, h;
};
struct PyramidView {
    int nlevels;
    LevelView lv[ORBX_MAX_LEVELS];
};

// ---- error plumbing ----
void set_error(const char* fmt, ...);

// ---- kernel launchers (one per .hip file) ----
hipError_t launch_repack_frames(const uint8_t* src, long long srcRowStride, long long srcFrameStride, uint8_t* dst,
                                long long dstRowStride, long long dstFrameStride, int w, int h, int nframes,
                                hipStream_t s);
hipError_t launch_pyramid(const FrameSet& fs, const PyramidGeom& g, const ResizeTab* tabs, int nframes,
                          hipStream_t s, const FusedPlan* fused = nullptr);
struct HostResizeTab;
// Plans the grouped launches for a geometry (tileW x tileH = tile of a group's last level). Leaves the plan empty
// when a level does not qualify (step between levels above 2, tiles wider than the kernel's lanes).
void plan_fused_pyramid(const PyramidGeom& g, const std::vector<HostResizeTab>& ht, const ResizeTab* devTabs,
                        int tileW, int tileH, int groupLevels, FusedPlan& plan);
bool check_fused_plan(const PyramidGeom& g, const std::vector<HostResizeTab>& ht, const FusedPlan& plan,
                      std::string& err);
// Largest source-column step inside a group of four destination columns (ResizeTab::maxSpan).
int resize_max_span(const HostResizeTab& t);
hipError_t launch_fast_cells(const FrameSet& fs, const PyramidGeom& g, int nframes, int iniTh, int minTh,
                             hipStream_t s);
hipError_t launch_octree(const FrameSet& fs, const PyramidGeom& g, int nframes, hipStream_t s);
hipError_t launch_orient_desc(const FrameSet& fs, const PyramidGeom& g, int nframes, int blurTaps,
                              orbx_keypoint* kps, uint8_t* desc, int cap, int32_t* counts, hipStream_t s,
                              uint8_t* blDump = nullptr);
// Shape of one keypoint's blurred patch in the probe output of launch_orient_desc (blDump): rows x row pitch, with
// the keypoint at (row, column) = (kBlurPatchCy, kBlurPatchCx).
constexpr int kBlurPatchRows = ORBX_BLUR_PATCH_ROWS, kBlurPatchPitch = ORBX_BLUR_PATCH_PITCH,
              kBlurPatchCy = ORBX_BLUR_PATCH_CY, kBlurPatchCx = ORBX_BLUR_PATCH_CX;
size_t octree_lds_bytes(const PyramidGeom& g);

// Results go to idx / d1 / d2, or, when hits != nullptr, to nq packed orbx_knn2_hit records.
hipError_t launch_knn2(const uint8_t* q, int64_t nq, const uint8_t* db, int64_t ndb, int32_t init,
                       int64_t indexBase, int64_t* idx, int32_t* d1, int32_t* d2, void* hits, void* scratch,
                       size_t scratchBytes, hipStream_t s);
hipError_t launch_knn2_merge_packed(int nparts, int64_t nq, const void* hits, int32_t init, int64_t* oidx,
                                    int32_t* od1, int32_t* od2, hipStream_t s);
size_t knn2_scratch_bytes(int64_t nq, int64_t ndb);
// The scan with its per-chunk partial results in a fixed device array of the library (no allocation).
int64_t knn2_static_max_partials();
bool knn2_static_fits(int64_t nq, int64_t ndb);
hipError_t knn2_static_init(int device);
hipError_t launch_knn2_static(int device, const uint8_t* q, int64_t nq, const uint8_t* db, int64_t ndb,
                              int32_t init, int64_t indexBase, int64_t* idx, int32_t* d1, int32_t* d2,
                              hipStream_t s);
hipError_t launch_knn2_merge(int nparts, int64_t nq, const int64_t* idx, const int32_t* d1,
                             const int32_t* d2, int32_t init, int64_t* oidx, int32_t* od1, int32_t* od2,
                             hipStream_t s);
hipError_t launch_ratio_test(int64_t nq, const int32_t* d1, const int32_t* d2, int32_t th, int inclusive,
                             float ratio, uint8_t* accept, hipStream_t s);
size_t stereo_band_scratch_bytes(int npairs, int cap);
hipError_t launch_stereo_band(int npairs, int cap, const orbx_keypoint* kl, const uint8_t* dl,
                              const int32_t* nl, const orbx_keypoint* kr, const uint8_t* dr,
                              const int32_t* nr, const float* scale, int nlevels, float maxScale, float maxD,
                              void* work, int32_t* bestIdx, int32_t* bestDist, hipStream_t s);

hipError_t launch_stereo_refine(int npairs, int cap, const PyramidView& left, const PyramidView& right,
                                const orbx_keypoint* kl, const int32_t* nl, const orbx_keypoint* kr,
                                const int32_t* bestIdx, const int32_t* bestDist, const float* scale,
                                const float* invScale, float bf, float maxD, float* uRight, float* depth,
                                int32_t* sadDist, void* work, hipStream_t s);

hipError_t launch_undistort_grid(int nframes, int cap, const orbx_keypoint* kps, const int32_t* counts,
                                 orbx_camera cam, float minX, float minY, float invW, float invH,
                                 orbx_keypoint* kpsUn, int32_t* cellStart, int32_t* cellItems, hipStream_t s);
// One distance row of SearchByBoW: left feature idx1 against count2 right-hand features listed from
// items2[start2]; its distances go to dist[out ..].
struct BowRow {
    int32_t idx1, start2, count2, out;
};
constexpr int kBowTop = 4;               // entries of a row's ordered short list (bow_distance_kernel)
constexpr uint32_t kBowListEnd = 0xFFFFFFFFu, kBowNoList = 0xFFFFFFFEu;
// valid2 (device-readable, may be null): right-hand features that may be matched; the others stay off the lists.
hipError_t launch_bow_distances(int nrows, const BowRow* rows, const uint8_t* d1, const uint32_t* items2,
                                const uint8_t* d2, const uint8_t* valid2, uint16_t* dist, uint32_t* top,
                                hipStream_t s);
// Buffers of the host pass that keep their storage from call to call.
struct BowReplayScratch {
    std::vector<uint8_t> matched2;
    std::vector<std::pair<int, int>> binOf;
};
bool build_bow_rows(const uint32_t* nodes1, const int32_t* start1, const uint32_t* items1, int nn1,
                    const uint8_t* valid1, int n1, const uint32_t* nodes2, const int32_t* start2,
                    const uint32_t* items2, int nn2, int n2, std::vector<BowRow>& rows, long long& npairs);
int replay_search_by_bow(int mode, const std::vector<BowRow>& rows, const uint16_t* dist, const uint32_t* top,
                         const uint32_t* items2, const orbx_keypoint* k1, const orbx_keypoint* k2,
                         const uint8_t* valid2, int n1, int n2, float nnratio, bool checkOri, int32_t* matches,
                         BowReplayScratch& tmp);
// One windowed search over a frame's grid (Frame::GetFeaturesInArea, src/Frame.cc:365-418). r < 0: skipped.
// ur is only read when the searched frame has right-image coordinates.
struct WindowQuery {
    float x, y, r, ur;
    int32_t minLevel, maxLevel;
};
hipError_t launch_window_candidates(int n1, const WindowQuery* queries, const uint8_t* d1, const orbx_keypoint* k2,
                                    const uint8_t* d2, const float* uRight2, const int32_t* cellStart,
                                    const int32_t* cellItems, float minX, float minY, float invW, float invH,
                                    int n2, int capTotal, int32_t* cursor, int32_t* candCount, int32_t* candOffset,
                                    uint32_t* cand, hipStream_t s);
// window_top_kernel: per active query {count, kWinTop ordered entries}; see search_init.hip.
constexpr int kWinTop = 8, kWinStride = kWinTop + 1, kWinThreads = 512;
constexpr uint32_t kWinListEnd = 0xFFFFFFFFu;
hipError_t launch_window_top(int nq, const WindowQuery* queries, const int32_t* queryIndex, const uint8_t* d1,
                             const orbx_keypoint* k2, const uint8_t* d2, const int32_t* cellStart,
                             const int32_t* cellItems, float minX, float minY, float invW, float invH, int n2,
                             uint32_t* cand, uint32_t* out, hipStream_t s);
struct InitReplayScratch {
    std::vector<int> matchedDistance, matches21;
    std::vector<std::pair<int, int>> binOf;
    std::vector<uint32_t> list;
};
// Returns the number of matches, or -1 when a record or a list holds an index outside the searched frame.
int replay_search_for_initialization_top(const orbx_keypoint* k1, int n1, const orbx_keypoint* k2, int n2, int nq,
                                         const int32_t* queryIndex, const uint32_t* records,
                                         const std::function<const uint32_t*(int, int)>& fetch, float* prev,
                                         float nnratio, bool checkOri, int32_t* matches12, InitReplayScratch& tmp);
int replay_search_by_projection_frame(int npts, const int32_t* candCount, const int32_t* candOffset,
                                      const uint32_t* cand, const float* lastAngle, const uint8_t* hasObs,
                                      const orbx_keypoint* k2, const uint8_t* occupied2, int n2, int maxDist,
                                      bool checkOri, int32_t* match2);
int replay_search_by_projection_points(int npts, const int32_t* candCount, const int32_t* candOffset,
                                       const uint32_t* cand, const uint8_t* hasObs, const orbx_keypoint* k2,
                                       const uint8_t* occupied2, int n2, float nnratio, int32_t* match2);
void undistort_point_host(const orbx_camera& cam, float u, float v, float* xo, float* yo);

}  // namespace orbx
