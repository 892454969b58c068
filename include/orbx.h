/*
 * orbx.h — C ABI of the MI355X (gfx950) ORB front end.
 *
 * Drop-in boundary for two pieces of One-sixth/my_orb_slam2 (paths below are relative to
 * my_orb_slam2/ in that repository):
 *   - ORBextractor::operator()(image, mask, keypoints, descriptors)   include/ORBextractor.h:66-68
 *   - ORBmatcher::DescriptorDistance + the brute-force best/second-best scans that call it
 *                                                                      include/ORBmatcher.h:47
 * Plain pointers and sizes only; no C++ or torch types cross this boundary; nothing throws.
 * Every function returns ORBX_OK (0) or a negative ORBX_ERR_* code; orbx_last_error() gives the
 * text of the calling thread's last failure.
 *
 * Threading: one orbx_ctx owns one HIP stream and its device buffers. Different contexts may be
 * used from different threads at the same time (the reference runs its left and right extractor
 * on two threads, src/Frame.cc:97-110); a single context is not re-entrant, exactly like an
 * ORBextractor instance (its mvImagePyramid member is per-call state, include/ORBextractor.h:100).
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_ERR_INVALID (-1)   /* bad argument */
#define ORBX_ERR_HIP (-2)       /* a HIP runtime call failed (no GPU, out of memory, ...) */
#define ORBX_ERR_CAPACITY (-3)  /* output capacity too small; *n holds the required count */
#define ORBX_ERR_STATE (-4)     /* call order: nothing extracted yet, frame index out of range */

#define ORBX_MAX_LEVELS 16
#define ORBX_DESC_BYTES 32

/* Gaussian tap sets of cv::GaussianBlur(7x7, sigma 2) in 8.8 fixed point (src/ORBextractor.cc:1078).
 * OpenCV 4.0.0 (the version the reference pins) rounds each tap on its own: 18 34 49 55 49 34 18.
 * OpenCV >= 4.2 forces the sum to 256: 18 34 48 56 48 34 18. */
#define ORBX_BLUR_CV_4_0_0 0
#define ORBX_BLUR_CV_4_2_PLUS 1

/* Bit-compatible with cv::KeyPoint (28 bytes), so the reference's map file keypoint block
 * (src/KeyFrame.cc:934-942) keeps its layout. */
typedef struct orbx_keypoint {
    float x, y;       /* pt, in level-0 pixel units (level coords * scale, src/ORBextractor.cc:1092) */
    float size;       /* (int)(31 * scale[octave])                      (:829, :838) */
    float angle;      /* intensity-centroid angle in degrees, [0, 360)  (:69-96)    */
    float response;   /* FAST corner score                                           */
    int32_t octave;   /* pyramid level                                  (:837)       */
    int32_t class_id; /* always -1                                                   */
} orbx_keypoint;

/* Constructor arguments of ORBextractor (include/ORBextractor.h:52-53, src/ORBextractor.cc:402-406)
 * plus what a device context needs to size its buffers. */
typedef struct orbx_config {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t device;     /* HIP device ordinal */
    int32_t max_width;  /* largest image accepted (<= 4096) */
    int32_t max_height;
    int32_t max_batch;  /* frames per batched call */
    int32_t blur_taps;  /* ORBX_BLUR_* */
} orbx_config;

typedef struct orbx_ctx orbx_ctx;

const char* orbx_last_error(void);

/* ---- extractor --------------------------------------------------------------------------- */

/* Replaces `new ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)`
 * (src/Tracking.cc:116-122). Fails with ORBX_ERR_HIP when the device is not usable. */
int orbx_create(const orbx_config* cfg, orbx_ctx** out);
void orbx_destroy(orbx_ctx* ctx);

/* Getters of include/ORBextractor.h:70-98 (read by every Frame constructor, src/Frame.cc:86-92).
 * Arrays receive nlevels floats / ints. Host-side only. */
int orbx_get_levels(const orbx_ctx* ctx);
float orbx_get_scale_factor(const orbx_ctx* ctx);
int orbx_get_scale_factors(const orbx_ctx* ctx, float* out);
int orbx_get_inverse_scale_factors(const orbx_ctx* ctx, float* out);
int orbx_get_scale_sigma_squares(const orbx_ctx* ctx, float* out);
int orbx_get_inverse_scale_sigma_squares(const orbx_ctx* ctx, float* out);
int orbx_get_features_per_level(const orbx_ctx* ctx, int32_t* out); /* mnFeaturesPerLevel (:427-438) */
int orbx_get_umax(const orbx_ctx* ctx, int32_t* out16);              /* umax (:446-461) */
/* A capacity no image the context accepts can exceed: sum over levels of max(quota + 3, 4 * nIniMax). A level
 * yields at most quota + 3 keypoints once the node list nears the quota (src/ORBextractor.cc:661-723), but the
 * first sweep splits all nIni = round(width / height) root nodes whatever the quota (:598-657); nIniMax is the
 * nIni of the flattest strip of max_width that still has a row of FAST cells. */
int orbx_keypoint_capacity(const orbx_ctx* ctx);

/* ORBextractor::operator() (src/ORBextractor.cc:1035-1097) on one host image (CV_8UC1 rows of
 * `stride` bytes). The mask argument of the reference is ignored there (include/ORBextractor.h:65)
 * and has no counterpart here. Writes up to `cap` keypoints and cap*32 descriptor bytes to
 * caller-owned host buffers and the number found to *n. An empty image (NULL or w/h <= 0) returns
 * ORBX_OK and leaves every output untouched (:1038-1039). Blocks until the result is on the host. */
int orbx_extract(orbx_ctx* ctx, const uint8_t* image, int width, int height, size_t stride,
                 orbx_keypoint* keypoints, uint8_t* descriptors, int cap, int* n);

/* The same on `nframes` (<= max_batch) equally sized frames that already live in device memory;
 * frame f starts at d_images + f*frame_stride. Outputs are device buffers: d_keypoints[f*cap ..],
 * d_descriptors[(f*cap ..)*32], d_counts[f] = keypoints found in frame f (only the first `cap` are
 * stored). Work is enqueued on `stream` (a hipStream_t; NULL = the context's own stream) and the
 * call returns without waiting. Frames are read in place when d_images, row_stride and frame_stride are
 * multiples of 4; otherwise the call first copies them into an aligned buffer of its own (one extra pass
 * over the pixels), so tightly packed odd widths and views into larger images work, a little slower.
 * Frames read in place are read in whole dwords up to row_stride in every row: height * row_stride bytes per
 * frame must be readable, also in the last row of the last frame (a view into a larger allocation satisfies
 * this by itself; the host entry points copy exactly `width` bytes per row and have no such requirement).
 * A context may be driven from different streams in turn: a call on another stream than the previous one first
 * waits (on the device) for the previous call, since all calls share the context's buffers. */
int orbx_extract_batch_device(orbx_ctx* ctx, const uint8_t* d_images, int nframes, int width,
                              int height, size_t row_stride, size_t frame_stride,
                              orbx_keypoint* d_keypoints, uint8_t* d_descriptors, int cap,
                              int32_t* d_counts, void* stream);
/* Host-buffer form of the batched call (H2D, extract, D2H; blocks). */
int orbx_extract_batch(orbx_ctx* ctx, const uint8_t* images, int nframes, int width, int height,
                       size_t row_stride, size_t frame_stride, orbx_keypoint* keypoints,
                       uint8_t* descriptors, int cap, int32_t* counts);
int orbx_synchronize(orbx_ctx* ctx);
/* The context's own stream (a hipStream_t), for callers that order their work against the asynchronous
 * calls above (events, or an allocator that must not recycle a buffer still being read). The reference
 * has no counterpart: its operator() returns when the work is done (src/ORBextractor.cc:1035-1097). */
int orbx_get_stream(orbx_ctx* ctx, void** stream);

/* Stage timing. The reference has no profiler; its examples only wrap std::chrono around Track*
 * (orb_slam2_mono_example/mono_euroc.cc:96-105). With profiling enabled every batched call records
 * HIP events on its launch stream around the four stages; orbx_get_stage_ms() waits for the last
 * call and returns milliseconds for {pyramid (all resize launches), FAST cells, quadtree,
 * orientation + descriptors}. */
#define ORBX_NUM_STAGES 4
int orbx_set_profiling(orbx_ctx* ctx, int enabled);
int orbx_get_stage_ms(orbx_ctx* ctx, float* ms);

/* mvImagePyramid[level] of frame `frame` of the last call (include/ORBextractor.h:100; read by
 * src/Frame.cc:510,600,612,617). With border != 0 the copy is (w+38) x (h+38) and carries the
 * 19-px BORDER_REFLECT_101 frame of src/ORBextractor.cc:1105-1120. dst is a host buffer. */
int orbx_level_size(const orbx_ctx* ctx, int level, int* width, int* height);
int orbx_get_level(orbx_ctx* ctx, int frame, int level, int border, uint8_t* dst, size_t dst_stride);

/* Stage probes of the last call, for parity tests: the per-cell FAST candidates in vToDistributeKeys order
 * (:770-821, coordinates relative to the 16-px border), the DistributeOctTree selection in
 * list order (:826-839, level coordinates, response set, angle -1) and the blurred pixels the descriptors
 * sample (GaussianBlur of the level, src/ORBextractor.cc:1077-1078). Host outputs.
 * orbx_debug_keypoint_blur: the extractor blurs only the patch around each keypoint, inside the kernel that
 * makes the descriptors; this probe re-runs that kernel on one frame of the last call and returns, for the
 * frame's keypoints in output order, the patch exactly as the descriptor stage read it: ORBX_BLUR_PATCH_ROWS
 * rows of ORBX_BLUR_PATCH_PITCH bytes per keypoint, the keypoint at row ORBX_BLUR_PATCH_CY, column
 * ORBX_BLUR_PATCH_CX. Only offsets the rotated pattern can reach (within sqrt(338) + 1/2 pixel of the keypoint
 * in both coordinates' unit boxes) hold blurred values; the rest is unspecified. *n = keypoints of the frame;
 * at most cap patches are written. */
#define ORBX_BLUR_PATCH_ROWS 37
#define ORBX_BLUR_PATCH_PITCH 44
#define ORBX_BLUR_PATCH_CY 18
#define ORBX_BLUR_PATCH_CX 23
int orbx_debug_keypoint_blur(orbx_ctx* ctx, int frame, uint8_t* dst, int cap, int* n);
/* Host-only check of the work plan of the one-launch pyramid kernel (the kernel small calls use in place of one
 * launch per cv::resize, src/ORBextractor.cc:1112): builds the plan for an image size and pyramid shape and
 * verifies that every pixel of every level is written by exactly one tile and that every tile finds the source
 * pixels of what it computes in its own tile of the level above. Needs no GPU. *ngroups = launches the plan makes
 * (0: this pyramid shape takes the per-level kernel). tile_w x tile_h = tile on a group's last level,
 * group_levels = levels per launch; the library itself uses 16, 8, 8. */
int orbx_debug_check_pyramid_plan(int width, int height, float scale_factor, int nlevels, int tile_w, int tile_h,
                                  int group_levels, int* ngroups);
int orbx_debug_candidates(orbx_ctx* ctx, int frame, int level, orbx_keypoint* out, int cap, int* n);
int orbx_debug_selected(orbx_ctx* ctx, int frame, int level, orbx_keypoint* out, int cap, int* n);

/* ---- matcher ----------------------------------------------------------------------------- */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1698-1714) for host callers. */
int orbx_hamming256(const uint8_t* a, const uint8_t* b);

typedef struct orbx_matcher orbx_matcher;
int orbx_matcher_create(int device, orbx_matcher** out);
void orbx_matcher_destroy(orbx_matcher* m);

/* Brute-force best / second-best scan with the update rule of src/ORBmatcher.cc:204-229
 * (also :440-465, :587-617): every query against every database row in index order,
 * `dist < best1` shifts best1 into best2, else `dist < best2` replaces best2; best1 = best2 = init
 * and idx = -1 before the scan (init is 256 at :204-206, INT_MAX at :440-441). idx/d1/d2 receive
 * nq int32 each. Device pointers; enqueued on `stream` (NULL = the matcher's own). ndb <= 2^31-1.
 * db_index_base is added to every returned index (for a database shard). */
int orbx_hamming_knn2_device(orbx_matcher* m, const uint8_t* d_query, int64_t nq,
                             const uint8_t* d_db, int64_t ndb, int32_t init, int64_t db_index_base,
                             int64_t* d_idx, int32_t* d_d1, int32_t* d_d2, void* stream);
/* The same scan for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. No counterpart in the reference. orbx_hamming_knn2_device keeps its
 * per-chunk partial results in a buffer it grows with hipMalloc; this form keeps them in a fixed 32 MiB device
 * array that is part of the library (one copy per GPU, held until the process ends) and calls no hipMalloc,
 * hipFree or other allocating routine. Same arguments and results. Limit: nq * ceil(ndb / 4194304) must not
 * exceed 2,097,152 (e.g. 65,536 queries x 134 M rows, or 2 M queries x 4 M rows); larger problems return
 * ORBX_ERR_CAPACITY and change nothing. All calls on one GPU share that GPU's array: calls that overlap in time on
 * one GPU (different streams or matchers) race; calls on one stream, or ordered by events, are safe. */
int orbx_hamming_knn2_noalloc_device(orbx_matcher* m, const uint8_t* d_query, int64_t nq,
                                     const uint8_t* d_db, int64_t ndb, int32_t init, int64_t db_index_base,
                                     int64_t* d_idx, int32_t* d_d1, int32_t* d_d2, void* stream);
/* Host-buffer form (H2D, scan, D2H; blocks). */
int orbx_hamming_knn2(orbx_matcher* m, const uint8_t* query, int64_t nq, const uint8_t* db,
                      int64_t ndb, int32_t init, int64_t* idx, int32_t* d1, int32_t* d2);
/* Merges `nparts` partial results of the same queries taken over consecutive database shards
 * (part p holds nq entries at offset p*nq) into the result of one scan over the whole database:
 * best = smallest d1, earliest shard on ties; second = second smallest of all d1 and d2. */
int orbx_knn2_merge_device(orbx_matcher* m, int nparts, int64_t nq, const int64_t* d_idx,
                           const int32_t* d_d1, const int32_t* d_d2, int32_t init,
                           int64_t* d_out_idx, int32_t* d_out_d1, int32_t* d_out_d2, void* stream);
/* The same scan and merge with one 16-byte record per query, for a database sharded over several GPUs: every
 * rank scans its shard into nq records, ONE all-gather of nq * 16 bytes per rank (ncclAllGather over the rank
 * order = shard order) leaves part p of query i at d_hits[p * nq + i], and the merge folds them exactly as a
 * single scan over the whole database would have (src/ORBmatcher.cc:204-229: lowest row wins ties, second best
 * = second smallest of all d1 and d2). A record with idx = -1 carries d1 = d2 = init. */
typedef struct orbx_knn2_hit {
    int32_t d1, d2;
    int64_t idx;
} orbx_knn2_hit;
int orbx_hamming_knn2_packed_device(orbx_matcher* m, const uint8_t* d_query, int64_t nq, const uint8_t* d_db,
                                    int64_t ndb, int32_t init, int64_t db_index_base, orbx_knn2_hit* d_hits,
                                    void* stream);
int orbx_knn2_merge_packed_device(orbx_matcher* m, int nparts, int64_t nq, const orbx_knn2_hit* d_hits,
                                  int32_t init, int64_t* d_out_idx, int32_t* d_out_d1, int32_t* d_out_d2,
                                  void* stream);
/* Acceptance test of the callers: d1 <= th (inclusive != 0: src/ORBmatcher.cc:231,467) or d1 < th
 * (:619), then (float)d1 < ratio*(float)d2 (:233,469,621). accept receives nq bytes (0/1). */
int orbx_ratio_test_device(orbx_matcher* m, int64_t nq, const int32_t* d_d1, const int32_t* d_d2,
                           int32_t th, int inclusive, float ratio, uint8_t* d_accept, void* stream);

/* Descriptor half of Frame::ComputeStereoMatches (src/Frame.cc:503-586) for `npairs` stereo pairs:
 * pair p has nl[p] left and nr[p] right keypoints stored at kl + p*cap etc. For every left
 * keypoint: the right keypoint with the smallest distance among those in its row band
 * (|yR - row| within 2*scale[octaveR], :520-530), octave within +-1 (:570) and
 * uL - bf/b <= uR <= uL (:553-554,575); strict `<` starting from TH_HIGH = 100, lowest right index
 * wins ties. best_idx = -1 and best_dist = 100 when nothing qualifies. Device pointers. */
int orbx_stereo_band_match_device(orbx_matcher* m, int npairs, int cap,
                                  const orbx_keypoint* d_kl, const uint8_t* d_dl, const int32_t* d_nl,
                                  const orbx_keypoint* d_kr, const uint8_t* d_dr, const int32_t* d_nr,
                                  const float* scale_factors, int nlevels, float bf, float b,
                                  int32_t* d_best_idx, int32_t* d_best_dist, void* stream);
/* The whole of Frame::ComputeStereoMatches (src/Frame.cc:503-677) for `npairs` stereo pairs: the band
 * search above, then the 11x11 SAD slide over +-5 px on the pyramid level of the left keypoint, the
 * parabola fit, disparity / depth (:589-661) and the median-based outlier cut (:663-676). The left and
 * right images must be frames 0..npairs-1 of the last batched call of `left` and `right` (their
 * pyramids are read in place; the call waits for those extractions first). d_uright / d_depth receive
 * mvuRight / mvDepth: cap floats per pair, -1 where a keypoint has no stereo match. */
int orbx_compute_stereo_matches_device(orbx_matcher* m, orbx_ctx* left, orbx_ctx* right, int npairs, int cap,
                                       const orbx_keypoint* d_kl, const uint8_t* d_dl, const int32_t* d_nl,
                                       const orbx_keypoint* d_kr, const uint8_t* d_dr, const int32_t* d_nr,
                                       float bf, float b, float* d_uright, float* d_depth, void* stream);
int orbx_matcher_synchronize(orbx_matcher* m);
/* The matcher's own stream (a hipStream_t); see orbx_get_stream. */
int orbx_matcher_get_stream(orbx_matcher* m, void** stream);

/* ---- the step after extraction: undistort + grid ---------------------------------------------- */

/* Camera.fx/fy/cx/cy and Camera.k1/k2/p1/p2/k3 as Tracking reads them (src/Tracking.cc:51-79) into
 * mK and mDistCoef. */
typedef struct orbx_camera {
    float fx, fy, cx, cy;
    float k1, k2, p1, p2, k3;
} orbx_camera;

#define ORBX_GRID_COLS 64 /* FRAME_GRID_COLS, include/PresetParameters.h:8 */
#define ORBX_GRID_ROWS 48 /* FRAME_GRID_ROWS, include/PresetParameters.h:7 */

/* Frame::ComputeImageBounds (src/Frame.cc:474-501): undistorted image corners -> min/max; host code. */
int orbx_image_bounds(const orbx_camera* cam, int width, int height, float* min_x, float* min_y,
                      float* max_x, float* max_y);

/* Frame::UndistortKeyPoints (src/Frame.cc:442-472: cv::undistortPoints(pts, pts, mK, mDistCoef,
 * Mat(), mK), five fixed-point iterations in double; a plain copy when k1 == 0) followed by
 * Frame::AssignFeaturesToGrid (:268-283) with PosInGrid (:420-430), for `nframes` keypoint sets of
 * d_counts[f] (<= cap) keypoints each. d_keypoints_un receives mvKeysUn. The grid comes back in
 * compressed form: keypoints of cell (ix, iy) are d_cell_items[f*cap + k] for
 * k in [d_cell_start[f*3073 + ix*48 + iy], d_cell_start[f*3073 + ix*48 + iy + 1]), ascending, exactly
 * the contents of mGrid[ix][iy]. Device pointers; enqueued on `stream` (NULL = the matcher's own). */
int orbx_undistort_assign_grid_device(orbx_matcher* m, int nframes, int cap, const orbx_keypoint* d_keypoints,
                                      const int32_t* d_counts, const orbx_camera* cam, float min_x, float min_y,
                                      float max_x, float max_y, orbx_keypoint* d_keypoints_un,
                                      int32_t* d_cell_start, int32_t* d_cell_items, void* stream);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:413-530) between two frames, host arrays in and
 * out. For every level-0 keypoint of frame 1 the candidates are Frame::GetFeaturesInArea
 * (src/Frame.cc:365-418) of frame 2 around prev_matched[i1] with radius window_size, levels 0..0; all
 * candidate lists and their 256-bit distances are produced on the GPU in the reference's visiting order
 * (grid columns, then rows, then cell contents), then the order-dependent part runs on the host exactly as
 * written there: best / second-best with the vMatchedDistance skip (:452-453), TH_LOW and ratio tests,
 * replacement of an earlier match of the same frame-2 keypoint, rotation histogram with
 * ComputeThreeMaxima (:1652-1693) and the final update of prev_matched (:524-527).
 * cell_start2 / cell_items2 are frame 2's grid as orbx_undistort_assign_grid_device returns it (host
 * copies). matches12 receives n1 indices into frame 2 (-1 = none); *nmatches the return value. */
int orbx_search_for_initialization(orbx_matcher* m, const orbx_keypoint* keys1_un, const uint8_t* desc1, int n1,
                                   const orbx_keypoint* keys2_un, const uint8_t* desc2, int n2,
                                   const int32_t* cell_start2, const int32_t* cell_items2, float min_x,
                                   float min_y, float max_x, float max_y, float* prev_matched, int window_size,
                                   float nnratio, int check_orientation, int32_t* matches12, int* nmatches);

/* The same function (src/ORBmatcher.cc:413-530) for frames whose data never left the GPU: d_desc1 and d_desc2 are
 * the descriptor arrays orbx_extract_batch_device wrote for the two frames, d_keys2_un / d_cell_start2 /
 * d_cell_items2 frame 2's mvKeysUn and grid as orbx_undistort_assign_grid_device left them (device pointers, the
 * slices of one frame; descriptor arrays 16-byte aligned). Nothing is uploaded but one query per level-0 keypoint
 * of frame 1 (24 bytes each, read by the kernel from pinned host memory) and nothing comes back but, per query,
 * the number of its candidates and the eight closest of them in order, written into pinned host memory: one
 * launch and one wait. The candidate lists themselves stay in device memory; the host pass reads one back only
 * when those eight do not settle its query. keys1_un, keys2_un (host copies of mvKeysUn,
 * which a Frame holds anyway) feed the order-dependent host pass (octaves, angles, the prev_matched update).
 * The device inputs must be complete, or queued on the matcher's stream, when the call is made; it returns
 * after the result is final. Indices in the device grid that fall outside frame 2 are skipped by the kernel. */
int orbx_search_for_initialization_device(orbx_matcher* m, const orbx_keypoint* keys1_un, const uint8_t* d_desc1,
                                          int n1, const orbx_keypoint* keys2_un, const orbx_keypoint* d_keys2_un,
                                          const uint8_t* d_desc2, int n2, const int32_t* d_cell_start2,
                                          const int32_t* d_cell_items2, float min_x, float min_y, float max_x,
                                          float max_y, float* prev_matched, int window_size, float nnratio,
                                          int check_orientation, int32_t* matches12, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1372-1516; called from Tracking::TrackWithMotionModel, src/Tracking.cc:929-940), host arrays
 * in and out, from the projection onwards. The pose algebra stays with the caller (:1382-1409): for each of the
 * npts features of the last frame that carry a live map point and are no outliers, in LastFrame order, it passes
 * the projection into the current frame proj_uv[2i], proj_uv[2i+1], inv_z[i] = invzc, the feature's octave
 * (LastFrame.mvKeys[i].octave) and angle (LastFrame.mvKeysUn[i].angle), the map point's descriptor
 * (pMP->GetDescriptor()) and has_observations[i] = pMP->Observations() > 0. direction: 0 = neither bForward nor
 * bBackward (levels o-1..o+1), 1 = bForward (levels >= o), 2 = bBackward (levels <= o) (:1392-1393,
 * :1429-1434). Current frame: keys_un = mvKeysUn, desc = mDescriptors, u_right = mvuRight (NULL = monocular),
 * occupied[i] != 0 where mvpMapPoints[i] already holds a point with observations (NULL = none), cell_start /
 * cell_items its grid. The frustum and image-bound tests (:1411-1420), the search radius th * scale_factors[o]
 * (:1425), Frame::GetFeaturesInArea, the stereo consistency test (:1451-1457) and all distances run on the
 * GPU in the reference's visiting order; the order-dependent part (what a feature already holds, best distance
 * <= max_dist, rotation histogram) runs on the host as written there. max_dist is TH_HIGH = 100 for this
 * overload (:1470). matches receives, per current-frame feature, the index into the npts list of the map point
 * assigned to it (-1 = none); *nmatches the return value.
 * The relocalisation overload SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (:1518-1650,
 * src/Tracking.cc:1501,1515) is the same loop: pass the points that survive :1541-1570 with last_octave = the
 * predicted level, last_angle = pKF->mvKeysUn[i].angle, inv_z = any positive value (that overload has no test on
 * it), has_observations all 1, occupied[i] = (mvpMapPoints[i] != nullptr), u_right NULL, direction 0 and
 * max_dist = ORBdist. */
int orbx_search_by_projection_frame(orbx_matcher* m, const float* proj_uv, const float* inv_z,
                                    const int32_t* last_octave, const float* last_angle, const uint8_t* desc_mp,
                                    const uint8_t* has_observations, int npts, const orbx_keypoint* keys_un,
                                    const uint8_t* desc, const float* u_right, const uint8_t* occupied, int n,
                                    const int32_t* cell_start, const int32_t* cell_items, float min_x, float min_y,
                                    float max_x, float max_y, const float* scale_factors, int nlevels, float th,
                                    float mbf, int direction, int max_dist, int check_orientation, int32_t* matches,
                                    int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th) (src/ORBmatcher.cc:41-123;
 * called from Tracking::SearchLocalPoints, src/Tracking.cc:1253-1263), host arrays in and out. The caller keeps
 * Frame::isInFrustum (src/Frame.cc:293-363) and passes, for each of the npts map points that pass :51 (not bad,
 * mbTrackInView) in vpMapPoints order: proj_xy = mTrackProjX / mTrackProjY, proj_xr = mTrackProjXR (read only when
 * u_right is given), view_cos = mTrackViewCos, level = mnTrackScaleLevel, the descriptor and
 * has_observations. Window radius RadiusByViewingCos(view_cos) [* th if th != 1] * scale_factors[level], levels
 * level-1..level (:57-63), stereo bound (:85-90) and distances on the GPU; best / second best with their
 * levels, TH_HIGH and the same-level ratio test (:96-119) on the host in vpMapPoints order. The other arguments
 * and `matches` as in orbx_search_by_projection_frame (the threshold is TH_HIGH, :112). */
int orbx_search_by_projection_points(orbx_matcher* m, const float* proj_xy, const float* proj_xr,
                                     const float* view_cos, const int32_t* level, const uint8_t* desc_mp,
                                     const uint8_t* has_observations, int npts, const orbx_keypoint* keys_un,
                                     const uint8_t* desc, const float* u_right, const uint8_t* occupied, int n,
                                     const int32_t* cell_start, const int32_t* cell_items, float min_x, float min_y,
                                     float max_x, float max_y, const float* scale_factors, int nlevels, float th,
                                     float nnratio, int32_t* matches, int* nmatches);

/* ORBmatcher::SearchByBoW, host arrays in and out. mode ORBX_BOW_KEYFRAME_FRAME follows
 * SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:153-293): side 1 is the keyframe
 * (keys1_un = mvKeysUn, valid1[i] != 0 where the keyframe holds a live map point), side 2 the frame
 * (keys2 = mvKeys; valid2 is not read); matches receives n2 entries: the keyframe feature whose map point
 * frame feature i2 gets, or -1; acceptance bestDist1 <= TH_LOW (:231). mode ORBX_BOW_KEYFRAME_KEYFRAME
 * follows SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (:532-690): both sides carry valid flags, matches
 * receives n1 entries (index into side 2 or -1), acceptance bestDist1 < TH_LOW (:619).
 * A feature vector (DBoW3::FeatureVector, a map node id -> feature indices) is passed flattened:
 * fvK_nodes ascending node ids, features of node j are fvK_items[fvK_start[j] .. fvK_start[j+1]).
 * The 256-bit distances of every (live left feature, right feature) pair inside a common node are made on
 * the GPU; the greedy, order-dependent pass over them (skip of already matched features, best / second best,
 * TH_LOW and ratio tests, rotation histogram with ComputeThreeMaxima :1652-1693) runs on the host as
 * written in the reference. *nmatches receives the return value. */
#define ORBX_BOW_KEYFRAME_FRAME 0
#define ORBX_BOW_KEYFRAME_KEYFRAME 1
int orbx_search_by_bow(orbx_matcher* m, int mode, const orbx_keypoint* keys1_un, const uint8_t* desc1,
                       const uint8_t* valid1, int n1, const uint32_t* fv1_nodes, const int32_t* fv1_start,
                       const uint32_t* fv1_items, int fv1_n, const orbx_keypoint* keys2, const uint8_t* desc2,
                       const uint8_t* valid2, int n2, const uint32_t* fv2_nodes, const int32_t* fv2_start,
                       const uint32_t* fv2_items, int fv2_n, float nnratio, int check_orientation,
                       int32_t* matches, int* nmatches);

/* orbx_search_by_bow with the two descriptor arrays already on the device (d_desc1, d_desc2: what
 * orbx_extract_batch_device wrote for the two images, 16-byte aligned; complete, or queued on the matcher's stream,
 * when the call is made). Every other argument is a host array with the meaning above. The kernel reads the
 * distance rows' descriptions from pinned host memory and writes the distances, and each row's four closest
 * matchable entries in order, back into it, so a call is one launch and one wait with no copy in either direction
 * (src/ORBmatcher.cc:153-293, :532-690). */
int orbx_search_by_bow_device(orbx_matcher* m, int mode, const orbx_keypoint* keys1_un, const uint8_t* d_desc1,
                              const uint8_t* valid1, int n1, const uint32_t* fv1_nodes, const int32_t* fv1_start,
                              const uint32_t* fv1_items, int fv1_n, const orbx_keypoint* keys2,
                              const uint8_t* d_desc2, const uint8_t* valid2, int n2, const uint32_t* fv2_nodes,
                              const int32_t* fv2_start, const uint32_t* fv2_items, int fv2_n, float nnratio,
                              int check_orientation, int32_t* matches, int* nmatches);

/* ---- synthetic frames (host) --------------------------------------------------------------- */
#define ORBX_SYNTH_SCENE 0
#define ORBX_SYNTH_ZERO 1
#define ORBX_SYNTH_CONST128 2
#define ORBX_SYNTH_NOISE 3
/* Deterministic test frame `index` (no counterpart in the reference, whose examples read EuRoC
 * PNGs that are not redistributable offline: orb_slam2_mono_example/mono_euroc.cc:86). */
int orbx_synth_frame(uint64_t index, int width, int height, int kind, int nrect, int ncheck,
                     int right_view, uint8_t* out, size_t stride);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
