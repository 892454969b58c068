/*
 * orb_oracle.h — CPU restatement of the ORB front end of One-sixth/my_orb_slam2.
 *
 * THIS IS TEST INFRASTRUCTURE.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors for
 * this path and it cannot be compiled here (it needs OpenCV 4.0.0, DBoW3, g2o).
 * The arithmetic that matters lives in OpenCV 4.0.0 (cv::resize, cv::FAST,
 * cv::GaussianBlur, cv::fastAtan2, cvRound), which is not vendored under
 * /root/reference; it is restated here from its published algorithm and pinned
 * only by analytic known-answer tests (tests/test_oracle_kat.py).
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/my_orb_slam2/).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct orc_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

enum { ORC_BLUR_CV_4_0_0 = 0, ORC_BLUR_CV_4_2_PLUS = 1 };

/* ---- primitives (OpenCV 4.0.0 semantics) -------------------------------- */
int   orc_cv_round_f(float v);                       /* cvRound: round half to even */
float orc_fast_atan2(float y, float x);              /* cv::fastAtan2, degrees [0,360) */
void  orc_resize_linear_u8(const uint8_t* src, int sw, int sh, long sstride,
                           uint8_t* dst, int dw, int dh, long dstride);
void  orc_gaussian7_u8(const uint8_t* src, int w, int h, long sstride,
                       uint8_t* dst, long dstride, int variant);
/* cv::FAST(img, kps, threshold, nonmaxSuppression) with TYPE_9_16. Returns the number found
 * (may exceed cap; only the first cap are stored). */
int   orc_fast9_16(const uint8_t* img, int w, int h, long stride, int threshold, int nms,
                   orc_keypoint* out, int cap);
/* DistributeOctTree (src/ORBextractor.cc:531-755). Coordinates are relative to (minX,minY).
 * Returns the number selected, written in list order. */
int   orc_distribute_octree(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY,
                            int N, orc_keypoint* out, int cap);

/* ---- extractor (src/ORBextractor.cc) ------------------------------------ */
typedef struct orc_extractor orc_extractor;
orc_extractor* orc_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                          int minThFAST, int blur_variant);
void  orc_destroy(orc_extractor* e);
/* How computeOrbDescriptor (src/ORBextractor.cc:103-104) takes cos / sin of the keypoint angle, for every
 * extractor of the process: 0 (default) = the correctly rounded float value, 1 = this C runtime's cosf / sinf
 * (what the reference's call resolves to, with the last-bit behaviour of whatever runtime it is built against). */
void  orc_set_trig_mode(int libm_float);
/* ORBextractor::operator() (:1035-1097). Returns the number of keypoints (<= cap stored),
 * or -1 on bad arguments. An empty image returns 0 and leaves outputs untouched. */
int   orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, long stride,
                  orc_keypoint* kps, uint8_t* desc, int cap);
/* Same, without keeping stage dumps (used for timing). */
int   orc_extract_fast(orc_extractor* e, const uint8_t* img, int w, int h, long stride,
                       orc_keypoint* kps, uint8_t* desc, int cap);

/* tables (ORBextractor.cc:402-462) */
int   orc_levels(const orc_extractor* e);
void  orc_scale_factors(const orc_extractor* e, float* scale, float* inv_scale,
                        float* sigma2, float* inv_sigma2);
void  orc_features_per_level(const orc_extractor* e, int* out);
void  orc_umax(const orc_extractor* e, int* out16);

/* stage dumps of the last orc_extract() call */
int   orc_level_size(const orc_extractor* e, int level, int* w, int* h);
/* Copies the level with its 19-px BORDER_REFLECT_101 frame: (w+38) x (h+38). */
int   orc_level_padded(const orc_extractor* e, int level, uint8_t* dst, long dstride);
int   orc_level_image(const orc_extractor* e, int level, uint8_t* dst, long dstride);
int   orc_level_blurred(const orc_extractor* e, int level, uint8_t* dst, long dstride);
/* vToDistributeKeys of a level (:770-821): coordinates relative to the 16-px border. */
int   orc_level_candidates(const orc_extractor* e, int level, orc_keypoint* out, int cap);
/* keypoints of a level after DistributeOctTree + border + orientation (:823-844), level coords. */
int   orc_level_selected(const orc_extractor* e, int level, orc_keypoint* out, int cap);

/* ---- matcher ------------------------------------------------------------- */
/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1698-1714), SWAR popcount. */
int   orc_hamming256(const uint8_t* a, const uint8_t* b);
int   orc_hamming256_popcnt(const uint8_t* a, const uint8_t* b);
/* best / second-best scan of src/ORBmatcher.cc:204-229: for each query the full database is
 * scanned in index order; d1,d2 start at `init` (256 there), idx at -1. use_popcnt selects the
 * distance routine (same results). */
void  orc_knn2(const uint8_t* q, long nq, const uint8_t* db, long ndb, int init, int use_popcnt,
               int32_t* idx, int32_t* d1, int32_t* d2);
/* acceptance test of :231-233 / :467-469 (inclusive!=0) and :619-621 (inclusive==0) */
int   orc_ratio_accept(int d1, int d2, int th, int inclusive, float ratio);
/* Frame::ComputeStereoMatches band search (src/Frame.cc:503-586): per left keypoint the best
 * right index and distance; best_dist = TH_HIGH (100) and best_idx = -1 when nothing beat it
 * (the reference leaves bestIdxR = 0 there; it is never used since 100 >= thOrbDist).
 * rows = image rows of level 0, scale = mvScaleFactors. Returns 0, or -1 if a row index falls
 * outside [0,rows) (the reference would write out of bounds). */
int   orc_stereo_band_match(const orc_keypoint* kl, const uint8_t* dl, int nl,
                            const orc_keypoint* kr, const uint8_t* dr, int nr,
                            int rows, const float* scale, int nlevels, float bf, float b,
                            int32_t* best_idx, int32_t* best_dist);

/* The whole of Frame::ComputeStereoMatches (src/Frame.cc:503-677): band search, SAD slide on the
 * pyramids that `left` / `right` kept from their last orc_extract() call, parabola fit, depth, and the
 * median-based outlier cut. uright / depth receive mvuRight / mvDepth (nl floats, -1 = no match).
 * Returns the number of matches kept, or -1 where the reference would read out of bounds. */
int   orc_compute_stereo_matches(const orc_extractor* left, const orc_extractor* right,
                                 const orc_keypoint* kl, const uint8_t* dl, int nl,
                                 const orc_keypoint* kr, const uint8_t* dr, int nr,
                                 float bf, float b, float* uright, float* depth);

/* cv::undistortPoints(pts, pts, K, D, Mat(), K) of OpenCV 4.0.0 as called by Frame::UndistortKeyPoints
 * (src/Frame.cc:442-472) and Frame::ComputeImageBounds (:474-501): K = fx, fy, cx, cy,
 * D = k1, k2, p1, p2, k3; five iterations in double. In place on n (x, y) float pairs. */
void  orc_undistort_points(float* xy, int n, const float* K4, const float* D5);
/* Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:268-283, 420-430): cell_start receives
 * 64*48+1 offsets (cell = ix*48 + iy), cell_items the keypoint indices cell by cell in push_back order. */
void  orc_assign_grid(const orc_keypoint* kps, int n, float minX, float minY, float maxX, float maxY,
                      int32_t* cell_start, int32_t* cell_items);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:413-530) with Frame::GetFeaturesInArea
 * (src/Frame.cc:365-418) and ComputeThreeMaxima (src/ORBmatcher.cc:1652-1693). Frame 2's grid is built
 * internally with orc_assign_grid. prev_matched (n1 x 2) is updated in place. Returns nmatches. */
int   orc_search_for_initialization(const orc_keypoint* k1, const uint8_t* d1, int n1,
                                    const orc_keypoint* k2, const uint8_t* d2, int n2,
                                    float minX, float minY, float maxX, float maxY, float* prev_matched,
                                    int window_size, float nnratio, int check_orientation, int32_t* matches12);

/* ORBmatcher::SearchByBoW: mode 0 the (KeyFrame*, Frame&) overload (src/ORBmatcher.cc:153-293; out has n2
 * entries = side-1 index per side-2 feature), mode 1 the (KeyFrame*, KeyFrame*) overload (:532-690; out has n1
 * entries = side-2 index per side-1 feature). validK[i] != 0 where the feature holds a live MapPoint
 * (valid2 unread in mode 0). Feature vectors flattened: ascending node ids, items[start[j]..start[j+1]). */
int   orc_search_by_bow(int mode, const orc_keypoint* k1, const uint8_t* d1, const uint8_t* valid1, int n1,
                        const uint32_t* nodes1, const int32_t* start1, const uint32_t* items1, int nn1,
                        const orc_keypoint* k2, const uint8_t* d2, const uint8_t* valid2, int n2,
                        const uint32_t* nodes2, const int32_t* start2, const uint32_t* items2, int nn2,
                        float nnratio, int check_orientation, int32_t* out);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1372-1516) from the projection onwards: the caller has run lines :1382-1409 and hands over,
 * for each of the npts last-frame features that carry a live map point and are no outliers (in LastFrame order),
 * the projection (u, v), invzc, the feature's octave and angle (LastFrame.mvKeys / mvKeysUn), the map point's
 * descriptor and whether the map point has observations. direction: 0 = neither forward nor backward
 * (levels o-1..o+1), 1 = bForward (levels >= o), 2 = bBackward (levels <= o) (:1392-1393, :1429-1434).
 * uright2 NULL = monocular (mvuRight all -1); occupied2[i] != 0 where CurrentFrame.mvpMapPoints[i] already
 * holds a point with observations (NULL = none). match2[i] receives the index (into the npts list) of the
 * map point assigned to current feature i, or -1. Returns nmatches. max_dist = TH_HIGH (100) for this overload.
 * SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (:1518-1650) is the same loop with
 * max_dist = ORBdist, no stereo test, levels l-1..l+1 around the predicted level (passed as last_octave), no
 * test on invzc (pass any positive value), every feature that holds a point blocked (occupied2) and every
 * assignment blocking (has_obs all 1). */
int   orc_search_by_projection_frame(const float* uv, const float* invz, const int32_t* last_octave,
                                     const float* last_angle, const uint8_t* desc_mp, const uint8_t* has_obs,
                                     int npts, const orc_keypoint* k2, const uint8_t* d2, const float* uright2,
                                     const uint8_t* occupied2, int n2, float minX, float minY, float maxX,
                                     float maxY, const float* scale_factors, int nlevels, float th, float mbf,
                                     int direction, int max_dist, int check_orientation, int32_t* match2);

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th) (src/ORBmatcher.cc:41-123)
 * for the npts map points that pass :51 (not bad, mbTrackInView), in vpMapPoints order: proj[2i], proj[2i+1] =
 * mTrackProjX/Y, proj_xr = mTrackProjXR, view_cos = mTrackViewCos, level = mnTrackScaleLevel. The other arguments
 * and match2 as in orc_search_by_projection_frame. Returns nmatches. */
int   orc_search_by_projection_points(const float* proj, const float* proj_xr, const float* view_cos,
                                      const int32_t* level, const uint8_t* desc_mp, const uint8_t* has_obs, int npts,
                                      const orc_keypoint* k2, const uint8_t* d2, const float* uright2,
                                      const uint8_t* occupied2, int n2, float minX, float minY, float maxX,
                                      float maxY, const float* scale_factors, int nlevels, float th,
                                      float nnratio, int32_t* match2);

#ifdef __cplusplus
}
#endif
#endif
