/*
 * orb_oracle.cpp — CPU restatement of the reference's ORB front end (see orb_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference fixtures exist; OpenCV 4.0.0, where
 * the arithmetic lives, is neither vendored in the reference nor installed here).
 *
 * Citations are file:line under /root/reference/my_orb_slam2/.  OpenCV semantics are restated
 * from the published OpenCV 4.0.0 algorithms (imgproc resize.cpp / smooth.cpp, features2d
 * fast.cpp + fast_score.cpp, core mathfuncs_core) and marked "cv:".
 *
 * Build: g++ -O3 -ffp-contract=off (no FMA contraction: the x86 scalar build of OpenCV and of
 * the reference rounds every float product separately).
 */
#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <map>
#include <vector>

namespace {

/* src/ORBextractor.cc:64-66 */
const int kPatchSize = 31;
const int kHalfPatch = 15;
const int kEdge = 19;

const int kPattern[1024] = {
#include "../include/orbx_pattern31.inc"
};

/* cv: cvRound(double/float) = SSE cvtsd2si / cvtss2si under the default MXCSR = nearest-even. */
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_round(float v) { return (int)std::nearbyintf(v); }
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int cv_ceil(float v) { return (int)std::ceil(v); }
inline short sat_short(float v) {
    int i = cv_round(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
/* cv: borderInterpolate(p, len, BORDER_REFLECT_101) */
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

struct Image {
    int w = 0, h = 0;
    long stride = 0;
    std::vector<uint8_t> buf;
    void alloc(int w_, int h_) { w = w_; h = h_; stride = w_; buf.assign((size_t)w_ * h_, 0); }
    uint8_t* row(int y) { return buf.data() + (size_t)y * stride; }
    const uint8_t* row(int y) const { return buf.data() + (size_t)y * stride; }
};

}  // namespace

extern "C" int orc_cv_round_f(float v) { return cv_round(v); }

/* cv: fastAtan2(float y, float x) -> atan_f32, the scalar polynomial; all float32, each product
 * and sum rounded separately. Called at src/ORBextractor.cc:95. */
extern "C" float orc_fast_atan2(float y, float x) {
    const float k = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k;
    const float p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k;
    const float p7 = -0.04432655554792128f * k;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* cv: resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1: HResizeLinear<uchar,int,short,2048>
 * + VResizeLinear with FixedPtCast<int,uchar,22>. Called at src/ORBextractor.cc:1112. */
extern "C" void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, long sstride,
                                     uint8_t* dst, int dw, int dh, long dstride) {
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return;
    const double scale_x = 1.0 / ((double)dw / sw);
    const double scale_y = 1.0 / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * (size_t)dw), ibeta(2 * (size_t)dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { xmax = std::min(xmax, dx); fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = sat_short(fy * 2048);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hline = [&](int sy, std::vector<int>& out) {
        sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy); /* cv: clip(sy, 0, ssize.height) */
        const uint8_t* S = src + (size_t)sy * sstride;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            out[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
        }
        for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        hline(yofs[dy], r0);
        hline(yofs[dy] + 1, r1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++) {
            int v = (((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2;
            D[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

/* cv: GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8UC1, non-submatrix:
 * fixedSmoothInvoker<uint8_t, ufixedpoint16>: 8.8 fixed-point taps, u16 horizontal sums,
 * 16.16 vertical sums rounded to u8. Called at src/ORBextractor.cc:1078. */
extern "C" void orc_gaussian7_u8(const uint8_t* src, int w, int h, long sstride, uint8_t* dst,
                                 long dstride, int variant) {
    static const int taps400[7] = {18, 34, 49, 55, 49, 34, 18};   /* 4.0.0: each tap rounded */
    static const int taps42[7] = {18, 34, 48, 56, 48, 34, 18};    /* >= 4.2: sum forced to 256 */
    const int* K = variant == ORC_BLUR_CV_4_0_0 ? taps400 : taps42;
    if (w <= 0 || h <= 0) return;
    std::vector<uint16_t> t((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int k = 0; k < 7; k++) acc += (unsigned)K[k] * S[reflect101(x + k - 3, w)];
            t[(size_t)y * w + x] = (uint16_t)(acc > 0xFFFF ? 0xFFFF : acc);
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = 0; k < 7; k++)
                acc += (uint32_t)K[k] * t[(size_t)reflect101(y + k - 3, h) * w + x];
            uint32_t v = (acc + 32768u) >> 16;
            D[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

namespace {

/* cv: cornerScore<16>(ptr, pixel, threshold) */
int corner_score16(const uint8_t* p, const int* off, int threshold) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

/* cv: FAST_t<16>(img, keypoints, threshold, nonmax_suppression) — row-streaming detector with a
 * three-row score ring; a corner of row i-1 is emitted while row i is processed. */
void fast9_16(const uint8_t* img, int w, int h, long stride, int threshold, bool nms,
              std::vector<orc_keypoint>& out) {
    static const int cx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    static const int cy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    out.clear();
    if (w < 7 || h < 7) return;
    int off[25];
    for (int k = 0; k < 25; k++) off[k] = (int)(cy[k % 16] * stride + cx[k % 16]);
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++)
        tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    std::vector<uint8_t> ring((size_t)3 * w, 0);
    std::vector<int> pos((size_t)3 * (w + 1), 0);
    uint8_t* buf[3] = {ring.data(), ring.data() + w, ring.data() + 2 * (size_t)w};
    int* cp[3] = {pos.data() + 1, pos.data() + (w + 1) + 1, pos.data() + 2 * (size_t)(w + 1) + 1};
    for (int i = 3; i < h - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cpos = cp[(i - 3) % 3];
        std::memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; j++) {
                const uint8_t* p = ptr + j;
                const int v = p[0];
                const uint8_t* tb = tab - v + 255;
                int d = tb[p[off[0]]] | tb[p[off[8]]];
                if (d == 0) continue;
                d &= tb[p[off[2]]] | tb[p[off[10]]];
                d &= tb[p[off[4]]] | tb[p[off[12]]];
                d &= tb[p[off[6]]] | tb[p[off[14]]];
                if (d == 0) continue;
                d &= tb[p[off[1]]] | tb[p[off[9]]];
                d &= tb[p[off[3]]] | tb[p[off[11]]];
                d &= tb[p[off[5]]] | tb[p[off[13]]];
                d &= tb[p[off[7]]] | tb[p[off[15]]];
                bool corner = false;
                if (d & 1) {
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < 25; k++) {
                        if (p[off[k]] < vt) {
                            if (++count > 8) { corner = true; break; }
                        } else count = 0;
                    }
                }
                if (!corner && (d & 2)) {
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < 25; k++) {
                        if (p[off[k]] > vt) {
                            if (++count > 8) { corner = true; break; }
                        } else count = 0;
                    }
                }
                if (corner) {
                    cpos[ncorners++] = j;
                    if (nms) curr[j] = (uint8_t)corner_score16(p, off, threshold);
                }
            }
        }
        cpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        const int* ppos = cp[(i - 4 + 3) % 3];
        const int n = ppos[-1];
        for (int k = 0; k < n; k++) {
            const int j = ppos[k];
            const int score = prev[j];
            if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                         score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                         score > curr[j] && score > curr[j + 1])) {
                orc_keypoint kp = {(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
                out.push_back(kp);
            }
        }
    }
}

/* ExtractorNode (include/ORBextractor.h:33-44) with the box kept as x0,y0,x1,y1. */
struct Node {
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    std::vector<orc_keypoint> keys;
    bool no_more = false;
    long seq = 0; /* creation sequence: replaces the heap address in the sort at :676 */
    std::list<Node>::iterator self;
};

/* ExtractorNode::DivideNode (src/ORBextractor.cc:473-529) */
void divide(const Node& n, Node c[4]) {
    const int halfX = (int)std::ceil((float)(n.x1 - n.x0) / 2);
    const int halfY = (int)std::ceil((float)(n.y1 - n.y0) / 2);
    const int xm = n.x0 + halfX, ym = n.y0 + halfY;
    c[0].x0 = n.x0; c[0].y0 = n.y0; c[0].x1 = xm;   c[0].y1 = ym;
    c[1].x0 = xm;   c[1].y0 = n.y0; c[1].x1 = n.x1; c[1].y1 = ym;
    c[2].x0 = n.x0; c[2].y0 = ym;   c[2].x1 = xm;   c[2].y1 = n.y1;
    c[3].x0 = xm;   c[3].y0 = ym;   c[3].x1 = n.x1; c[3].y1 = n.y1;
    for (const orc_keypoint& kp : n.keys) {
        if (kp.x < (float)xm) {
            if (kp.y < (float)ym) c[0].keys.push_back(kp);
            else c[2].keys.push_back(kp);
        } else if (kp.y < (float)ym) c[1].keys.push_back(kp);
        else c[3].keys.push_back(kp);
    }
    for (int k = 0; k < 4; k++) c[k].no_more = c[k].keys.size() == 1;
}

/* ORBextractor::DistributeOctTree (src/ORBextractor.cc:531-755).
 * Tie-break rule: the reference sorts (pointCount, node address) at :676, which is not
 * reproducible; here the second key is the node's creation sequence, ascending, so among equal
 * counts the newest node sits at the back and is expanded first. */
std::vector<orc_keypoint> distribute_octree(const std::vector<orc_keypoint>& in, int minX, int maxX,
                                            int minY, int maxY, int N) {
    std::vector<orc_keypoint> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1) return result; /* the reference divides by zero here; treat as "no keypoints" */
    const float hX = (float)(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node n;
        n.x0 = (int)(hX * (float)i);
        n.x1 = (int)(hX * (float)(i + 1));
        n.y0 = 0;
        n.y1 = maxY - minY;
        n.seq = seq++;
        nodes.push_back(n);
        ini[i] = &nodes.back();
    }
    for (const orc_keypoint& kp : in) {
        int r = (int)(kp.x / hX);
        if (r < 0 || r >= nIni) return std::vector<orc_keypoint>(); /* out of bounds in the reference */
        ini[r]->keys.push_back(kp);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->no_more = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    typedef std::pair<int, Node*> SizeNode;
    auto by_size_then_seq = [](const SizeNode& a, const SizeNode& b) {
        return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq;
    };
    /* pushes the non-empty children to the front of the list in the order n1..n4 (:612-652) */
    auto add_children = [&](Node c[4], std::vector<SizeNode>& expandable, int* nToExpand) {
        for (int k = 0; k < 4; k++) {
            if (c[k].keys.empty()) continue;
            c[k].seq = seq++;
            nodes.push_front(c[k]);
            if (c[k].keys.size() > 1) {
                if (nToExpand) ++*nToExpand;
                expandable.push_back(SizeNode((int)c[k].keys.size(), &nodes.front()));
                nodes.front().self = nodes.begin();
            }
        }
    };
    bool finish = false;
    std::vector<SizeNode> expandable;
    while (!finish) {
        const int prevSize = (int)nodes.size();
        int nToExpand = 0;
        expandable.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->no_more) { ++it; continue; }
            Node c[4];
            divide(*it, c);
            add_children(c, expandable, &nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                const int prev2 = (int)nodes.size();
                std::vector<SizeNode> prev = expandable;
                expandable.clear();
                std::sort(prev.begin(), prev.end(), by_size_then_seq);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node c[4];
                    divide(*prev[j].second, c);
                    add_children(c, expandable, nullptr);
                    nodes.erase(prev[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prev2) finish = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (const Node& n : nodes) { /* :736-752: best response per node, first maximum wins */
        const orc_keypoint* best = &n.keys[0];
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > best->response) best = &n.keys[k];
        result.push_back(*best);
    }
    return result;
}

}  // namespace

extern "C" int orc_fast9_16(const uint8_t* img, int w, int h, long stride, int threshold, int nms,
                            orc_keypoint* out, int cap) {
    std::vector<orc_keypoint> v;
    fast9_16(img, w, h, stride, threshold, nms != 0, v);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

extern "C" int orc_distribute_octree(const orc_keypoint* in, int n, int minX, int maxX, int minY,
                                     int maxY, int N, orc_keypoint* out, int cap) {
    std::vector<orc_keypoint> v(in, in + n);
    std::vector<orc_keypoint> r = distribute_octree(v, minX, maxX, minY, maxY, N);
    for (int i = 0; i < (int)r.size() && i < cap; i++) out[i] = r[i];
    return (int)r.size();
}

/* ------------------------------------------------------------------------- */
struct orc_extractor {
    int nfeatures, nlevels, iniTh, minTh, blur_variant;
    double scaleFactor; /* include/ORBextractor.h:113 keeps it as double */
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> per_level, umax;
    /* per-call state */
    std::vector<Image> padded;           /* mvImagePyramid backing stores, (w+38) x (h+38) */
    std::vector<int> lw, lh;
    std::vector<Image> blurred;
    std::vector<std::vector<orc_keypoint>> candidates, selected;
    bool keep_dumps = true;
};

/* ORBextractor::ORBextractor (src/ORBextractor.cc:402-462) */
extern "C" orc_extractor* orc_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                                     int minThFAST, int blur_variant) {
    if (nlevels < 1 || nfeatures < 0 || !(scaleFactor > 1.0f)) return nullptr;
    orc_extractor* e = new orc_extractor();
    e->nfeatures = nfeatures;
    e->scaleFactor = scaleFactor;
    e->nlevels = nlevels;
    e->iniTh = iniThFAST;
    e->minTh = minThFAST;
    e->blur_variant = blur_variant;
    e->scale.resize(nlevels);
    e->sigma2.resize(nlevels);
    e->scale[0] = 1.0f;
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->scale[i] = (float)(e->scale[i - 1] * e->scaleFactor);
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    e->inv_scale.resize(nlevels);
    e->inv_sigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    e->per_level.resize(nlevels);
    float factor = (float)(1.0f / e->scaleFactor);
    float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->per_level[l] = cv_round(desired);
        sum += e->per_level[l];
        desired *= factor;
    }
    e->per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
    /* :446-461 */
    e->umax.assign(kHalfPatch + 2, 0);
    int vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) e->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    e->umax.resize(kHalfPatch + 1);
    return e;
}

extern "C" void orc_destroy(orc_extractor* e) { delete e; }

extern "C" int orc_levels(const orc_extractor* e) { return e->nlevels; }
extern "C" void orc_scale_factors(const orc_extractor* e, float* s, float* is, float* g, float* ig) {
    for (int i = 0; i < e->nlevels; i++) {
        if (s) s[i] = e->scale[i];
        if (is) is[i] = e->inv_scale[i];
        if (g) g[i] = e->sigma2[i];
        if (ig) ig[i] = e->inv_sigma2[i];
    }
}
extern "C" void orc_features_per_level(const orc_extractor* e, int* out) {
    for (int i = 0; i < e->nlevels; i++) out[i] = e->per_level[i];
}
extern "C" void orc_umax(const orc_extractor* e, int* out16) {
    for (int i = 0; i <= kHalfPatch; i++) out16[i] = e->umax[i];
}

namespace {

/* cv: copyMakeBorder(..., BORDER_REFLECT_101) of the interior of a padded level onto its frame */
void fill_border101(Image& p, int w, int h) {
    for (int y = 0; y < h + 2 * kEdge; y++) {
        const int sy = reflect101(y - kEdge, h) + kEdge;
        uint8_t* D = p.row(y);
        const uint8_t* S = p.row(sy);
        for (int x = 0; x < w + 2 * kEdge; x++) {
            const int sx = reflect101(x - kEdge, w) + kEdge;
            if (y != sy || x != sx) D[x] = S[sx];
        }
    }
}

/* ORBextractor::ComputePyramid (src/ORBextractor.cc:1099-1124) */
void compute_pyramid(orc_extractor* e, const uint8_t* img, int w, int h, long stride) {
    e->padded.resize(e->nlevels);
    e->lw.resize(e->nlevels);
    e->lh.resize(e->nlevels);
    for (int l = 0; l < e->nlevels; l++) {
        const float s = e->inv_scale[l];
        const int lw = cv_round((float)w * s), lh = cv_round((float)h * s);
        e->lw[l] = lw;
        e->lh[l] = lh;
        Image& p = e->padded[l];
        p.alloc(lw + 2 * kEdge, lh + 2 * kEdge);
        if (lw <= 0 || lh <= 0) continue;
        if (l == 0) {
            for (int y = 0; y < h; y++) std::memcpy(p.row(y + kEdge) + kEdge, img + (size_t)y * stride, w);
        } else {
            Image& q = e->padded[l - 1];
            orc_resize_linear_u8(q.row(kEdge) + kEdge, e->lw[l - 1], e->lh[l - 1], q.stride,
                                 p.row(kEdge) + kEdge, lw, lh, p.stride);
        }
        fill_border101(p, lw, lh);
    }
}

/* IC_Angle (src/ORBextractor.cc:69-96) */
float ic_angle(const uint8_t* center, long step, const std::vector<int>& umax) {
    int m_01 = 0, m_10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor (src/ORBextractor.cc:99-139).  The reference writes cos(angle) / sin(angle) on a float
 * under `using namespace std` (include/ORBextractor.h:28), i.e. the float overloads = cosf / sinf of whatever C
 * runtime it is built with (MSVC's there). Their last bit is not specified: glibc's cosf / sinf differ from the
 * correctly rounded value for 0.04 % / 0.09 % of the floats in [0, 2 pi] and pick FMA or non-FMA variants by CPU.
 * g_trig_mode 0 (default) takes the correctly rounded value, which no runtime can improve on: (float) of the
 * double-precision result; mode 1 calls this runtime's cosf / sinf, to measure how much the choice matters.
 * Products and the sum are separate float roundings (no FMA). */
int g_trig_mode = 0;
void orb_descriptor(const orc_keypoint& kp, const uint8_t* img, long step, uint8_t* desc) {
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float angle = kp.angle * factorPI;
    const float a = g_trig_mode ? cosf(angle) : (float)std::cos((double)angle);
    const float b = g_trig_mode ? sinf(angle) : (float)std::sin((double)angle);
    const uint8_t* center = img + (long)cv_round(kp.y) * step + cv_round(kp.x);
    const int* pat = kPattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const float x0 = (float)pat[4 * k], y0 = (float)pat[4 * k + 1];
            const float x1 = (float)pat[4 * k + 2], y1 = (float)pat[4 * k + 3];
            const int t0 = center[(long)cv_round(x0 * b + y0 * a) * step + cv_round(x0 * a - y0 * b)];
            const int t1 = center[(long)cv_round(x1 * b + y1 * a) * step + cv_round(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ORBextractor::ComputeKeyPointsOctTree (src/ORBextractor.cc:757-845), one level */
void level_keypoints(orc_extractor* e, int level, std::vector<orc_keypoint>& cand,
                     std::vector<orc_keypoint>& sel) {
    cand.clear();
    sel.clear();
    const Image& p = e->padded[level];
    const int cols = e->lw[level], rows = e->lh[level];
    const uint8_t* roi = p.row(kEdge) + kEdge;
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = cols - kEdge + 3, maxBY = rows - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
    if (nCols < 1 || nRows < 1) return; /* the reference divides by zero for such tiny levels */
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<orc_keypoint> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
            const uint8_t* sub = roi + (long)y0 * p.stride + x0;
            fast9_16(sub, cw, ch, p.stride, e->iniTh, true, cell);
            if (cell.empty()) fast9_16(sub, cw, ch, p.stride, e->minTh, true, cell);
            for (orc_keypoint kp : cell) {
                kp.x += j * wCell;
                kp.y += i * hCell;
                cand.push_back(kp);
            }
        }
    }
    sel = distribute_octree(cand, minBX, maxBX, minBY, maxBY, e->per_level[level]);
    const int scaledPatch = (int)(kPatchSize * e->scale[level]);
    for (orc_keypoint& kp : sel) {
        kp.x += minBX;
        kp.y += minBY;
        kp.octave = level;
        kp.size = (float)scaledPatch;
    }
    for (orc_keypoint& kp : sel) /* computeOrientation (:464-471) on the unblurred level */
        kp.angle = ic_angle(roi + (long)cv_round(kp.y) * p.stride + cv_round(kp.x), p.stride, e->umax);
}

int extract_impl(orc_extractor* e, const uint8_t* img, int w, int h, long stride, orc_keypoint* kps,
                 uint8_t* desc, int cap) {
    if (!e) return -1;
    if (!img || w <= 0 || h <= 0) return 0; /* :1038-1039 */
    compute_pyramid(e, img, w, h, stride);
    e->candidates.assign(e->nlevels, std::vector<orc_keypoint>());
    e->selected.assign(e->nlevels, std::vector<orc_keypoint>());
    e->blurred.assign(e->nlevels, Image());
    for (int l = 0; l < e->nlevels; l++) level_keypoints(e, l, e->candidates[l], e->selected[l]);
    int total = 0, offset = 0;
    for (int l = 0; l < e->nlevels; l++) total += (int)e->selected[l].size();
    for (int l = 0; l < e->nlevels; l++) { /* :1068-1096 */
        std::vector<orc_keypoint>& k = e->selected[l];
        if (k.empty()) continue;
        const Image& p = e->padded[l];
        Image& bl = e->blurred[l];
        bl.alloc(e->lw[l], e->lh[l]);
        orc_gaussian7_u8(p.row(kEdge) + kEdge, e->lw[l], e->lh[l], p.stride, bl.row(0), bl.stride,
                         e->blur_variant);
        for (size_t i = 0; i < k.size(); i++) {
            const int o = offset + (int)i;
            if (o >= cap) continue;
            orb_descriptor(k[i], bl.row(0), bl.stride, desc + (size_t)o * 32);
            orc_keypoint kp = k[i];
            if (l != 0) {
                const float s = e->scale[l];
                kp.x *= s;
                kp.y *= s;
            }
            kps[o] = kp;
        }
        offset += (int)k.size();
    }
    if (!e->keep_dumps) {
        e->blurred.clear();
    }
    return total;
}

}  // namespace

extern "C" void orc_set_trig_mode(int libm_float) { g_trig_mode = libm_float != 0; }

extern "C" int orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, long stride,
                           orc_keypoint* kps, uint8_t* desc, int cap) {
    if (e) e->keep_dumps = true;
    return extract_impl(e, img, w, h, stride, kps, desc, cap);
}
extern "C" int orc_extract_fast(orc_extractor* e, const uint8_t* img, int w, int h, long stride,
                                orc_keypoint* kps, uint8_t* desc, int cap) {
    if (e) e->keep_dumps = false;
    return extract_impl(e, img, w, h, stride, kps, desc, cap);
}

extern "C" int orc_level_size(const orc_extractor* e, int level, int* w, int* h) {
    if (!e || level < 0 || level >= (int)e->lw.size()) return -1;
    *w = e->lw[level];
    *h = e->lh[level];
    return 0;
}
extern "C" int orc_level_padded(const orc_extractor* e, int level, uint8_t* dst, long dstride) {
    if (!e || level < 0 || level >= (int)e->padded.size()) return -1;
    const Image& p = e->padded[level];
    for (int y = 0; y < p.h; y++) std::memcpy(dst + (size_t)y * dstride, p.row(y), p.w);
    return 0;
}
extern "C" int orc_level_image(const orc_extractor* e, int level, uint8_t* dst, long dstride) {
    if (!e || level < 0 || level >= (int)e->padded.size()) return -1;
    const Image& p = e->padded[level];
    for (int y = 0; y < e->lh[level]; y++)
        std::memcpy(dst + (size_t)y * dstride, p.row(y + kEdge) + kEdge, e->lw[level]);
    return 0;
}
extern "C" int orc_level_blurred(const orc_extractor* e, int level, uint8_t* dst, long dstride) {
    if (!e || level < 0 || level >= (int)e->blurred.size()) return -1;
    const Image& b = e->blurred[level];
    if (b.w == 0) return 1; /* level had no keypoints: the reference never blurs it (:1073-1074) */
    for (int y = 0; y < b.h; y++) std::memcpy(dst + (size_t)y * dstride, b.row(y), b.w);
    return 0;
}
extern "C" int orc_level_candidates(const orc_extractor* e, int level, orc_keypoint* out, int cap) {
    if (!e || level < 0 || level >= (int)e->candidates.size()) return -1;
    const std::vector<orc_keypoint>& v = e->candidates[level];
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
extern "C" int orc_level_selected(const orc_extractor* e, int level, orc_keypoint* out, int cap) {
    if (!e || level < 0 || level >= (int)e->selected.size()) return -1;
    const std::vector<orc_keypoint>& v = e->selected[level];
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

/* ------------------------------------------------------------------------- */
/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1698-1714) */
extern "C" int orc_hamming256(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4);
        std::memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return dist;
}

extern "C" int orc_hamming256_popcnt(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8);
        std::memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

/* best/second-best update of src/ORBmatcher.cc:219-228 over the whole database */
extern "C" void orc_knn2(const uint8_t* q, long nq, const uint8_t* db, long ndb, int init,
                         int use_popcnt, int32_t* idx, int32_t* d1, int32_t* d2) {
    for (long i = 0; i < nq; i++) {
        const uint8_t* a = q + 32 * i;
        int best1 = init, best2 = init, bi = -1;
        for (long j = 0; j < ndb; j++) {
            const int dist = use_popcnt ? orc_hamming256_popcnt(a, db + 32 * j)
                                        : orc_hamming256(a, db + 32 * j);
            if (dist < best1) {
                best2 = best1;
                best1 = dist;
                bi = (int)j;
            } else if (dist < best2) {
                best2 = dist;
            }
        }
        idx[i] = bi;
        d1[i] = best1;
        d2[i] = best2;
    }
}

extern "C" int orc_ratio_accept(int d1, int d2, int th, int inclusive, float ratio) {
    if (inclusive ? !(d1 <= th) : !(d1 < th)) return 0;
    return (float)d1 < ratio * (float)d2 ? 1 : 0;
}

/* Frame::ComputeStereoMatches, descriptor band search (src/Frame.cc:503-586) */
extern "C" int orc_stereo_band_match(const orc_keypoint* kl, const uint8_t* dl, int nl,
                                     const orc_keypoint* kr, const uint8_t* dr, int nr, int rows,
                                     const float* scale, int nlevels, float bf, float b,
                                     int32_t* best_idx, int32_t* best_dist) {
    const int TH_HIGH = 100; /* include/PresetParameters.h:11-13 */
    std::vector<std::vector<int>> rowIdx(rows);
    for (int iR = 0; iR < nr; iR++) {
        const float y = kr[iR].y;
        if (kr[iR].octave < 0 || kr[iR].octave >= nlevels) return -1;
        const float r = 2.0f * scale[kr[iR].octave];
        const int maxr = (int)std::ceil(y + r), minr = (int)std::floor(y - r);
        if (minr < 0 || maxr >= rows) return -1;
        for (int yi = minr; yi <= maxr; yi++) rowIdx[yi].push_back(iR);
    }
    const float minZ = b, minD = 0, maxD = bf / minZ;
    for (int iL = 0; iL < nl; iL++) {
        best_idx[iL] = -1;
        best_dist[iL] = TH_HIGH;
        const float vL = kl[iL].y, uL = kl[iL].x;
        const int levelL = kl[iL].octave;
        const long row = (long)vL;
        if (row < 0 || row >= rows) return -1;
        const std::vector<int>& c = rowIdx[row];
        if (c.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH, bestIdx = -1;
        for (size_t k = 0; k < c.size(); k++) {
            const int iR = c[k];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(dl + 32 * (size_t)iL, dr + 32 * (size_t)iR);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdx = iR;
                }
            }
        }
        best_idx[iL] = bestIdx;
        best_dist[iL] = bestDist;
    }
    return 0;
}

/* Frame::ComputeStereoMatches (src/Frame.cc:503-677) */
extern "C" int orc_compute_stereo_matches(const orc_extractor* left, const orc_extractor* right,
                                          const orc_keypoint* kl, const uint8_t* dl, int nl,
                                          const orc_keypoint* kr, const uint8_t* dr, int nr, float bf, float b,
                                          float* uright, float* depth) {
    if (!left || !right || left->padded.empty() || right->padded.empty()) return -1;
    const int TH_HIGH = 100, TH_LOW = 50;
    for (int i = 0; i < nl; i++) uright[i] = depth[i] = -1.0f;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = left->lh[0];
    std::vector<int32_t> bi(nl), bd(nl);
    if (orc_stereo_band_match(kl, dl, nl, kr, dr, nr, nRows, left->scale.data(), left->nlevels, bf, b,
                              bi.data(), bd.data()) != 0)
        return -1;
    const float minD = 0, maxD = bf / b;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < nl; iL++) {
        if (!(bd[iL] < thOrbDist) || bi[iL] < 0) continue;
        const orc_keypoint& kpL = kl[iL];
        const float uL = kpL.x;
        const float uR0 = kr[bi[iL]].x;
        const float scaleFactor = left->inv_scale[kpL.octave];
        const float scaleduL = std::round(kpL.x * scaleFactor);
        const float scaledvL = std::round(kpL.y * scaleFactor);
        const float scaleduR0 = std::round(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const Image& PL = left->padded[kpL.octave];
        const Image& PR = right->padded[kpL.octave];
        const int colsR = right->lw[kpL.octave];
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= colsR) continue;
        auto at = [&](const Image& P, int y, int x) { return (float)P.row(y + kEdge)[x + kEdge]; };
        const int cv = (int)scaledvL, cu = (int)scaleduL, cr = (int)scaleduR0;
        float IL[11][11];
        const float cL = at(PL, cv, cu);
        for (int dy = -w; dy <= w; dy++)
            for (int dx = -w; dx <= w; dx++) IL[dy + w][dx + w] = at(PL, cv + dy, cu + dx) - cL;
        int bestDist = INT_MAX, bestincR = 0;
        std::vector<float> vDists(2 * L + 1);
        for (int incR = -L; incR <= +L; incR++) {
            const float cR = at(PR, cv, cr + incR);
            double acc = 0; /* cv::norm(NORM_L1) on CV_32F accumulates in double; the terms are integers */
            for (int dy = -w; dy <= w; dy++)
                for (int dx = -w; dx <= w; dx++)
                    acc += std::fabs(IL[dy + w][dx + w] - (at(PR, cv + dy, cr + incR + dx) - cR));
            const float dist = (float)acc;
            if (dist < bestDist) {
                bestDist = (int)dist;
                bestincR = incR;
            }
            vDists[L + incR] = dist;
        }
        if (bestincR == -L || bestincR == L) continue;
        const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = left->scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = (uL - bestuR);
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
                disparity = 0.01;
                bestuR = uL - 0.01;
            }
            depth[iL] = bf / disparity;
            uright[iL] = bestuR;
            vDistIdx.push_back(std::pair<int, int>(bestDist, iL));
        }
    }
    if (vDistIdx.empty()) return 0; /* the reference indexes the empty vector here */
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uright[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
        kept--;
    }
    return kept;
}

/* cv::undistortPoints (imgproc undistort.cpp, cvUndistortPointsInternal) with R = I, P = K and
 * TermCriteria(MAX_ITER, 5, 0.01) as called at src/Frame.cc:460,486 */
extern "C" void orc_undistort_points(float* xy, int n, const float* K4, const float* D5) {
    double k[14] = {0};
    for (int i = 0; i < 5; i++) k[i] = D5[i];
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
    for (int i = 0; i < n; i++) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        xy[2 * i] = (float)(xx * ww);
        xy[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::AssignFeaturesToGrid (src/Frame.cc:268-283) with PosInGrid (:420-430) */
extern "C" void orc_assign_grid(const orc_keypoint* kps, int n, float minX, float minY, float maxX, float maxY,
                                int32_t* cell_start, int32_t* cell_items) {
    const int COLS = 64, ROWS = 48; /* include/PresetParameters.h:7-8 */
    const float invW = static_cast<float>(COLS) / (maxX - minX);   /* src/Frame.cc:133-134 */
    const float invH = static_cast<float>(ROWS) / (maxY - minY);
    std::vector<std::vector<int>> grid(COLS * ROWS);
    for (int i = 0; i < n; i++) {
        const int posX = (int)std::round((kps[i].x - minX) * invW);
        const int posY = (int)std::round((kps[i].y - minY) * invH);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[posX * ROWS + posY].push_back(i);
    }
    int o = 0;
    for (int c = 0; c < COLS * ROWS; c++) {
        cell_start[c] = o;
        for (int i : grid[c]) cell_items[o++] = i;
    }
    cell_start[COLS * ROWS] = o;
}

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:413-530) */
extern "C" int orc_search_for_initialization(const orc_keypoint* k1, const uint8_t* d1, int n1,
                                             const orc_keypoint* k2, const uint8_t* d2, int n2, float minX,
                                             float minY, float maxX, float maxY, float* prev, int windowSize,
                                             float mfNNratio, int checkOri, int32_t* vnMatches12) {
    const int COLS = 64, ROWS = 48, HISTO_LENGTH = 30, TH_LOW = 50;
    const float invW = static_cast<float>(COLS) / (maxX - minX), invH = static_cast<float>(ROWS) / (maxY - minY);
    /* mGrid of frame 2 (src/Frame.cc:268-283) */
    std::vector<std::vector<uint32_t>> grid(COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k2[i].x - minX) * invW), posY = (int)std::round((k2[i].y - minY) * invH);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[posX * ROWS + posY].push_back((uint32_t)i);
    }
    /* Frame::GetFeaturesInArea (src/Frame.cc:365-418) */
    auto features_in_area = [&](float x, float y, float r, int minLevel, int maxLevel) {
        std::vector<size_t> v;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
        if (nMinCellX >= COLS) return v;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - minX + r) * invW));
        if (nMaxCellX < 0) return v;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
        if (nMinCellY >= ROWS) return v;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - minY + r) * invH));
        if (nMaxCellY < 0) return v;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (uint32_t j : grid[ix * ROWS + iy]) {
                    const orc_keypoint& kp = k2[j];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(j);
                }
        return v;
    };
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        std::vector<size_t> vIndices2 = features_in_area(prev[2 * i1], prev[2 * i1 + 1], (float)windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            const int dist = orc_hamming256(d1 + 32 * (size_t)i1, d2 + 32 * i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = (int)i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot / (360 * factor));
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) { /* ComputeThreeMaxima (:1652-1693) */
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) {
            prev[2 * i1] = k2[vnMatches12[i1]].x;
            prev[2 * i1 + 1] = k2[vnMatches12[i1]].y;
        }
    return nmatches;
}

/* ORBmatcher::SearchByBoW: mode 0 = (KeyFrame*, Frame&, ...) src/ORBmatcher.cc:153-293, mode 1 =
 * (KeyFrame*, KeyFrame*, ...) :532-690. DBoW3::FeatureVector is a std::map<node id, vector<feature index>>;
 * it is rebuilt here from the flattened form so that the walk uses the same begin / lower_bound / ++ steps. */
extern "C" int orc_search_by_bow(int mode, const orc_keypoint* k1, const uint8_t* d1, const uint8_t* valid1, int n1,
                                 const uint32_t* nodes1, const int32_t* start1, const uint32_t* items1, int nn1,
                                 const orc_keypoint* k2, const uint8_t* d2, const uint8_t* valid2, int n2,
                                 const uint32_t* nodes2, const int32_t* start2, const uint32_t* items2, int nn2,
                                 float mfNNratio, int checkOri, int32_t* out) {
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    typedef std::map<uint32_t, std::vector<unsigned int>> FeatureVector;
    FeatureVector fv1, fv2;
    for (int j = 0; j < nn1; j++) fv1[nodes1[j]].assign(items1 + start1[j], items1 + start1[j + 1]);
    for (int j = 0; j < nn2; j++) fv2[nodes2[j]].assign(items2 + start2[j], items2 + start2[j + 1]);
    const int nout = mode == 0 ? n2 : n1;
    for (int i = 0; i < nout; i++) out[i] = -1;
    std::vector<bool> vbMatched2(n2, false);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    FeatureVector::const_iterator it1 = fv1.begin(), it2 = fv2.begin(), end1 = fv1.end(), end2 = fv2.end();
    while (it1 != end1 && it2 != end2) {
        if (it1->first == it2->first) {
            const std::vector<unsigned int>& v1 = it1->second;
            const std::vector<unsigned int>& v2 = it2->second;
            for (size_t i1 = 0; i1 < v1.size(); i1++) {
                const unsigned int idx1 = v1[i1];
                if (!valid1[idx1]) continue; /* no MapPoint or a bad one */
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (size_t i2 = 0; i2 < v2.size(); i2++) {
                    const unsigned int idx2 = v2[i2];
                    if (mode == 0) {
                        if (out[idx2] >= 0) continue;            /* :212 vpMapPointMatches[realIdxF] */
                    } else {
                        if (vbMatched2[idx2] || !valid2[idx2]) continue; /* :597-601 */
                    }
                    const int dist = orc_hamming256(d1 + 32 * (size_t)idx1, d2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = (int)idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                const bool close = mode == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW);
                if (close) {
                    if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                        if (mode == 0) {
                            out[bestIdx2] = (int)idx1;
                        } else {
                            out[idx1] = bestIdx2;
                            vbMatched2[bestIdx2] = true;
                        }
                        if (checkOri) {
                            float rot = k1[idx1].angle - k2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot / (360 * factor));
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(mode == 0 ? bestIdx2 : (int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            it1++;
            it2++;
        } else if (it1->first < it2->first) {
            it1 = fv1.lower_bound(it2->first);
        } else {
            it2 = fv2.lower_bound(it1->first);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) { /* ComputeThreeMaxima (:1652-1693) */
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                out[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), src/ORBmatcher.cc:1395-1515 (the pose algebra
 * of :1382-1409 is the caller's). */
extern "C" int orc_search_by_projection_frame(const float* uv, const float* invz, const int32_t* last_octave,
                                              const float* last_angle, const uint8_t* desc_mp,
                                              const uint8_t* has_obs, int npts, const orc_keypoint* k2,
                                              const uint8_t* d2, const float* uright2, const uint8_t* occupied2,
                                              int n2, float minX, float minY, float maxX, float maxY,
                                              const float* scale_factors, int nlevels, float th, float mbf,
                                              int direction, int maxDist, int checkOri, int32_t* match2) {
    const int COLS = 64, ROWS = 48, HISTO_LENGTH = 30;
    const float invW = static_cast<float>(COLS) / (maxX - minX), invH = static_cast<float>(ROWS) / (maxY - minY);
    std::vector<std::vector<uint32_t>> grid(COLS * ROWS);   /* mGrid of the current frame (src/Frame.cc:268-283) */
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k2[i].x - minX) * invW), posY = (int)std::round((k2[i].y - minY) * invH);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[posX * ROWS + posY].push_back((uint32_t)i);
    }
    /* Frame::GetFeaturesInArea (src/Frame.cc:365-418) */
    auto features_in_area = [&](float x, float y, float r, int minLevel, int maxLevel) {
        std::vector<size_t> v;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
        if (nMinCellX >= COLS) return v;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - minX + r) * invW));
        if (nMaxCellX < 0) return v;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
        if (nMinCellY >= ROWS) return v;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - minY + r) * invH));
        if (nMaxCellY < 0) return v;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (uint32_t j : grid[ix * ROWS + iy]) {
                    const orc_keypoint& kp = k2[j];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(j);
                }
        return v;
    };
    /* CurrentFrame.mvpMapPoints as far as this function looks at it: who sits there, and has it observations */
    std::vector<int32_t> holder(n2, -1);
    std::vector<uint8_t> blocked(n2, 0);
    for (int i = 0; i < n2; i++) blocked[i] = occupied2 ? (occupied2[i] != 0) : 0;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    const bool bForward = direction == 1, bBackward = direction == 2;
    for (int i = 0; i < npts; i++) {
        const float invzc = invz[i];
        if (invzc < 0) continue;                                            /* :1411 */
        const float u = uv[2 * i], v = uv[2 * i + 1];
        if (u < minX || u > maxX) continue;                                 /* :1417-1420 */
        if (v < minY || v > maxY) continue;
        const int nLastOctave = last_octave[i];
        if (nLastOctave < 0 || nLastOctave >= nlevels) continue;            /* (not a value the extractor produces) */
        const float radius = th * scale_factors[nLastOctave];               /* :1425 */
        std::vector<size_t> vIndices2;
        if (bForward) vIndices2 = features_in_area(u, v, radius, nLastOctave, -1);
        else if (bBackward) vIndices2 = features_in_area(u, v, radius, 0, nLastOctave);
        else vIndices2 = features_in_area(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = desc_mp + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (blocked[i2]) continue;                                      /* :1447-1449 */
            if (uright2 && uright2[i2] > 0) {                               /* :1451-1457 */
                const float ur = u - mbf * invzc;
                const float er = std::fabs(ur - uright2[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_hamming256(dMP, d2 + 32 * i2);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = (int)i2;
            }
        }
        if (bestDist <= maxDist) {                                          /* TH_HIGH :1470, ORBdist :1604 */
            holder[bestIdx2] = i;                                           /* :1472 */
            blocked[bestIdx2] = has_obs[i] != 0;
            nmatches++;
            if (checkOri) {
                float rot = last_angle[i] - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot / (360 * factor));
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) { /* ComputeThreeMaxima (:1652-1693) */
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx2 : rotHist[i]) {                                   /* :1506-1510: unconditional */
                holder[idx2] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < n2; i++) match2[i] = holder[i];
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:41-123 */
extern "C" int orc_search_by_projection_points(const float* proj, const float* proj_xr, const float* view_cos,
                                               const int32_t* level, const uint8_t* desc_mp,
                                               const uint8_t* has_obs, int npts, const orc_keypoint* k2,
                                               const uint8_t* d2, const float* uright2, const uint8_t* occupied2,
                                               int n2, float minX, float minY, float maxX, float maxY,
                                               const float* scale_factors, int nlevels, float th, float mfNNratio,
                                               int32_t* match2) {
    const int COLS = 64, ROWS = 48, TH_HIGH = 100;
    const float invW = static_cast<float>(COLS) / (maxX - minX), invH = static_cast<float>(ROWS) / (maxY - minY);
    std::vector<std::vector<uint32_t>> grid(COLS * ROWS);
    for (int i = 0; i < n2; i++) {
        const int posX = (int)std::round((k2[i].x - minX) * invW), posY = (int)std::round((k2[i].y - minY) * invH);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[posX * ROWS + posY].push_back((uint32_t)i);
    }
    auto features_in_area = [&](float x, float y, float r, int minLevel, int maxLevel) {   /* src/Frame.cc:365-418 */
        std::vector<size_t> v;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
        if (nMinCellX >= COLS) return v;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - minX + r) * invW));
        if (nMaxCellX < 0) return v;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
        if (nMinCellY >= ROWS) return v;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - minY + r) * invH));
        if (nMaxCellY < 0) return v;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (uint32_t j : grid[ix * ROWS + iy]) {
                    const orc_keypoint& kp = k2[j];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(j);
                }
        return v;
    };
    std::vector<int32_t> holder(n2, -1);
    std::vector<uint8_t> blocked(n2, 0);
    for (int i = 0; i < n2; i++) blocked[i] = occupied2 ? (occupied2[i] != 0) : 0;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < npts; iMP++) {
        const int nPredictedLevel = level[iMP];
        if (nPredictedLevel < 0 || nPredictedLevel >= nlevels) continue;    /* (not a value PredictScale returns) */
        float r = view_cos[iMP] > 0.998 ? 2.5 : 4.0;                        /* RadiusByViewingCos (:125-131) */
        if (bFactor) r *= th;
        const std::vector<size_t> vIndices = features_in_area(proj[2 * iMP], proj[2 * iMP + 1],
                                                              r * scale_factors[nPredictedLevel],
                                                              nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = desc_mp + 32 * (size_t)iMP;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t idx : vIndices) {
            if (blocked[idx]) continue;
            if (uright2 && uright2[idx] > 0) {
                const float er = std::fabs(proj_xr[iMP] - uright2[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = orc_hamming256(MPdescriptor, d2 + 32 * idx);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = k2[idx].octave;
                bestIdx = (int)idx;
            } else if (dist < bestDist2) {
                bestLevel2 = k2[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
            holder[bestIdx] = iMP;
            blocked[bestIdx] = has_obs[iMP] != 0;
            nmatches++;
        }
    }
    for (int i = 0; i < n2; i++) match2[i] = holder[i];
    return nmatches;
}

