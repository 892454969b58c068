// Host-side tables: scale factors, per-level quotas, pyramid / cell-grid geometry and the
// fixed-point coefficient tables of the bilinear resize. Shared by every frame of a batch, so the
// coefficient rounding is done once, on the host, in the exact float arithmetic the reference uses.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "orbx_internal.h"

namespace orbx {

namespace {
thread_local char g_err[512] = "";

inline int cv_round(double v) { return (int)std::nearbyint(v); }   // round half to even
inline int cv_round(float v) { return (int)std::nearbyintf(v); }
inline int16_t sat_short(float v) {
    int i = cv_round(v);
    return (int16_t)std::min(32767, std::max(-32768, i));
}
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
}  // namespace

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* orbx_last_error(void) { return g_err; }

// ORBextractor::ORBextractor (src/ORBextractor.cc:402-462). scaleFactor is a double member there
// (include/ORBextractor.h:113), so the cumulative product and 1/scaleFactor go through double.
void build_scale_tables(int nfeatures, float scaleFactor, int nlevels, ScaleTables& t) {
    const double sf = scaleFactor;
    t.scale.assign(nlevels, 1.0f);
    t.sigma2.assign(nlevels, 1.0f);
    for (int i = 1; i < nlevels; i++) {
        t.scale[i] = (float)(t.scale[i - 1] * sf);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    t.invScale.resize(nlevels);
    t.invSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        t.invScale[i] = 1.0f / t.scale[i];
        t.invSigma2[i] = 1.0f / t.sigma2[i];
    }
    t.perLevel.assign(nlevels, 0);
    const float factor = (float)(1.0f / sf);
    float want = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        t.perLevel[l] = cv_round(want);
        sum += t.perLevel[l];
        want *= factor;
    }
    t.perLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

    // end of each row of the radius-15 disc used by the intensity centroid (:446-461)
    int um[kHalfPatch + 2] = {0};
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = (double)kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) um[v] = cv_round(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (um[v0] == um[v0 + 1]) ++v0;
        um[v] = v0;
        ++v0;
    }
    for (int v = 0; v <= kHalfPatch; v++) t.umax[v] = um[v];
}

// Level sizes (src/ORBextractor.cc:1103-1104), FAST cell grid (:765-779) and quadtree roots
// (:535-537) for one image size, plus where each level's data lives inside a frame's buffers.
// Cell rectangles of ORBextractor::ComputeKeyPointsOctTree (src/ORBextractor.cc:783-798): cell (i, j) of a level starts at
// minBorder + (i * hCell, j * wCell), is wCell + 6 by hCell + 6 pixels and is cut at the level's far borders; the
// reference skips a cell that starts within 3 rows (6 columns) of them.
bool build_cell_table(const PyramidGeom& g, std::vector<CellRecord>& cells) {
    cells.assign((size_t)std::max(g.totalCells, 0), CellRecord{});
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
        for (int ci = 0; ci < L.nRows; ci++)
            for (int cj = 0; cj < L.nCols; cj++) {
                const int c = ci * L.nCols + cj;
                const int iniX = kBorder + cj * L.wCell, iniY = kBorder + ci * L.hCell;
                const int maxX = std::min(iniX + L.wCell + 6, maxBX), maxY = std::min(iniY + L.hCell + 6, maxBY);
                const int sw = maxX - iniX - 6, sh = maxY - iniY - 6;
                const bool empty = iniY >= maxBY - 3 || iniX >= maxBX - 6 || sw <= 0 || sh <= 0;
                const long long slotOff = L.slotBase + (long long)c * L.slotCap;
                if (L.cellBase + c >= g.totalCells || iniX > 0xFFFF || iniY > 0xFFFF || slotOff > 0x7FFFFFFFll ||
                    L.offset > 0x7FFFFFFFll || L.h > 0xFFFFFF || L.slotCap > 0xFFFF)
                    return false;
                CellRecord& r = cells[(size_t)L.cellBase + c];
                r.ini = (uint32_t)iniX | (uint32_t)iniY << 16;
                // an empty cell's far corner may lie left of or above its first one: it is never looked at
                r.max = empty ? r.ini : ((uint32_t)maxX | (uint32_t)maxY << 16);
                r.slotOff = (uint32_t)slotOff;
                r.levelOff = l ? (uint32_t)L.offset : 0u;
                r.stride = l ? (uint32_t)L.stride : 0u;
                r.levelH = (uint32_t)l | (uint32_t)L.h << 8;
                r.slotCap = (uint32_t)L.slotCap | (empty ? 0x80000000u : 0u);
            }
    }
    return true;
}

bool build_geometry(int width, int height, const ScaleTables& t, PyramidGeom& g, std::string& err) {
    const int nlevels = (int)t.scale.size();
    g = PyramidGeom();
    g.nlevels = nlevels;
    g.width = width;
    g.height = height;
    long long pyr = 0, slots = 0;
    int cells = 0, sel = 0, maxWCell = 0, maxHCell = 0;
    g.nodeCap = 16;
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& L = g.lv[l];
        L.w = cv_round((float)width * t.invScale[l]);
        L.h = cv_round((float)height * t.invScale[l]);
        if (L.w < 1 || L.h < 1) { L.w = 0; L.h = 0; }
        L.stride = align_up(std::max(L.w, 1), 64);
        L.offset = pyr;
        if (l > 0) pyr += (long long)L.stride * L.h;
        pyr = (pyr + 255) / 256 * 256;
        L.quota = t.perLevel[l];
        L.scale = t.scale[l];
        L.kpSize = (float)(int)(kPatchSize * t.scale[l]);
        const int minB = kBorder, maxBX = L.w - kBorder, maxBY = L.h - kBorder;
        const float fw = (float)(maxBX - minB), fh = (float)(maxBY - minB);
        int nCols = maxBX > minB ? (int)(fw / kCellSize) : 0;
        int nRows = maxBY > minB ? (int)(fh / kCellSize) : 0;
        int nIni = 0;
        if (nCols >= 1 && nRows >= 1) nIni = (int)std::round(fw / fh);
        // The reference divides by zero when a level is narrower than one cell or much taller
        // than wide; such levels produce no keypoints here.
        if (nCols < 1 || nRows < 1 || nIni < 1) { nCols = nRows = nIni = 0; }
        L.nCols = nCols;
        L.colsMagic = nCols > 1 ? (uint32_t)((1ull << 32) / (unsigned)nCols + 1) : 0u;   // exact for indices below 2^16
        L.nRows = nRows;
        L.wCell = nCols ? (int)std::ceil(fw / nCols) : 0;
        L.hCell = nRows ? (int)std::ceil(fh / nRows) : 0;
        L.nIni = nIni;
        L.hX = nIni ? fw / nIni : 0.f;
        L.cellBase = cells;
        L.slotCap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
        L.slotBase = slots;
        L.selBase = sel;
        cells += nCols * nRows;
        slots += (long long)nCols * nRows * L.slotCap;
        L.selCap = std::max(L.quota + kSelPad, 4 * nIni + 1);
        sel += L.selCap;
        g.nodeCap = std::max(g.nodeCap, std::max(L.quota + 8, 4 * nIni + 8));
        if (nCols) {
            maxWCell = std::max(maxWCell, L.wCell);
            maxHCell = std::max(maxHCell, L.hCell);
            g.maxScanPix = std::max(g.maxScanPix, L.wCell * L.hCell);
        }
        if (L.w > 4096 + 2 * kBorder || L.h > 4096 + 2 * kBorder) {
            err = "image larger than 4096 px per side";
            return false;
        }
    }
    // A staged cell row holds one byte of alignment slack + the cell + 6 px, rounded up to dwords; every level of a
    // launch uses the same pitch: 40 bytes when all cells fit (cells up to 33 px wide: the usual 30-32 px cells),
    // else 48, else 80. The smaller the pitch, the more workgroups share a CU's LDS.
    {
        const int rowBytes = (1 + maxWCell + 6 + 3) & ~3;
        g.tilePitch = rowBytes <= 40 ? 40 : ((3 + maxWCell + 6 + 3) & ~3) <= 48 ? 48 : 80;
    }
    if (maxWCell > 64 || maxHCell > 64) {   // cannot happen: a cell is width / floor(width / 30) < 60 px
        err = "FAST cell larger than 64 px";
        return false;
    }
    g.maxTileBytes = g.tilePitch * ((maxHCell + 6 + 39) / 40 * 40);   // fast_cells stages rows in batches of 40
    g.maxScoreBytes = align_up(g.tilePitch * (maxHCell + 2), 16);      // the score map has the tile's pitch
    g.nodeCap = (g.nodeCap + 3) & ~3;         // the quadtree reads node keys four at a time
    g.totalCells = cells;
    g.totalSlots = slots;
    g.totalSel = sel;
    g.pyrBytes = std::max(pyr, 256LL);
    return true;
}

// Coefficient tables of cv::resize(INTER_LINEAR) for CV_8U: offsets from
// fx = (float)((dx+0.5)*scale - 0.5), weights rounded to 11-bit fixed point (cv: resize.cpp,
// INTER_RESIZE_COEF_BITS = 11). Called for src/ORBextractor.cc:1112.
void build_resize_table(int sw, int sh, int dw, int dh, HostResizeTab& t) {
    t.xofs.assign(std::max(dw, 1), 0);
    t.yofs.assign(std::max(dh, 1), 0);
    t.ialpha.assign(2 * (size_t)std::max(dw, 1), 0);
    t.ibeta.assign(2 * (size_t)std::max(dh, 1), 0);
    t.xmax = dw;
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return;
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { t.xmax = std::min(t.xmax, dx); fx = 0; sx = sw - 1; }
        t.xofs[dx] = sx;
        t.ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
        t.ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        t.yofs[dy] = sy;
        t.ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
        t.ibeta[2 * dy + 1] = sat_short(fy * 2048);
    }
}

int resize_max_span(const HostResizeTab& t) {
    const size_t n = t.xofs.size(), n4 = (n + 3) / 4 * 4;
    int maxSpan = 0;
    for (size_t i = 0; i < n4; i++) {
        const size_t j = std::min(i, n - 1), j0 = std::min(i & ~(size_t)3, n - 1);
        maxSpan = std::max(maxSpan, t.xofs[j] - t.xofs[j0]);
    }
    return maxSpan;
}

}  // namespace orbx
