// C ABI of liborbx (include/orbx.h): context management, buffer layout and kernel sequencing.
// Host code only; kernels live in pyramid.hip, fast_cells.hip, octree.hip, orient_desc.hip and
// match.hip. There is no CPU fallback: without a usable HIP device every compute entry point
// returns ORBX_ERR_HIP.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "orbx_internal.h"

using namespace orbx;

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                      __LINE__);                                                        \
            return ORBX_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

// Stream ordering. Several streams work on one context's buffers (the caller's streams in turn, the matcher's
// stream during stereo matching) and are ordered by events only, never by the host. The rule every path here obeys:
// a hipStreamWaitEvent names an event whose hipEventRecord THIS host thread has already submitted. Waits then depend
// only on work submitted earlier, so the dependencies form no cycle, and on a device that multiplexes many streams
// over a few hardware queues (four per process here) a wait packet can never sit in front of the record it waits
// for. (An arrangement that queued one stream's whole chain, waits included, before the other stream's records
// did hang on this box.) OrderedEvent keeps the "has been recorded" fact and wait_event() refuses a wait that
// would break the rule instead of queuing it.
struct OrderedEvent {
    hipEvent_t ev = nullptr;
    unsigned long long recorded = 0;      // sequence number of the last record, 0 = never
};
namespace {
std::atomic<unsigned long long> g_eventSeq{0};   // contexts may be driven from different host threads (one each)
hipError_t record_event(OrderedEvent& e, hipStream_t s) {
    hipError_t r = hipEventRecord(e.ev, s);
    if (r == hipSuccess) e.recorded = ++g_eventSeq;
    return r;
}
int wait_event(hipStream_t s, const OrderedEvent& e, const char* what) {
    if (!e.recorded) {
        set_error("stream order: wait for '%s' before any record of it was submitted", what);
        return ORBX_ERR_STATE;
    }
    HIP_TRY(hipStreamWaitEvent(s, e.ev, 0));
    return ORBX_OK;
}
}  // namespace

struct orbx_ctx {
    orbx_config cfg;
    ScaleTables tab;
    hipStream_t stream = nullptr;
    // geometry of the image size seen last
    PyramidGeom geom;
    bool haveGeom = false;
    ResizeTab resize[ORBX_MAX_LEVELS];
    void* resizeBlob = nullptr;
    void* cellBlob = nullptr;     // CellRecord[totalCells], then SelRecord[totalSel], of the current geometry
    FusedPlan fused;              // grouped pyramid launches for this geometry (empty: one launch per level)
    void* fusedBlob = nullptr;
    // device buffers sized for max_width x max_height x max_batch
    PyramidGeom maxGeom;
    FrameSet fs;
    // Host-buffer entry points: frames go through stageImg; counts, keypoints and descriptors of a call are
    // packed into one device block (stageOut) that comes back in one copy to its pinned twin (pinOut). Small
    // inputs are packed into pinned memory first (pinIn), whatever the caller's strides.
    uint8_t* stageImg = nullptr;
    uint8_t* stageOut = nullptr;
    uint8_t* pinOut = nullptr;
    uint8_t* pinIn = nullptr;
    size_t pinInBytes = 0;
    int stageCap = 0;
    // last call
    int lastFrames = 0;
    hipStream_t lastStream = nullptr;
    // stage timing
    bool profiling = false;
    bool haveTimes = false;
    hipEvent_t ev[ORBX_NUM_STAGES + 1] = {};
    // recorded behind every batched call: a later call on another stream waits for it before it reuses the
    // context's buffers
    OrderedEvent evDone;
    // Host-buffer calls of a few frames replay their chain (copy in, four kernels, copy out) as one HIP graph per
    // (size, frame count, capacity); every address in it belongs to the context's own staging buffers. Dropped
    // whenever a buffer or table the kernels were given is reallocated.
    struct Replay {
        int w, h, n, cap;
        hipGraphExec_t exec;
    };
    std::vector<Replay> replays;
    bool capturing = false;      // the batched entry point is being recorded into a graph: no events of its own
    int useGraph = -1;           // ORBX_GRAPH=0 switches the replay path off (-1: not read yet)
};

namespace {
void drop_replays(orbx_ctx* c) {
    for (auto& r : c->replays) (void)hipGraphExecDestroy(r.exec);
    c->replays.clear();
}
}  // namespace

namespace {

template <typename T>
int dev_alloc(T** p, size_t count) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(count * sizeof(T), 256)));
    *p = static_cast<T*>(q);
    return ORBX_OK;
}

// Set-up steps that allocate, free or copy synchronously (table uploads, staging buffers) and the recording of a launch
// chain into a HIP graph exclude each other across the host threads of the process: a device-wide synchronisation
// issued by one thread while another thread's stream is recording invalidates that recording. Both are rare (once per
// context and geometry) and short.
std::mutex g_setupMutex;

int ensure_geometry(orbx_ctx* c, int w, int h) {
    if (c->haveGeom && c->geom.width == w && c->geom.height == h) return ORBX_OK;
    std::lock_guard<std::mutex> setup(g_setupMutex);
    std::string err;
    PyramidGeom g;
    if (!build_geometry(w, h, c->tab, g, err)) {
        set_error("%s", err.c_str());
        return ORBX_ERR_INVALID;
    }
    if (g.totalCells > c->maxGeom.totalCells || g.totalSlots > c->maxGeom.totalSlots ||
        g.totalSel > c->maxGeom.totalSel || octree_lds_bytes(g) > 160 * 1024 ||
        g.pyrBytes > c->maxGeom.pyrBytes || w > c->cfg.max_width || h > c->cfg.max_height) {
        set_error("image %dx%d exceeds the context's maximum %dx%d", w, h, c->cfg.max_width,
                  c->cfg.max_height);
        return ORBX_ERR_INVALID;
    }
    // resize tables for levels 1..n-1, one blob
    std::vector<HostResizeTab> ht(g.nlevels);
    size_t bytes = 0;
    auto up = [](size_t v) { return (v + 63) / 64 * 64; };
    for (int l = 1; l < g.nlevels; l++) {
        build_resize_table(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, ht[l]);
        bytes += up(ht[l].xofs.size() * 4) + up(ht[l].yofs.size() * 4) + up(ht[l].ialpha.size() * 2) +
                 up(ht[l].ibeta.size() * 2) + up((ht[l].xofs.size() + 4) * 8);
    }
    std::vector<uint8_t> blob(std::max<size_t>(bytes, 64));
    HIP_TRY(hipStreamSynchronize(c->stream));   // the old tables may still be in use
    if (c->lastStream && c->lastStream != c->stream) HIP_TRY(hipStreamSynchronize(c->lastStream));
    drop_replays(c);                            // recorded launches carry the old tables' addresses
    c->haveGeom = false;                        // until the new tables are all in place
    if (c->resizeBlob) HIP_TRY(hipFree(c->resizeBlob));
    c->resizeBlob = nullptr;
    HIP_TRY(hipMalloc(&c->resizeBlob, blob.size()));
    size_t off = 0;
    uint8_t* dbase = static_cast<uint8_t*>(c->resizeBlob);
    for (int l = 1; l < g.nlevels; l++) {
        auto put = [&](const void* src, size_t n) {
            std::memcpy(blob.data() + off, src, n);
            const uint8_t* d = dbase + off;
            off += up(n);
            return d;
        };
        c->resize[l].xofs = reinterpret_cast<const int32_t*>(put(ht[l].xofs.data(), ht[l].xofs.size() * 4));
        c->resize[l].yofs = reinterpret_cast<const int32_t*>(put(ht[l].yofs.data(), ht[l].yofs.size() * 4));
        c->resize[l].ialpha = reinterpret_cast<const int16_t*>(put(ht[l].ialpha.data(), ht[l].ialpha.size() * 2));
        c->resize[l].ibeta = reinterpret_cast<const int16_t*>(put(ht[l].ibeta.data(), ht[l].ibeta.size() * 2));
        c->resize[l].xmax = ht[l].xmax;
        // packed per-column records, padded to a multiple of four with copies of the last column. The
        // first column of every group of four carries its source column; the others carry the v_perm_b32
        // selector that picks bytes d and d+1 (d = their source column minus the group's first) out of an
        // 8-byte window into the low and high 16-bit halves (0x0c = constant zero byte).
        const size_t n = ht[l].xofs.size(), n4 = (n + 3) / 4 * 4;
        std::vector<int32_t> xe(2 * n4);
        int maxSpan = 0;
        for (size_t i = 0; i < n4; i++) {
            const size_t j = std::min(i, n - 1), j0 = std::min(i & ~(size_t)3, n - 1);
            const int d = ht[l].xofs[j] - ht[l].xofs[j0];
            maxSpan = std::max(maxSpan, d);
            const uint32_t dd = (uint32_t)std::min(std::max(d, 0), 6);
            xe[2 * i] = (i & 3) == 0 ? ht[l].xofs[j] : (int32_t)(0x0c000c00u | dd | ((dd + 1) << 16));
            xe[2 * i + 1] = (int32_t)((uint32_t)(uint16_t)ht[l].ialpha[2 * j] |
                                      ((uint32_t)(uint16_t)ht[l].ialpha[2 * j + 1] << 16));
        }
        c->resize[l].maxSpan = maxSpan;
        c->resize[l].xentry = reinterpret_cast<const int32_t*>(put(xe.data(), xe.size() * 4));
    }
    HIP_TRY(hipMemcpy(c->resizeBlob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    {
        std::vector<CellRecord> cells;
        if (!build_cell_table(g, cells)) {
            set_error("image %dx%d: a FAST cell does not fit its table record", w, h);
            return ORBX_ERR_INVALID;
        }
        // behind the cells: level and level geometry of every entry of a frame's selected-keypoint regions (the
        // descriptor kernel's waves are numbered by entry)
        std::vector<SelRecord> selRec((size_t)std::max(g.totalSel, 0), SelRecord{});
        for (int l = 0; l < g.nlevels; l++) {
            const LevelGeom& L = g.lv[l];
            if (L.offset > 0x7FFFFFFFll || L.w > 0xFFFF || L.h > 0xFFFF) {
                set_error("image %dx%d: a level does not fit its table record", w, h);
                return ORBX_ERR_INVALID;
            }
            for (int i = L.selBase; i < g.totalSel; i++) {
                SelRecord& r = selRec[(size_t)i];
                r.levelK = (uint32_t)l | (uint32_t)(i - L.selBase) << 8;
                r.levelOff = l ? (uint32_t)L.offset : 0u;
                r.stride = l ? (uint32_t)L.stride : 0u;
                r.wh = (uint32_t)L.w | (uint32_t)L.h << 16;
                r.scale = L.scale;
                r.kpSize = L.kpSize;
            }
        }
        if (c->cellBlob) HIP_TRY(hipFree(c->cellBlob));
        c->cellBlob = nullptr;
        c->fs.cells = nullptr;
        c->fs.selRec = nullptr;
        const size_t cellBytes = cells.size() * sizeof(CellRecord), selBytes = selRec.size() * sizeof(SelRecord);
        HIP_TRY(hipMalloc(&c->cellBlob, std::max<size_t>(cellBytes + selBytes, 64)));
        if (cellBytes) HIP_TRY(hipMemcpy(c->cellBlob, cells.data(), cellBytes, hipMemcpyHostToDevice));
        if (selBytes) HIP_TRY(hipMemcpy(static_cast<uint8_t*>(c->cellBlob) + cellBytes, selRec.data(), selBytes, hipMemcpyHostToDevice));
        c->fs.cells = static_cast<const CellRecord*>(c->cellBlob);
        c->fs.selRec = reinterpret_cast<const SelRecord*>(static_cast<uint8_t*>(c->cellBlob) + cellBytes);
    }
    // grouped pyramid launches: range tables next to the resize tables
    if (c->fusedBlob) HIP_TRY(hipFree(c->fusedBlob));
    c->fusedBlob = nullptr;
    // (tiles of 16 x 8 pixels on the smallest level, all levels in one launch: the shape that gave the shortest
    // single-frame pyramid, 17 us against 44 us for seven dependent launches)
    plan_fused_pyramid(g, ht, c->resize, 16, 8, kFusedMaxLevels, c->fused);
    if (!c->fused.groups.empty()) {
        const size_t bytes = c->fused.ranges.size() * sizeof(FusedRange);
        HIP_TRY(hipMalloc(&c->fusedBlob, bytes));
        HIP_TRY(hipMemcpy(c->fusedBlob, c->fused.ranges.data(), bytes, hipMemcpyHostToDevice));
        const FusedRange* base = static_cast<const FusedRange*>(c->fusedBlob);
        for (FusedGroup& G : c->fused.groups) {
            G.cols = base + reinterpret_cast<size_t>(G.cols);
            G.rows = base + reinterpret_cast<size_t>(G.rows);
        }
    }
    c->geom = g;
    c->haveGeom = true;
    return ORBX_OK;
}

int max_root_nodes(const orbx_config& cfg);
constexpr int kFusedMaxFrames = 8;      // largest batch that takes the one-launch pyramid

int check_ctx(const orbx_ctx* c) {
    if (!c) {
        set_error("null context");
        return ORBX_ERR_INVALID;
    }
    return ORBX_OK;
}

}  // namespace

extern "C" int orbx_create(const orbx_config* cfg, orbx_ctx** out) {
    if (!cfg || !out) {
        set_error("null argument");
        return ORBX_ERR_INVALID;
    }
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > ORBX_MAX_LEVELS || cfg->nfeatures < 1 ||
        !(cfg->scale_factor > 1.0f) || cfg->max_width < 1 || cfg->max_height < 1 || cfg->max_batch < 1 ||
        cfg->max_width > 4096 || cfg->max_height > 4096 || cfg->ini_th_fast < 1 || cfg->min_th_fast < 1 ||
        cfg->ini_th_fast > 255 || cfg->min_th_fast > 255 ||
        (cfg->blur_taps != ORBX_BLUR_CV_4_0_0 && cfg->blur_taps != ORBX_BLUR_CV_4_2_PLUS)) {
        set_error("invalid orbx_config");
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(cfg->device));
    orbx_ctx* c = new (std::nothrow) orbx_ctx();
    if (!c) {
        set_error("out of host memory");
        return ORBX_ERR_INVALID;
    }
    c->cfg = *cfg;
    build_scale_tables(cfg->nfeatures, cfg->scale_factor, cfg->nlevels, c->tab);
    std::string err;
    if (!build_geometry(cfg->max_width, cfg->max_height, c->tab, c->maxGeom, err)) {
        set_error("%s", err.c_str());
        delete c;
        return ORBX_ERR_INVALID;
    }
    if (octree_lds_bytes(c->maxGeom) > 160 * 1024) {
        set_error("nfeatures too large for the quadtree's LDS node table");
        delete c;
        return ORBX_ERR_INVALID;
    }
    // Cell grids are not monotone in the image size (wCell = ceil(width / nCols)), so leave head
    // room: smaller images can have a few more cells or larger per-cell slot capacities.
    PyramidGeom& mg = c->maxGeom;
    mg.totalCells = mg.totalCells + mg.totalCells / 4 + 64;
    mg.totalSlots = mg.totalSlots + mg.totalSlots / 2 + 4096;
    mg.pyrBytes += 4096;
    // The selected-keypoint region of a level holds max(quota + 4, 4 * nIni + 1) entries and nIni =
    // round(width / height) of the level grows with the aspect ratio, not with the size: a flat strip of the
    // largest width has the most roots any admissible image can have (a level needs 30 rows inside its border to
    // have cells at all).
    {
        const int nIniMax = max_root_nodes(*cfg);
        int sel = 0;
        for (int l = 0; l < mg.nlevels; l++) sel += std::max(c->tab.perLevel[l] + kSelPad, 4 * nIniMax + 1);
        mg.totalSel = std::max(mg.totalSel, sel);
    }
    const size_t B = (size_t)cfg->max_batch;
    int rc = ORBX_OK;
    auto fail = [&](int code) {
        orbx_destroy(c);
        return code;
    };
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(he));
        return fail(ORBX_ERR_HIP);
    }
    he = hipEventCreateWithFlags(&c->evDone.ev, hipEventDisableTiming);
    if (he != hipSuccess) {
        set_error("hipEventCreate failed: %s", hipGetErrorString(he));
        return fail(ORBX_ERR_HIP);
    }
    if ((rc = dev_alloc(&c->fs.pyr, B * (size_t)mg.pyrBytes))) return fail(rc);
    if ((rc = dev_alloc(&c->fs.cellCount, B * (size_t)mg.totalCells))) return fail(rc);
    if ((rc = dev_alloc(&c->fs.slots, B * (size_t)mg.totalSlots))) return fail(rc);
    if ((rc = dev_alloc(&c->fs.cand, B * (size_t)mg.totalSlots))) return fail(rc);
    if ((rc = dev_alloc(&c->fs.nid, B * (size_t)mg.totalSlots))) return fail(rc);
    if ((rc = dev_alloc(&c->fs.sel, B * (size_t)mg.totalSel))) return fail(rc);
    if ((rc = dev_alloc(&c->fs.selCount, 2 * B * (size_t)cfg->nlevels))) return fail(rc);
    c->fs.selBefore = c->fs.selCount + B * (size_t)cfg->nlevels;
    *out = c;
    return ORBX_OK;
}

extern "C" void orbx_destroy(orbx_ctx* c) {
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_replays(c);
    void* bufs[] = {c->fs.pyr, c->fs.cellCount, c->fs.slots, c->fs.cand, c->fs.nid, c->fs.sel,
                    c->fs.selCount, c->resizeBlob, c->cellBlob, c->fusedBlob, c->stageImg, c->stageOut};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (c->pinOut) (void)hipHostFree(c->pinOut);
    if (c->pinIn) (void)hipHostFree(c->pinIn);
    for (hipEvent_t e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->evDone.ev) (void)hipEventDestroy(c->evDone.ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orbx_get_levels(const orbx_ctx* c) { return c ? c->cfg.nlevels : 0; }
extern "C" float orbx_get_scale_factor(const orbx_ctx* c) { return c ? c->cfg.scale_factor : 0.f; }
#define ORBX_GETTER(name, field)                                          \
    extern "C" int name(const orbx_ctx* c, float* out) {                  \
        if (!c || !out) return ORBX_ERR_INVALID;                          \
        std::copy(c->tab.field.begin(), c->tab.field.end(), out);         \
        return ORBX_OK;                                                   \
    }
ORBX_GETTER(orbx_get_scale_factors, scale)
ORBX_GETTER(orbx_get_inverse_scale_factors, invScale)
ORBX_GETTER(orbx_get_scale_sigma_squares, sigma2)
ORBX_GETTER(orbx_get_inverse_scale_sigma_squares, invSigma2)
extern "C" int orbx_get_features_per_level(const orbx_ctx* c, int32_t* out) {
    if (!c || !out) return ORBX_ERR_INVALID;
    std::copy(c->tab.perLevel.begin(), c->tab.perLevel.end(), out);
    return ORBX_OK;
}
extern "C" int orbx_get_umax(const orbx_ctx* c, int32_t* out16) {
    if (!c || !out16) return ORBX_ERR_INVALID;
    std::copy(c->tab.umax, c->tab.umax + kHalfPatch + 1, out16);
    return ORBX_OK;
}
namespace {
// Most quadtree roots any image a context accepts can have: nIni = round(width / height) of a level inside its
// border (src/ORBextractor.cc:535), largest for a strip of the full width and the 30 rows a level needs for one cell.
int max_root_nodes(const orbx_config& cfg) {
    return (int)std::lround((double)std::max(cfg.max_width - 2 * kBorder, 0) / kCellSize) + 1;
}
}  // namespace

extern "C" int orbx_keypoint_capacity(const orbx_ctx* c) {
    if (!c) return 0;
    // per level max(quota + 3, 4 * nIni), with the largest nIni any admissible image has
    const int nIniMax = max_root_nodes(c->cfg);
    int cap = 0;
    for (int l = 0; l < c->maxGeom.nlevels; l++) cap += std::max(c->tab.perLevel[l] + 3, 4 * nIniMax);
    return cap;
}

namespace {
constexpr size_t kPinInMax = 16u << 20;   // host inputs up to this size are packed through pinned memory

// Device staging for frames (max_batch frames of the largest size, 4-byte-multiple row pitch) and its small
// pinned twin on the host.
int ensure_stage_img(orbx_ctx* c) {
    if (c->stageImg) return ORBX_OK;
    const size_t all = (size_t)c->cfg.max_batch * (((size_t)c->cfg.max_width + 3) & ~(size_t)3) * c->cfg.max_height;
    int rc = dev_alloc(&c->stageImg, all);
    if (rc) return rc;
    void* p = nullptr;
    c->pinInBytes = std::min(all, kPinInMax);
    HIP_TRY(hipHostMalloc(&p, c->pinInBytes, hipHostMallocDefault));
    c->pinIn = static_cast<uint8_t*>(p);
    return ORBX_OK;
}
}  // namespace

extern "C" int orbx_extract_batch_device(orbx_ctx* c, const uint8_t* d_images, int nframes, int width,
                                         int height, size_t row_stride, size_t frame_stride,
                                         orbx_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts,
                                         void* stream) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (nframes == 0) return ORBX_OK;
    if (!d_images || !d_kps || !d_desc || !d_counts || nframes < 0 || nframes > c->cfg.max_batch ||
        width < 1 || height < 1 || row_stride < (size_t)width || cap < 1 ||
        (nframes > 1 && frame_stride < row_stride * (size_t)(height - 1) + (size_t)width)) {
        set_error("orbx_extract_batch_device: bad argument (nframes %d of max %d, %dx%d, cap %d)", nframes,
                  c->cfg.max_batch, width, height, cap);
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if ((rc = ensure_geometry(c, width, height))) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    // The context's buffers (pyramid, candidates, selection, staging) are shared by all its calls: work queued on
    // another stream than the previous call's starts only after that call has finished with them.
    if (!c->capturing && c->lastStream && c->lastStream != s && (rc = wait_event(s, c->evDone, "context done"))) return rc;
    FrameSet fs = c->fs;
    if (nframes == 1) frame_stride = 0;                 // never used to address anything
    if ((uintptr_t)d_images % 4 || row_stride % 4 || frame_stride % 4) {
        // The kernels read level 0 in place with dword loads: frames that are not 4-byte aligned (odd widths
        // packed tightly, views into larger images) are first copied into the aligned staging layout.
        if ((rc = ensure_stage_img(c))) return rc;
        const size_t pitch = ((size_t)width + 3) & ~(size_t)3;
        HIP_TRY(launch_repack_frames(d_images, (long long)row_stride, (long long)frame_stride, c->stageImg,
                                     (long long)pitch, (long long)(pitch * (size_t)height), width, height, nframes, s));
        d_images = c->stageImg;
        row_stride = pitch;
        frame_stride = pitch * (size_t)height;
    }
    fs.img = d_images;
    fs.imgRowStride = (long long)row_stride;
    fs.imgFrameStride = (long long)frame_stride;
    c->fs = fs;
    c->lastFrames = nframes;
    const PyramidGeom& g = c->geom;
    const bool prof = c->profiling && !c->capturing;
    // A handful of frames cannot fill the GPU whatever the kernel: what counts then is the length of the chain of
    // dependent launches, so the pyramid comes from the one-launch kernel (more arithmetic, far less waiting).
    // Batches go level by level through the kernel that spends the fewest instructions per pixel.
    auto queue = [&]() -> int {
        if (prof) HIP_TRY(hipEventRecord(c->ev[0], s));
        HIP_TRY(launch_pyramid(fs, g, c->resize, nframes, s, nframes <= kFusedMaxFrames ? &c->fused : nullptr));
        if (prof) HIP_TRY(hipEventRecord(c->ev[1], s));
        HIP_TRY(launch_fast_cells(fs, g, nframes, c->cfg.ini_th_fast, c->cfg.min_th_fast, s));
        if (prof) HIP_TRY(hipEventRecord(c->ev[2], s));
        HIP_TRY(launch_octree(fs, g, nframes, s));
        if (prof) HIP_TRY(hipEventRecord(c->ev[3], s));
        HIP_TRY(launch_orient_desc(fs, g, nframes, c->cfg.blur_taps, d_kps, d_desc, cap, d_counts, s));
        if (prof) HIP_TRY(hipEventRecord(c->ev[4], s));
        return ORBX_OK;
    };
    rc = queue();
    if (c->capturing) return rc;                // the caller records the event behind the replayed graph
    // Whatever was queued (all of the call, or its first kernels when a launch failed) uses the context's buffers:
    // the event goes behind it either way, and only then does this stream become "the stream of the last call".
    const hipError_t recorded = record_event(c->evDone, s);
    c->lastStream = s;
    if (rc) return rc;
    HIP_TRY(recorded);
    c->haveTimes = prof;
    return ORBX_OK;
}

namespace {
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// Offsets of the three outputs of one call inside the output block.
struct OutLayout {
    size_t kps, desc, total;
    OutLayout(size_t nframes, size_t cap) {
        kps = up256(nframes * sizeof(int32_t));
        desc = kps + up256(nframes * cap * sizeof(orbx_keypoint));
        total = desc + nframes * cap * ORBX_DESC_BYTES;
    }
};

int ensure_staging(orbx_ctx* c, int cap) {
    const size_t B = (size_t)c->cfg.max_batch;
    int rc = ensure_stage_img(c);
    if (rc) return rc;
    if (cap > c->stageCap) {
        std::lock_guard<std::mutex> setup(g_setupMutex);
        HIP_TRY(hipStreamSynchronize(c->stream));
        drop_replays(c);
        if (c->stageOut) HIP_TRY(hipFree(c->stageOut));
        c->stageOut = nullptr;
        if (c->pinOut) HIP_TRY(hipHostFree(c->pinOut));
        c->pinOut = nullptr;
        c->stageCap = 0;
        const OutLayout lay(B, (size_t)cap);
        if ((rc = dev_alloc(&c->stageOut, lay.total))) return rc;
        void* p = nullptr;
        HIP_TRY(hipHostMalloc(&p, lay.total, hipHostMallocDefault));
        c->pinOut = static_cast<uint8_t*>(p);
        c->stageCap = cap;
    }
    return ORBX_OK;
}
}  // namespace

extern "C" int orbx_extract_batch(orbx_ctx* c, const uint8_t* images, int nframes, int width, int height,
                                  size_t row_stride, size_t frame_stride, orbx_keypoint* kps, uint8_t* desc,
                                  int cap, int32_t* counts) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (nframes == 0) return ORBX_OK;
    if (!images || !kps || !desc || !counts || nframes < 0 || nframes > c->cfg.max_batch || width < 1 ||
        height < 1 || width > c->cfg.max_width || height > c->cfg.max_height || row_stride < (size_t)width ||
        cap < 1) {
        set_error("orbx_extract_batch: bad argument");
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    if ((rc = ensure_staging(c, cap))) return rc;
    // frames are packed on the device with a 4-byte-multiple pitch
    const size_t pitch = ((size_t)width + 3) & ~(size_t)3;
    const size_t fstride = pitch * (size_t)height;
    const size_t tail = pitch - (size_t)width;      // the last row of the caller's buffer may end at `width`
    if (fstride * (size_t)nframes <= c->pinInBytes) {
        // Small input (the one-frame-per-call case above all): repack into pinned memory on the host, whatever
        // the caller's strides, and send it with one asynchronous copy. The previous call has been waited for,
        // so the pinned buffer is free.
        if (row_stride == pitch && frame_stride == fstride) {
            std::memcpy(c->pinIn, images, fstride * (size_t)nframes - tail);
        } else {
            for (int f = 0; f < nframes; f++) {
                const uint8_t* src = images + (size_t)f * frame_stride;
                uint8_t* dst = c->pinIn + (size_t)f * fstride;
                if (row_stride == pitch) {
                    std::memcpy(dst, src, fstride - tail);
                } else {
                    for (int y = 0; y < height; y++) std::memcpy(dst + y * pitch, src + y * row_stride, (size_t)width);
                }
            }
        }
        if (c->useGraph < 0) {
            const char* e = getenv("ORBX_GRAPH");
            c->useGraph = e && e[0] == '0' ? 0 : 1;
        }
        if (c->useGraph == 1 && nframes <= kFusedMaxFrames && !c->profiling &&
            (!c->lastStream || c->lastStream == c->stream)) {
            const OutLayout lay((size_t)nframes, (size_t)cap);
            if ((rc = ensure_geometry(c, width, height))) return rc;       // tables are made outside the recording
            hipGraphExec_t exec = nullptr;
            for (auto& r : c->replays)
                if (r.w == width && r.h == height && r.n == nframes && r.cap == cap) exec = r.exec;
            if (!exec) {
                // Recorded once per (geometry, frames, capacity). A recording that does not come out whole (another
                // part of the process may have synchronised the device meanwhile) is dropped and this context goes
                // on without replays: the plain path below issues the same launches.
                std::lock_guard<std::mutex> setup(g_setupMutex);
                hipGraph_t graph = nullptr;
                hipError_t e0 = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
                int r1 = ORBX_OK;
                hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess, e4 = hipSuccess;
                if (e0 == hipSuccess) {
                    c->capturing = true;
                    e1 = hipMemcpyAsync(c->stageImg, c->pinIn, fstride * (size_t)nframes, hipMemcpyHostToDevice, c->stream);
                    // The last kernel writes counts, keypoints and descriptors straight into the pinned host block
                    // (its records leave as wide stores): no copy back, one node less in the chain.
                    uint8_t* outBlock = c->stageOut;
                    void* mapped = nullptr;
                    const char* zc = getenv("ORBX_ZEROCOPY_OUT");
                    const bool direct = !(zc && zc[0] == '0') &&
                                        hipHostGetDevicePointer(&mapped, c->pinOut, 0) == hipSuccess && mapped;
                    (void)hipGetLastError();
                    if (direct) outBlock = static_cast<uint8_t*>(mapped);
                    r1 = orbx_extract_batch_device(c, c->stageImg, nframes, width, height, pitch, fstride,
                                                   reinterpret_cast<orbx_keypoint*>(outBlock + lay.kps),
                                                   outBlock + lay.desc, cap, reinterpret_cast<int32_t*>(outBlock),
                                                   c->stream);
                    if (!direct)
                        e2 = hipMemcpyAsync(c->pinOut, c->stageOut, lay.total, hipMemcpyDeviceToHost, c->stream);
                    c->capturing = false;
                    e3 = hipStreamEndCapture(c->stream, &graph);
                }
                if (e0 == hipSuccess && r1 == ORBX_OK && e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && graph)
                    e4 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                else
                    e4 = hipErrorUnknown;
                if (graph) (void)hipGraphDestroy(graph);
                if (e4 != hipSuccess || !exec) {
                    (void)hipGetLastError();
                    exec = nullptr;
                    c->useGraph = false;
                } else {
                    if (c->replays.size() >= 4) {
                        (void)hipGraphExecDestroy(c->replays.front().exec);
                        c->replays.erase(c->replays.begin());
                    }
                    c->replays.push_back(orbx_ctx::Replay{width, height, nframes, cap, exec});
                }
            }
            if (exec) {
                HIP_TRY(hipGraphLaunch(exec, c->stream));
                c->lastStream = c->stream;
                HIP_TRY(record_event(c->evDone, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                const int32_t* hc = reinterpret_cast<const int32_t*>(c->pinOut);
                for (int f = 0; f < nframes; f++) {
                    counts[f] = hc[f];
                    const size_t n = (size_t)std::min(std::max(hc[f], 0), cap);
                    std::memcpy(kps + (size_t)f * cap, c->pinOut + lay.kps + (size_t)f * cap * sizeof(orbx_keypoint),
                                n * sizeof(orbx_keypoint));
                    std::memcpy(desc + (size_t)f * cap * ORBX_DESC_BYTES, c->pinOut + lay.desc + (size_t)f * cap * ORBX_DESC_BYTES,
                                n * ORBX_DESC_BYTES);
                }
                return ORBX_OK;
            }
        }
        HIP_TRY(hipMemcpyAsync(c->stageImg, c->pinIn, fstride * (size_t)nframes, hipMemcpyHostToDevice, c->stream));
    } else if (row_stride == pitch && frame_stride == fstride) {
        // Large input: plain 1-D copies wherever the caller's layout allows (2-D copies from pageable memory are
        // an order of magnitude slower): the whole batch at once, else frame by frame, else row-pitch conversion.
        HIP_TRY(hipMemcpyAsync(c->stageImg, images, fstride * (size_t)nframes - tail, hipMemcpyHostToDevice, c->stream));
    } else {
        for (int f = 0; f < nframes; f++) {
            if (row_stride == pitch)
                HIP_TRY(hipMemcpyAsync(c->stageImg + f * fstride, images + f * frame_stride, fstride - tail,
                                       hipMemcpyHostToDevice, c->stream));
            else
                HIP_TRY(hipMemcpy2DAsync(c->stageImg + f * fstride, pitch, images + f * frame_stride, row_stride,
                                         (size_t)width, (size_t)height, hipMemcpyHostToDevice, c->stream));
        }
    }
    const OutLayout lay((size_t)nframes, (size_t)cap);
    rc = orbx_extract_batch_device(c, c->stageImg, nframes, width, height, pitch, fstride,
                                   reinterpret_cast<orbx_keypoint*>(c->stageOut + lay.kps), c->stageOut + lay.desc, cap,
                                   reinterpret_cast<int32_t*>(c->stageOut), c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->pinOut, c->stageOut, lay.total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // only the keypoints that exist are handed over; the rest of the caller's arrays stays untouched
    const int32_t* hc = reinterpret_cast<const int32_t*>(c->pinOut);
    for (int f = 0; f < nframes; f++) {
        counts[f] = hc[f];
        const size_t n = (size_t)std::min(std::max(hc[f], 0), cap);
        std::memcpy(kps + (size_t)f * cap, c->pinOut + lay.kps + (size_t)f * cap * sizeof(orbx_keypoint),
                    n * sizeof(orbx_keypoint));
        std::memcpy(desc + (size_t)f * cap * ORBX_DESC_BYTES, c->pinOut + lay.desc + (size_t)f * cap * ORBX_DESC_BYTES,
                    n * ORBX_DESC_BYTES);
    }
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_ctx* c, const uint8_t* image, int width, int height, size_t stride,
                            orbx_keypoint* kps, uint8_t* desc, int cap, int* n) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!image || width <= 0 || height <= 0) return ORBX_OK;   // empty image: outputs untouched
    if (!n) {
        set_error("orbx_extract: null count pointer");
        return ORBX_ERR_INVALID;
    }
    int32_t count = 0;
    rc = orbx_extract_batch(c, image, 1, width, height, stride, stride * (size_t)height, kps, desc, cap, &count);
    if (rc) return rc;
    *n = count;
    if (count > cap) {
        set_error("orbx_extract: %d keypoints found, capacity %d", count, cap);
        return ORBX_ERR_CAPACITY;
    }
    return ORBX_OK;
}

extern "C" int orbx_get_stream(orbx_ctx* c, void** stream) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!stream) {
        set_error("orbx_get_stream: null argument");
        return ORBX_ERR_INVALID;
    }
    *stream = c->stream;
    return ORBX_OK;
}

extern "C" int orbx_synchronize(orbx_ctx* c) {
    int rc = check_ctx(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->lastStream && c->lastStream != c->stream) HIP_TRY(hipStreamSynchronize(c->lastStream));
    return ORBX_OK;
}

extern "C" int orbx_set_profiling(orbx_ctx* c, int enabled) {
    int rc = check_ctx(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (enabled && !c->ev[0])
        for (hipEvent_t& e : c->ev) HIP_TRY(hipEventCreate(&e));
    c->profiling = enabled != 0;
    c->haveTimes = false;
    return ORBX_OK;
}

extern "C" int orbx_get_stage_ms(orbx_ctx* c, float* ms) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!ms || !c->haveTimes) {
        set_error("orbx_get_stage_ms: no profiled call yet");
        return ORBX_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipEventSynchronize(c->ev[ORBX_NUM_STAGES]));
    for (int i = 0; i < ORBX_NUM_STAGES; i++) HIP_TRY(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    return ORBX_OK;
}

extern "C" int orbx_level_size(const orbx_ctx* c, int level, int* width, int* height) {
    if (!c || !c->haveGeom || level < 0 || level >= c->geom.nlevels || !width || !height) {
        set_error("orbx_level_size: no geometry yet or bad level");
        return ORBX_ERR_STATE;
    }
    *width = c->geom.lv[level].w;
    *height = c->geom.lv[level].h;
    return ORBX_OK;
}

namespace {
int check_probe(orbx_ctx* c, int frame, int level) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!c->haveGeom || c->lastFrames == 0 || frame < 0 || frame >= c->lastFrames || level < 0 ||
        level >= c->geom.nlevels) {
        set_error("probe: frame %d / level %d not available", frame, level);
        return ORBX_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->lastStream));
    return ORBX_OK;
}

void level_ptr(const orbx_ctx* c, int frame, int level, const uint8_t** p, size_t* stride) {
    const PyramidGeom& g = c->geom;
    if (level == 0) {
        *p = c->fs.img + (size_t)frame * c->fs.imgFrameStride;
        *stride = (size_t)c->fs.imgRowStride;
    } else {
        *p = c->fs.pyr + (size_t)frame * g.pyrBytes + g.lv[level].offset;
        *stride = (size_t)g.lv[level].stride;
    }
}

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
}  // namespace

extern "C" int orbx_get_level(orbx_ctx* c, int frame, int level, int border, uint8_t* dst, size_t dst_stride) {
    int rc = check_probe(c, frame, level);
    if (rc) return rc;
    const LevelGeom& L = c->geom.lv[level];
    if (!dst || L.w < 1 || L.h < 1) {
        set_error("orbx_get_level: null destination or empty level");
        return ORBX_ERR_INVALID;
    }
    const uint8_t* src;
    size_t stride;
    level_ptr(c, frame, level, &src, &stride);
    // one 1-D copy of the level with its device pitch (2-D copies into pageable memory are far slower); the
    // last row is copied only as far as its last pixel
    const size_t bytes = stride * (size_t)(L.h - 1) + (size_t)L.w;
    std::vector<uint8_t> tmp(bytes);
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpy(tmp.data(), src, bytes, hipMemcpyDeviceToHost));
    if (!border) {
        for (int y = 0; y < L.h; y++) std::memcpy(dst + (size_t)y * dst_stride, tmp.data() + (size_t)y * stride, L.w);
        return ORBX_OK;
    }
    // the 19-px BORDER_REFLECT_101 frame of src/ORBextractor.cc:1114-1120
    for (int y = 0; y < L.h + 2 * kEdge; y++) {
        const uint8_t* S = tmp.data() + (size_t)reflect101(y - kEdge, L.h) * stride;
        uint8_t* D = dst + (size_t)y * dst_stride;
        for (int x = 0; x < L.w + 2 * kEdge; x++) D[x] = S[reflect101(x - kEdge, L.w)];
    }
    return ORBX_OK;
}

extern "C" int orbx_debug_keypoint_blur(orbx_ctx* c, int frame, uint8_t* dst, int cap, int* n) {
    int rc = check_probe(c, frame, 0);
    if (rc) return rc;
    if (!n || (!dst && cap > 0) || cap < 0) {
        set_error("orbx_debug_keypoint_blur: bad argument");
        return ORBX_ERR_INVALID;
    }
    // The orientation + descriptor kernel of the batched call runs once more on this frame alone, with its probe
    // output switched on; keypoints and descriptors go to scratch arrays.
    const PyramidGeom& g = c->geom;
    const int kcap = orbx_keypoint_capacity(c);
    const size_t patch = (size_t)kBlurPatchRows * kBlurPatchPitch;
    FrameSet fs = c->fs;
    fs.img += (size_t)frame * fs.imgFrameStride;
    fs.pyr += (size_t)frame * g.pyrBytes;
    fs.sel += (size_t)frame * g.totalSel;
    fs.selCount += (size_t)frame * g.nlevels;
    fs.selBefore += (size_t)frame * g.nlevels;
    uint8_t* tmp = nullptr;
    const size_t oDesc = up256((size_t)kcap * sizeof(orbx_keypoint)), oCnt = oDesc + up256((size_t)kcap * ORBX_DESC_BYTES),
                 oDump = oCnt + 256, total = oDump + (size_t)kcap * patch;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tmp), total));
    int32_t count = 0;
    hipError_t e = hipMemsetAsync(tmp + oDump, 0, (size_t)kcap * patch, c->stream);
    if (e == hipSuccess)
        e = launch_orient_desc(fs, g, 1, c->cfg.blur_taps, reinterpret_cast<orbx_keypoint*>(tmp), tmp + oDesc, kcap,
                               reinterpret_cast<int32_t*>(tmp + oCnt), c->stream, tmp + oDump);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(&count, tmp + oCnt, sizeof(count), hipMemcpyDeviceToHost);
    if (e == hipSuccess && count > 0 && cap > 0)
        e = hipMemcpy(dst, tmp + oDump, (size_t)std::min(count, cap) * patch, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    HIP_TRY(e);
    *n = count;
    return ORBX_OK;
}

extern "C" int orbx_debug_check_pyramid_plan(int width, int height, float scale_factor, int nlevels, int tile_w,
                                             int tile_h, int group_levels, int* ngroups) {
    if (width < 1 || height < 1 || !(scale_factor > 1.0f) || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || tile_w < 4 ||
        tile_h < 1 || group_levels < 1 || !ngroups) {
        set_error("orbx_debug_check_pyramid_plan: bad argument");
        return ORBX_ERR_INVALID;
    }
    ScaleTables tab;
    build_scale_tables(100, scale_factor, nlevels, tab);
    PyramidGeom g;
    std::string err;
    if (!build_geometry(width, height, tab, g, err)) {
        set_error("%s", err.c_str());
        return ORBX_ERR_INVALID;
    }
    std::vector<HostResizeTab> ht(g.nlevels);
    ResizeTab tabs[ORBX_MAX_LEVELS] = {};
    for (int l = 1; l < g.nlevels; l++) {
        build_resize_table(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, ht[l]);
        tabs[l].maxSpan = resize_max_span(ht[l]);
    }
    FusedPlan plan;
    plan_fused_pyramid(g, ht, tabs, tile_w, tile_h, group_levels, plan);
    *ngroups = (int)plan.groups.size();
    if (!check_fused_plan(g, ht, plan, err)) {
        set_error("pyramid plan for %dx%d, scale %g, %d levels: %s", width, height, (double)scale_factor, nlevels,
                  err.c_str());
        return ORBX_ERR_STATE;
    }
    return ORBX_OK;
}

extern "C" int orbx_debug_candidates(orbx_ctx* c, int frame, int level, orbx_keypoint* out, int cap, int* n) {
    int rc = check_probe(c, frame, level);
    if (rc) return rc;
    if (!n || (!out && cap > 0)) return ORBX_ERR_INVALID;
    const PyramidGeom& g = c->geom;
    const LevelGeom& L = g.lv[level];
    const int nCells = L.nCols * L.nRows;
    *n = 0;
    if (nCells == 0) return ORBX_OK;
    std::vector<int32_t> counts(nCells);
    std::vector<uint32_t> slots((size_t)nCells * L.slotCap);
    HIP_TRY(hipMemcpy(counts.data(), c->fs.cellCount + (size_t)frame * g.totalCells + L.cellBase,
                      sizeof(int32_t) * nCells, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(slots.data(), c->fs.slots + (size_t)frame * g.totalSlots + L.slotBase,
                      sizeof(uint32_t) * slots.size(), hipMemcpyDeviceToHost));
    int k = 0;
    for (int cell = 0; cell < nCells; cell++)
        for (int i = 0; i < counts[cell]; i++, k++) {
            if (k >= cap) continue;
            const uint32_t p = slots[(size_t)cell * L.slotCap + i];
            out[k] = orbx_keypoint{(float)cand_x(p), (float)cand_y(p), 7.f, -1.f, (float)cand_score(p), 0, -1};
        }
    *n = k;
    return ORBX_OK;
}

extern "C" int orbx_debug_selected(orbx_ctx* c, int frame, int level, orbx_keypoint* out, int cap, int* n) {
    int rc = check_probe(c, frame, level);
    if (rc) return rc;
    if (!n || (!out && cap > 0)) return ORBX_ERR_INVALID;
    const PyramidGeom& g = c->geom;
    const LevelGeom& L = g.lv[level];
    int32_t count = 0;
    HIP_TRY(hipMemcpy(&count, c->fs.selCount + (size_t)frame * g.nlevels + level, sizeof(int32_t),
                      hipMemcpyDeviceToHost));
    std::vector<uint32_t> sel(std::max(count, 1));
    HIP_TRY(hipMemcpy(sel.data(), c->fs.sel + (size_t)frame * g.totalSel + L.selBase, sizeof(uint32_t) * count,
                      hipMemcpyDeviceToHost));
    for (int i = 0; i < count && i < cap; i++)
        out[i] = orbx_keypoint{(float)(cand_x(sel[i]) + kBorder), (float)(cand_y(sel[i]) + kBorder), L.kpSize,
                               -1.f, (float)cand_score(sel[i]), level, -1};
    *n = count;
    return ORBX_OK;
}

// ---- matcher ------------------------------------------------------------------------------

extern "C" int orbx_hamming256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8);
        std::memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

struct orbx_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    void* scratch = nullptr;
    size_t scratchBytes = 0;
    uint8_t* pin = nullptr;   // pinned host staging of the host-array entry points: one copy in, one copy out
    size_t pinBytes = 0;
    float* scale = nullptr;   // device copy of scale factors [0..15] and inverse scale factors [16..31]
    float hostScale[2 * ORBX_MAX_LEVELS] = {0};   // what `scale` holds
    bool scaleValid = false;
    OrderedEvent evIn[2];                         // "the extractor contexts' work is done"
    OrderedEvent evOut;                           // "the matcher has read the contexts' pyramids"
    std::vector<BowRow> bowRows;                  // kept between calls of the vocabulary-node search for their storage
    BowReplayScratch bowTmp;
    std::vector<WindowQuery> winQueries;          // the same for the initialization search
    std::vector<int32_t> winQueryIndex;
    InitReplayScratch initTmp;
};

namespace {
int matcher_scratch(orbx_matcher* m, size_t bytes, hipStream_t s) {
    if (bytes <= m->scratchBytes) return ORBX_OK;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->scratch) HIP_TRY(hipFree(m->scratch));
    m->scratch = nullptr;
    m->scratchBytes = 0;
    HIP_TRY(hipMalloc(&m->scratch, bytes));
    m->scratchBytes = bytes;
    return ORBX_OK;
}

// Grows the pinned staging buffer (contents are lost).
int matcher_pinned(orbx_matcher* m, size_t bytes) {
    if (bytes <= m->pinBytes) return ORBX_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->pin) HIP_TRY(hipHostFree(m->pin));
    m->pin = nullptr;
    m->pinBytes = 0;
    bytes += bytes / 2;
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    m->pin = static_cast<uint8_t*>(p);
    m->pinBytes = bytes;
    return ORBX_OK;
}
}  // namespace

extern "C" int orbx_matcher_create(int device, orbx_matcher** out) {
    if (!out) return ORBX_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    orbx_matcher* m = new (std::nothrow) orbx_matcher();
    if (!m) return ORBX_ERR_INVALID;
    m->device = device;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->scale), sizeof(float) * 2 * ORBX_MAX_LEVELS);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&m->evIn[i].ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->evOut.ev, hipEventDisableTiming);
    if (e == hipSuccess) e = knn2_static_init(device);   // address of this device's fixed partial-result array
    if (e != hipSuccess) {
        set_error("orbx_matcher_create: %s", hipGetErrorString(e));
        orbx_matcher_destroy(m);
        return ORBX_ERR_HIP;
    }
    *out = m;
    return ORBX_OK;
}

extern "C" void orbx_matcher_destroy(orbx_matcher* m) {
    if (!m) return;
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->scratch) (void)hipFree(m->scratch);
    if (m->pin) (void)hipHostFree(m->pin);
    if (m->scale) (void)hipFree(m->scale);
    for (hipEvent_t ev : {m->evIn[0].ev, m->evIn[1].ev, m->evOut.ev})
        if (ev) (void)hipEventDestroy(ev);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int orbx_matcher_get_stream(orbx_matcher* m, void** stream) {
    if (!m || !stream) {
        set_error("orbx_matcher_get_stream: null argument");
        return ORBX_ERR_INVALID;
    }
    *stream = m->stream;
    return ORBX_OK;
}

extern "C" int orbx_matcher_synchronize(orbx_matcher* m) {
    if (!m) return ORBX_ERR_INVALID;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

extern "C" int orbx_hamming_knn2_device(orbx_matcher* m, const uint8_t* d_query, int64_t nq, const uint8_t* d_db,
                                        int64_t ndb, int32_t init, int64_t db_index_base, int64_t* d_idx,
                                        int32_t* d_d1, int32_t* d_d2, void* stream) {
    if (!m || nq < 0 || ndb < 0 || ndb > INT32_MAX || (nq > 0 && (!d_query || !d_idx || !d_d1 || !d_d2)) ||
        (ndb > 0 && !d_db) || ((uintptr_t)d_query % 16) || ((uintptr_t)d_db % 16)) {
        set_error("orbx_hamming_knn2_device: bad argument (descriptor arrays must be 16-byte aligned)");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    int rc = matcher_scratch(m, knn2_scratch_bytes(nq, ndb), s);
    if (rc) return rc;
    HIP_TRY(launch_knn2(d_query, nq, d_db, ndb, init, db_index_base, d_idx, d_d1, d_d2, nullptr, m->scratch,
                        m->scratchBytes, s));
    return ORBX_OK;
}

extern "C" int orbx_hamming_knn2_packed_device(orbx_matcher* m, const uint8_t* d_query, int64_t nq,
                                               const uint8_t* d_db, int64_t ndb, int32_t init, int64_t db_index_base,
                                               orbx_knn2_hit* d_hits, void* stream) {
    if (!m || nq < 0 || ndb < 0 || ndb > INT32_MAX || (nq > 0 && (!d_query || !d_hits)) || (ndb > 0 && !d_db) ||
        ((uintptr_t)d_query % 16) || ((uintptr_t)d_db % 16) || ((uintptr_t)d_hits % 16)) {
        set_error("orbx_hamming_knn2_packed_device: bad argument (descriptor and result arrays must be 16-byte aligned)");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    int rc = matcher_scratch(m, knn2_scratch_bytes(nq, ndb), s);
    if (rc) return rc;
    HIP_TRY(launch_knn2(d_query, nq, d_db, ndb, init, db_index_base, nullptr, nullptr, nullptr, d_hits, m->scratch,
                        m->scratchBytes, s));
    return ORBX_OK;
}

extern "C" int orbx_knn2_merge_packed_device(orbx_matcher* m, int nparts, int64_t nq, const orbx_knn2_hit* d_hits,
                                             int32_t init, int64_t* d_out_idx, int32_t* d_out_d1, int32_t* d_out_d2,
                                             void* stream) {
    if (!m || nparts < 1 || nq < 0 || (nq > 0 && (!d_hits || !d_out_idx || !d_out_d1 || !d_out_d2)) ||
        ((uintptr_t)d_hits % 16)) {
        set_error("orbx_knn2_merge_packed_device: bad argument");
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    HIP_TRY(launch_knn2_merge_packed(nparts, nq, d_hits, init, d_out_idx, d_out_d1, d_out_d2, s));
    return ORBX_OK;
}

extern "C" int orbx_hamming_knn2_noalloc_device(orbx_matcher* m, const uint8_t* d_query, int64_t nq,
                                                const uint8_t* d_db, int64_t ndb, int32_t init,
                                                int64_t db_index_base, int64_t* d_idx, int32_t* d_d1, int32_t* d_d2,
                                                void* stream) {
    if (!m || nq < 0 || ndb < 0 || ndb > INT32_MAX || (nq > 0 && (!d_query || !d_idx || !d_d1 || !d_d2)) ||
        (ndb > 0 && !d_db) || ((uintptr_t)d_query % 16) || ((uintptr_t)d_db % 16)) {
        set_error("orbx_hamming_knn2_noalloc_device: bad argument (descriptor arrays must be 16-byte aligned)");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    if (!knn2_static_fits(nq, ndb)) {
        set_error("orbx_hamming_knn2_noalloc_device: %lld queries x %lld rows do not fit the fixed partial-result "
                  "array (nq * ceil(ndb / 4194304) must not exceed %lld); use orbx_hamming_knn2_device",
                  (long long)nq, (long long)ndb, (long long)knn2_static_max_partials());
        return ORBX_ERR_CAPACITY;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    HIP_TRY(launch_knn2_static(m->device, d_query, nq, d_db, ndb, init, db_index_base, d_idx, d_d1, d_d2, s));
    return ORBX_OK;
}

extern "C" int orbx_hamming_knn2(orbx_matcher* m, const uint8_t* query, int64_t nq, const uint8_t* db, int64_t ndb,
                                 int32_t init, int64_t* idx, int32_t* d1, int32_t* d2) {
    if (!m || nq < 0 || ndb < 0 || (nq > 0 && (!query || !idx || !d1 || !d2)) || (ndb > 0 && !db)) {
        set_error("orbx_hamming_knn2: bad argument");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    if (ndb > INT32_MAX) {
        set_error("orbx_hamming_knn2: at most 2^31 - 1 database rows per call");
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // Inputs, outputs and the scan's partial results share the matcher's one device buffer, which only ever grows;
    // inputs and outputs of ordinary size go through its pinned staging buffer (one asynchronous copy each way,
    // pageable copies of the same bytes take several times as long).
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t qBytes = (size_t)nq * 32, dbBytes = (size_t)ndb * 32;
    const size_t oQ = 0, oDb = oQ + up(qBytes), oIdx = oDb + up(std::max<size_t>(dbBytes, 32)), oD1 = oIdx + up((size_t)nq * 8),
                 oD2 = oD1 + up((size_t)nq * 4), oPart = oD2 + up((size_t)nq * 4);
    const size_t partBytes = knn2_scratch_bytes(nq, ndb), total = oPart + partBytes;
    int rc = matcher_scratch(m, total, s);
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(m->scratch);
    const size_t inBytes = oIdx, outBytes = oPart - oIdx;
    const bool staged = inBytes <= ((size_t)64 << 20);
    if (staged) {
        if ((rc = matcher_pinned(m, std::max(inBytes, outBytes)))) return rc;
        std::memcpy(m->pin + oQ, query, qBytes);
        if (ndb) std::memcpy(m->pin + oDb, db, dbBytes);
        HIP_TRY(hipMemcpyAsync(base, m->pin, oDb + dbBytes, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(hipMemcpyAsync(base + oQ, query, qBytes, hipMemcpyHostToDevice, s));
        if (ndb) HIP_TRY(hipMemcpyAsync(base + oDb, db, dbBytes, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(launch_knn2(base + oQ, nq, base + oDb, ndb, init, 0, reinterpret_cast<int64_t*>(base + oIdx),
                        reinterpret_cast<int32_t*>(base + oD1), reinterpret_cast<int32_t*>(base + oD2), nullptr,
                        base + oPart, partBytes, s));
    if (staged) {
        HIP_TRY(hipMemcpyAsync(m->pin, base + oIdx, outBytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::memcpy(idx, m->pin, (size_t)nq * 8);
        std::memcpy(d1, m->pin + (oD1 - oIdx), (size_t)nq * 4);
        std::memcpy(d2, m->pin + (oD2 - oIdx), (size_t)nq * 4);
    } else {
        HIP_TRY(hipMemcpyAsync(idx, base + oIdx, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(d1, base + oD1, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(d2, base + oD2, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return ORBX_OK;
}

extern "C" int orbx_knn2_merge_device(orbx_matcher* m, int nparts, int64_t nq, const int64_t* d_idx,
                                      const int32_t* d_d1, const int32_t* d_d2, int32_t init, int64_t* d_out_idx,
                                      int32_t* d_out_d1, int32_t* d_out_d2, void* stream) {
    if (!m || nparts < 1 || nq < 0 || (nq > 0 && (!d_idx || !d_d1 || !d_d2 || !d_out_idx || !d_out_d1 || !d_out_d2))) {
        set_error("orbx_knn2_merge_device: bad argument");
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    HIP_TRY(launch_knn2_merge(nparts, nq, d_idx, d_d1, d_d2, init, d_out_idx, d_out_d1, d_out_d2, s));
    return ORBX_OK;
}

extern "C" int orbx_ratio_test_device(orbx_matcher* m, int64_t nq, const int32_t* d_d1, const int32_t* d_d2,
                                      int32_t th, int inclusive, float ratio, uint8_t* d_accept, void* stream) {
    if (!m || nq < 0 || (nq > 0 && (!d_d1 || !d_d2 || !d_accept))) {
        set_error("orbx_ratio_test_device: bad argument");
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    HIP_TRY(launch_ratio_test(nq, d_d1, d_d2, th, inclusive, ratio, d_accept, s));
    return ORBX_OK;
}

extern "C" int orbx_stereo_band_match_device(orbx_matcher* m, int npairs, int cap, const orbx_keypoint* d_kl,
                                             const uint8_t* d_dl, const int32_t* d_nl, const orbx_keypoint* d_kr,
                                             const uint8_t* d_dr, const int32_t* d_nr, const float* scale_factors,
                                             int nlevels, float bf, float b, int32_t* d_best_idx,
                                             int32_t* d_best_dist, void* stream) {
    if (!m || npairs < 0 || cap < 1 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !scale_factors || !(b > 0) ||
        (npairs > 0 && (!d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !d_best_idx || !d_best_dist)) ||
        ((uintptr_t)d_dl % 16) || ((uintptr_t)d_dr % 16)) {
        set_error("orbx_stereo_band_match_device: bad argument");
        return ORBX_ERR_INVALID;
    }
    if (npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    m->scaleValid = false;   // orbx_compute_stereo_matches_device re-uploads its own table
    HIP_TRY(hipMemcpyAsync(m->scale, scale_factors, sizeof(float) * nlevels, hipMemcpyHostToDevice, s));
    const float maxD = bf / b;   // mbf / minZ with minZ = mb (src/Frame.cc:533-535)
    int rc = matcher_scratch(m, stereo_band_scratch_bytes(npairs, cap), s);
    if (rc) return rc;
    float maxScale = 0.f;
    for (int l = 0; l < nlevels; l++) maxScale = std::max(maxScale, scale_factors[l]);
    HIP_TRY(launch_stereo_band(npairs, cap, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, m->scale, nlevels, maxScale, maxD,
                               m->scratch, d_best_idx, d_best_dist, s));
    return ORBX_OK;
}

namespace {
void pyramid_view(const orbx_ctx* c, PyramidView& v) {
    const PyramidGeom& g = c->geom;
    v.nlevels = g.nlevels;
    for (int l = 0; l < g.nlevels; l++) {
        LevelView& L = v.lv[l];
        L.w = g.lv[l].w;
        L.h = g.lv[l].h;
        if (l == 0) {
            L.base = c->fs.img;
            L.rowStride = c->fs.imgRowStride;
            L.frameStride = c->fs.imgFrameStride;
        } else {
            L.base = c->fs.pyr + g.lv[l].offset;
            L.rowStride = g.lv[l].stride;
            L.frameStride = g.pyrBytes;
        }
    }
}
}  // namespace

extern "C" int orbx_compute_stereo_matches_device(orbx_matcher* m, orbx_ctx* left, orbx_ctx* right, int npairs,
                                                  int cap, const orbx_keypoint* d_kl, const uint8_t* d_dl,
                                                  const int32_t* d_nl, const orbx_keypoint* d_kr,
                                                  const uint8_t* d_dr, const int32_t* d_nr, float bf, float b,
                                                  float* d_uright, float* d_depth, void* stream) {
    if (!m || !left || !right || npairs < 0 || cap < 1 || !(b > 0) ||
        (npairs > 0 && (!d_kl || !d_dl || !d_nl || !d_kr || !d_dr || !d_nr || !d_uright || !d_depth)) ||
        ((uintptr_t)d_dl % 16) || ((uintptr_t)d_dr % 16)) {
        set_error("orbx_compute_stereo_matches_device: bad argument");
        return ORBX_ERR_INVALID;
    }
    if (npairs == 0) return ORBX_OK;
    if (!left->haveGeom || !right->haveGeom || npairs > left->lastFrames || npairs > right->lastFrames ||
        left->cfg.nlevels != right->cfg.nlevels || left->cfg.device != m->device || right->cfg.device != m->device ||
        left->geom.width != right->geom.width || left->geom.height != right->geom.height) {
        set_error("orbx_compute_stereo_matches_device: the two contexts do not hold %d matching frames", npairs);
        return ORBX_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    // Order the streams on the device, without stopping the host: this call's kernels start after the two
    // extractions, and whatever the contexts are given next starts after these kernels have read their pyramids.
    orbx_ctx* side[2] = {left, right};
    for (int i = 0; i < 2; i++)
        if (side[i]->lastStream && side[i]->lastStream != s) {
            HIP_TRY(record_event(m->evIn[i], side[i]->lastStream));
            if (int wrc = wait_event(s, m->evIn[i], "extraction done")) return wrc;
        }
    const size_t n = (size_t)npairs * cap;
    const size_t bandBytes = (stereo_band_scratch_bytes(npairs, cap) + 255) / 256 * 256;
    int rc = matcher_scratch(m, bandBytes + n * 3 * sizeof(int32_t), s);
    if (rc) return rc;
    int32_t* bestIdx = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(m->scratch) + bandBytes);
    int32_t* bestDist = bestIdx + n;
    int32_t* sad = bestDist + n;
    const int nl = left->cfg.nlevels;
    float host[2 * ORBX_MAX_LEVELS] = {0};
    for (int l = 0; l < nl; l++) {
        host[l] = left->tab.scale[l];
        host[ORBX_MAX_LEVELS + l] = left->tab.invScale[l];
    }
    if (!m->scaleValid || memcmp(host, m->hostScale, sizeof(host)) != 0) {   // first call, or other extractor settings
        HIP_TRY(hipStreamSynchronize(s));                                    // earlier kernels may still read m->scale
        memcpy(m->hostScale, host, sizeof(host));
        HIP_TRY(hipMemcpyAsync(m->scale, m->hostScale, sizeof(host), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        m->scaleValid = true;
    }
    const float maxD = bf / b;          // mbf / minZ with minZ = mb (src/Frame.cc:533-535)
    float maxScale = 0.f;
    for (int l = 0; l < nl; l++) maxScale = std::max(maxScale, host[l]);
    HIP_TRY(launch_stereo_band(npairs, cap, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, m->scale, nl, maxScale, maxD,
                               m->scratch, bestIdx, bestDist, s));
    PyramidView lv, rv;
    pyramid_view(left, lv);
    pyramid_view(right, rv);
    HIP_TRY(launch_stereo_refine(npairs, cap, lv, rv, d_kl, d_nl, d_kr, bestIdx, bestDist, m->scale,
                                 m->scale + ORBX_MAX_LEVELS, bf, maxD, d_uright, d_depth, sad, m->scratch, s));
    HIP_TRY(record_event(m->evOut, s));
    for (int i = 0; i < 2; i++) {
        if (int wrc = wait_event(side[i]->stream, m->evOut, "pyramids read")) return wrc;
        if (side[i]->lastStream && side[i]->lastStream != side[i]->stream && side[i]->lastStream != s)
            if (int wrc = wait_event(side[i]->lastStream, m->evOut, "pyramids read")) return wrc;
        // A later extraction on any other stream waits for the context's evDone only: it now also stands for
        // "the stereo kernels have read this context's pyramid" (s has waited for the extraction itself above).
        HIP_TRY(record_event(side[i]->evDone, s));
    }
    return ORBX_OK;
}

extern "C" int orbx_image_bounds(const orbx_camera* cam, int width, int height, float* min_x, float* min_y,
                                 float* max_x, float* max_y) {
    if (!cam || !min_x || !min_y || !max_x || !max_y || width < 1 || height < 1) return ORBX_ERR_INVALID;
    if (cam->k1 != 0.0f) {
        float x[4], y[4];
        const float cu[4] = {0.f, (float)width, 0.f, (float)width}, cv[4] = {0.f, 0.f, (float)height, (float)height};
        for (int i = 0; i < 4; i++) undistort_point_host(*cam, cu[i], cv[i], &x[i], &y[i]);
        *min_x = std::min(x[0], x[2]);
        *max_x = std::max(x[1], x[3]);
        *min_y = std::min(y[0], y[1]);
        *max_y = std::max(y[2], y[3]);
    } else {
        *min_x = 0.0f;
        *max_x = (float)width;
        *min_y = 0.0f;
        *max_y = (float)height;
    }
    return ORBX_OK;
}

extern "C" int orbx_undistort_assign_grid_device(orbx_matcher* m, int nframes, int cap, const orbx_keypoint* d_kps,
                                                 const int32_t* d_counts, const orbx_camera* cam, float min_x,
                                                 float min_y, float max_x, float max_y, orbx_keypoint* d_kps_un,
                                                 int32_t* d_cell_start, int32_t* d_cell_items, void* stream) {
    if (!m || nframes < 0 || cap < 1 || cap >= 0xFFFF || !cam || !(max_x > min_x) || !(max_y > min_y) ||
        (nframes > 0 && (!d_kps || !d_counts || !d_kps_un || !d_cell_start || !d_cell_items))) {
        set_error("orbx_undistort_assign_grid_device: bad argument");
        return ORBX_ERR_INVALID;
    }
    if (nframes == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
    // fFrameGridElementWidthInv / HeightInv (src/Frame.cc:133-134)
    const float invW = static_cast<float>(ORBX_GRID_COLS) / (max_x - min_x);
    const float invH = static_cast<float>(ORBX_GRID_ROWS) / (max_y - min_y);
    HIP_TRY(launch_undistort_grid(nframes, cap, d_kps, d_counts, *cam, min_x, min_y, invW, invH, d_kps_un,
                                  d_cell_start, d_cell_items, s));
    return ORBX_OK;
}

namespace {
// Argument checks and the walk over the two node lists, shared by both forms of orbx_search_by_bow.
// Returns ORBX_OK with rows / npairs filled (possibly empty).
int bow_prepare(const char* who, orbx_matcher* m, int mode, const void* keys1_un, const void* desc1,
                const uint8_t* valid1, int n1, const uint32_t* fv1_nodes, const int32_t* fv1_start,
                const uint32_t* fv1_items, int fv1_n, const void* keys2, const void* desc2, const uint8_t* valid2,
                int n2, const uint32_t* fv2_nodes, const int32_t* fv2_start, const uint32_t* fv2_items, int fv2_n,
                int32_t* matches, int* nmatches, std::vector<BowRow>& rows, long long& npairs) {
    const bool kfkf = mode == ORBX_BOW_KEYFRAME_KEYFRAME;
    if (!m || (mode != ORBX_BOW_KEYFRAME_FRAME && !kfkf) || n1 < 0 || n2 < 0 || fv1_n < 0 || fv2_n < 0 || !nmatches ||
        (n1 > 0 && (!keys1_un || !desc1 || !valid1)) || (n2 > 0 && (!keys2 || !desc2 || (kfkf && !valid2))) ||
        (fv1_n > 0 && (!fv1_nodes || !fv1_start || !fv1_items)) ||
        (fv2_n > 0 && (!fv2_nodes || !fv2_start || !fv2_items)) || ((kfkf ? n1 : n2) > 0 && !matches)) {
        set_error("%s: bad argument", who);
        return ORBX_ERR_INVALID;
    }
    for (int j = 0; j + 1 < fv1_n; j++)
        if (!(fv1_nodes[j] < fv1_nodes[j + 1])) { set_error("%s: fv1_nodes not ascending", who); return ORBX_ERR_INVALID; }
    for (int j = 0; j + 1 < fv2_n; j++)
        if (!(fv2_nodes[j] < fv2_nodes[j + 1])) { set_error("%s: fv2_nodes not ascending", who); return ORBX_ERR_INVALID; }
    for (int side = 0; side < 2; side++) {
        const int32_t* st = side ? fv2_start : fv1_start;
        const int n = side ? fv2_n : fv1_n;
        bool ok = n == 0 || st[0] == 0;
        for (int j = 0; ok && j < n; j++) ok = st[j] <= st[j + 1];
        if (!ok) {
            set_error("%s: fv%d_start must rise from 0", who, side + 1);
            return ORBX_ERR_INVALID;
        }
    }
    *nmatches = 0;
    const int nout = kfkf ? n1 : n2;
    for (int i = 0; i < nout; i++) matches[i] = -1;
    if (!build_bow_rows(fv1_nodes, fv1_start, fv1_items, fv1_n, valid1, n1, fv2_nodes, fv2_start, fv2_items, fv2_n, n2,
                        rows, npairs)) {
        set_error("%s: feature vector index out of range", who);
        return ORBX_ERR_INVALID;
    }
    return ORBX_OK;
}
}  // namespace

extern "C" int orbx_search_by_bow(orbx_matcher* m, int mode, const orbx_keypoint* keys1_un, const uint8_t* desc1,
                                  const uint8_t* valid1, int n1, const uint32_t* fv1_nodes,
                                  const int32_t* fv1_start, const uint32_t* fv1_items, int fv1_n,
                                  const orbx_keypoint* keys2, const uint8_t* desc2, const uint8_t* valid2, int n2,
                                  const uint32_t* fv2_nodes, const int32_t* fv2_start, const uint32_t* fv2_items,
                                  int fv2_n, float nnratio, int check_orientation, int32_t* matches,
                                  int* nmatches) {
    if (!m) { set_error("orbx_search_by_bow: bad argument"); return ORBX_ERR_INVALID; }
    const bool kfkf = mode == ORBX_BOW_KEYFRAME_KEYFRAME;
    std::vector<BowRow>& rows = m->bowRows;
    long long npairs = 0;
    int rc = bow_prepare("orbx_search_by_bow", m, mode, keys1_un, desc1, valid1, n1, fv1_nodes, fv1_start, fv1_items,
                         fv1_n, keys2, desc2, valid2, n2, fv2_nodes, fv2_start, fv2_items, fv2_n, matches, nmatches,
                         rows, npairs);
    if (rc) return rc;
    if (rows.empty() || npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int nItems2 = fv2_start[fv2_n];
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    // inputs packed in pinned memory in the layout of the device scratch: one copy in, one copy out
    const size_t oD1 = 0, oD2 = oD1 + up(32 * (size_t)n1), oIt = oD2 + up(32 * (size_t)n2),
                 oRw = oIt + up(4 * (size_t)nItems2),
                 oV2 = oRw + up(sizeof(BowRow) * rows.size()), oDs = oV2 + up(kfkf ? (size_t)n2 : 0),
                 oTp = oDs + up(2 * (size_t)npairs), total = oTp + up(4 * kBowTop * rows.size());
    if ((rc = matcher_scratch(m, total, s))) return rc;
    if ((rc = matcher_pinned(m, total))) return rc;
    uint8_t* base = static_cast<uint8_t*>(m->scratch);
    std::memcpy(m->pin + oD1, desc1, 32 * (size_t)n1);
    std::memcpy(m->pin + oD2, desc2, 32 * (size_t)n2);
    std::memcpy(m->pin + oIt, fv2_items, 4 * (size_t)nItems2);
    std::memcpy(m->pin + oRw, rows.data(), sizeof(BowRow) * rows.size());
    if (kfkf) std::memcpy(m->pin + oV2, valid2, (size_t)n2);
    HIP_TRY(hipMemcpyAsync(base, m->pin, oDs, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_bow_distances((int)rows.size(), reinterpret_cast<const BowRow*>(base + oRw), base + oD1,
                                 reinterpret_cast<const uint32_t*>(base + oIt), base + oD2, kfkf ? base + oV2 : nullptr,
                                 reinterpret_cast<uint16_t*>(base + oDs), reinterpret_cast<uint32_t*>(base + oTp), s));
    HIP_TRY(hipMemcpyAsync(m->pin + oDs, base + oDs, total - oDs, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint16_t* dist = reinterpret_cast<const uint16_t*>(m->pin + oDs);
    *nmatches = replay_search_by_bow(mode, rows, dist, reinterpret_cast<const uint32_t*>(m->pin + oTp), fv2_items,
                                     keys1_un, keys2, valid2, n1, n2, nnratio, check_orientation != 0, matches,
                                     m->bowTmp);
    return ORBX_OK;
}

extern "C" int orbx_search_by_bow_device(orbx_matcher* m, int mode, const orbx_keypoint* keys1_un,
                                         const uint8_t* d_desc1, const uint8_t* valid1, int n1,
                                         const uint32_t* fv1_nodes, const int32_t* fv1_start,
                                         const uint32_t* fv1_items, int fv1_n, const orbx_keypoint* keys2,
                                         const uint8_t* d_desc2, const uint8_t* valid2, int n2,
                                         const uint32_t* fv2_nodes, const int32_t* fv2_start,
                                         const uint32_t* fv2_items, int fv2_n, float nnratio, int check_orientation,
                                         int32_t* matches, int* nmatches) {
    if (!m) { set_error("orbx_search_by_bow_device: bad argument"); return ORBX_ERR_INVALID; }
    const bool kfkf = mode == ORBX_BOW_KEYFRAME_KEYFRAME;
    std::vector<BowRow>& rows = m->bowRows;
    long long npairs = 0;
    int rc = bow_prepare("orbx_search_by_bow_device", m, mode, keys1_un, d_desc1, valid1, n1, fv1_nodes, fv1_start,
                         fv1_items, fv1_n, keys2, d_desc2, valid2, n2, fv2_nodes, fv2_start, fv2_items, fv2_n, matches,
                         nmatches, rows, npairs);
    if (rc) return rc;
    if (((uintptr_t)d_desc1 % 16) || ((uintptr_t)d_desc2 % 16)) {
        set_error("orbx_search_by_bow_device: descriptor arrays must be 16-byte aligned");
        return ORBX_ERR_INVALID;
    }
    if (rows.empty() || npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int nItems2 = fv2_start[fv2_n];
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    // The descriptors are already in HBM. The kernel reads its rows and the right-hand feature lists from pinned
    // host memory and writes the distance rows back into it: one launch and one wait, no copy either way.
    const size_t oIt = 0, oRw = oIt + up(4 * (size_t)nItems2), oV2 = oRw + up(sizeof(BowRow) * rows.size()),
                 oDs = oV2 + up(kfkf ? (size_t)n2 : 0), oTp = oDs + up(2 * (size_t)npairs), total = oTp + up(4 * kBowTop * rows.size());
    if ((rc = matcher_pinned(m, total))) return rc;
    std::memcpy(m->pin + oIt, fv2_items, 4 * (size_t)nItems2);
    std::memcpy(m->pin + oRw, rows.data(), sizeof(BowRow) * rows.size());
    if (kfkf) std::memcpy(m->pin + oV2, valid2, (size_t)n2);
    HIP_TRY(launch_bow_distances((int)rows.size(), reinterpret_cast<const BowRow*>(m->pin + oRw), d_desc1,
                                 reinterpret_cast<const uint32_t*>(m->pin + oIt), d_desc2,
                                 kfkf ? m->pin + oV2 : nullptr,
                                 reinterpret_cast<uint16_t*>(m->pin + oDs),
                                 reinterpret_cast<uint32_t*>(m->pin + oTp), s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint16_t* dist = reinterpret_cast<const uint16_t*>(m->pin + oDs);
    *nmatches = replay_search_by_bow(mode, rows, dist, reinterpret_cast<const uint32_t*>(m->pin + oTp), fv2_items,
                                     keys1_un, keys2, valid2, n1, n2, nnratio, check_orientation != 0, matches,
                                     m->bowTmp);
    return ORBX_OK;
}

namespace {
inline size_t up256z(size_t v) { return (v + 255) / 256 * 256; }

// Device-resident inputs of a windowed search.
struct WindowDev {
    const uint8_t* d1;              // descriptors of the queries
    const orbx_keypoint* k2;        // searched frame: undistorted keypoints, descriptors, right coordinates, grid
    const uint8_t* d2;
    const float* uRight2;
    const int32_t* cellStart;
    const int32_t* cellItems;
};

// What window_search_core needs beyond its caller's own use of the two staging buffers.
struct WindowCoreLayout {
    int n1;
    int budget;                     // candidate entries that come back without a second round trip
    long long capTotal;             // bound of the whole list: every active query meets every listed keypoint
    size_t pQ, pHd, pCd, pinBytes;  // offsets inside the pinned buffer (from the caller's first free byte)
    size_t sCur, sCd, scratchBytes; // offsets inside the device scratch buffer
    WindowCoreLayout(int n1_, long long capTotal_, size_t pinFree, size_t scratchFree) : n1(n1_), capTotal(capTotal_) {
        budget = (int)std::min<long long>(capTotal, std::max<long long>(1 << 16, 64LL * n1));
        pQ = up256z(pinFree);
        pHd = pQ + up256z(sizeof(WindowQuery) * (size_t)n1);
        pCd = pHd + up256z(4 * 2 * (size_t)n1);
        pinBytes = pCd + 4 * (size_t)budget;
        sCur = up256z(scratchFree);
        sCd = sCur + 256;
        scratchBytes = sCd + 4 * (size_t)capTotal;
    }
};

// The candidate kernel and the way back of its lists. The kernel reads the queries from pinned host memory and
// writes counts, offsets and (up to lay.budget) candidates straight into it, so the usual call is one launch and
// one wait, with no copy either way. Only when the lists outgrow the budget does the kernel run again into the
// device scratch region that holds any list, followed by a copy of what exists. The caller has sized both
// buffers with lay (matcher_pinned / matcher_scratch) and may have queued uploads on the stream.
// On return head = {counts[n1], offsets[n1]} and *candHost points at the packed lists.
int window_search_core(orbx_matcher* m, const char* who, const std::vector<WindowQuery>& queries, const WindowDev& in,
                       int n2, float min_x, float min_y, float max_x, float max_y, const WindowCoreLayout& lay,
                       std::vector<int32_t>& head, const uint32_t** candHost) {
    const int n1 = lay.n1;
    hipStream_t s = m->stream;
    uint8_t* base = static_cast<uint8_t*>(m->scratch);
    std::memcpy(m->pin + lay.pQ, queries.data(), sizeof(WindowQuery) * (size_t)n1);
    const float invW = static_cast<float>(ORBX_GRID_COLS) / (max_x - min_x);
    const float invH = static_cast<float>(ORBX_GRID_ROWS) / (max_y - min_y);
    int32_t* hd = reinterpret_cast<int32_t*>(m->pin + lay.pHd);
    auto launch = [&](uint32_t* cand, int cap) {
        return launch_window_candidates(n1, reinterpret_cast<const WindowQuery*>(m->pin + lay.pQ), in.d1, in.k2, in.d2,
                                        in.uRight2, in.cellStart, in.cellItems, min_x, min_y, invW, invH, n2, cap,
                                        reinterpret_cast<int32_t*>(base + lay.sCur), hd, hd + n1, cand, s);
    };
    HIP_TRY(launch(reinterpret_cast<uint32_t*>(m->pin + lay.pCd), lay.budget));
    HIP_TRY(hipStreamSynchronize(s));
    bool overflow = false;
    for (int i = 0; i < n1; i++) overflow |= hd[i] < 0;
    size_t ncand = 0;
    if (overflow) {
        HIP_TRY(launch(reinterpret_cast<uint32_t*>(base + lay.sCd), (int)lay.capTotal));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // the header leaves the pinned buffer before the buffer may grow for a long candidate list
    head.assign(hd, hd + 2 * (size_t)n1);
    for (int i = 0; i < n1; i++) {
        if (head[i] < 0 || head[n1 + i] < 0 || (long long)head[n1 + i] + head[i] > lay.capTotal) {
            set_error("%s: inconsistent candidate lists", who);
            return ORBX_ERR_STATE;
        }
        ncand = std::max(ncand, (size_t)head[n1 + i] + (size_t)head[i]);
    }
    if (!overflow) {
        *candHost = reinterpret_cast<const uint32_t*>(m->pin + lay.pCd);
        return ORBX_OK;
    }
    int rc = matcher_pinned(m, std::max<size_t>(4 * ncand, 4));
    if (rc) return rc;
    if (ncand > 0) {
        HIP_TRY(hipMemcpyAsync(m->pin, base + lay.sCd, 4 * ncand, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    *candHost = reinterpret_cast<const uint32_t*>(m->pin);
    return ORBX_OK;
}

// What the initialization search hands through run_window_search to its own host pass.
struct InitTail {
    const orbx_keypoint* keys1_un;
    int n1;
    const std::vector<int32_t>* queryIndex;       // frame-1 keypoint of each (active) query
    float* prev_matched;
    float nnratio;
    bool checkOri;
    int32_t* matches12;
    int* nmatches;
};

// Staging of the initialization search beyond its caller's own use of the two buffers: the active queries and
// their keypoint numbers go in through pinned memory, one record per query comes back into it, and the lists
// stay in the device scratch buffer, n2 entries apart.
struct InitLayout {
    int nq;
    size_t pQ, pIx, pRec, pinBytes, sCd, scratchBytes;
    InitLayout(int nq_, int n2, size_t pinFree, size_t scratchFree) : nq(nq_) {
        pQ = up256z(pinFree);
        pIx = pQ + up256z(sizeof(WindowQuery) * (size_t)nq);
        pRec = pIx + up256z(4 * (size_t)nq);
        pinBytes = pRec + up256z(4 * (size_t)kWinStride * (size_t)nq);
        sCd = up256z(scratchFree);
        scratchBytes = sCd + 4 * (size_t)nq * (size_t)n2;
    }
};

// The initialization search: one launch of window_top_kernel and one wait, then the host pass over the records;
// a query the records do not settle has its list read from the device.
int initialization_core(orbx_matcher* m, const char* who, const std::vector<WindowQuery>& queries, const WindowDev& in,
                        const orbx_keypoint* keys2_un, int n2, float min_x, float min_y, float max_x, float max_y,
                        const InitLayout& lay, const InitTail& tail) {
    const int nq = lay.nq;
    hipStream_t s = m->stream;
    uint8_t* base = static_cast<uint8_t*>(m->scratch);
    std::memcpy(m->pin + lay.pQ, queries.data(), sizeof(WindowQuery) * (size_t)nq);
    std::memcpy(m->pin + lay.pIx, tail.queryIndex->data(), 4 * (size_t)nq);
    const float invW = static_cast<float>(ORBX_GRID_COLS) / (max_x - min_x);
    const float invH = static_cast<float>(ORBX_GRID_ROWS) / (max_y - min_y);
    uint32_t* records = reinterpret_cast<uint32_t*>(m->pin + lay.pRec);
    uint32_t* cand = reinterpret_cast<uint32_t*>(base + lay.sCd);
    HIP_TRY(launch_window_top(nq, reinterpret_cast<const WindowQuery*>(m->pin + lay.pQ),
                              reinterpret_cast<const int32_t*>(m->pin + lay.pIx), in.d1, in.k2, in.d2, in.cellStart,
                              in.cellItems, min_x, min_y, invW, invH, n2, cand, records, s));
    HIP_TRY(hipStreamSynchronize(s));
    bool copyFailed = false;
    auto fetch = [&](int a, int count) -> const uint32_t* {
        if (a < 0 || a >= nq || count < 0 || count > n2) return nullptr;
        m->initTmp.list.resize((size_t)count);
        if (hipMemcpy(m->initTmp.list.data(), cand + (size_t)a * (size_t)n2, 4 * (size_t)count,
                      hipMemcpyDeviceToHost) != hipSuccess) {
            copyFailed = true;
            return nullptr;
        }
        return m->initTmp.list.data();
    };
    const int n = replay_search_for_initialization_top(tail.keys1_un, tail.n1, keys2_un, n2, nq,
                                                       tail.queryIndex->data(), records, fetch, tail.prev_matched,
                                                       tail.nnratio, tail.checkOri, tail.matches12, m->initTmp);
    if (n < 0) {
        for (int i = 0; i < tail.n1; i++) tail.matches12[i] = -1;
        if (copyFailed) {
            set_error("%s: reading a candidate list back failed: %s", who, hipGetErrorString(hipGetLastError()));
            return ORBX_ERR_HIP;
        }
        set_error("%s: inconsistent candidate lists", who);
        return ORBX_ERR_STATE;
    }
    *tail.nmatches = n;
    return ORBX_OK;
}

long long window_cap_total(const std::vector<WindowQuery>& queries, int nItems) {
    long long active = 0;
    for (const WindowQuery& q : queries) active += q.r >= 0.0f;
    return std::max(active * nItems, 1LL);
}

// Host-array form shared by the windowed searches: checks the grid, sends descriptors, keypoints and grid of the
// searched frame through the pinned staging buffer in one copy, then runs window_search_core.
int run_window_search(orbx_matcher* m, const char* who, const std::vector<WindowQuery>& queries, const uint8_t* desc1,
                      const orbx_keypoint* keys2, const uint8_t* desc2, const float* uright2, int n2,
                      const int32_t* cell_start2, const int32_t* cell_items2, float min_x, float min_y, float max_x,
                      float max_y, std::vector<int32_t>& head, const uint32_t** candHost,
                      const InitTail* tail = nullptr) {
    const int n1 = (int)queries.size();
    const int kCells = ORBX_GRID_COLS * ORBX_GRID_ROWS;
    const int nItems = cell_start2[kCells];
    {
        // The kernel follows these indices: a grid that is not a partition of (a subset of) the frame's keypoints
        // is refused here rather than read out of bounds there.
        bool ok = cell_start2[0] == 0 && nItems >= 0 && nItems <= n2;
        for (int c = 0; ok && c < kCells; c++) ok = cell_start2[c] <= cell_start2[c + 1];
        std::vector<uint8_t> seen(ok ? (size_t)n2 : 0, 0);
        for (int j = 0; ok && j < nItems; j++) {
            const int32_t i2 = cell_items2[j];
            ok = i2 >= 0 && i2 < n2 && !seen[i2];
            if (ok) seen[i2] = 1;
        }
        if (!ok) {
            set_error("%s: cell_start / cell_items do not describe a grid of the searched frame", who);
            return ORBX_ERR_INVALID;
        }
    }
    const long long capTotal = tail ? (long long)n1 * n2 : window_cap_total(queries, nItems);
    const int nDesc1 = tail ? tail->n1 : n1;              // the initialization search numbers its queries' descriptors by keypoint
    if (n2 > 0x7FFFFF || capTotal > 0x3FFFFFFF) {
        set_error("%s: too many keypoints (%d queries x %d)", who, n1, n2);
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // inputs packed in pinned memory in the layout of the device scratch: one copy in
    const size_t oD1 = 0, oK2 = oD1 + up256z(32 * (size_t)nDesc1), oD2 = oK2 + up256z(sizeof(orbx_keypoint) * n2),
                 oUr = oD2 + up256z(32 * (size_t)n2), oCs = oUr + up256z(uright2 ? 4 * (size_t)n2 : 0),
                 oCi = oCs + up256z(4 * (size_t)(kCells + 1)), oEnd = oCi + up256z(4 * (size_t)std::max(nItems, 1));
    const WindowCoreLayout lay(n1, capTotal, oEnd, oEnd);
    const InitLayout ilay(n1, n2, oEnd, oEnd);
    int rc = matcher_scratch(m, tail ? ilay.scratchBytes : lay.scratchBytes, s);
    if (rc) return rc;
    if ((rc = matcher_pinned(m, tail ? ilay.pinBytes : lay.pinBytes))) return rc;
    uint8_t* base = static_cast<uint8_t*>(m->scratch);
    std::memcpy(m->pin + oD1, desc1, 32 * (size_t)nDesc1);
    std::memcpy(m->pin + oK2, keys2, sizeof(orbx_keypoint) * n2);
    std::memcpy(m->pin + oD2, desc2, 32 * (size_t)n2);
    if (uright2) std::memcpy(m->pin + oUr, uright2, 4 * (size_t)n2);
    std::memcpy(m->pin + oCs, cell_start2, 4 * (size_t)(kCells + 1));
    if (nItems > 0) std::memcpy(m->pin + oCi, cell_items2, 4 * (size_t)nItems);
    HIP_TRY(hipMemcpyAsync(base, m->pin, oEnd, hipMemcpyHostToDevice, s));
    WindowDev in{base + oD1, reinterpret_cast<const orbx_keypoint*>(base + oK2), base + oD2,
                 uright2 ? reinterpret_cast<const float*>(base + oUr) : nullptr,
                 reinterpret_cast<const int32_t*>(base + oCs), reinterpret_cast<const int32_t*>(base + oCi)};
    if (tail) return initialization_core(m, who, queries, in, keys2, n2, min_x, min_y, max_x, max_y, ilay, *tail);
    return window_search_core(m, who, queries, in, n2, min_x, min_y, max_x, max_y, lay, head, candHost);
}

// SearchForInitialization's queries: one per frame-1 keypoint of level 0 (src/ORBmatcher.cc:427-431), a window
// around its previous match, level 0 only (:433); queryIndex[a] = the keypoint of query a.
void initialization_queries(const orbx_keypoint* keys1_un, int n1, const float* prev_matched, int window_size,
                            std::vector<WindowQuery>& queries, std::vector<int32_t>& queryIndex) {
    queries.clear();
    queryIndex.clear();
    for (int i = 0; i < n1; i++) {
        const int level1 = keys1_un[i].octave;
        if (level1 > 0) continue;                                                     // :429-431
        queries.push_back(WindowQuery{prev_matched[2 * i], prev_matched[2 * i + 1], (float)window_size, 0.0f, level1, level1});
        queryIndex.push_back(i);
    }
}
}  // namespace

extern "C" int orbx_search_for_initialization(orbx_matcher* m, const orbx_keypoint* keys1_un, const uint8_t* desc1,
                                              int n1, const orbx_keypoint* keys2_un, const uint8_t* desc2, int n2,
                                              const int32_t* cell_start2, const int32_t* cell_items2, float min_x,
                                              float min_y, float max_x, float max_y, float* prev_matched,
                                              int window_size, float nnratio, int check_orientation,
                                              int32_t* matches12, int* nmatches) {
    if (!m || n1 < 0 || n2 < 0 || !nmatches || !(max_x > min_x) || !(max_y > min_y) ||
        (n1 > 0 && (!keys1_un || !desc1 || !prev_matched || !matches12)) ||
        (n2 > 0 && (!keys2_un || !desc2 || !cell_start2 || !cell_items2))) {
        set_error("orbx_search_for_initialization: bad argument");
        return ORBX_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    // one query per frame-1 keypoint of level 0 (:427-431): window around its previous match, level 0 only (:433)
    std::vector<WindowQuery>& queries = m->winQueries;
    initialization_queries(keys1_un, n1, prev_matched, window_size, queries, m->winQueryIndex);
    if (queries.empty()) return ORBX_OK;
    std::vector<int32_t> head;
    const uint32_t* cand = nullptr;
    const InitTail tail{keys1_un, n1, &m->winQueryIndex, prev_matched, nnratio, check_orientation != 0, matches12, nmatches};
    return run_window_search(m, "orbx_search_for_initialization", queries, desc1, keys2_un, desc2, nullptr, n2,
                             cell_start2, cell_items2, min_x, min_y, max_x, max_y, head, &cand, &tail);
}

extern "C" int orbx_search_for_initialization_device(orbx_matcher* m, const orbx_keypoint* keys1_un,
                                                     const uint8_t* d_desc1, int n1, const orbx_keypoint* keys2_un,
                                                     const orbx_keypoint* d_keys2_un, const uint8_t* d_desc2, int n2,
                                                     const int32_t* d_cell_start2, const int32_t* d_cell_items2,
                                                     float min_x, float min_y, float max_x, float max_y,
                                                     float* prev_matched, int window_size, float nnratio,
                                                     int check_orientation, int32_t* matches12, int* nmatches) {
    if (!m || n1 < 0 || n2 < 0 || !nmatches || !(max_x > min_x) || !(max_y > min_y) ||
        (n1 > 0 && (!keys1_un || !d_desc1 || !prev_matched || !matches12)) ||
        (n2 > 0 && (!keys2_un || !d_keys2_un || !d_desc2 || !d_cell_start2 || !d_cell_items2)) ||
        ((uintptr_t)d_desc1 % 16) || ((uintptr_t)d_desc2 % 16)) {
        set_error("orbx_search_for_initialization_device: bad argument (descriptor arrays must be 16-byte aligned)");
        return ORBX_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    std::vector<WindowQuery>& queries = m->winQueries;
    initialization_queries(keys1_un, n1, prev_matched, window_size, queries, m->winQueryIndex);
    const int nq = (int)queries.size();
    if (nq == 0) return ORBX_OK;
    if (n2 > 0x7FFFFF || (long long)nq * n2 > 0x3FFFFFFF) {
        set_error("orbx_search_for_initialization_device: too many keypoints (%d queries x %d)", nq, n2);
        return ORBX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    const InitLayout lay(nq, n2, 0, 0);
    int rc = matcher_scratch(m, lay.scratchBytes, m->stream);
    if (rc) return rc;
    if ((rc = matcher_pinned(m, lay.pinBytes))) return rc;
    const WindowDev in{d_desc1, d_keys2_un, d_desc2, nullptr, d_cell_start2, d_cell_items2};
    const InitTail tail{keys1_un, n1, &m->winQueryIndex, prev_matched, nnratio, check_orientation != 0, matches12, nmatches};
    return initialization_core(m, "orbx_search_for_initialization_device", queries, in, keys2_un, n2, min_x, min_y,
                               max_x, max_y, lay, tail);
}

extern "C" int orbx_search_by_projection_frame(orbx_matcher* m, const float* proj_uv, const float* inv_z,
                                               const int32_t* last_octave, const float* last_angle,
                                               const uint8_t* desc_mp, const uint8_t* has_observations, int npts,
                                               const orbx_keypoint* keys_un, const uint8_t* desc,
                                               const float* u_right, const uint8_t* occupied, int n,
                                               const int32_t* cell_start, const int32_t* cell_items, float min_x,
                                               float min_y, float max_x, float max_y, const float* scale_factors,
                                               int nlevels, float th, float mbf, int direction, int max_dist,
                                               int check_orientation, int32_t* matches, int* nmatches) {
    if (!m || npts < 0 || n < 0 || !nmatches || !(max_x > min_x) || !(max_y > min_y) || !scale_factors ||
        nlevels < 1 || nlevels > ORBX_MAX_LEVELS || direction < 0 || direction > 2 ||
        (npts > 0 && (!proj_uv || !inv_z || !last_octave || !last_angle || !desc_mp || !has_observations)) ||
        (n > 0 && (!keys_un || !desc || !cell_start || !cell_items || !matches))) {
        set_error("orbx_search_by_projection_frame: bad argument");
        return ORBX_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n; i++) matches[i] = -1;
    if (npts == 0 || n == 0) return ORBX_OK;
    std::vector<WindowQuery> queries(npts);
    for (int i = 0; i < npts; i++) {
        WindowQuery q{0.f, 0.f, -1.0f, 0.f, 0, -1};
        const float invzc = inv_z[i], u = proj_uv[2 * i], v = proj_uv[2 * i + 1];
        const int o = last_octave[i];
        // (:1411-1420; an octave outside the scale table is not something the extractor produces)
        if (!(invzc < 0) && !(u < min_x || u > max_x) && !(v < min_y || v > max_y) && o >= 0 && o < nlevels) {
            q.x = u;
            q.y = v;
            q.r = th * scale_factors[o];                       // :1425
            q.ur = u - mbf * invzc;                            // :1453
            if (direction == 1) { q.minLevel = o; q.maxLevel = -1; }            // bForward (:1430)
            else if (direction == 2) { q.minLevel = 0; q.maxLevel = o; }        // bBackward (:1432)
            else { q.minLevel = o - 1; q.maxLevel = o + 1; }                    // (:1434)
            if (!(q.r >= 0.0f)) q.r = -1.0f;                   // a negative or NaN radius finds nothing
        }
        queries[i] = q;
    }
    std::vector<int32_t> head;
    const uint32_t* cand = nullptr;
    int rc = run_window_search(m, "orbx_search_by_projection_frame", queries, desc_mp, keys_un, desc, u_right, n,
                               cell_start, cell_items, min_x, min_y, max_x, max_y, head, &cand);
    if (rc) return rc;
    *nmatches = replay_search_by_projection_frame(npts, head.data(), head.data() + npts, cand, last_angle,
                                                  has_observations, keys_un, occupied, n, max_dist,
                                                  check_orientation != 0, matches);
    return ORBX_OK;
}

extern "C" int orbx_search_by_projection_points(orbx_matcher* m, const float* proj_xy, const float* proj_xr,
                                                const float* view_cos, const int32_t* level, const uint8_t* desc_mp,
                                                const uint8_t* has_observations, int npts,
                                                const orbx_keypoint* keys_un, const uint8_t* desc,
                                                const float* u_right, const uint8_t* occupied, int n,
                                                const int32_t* cell_start, const int32_t* cell_items, float min_x,
                                                float min_y, float max_x, float max_y, const float* scale_factors,
                                                int nlevels, float th, float nnratio, int32_t* matches,
                                                int* nmatches) {
    if (!m || npts < 0 || n < 0 || !nmatches || !(max_x > min_x) || !(max_y > min_y) || !scale_factors ||
        nlevels < 1 || nlevels > ORBX_MAX_LEVELS ||
        (npts > 0 && (!proj_xy || !view_cos || !level || !desc_mp || !has_observations || (u_right && !proj_xr))) ||
        (n > 0 && (!keys_un || !desc || !cell_start || !cell_items || !matches))) {
        set_error("orbx_search_by_projection_points: bad argument");
        return ORBX_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n; i++) matches[i] = -1;
    if (npts == 0 || n == 0) return ORBX_OK;
    const bool bFactor = th != 1.0;                                  // :45
    std::vector<WindowQuery> queries(npts);
    for (int i = 0; i < npts; i++) {
        WindowQuery q{0.f, 0.f, -1.0f, 0.f, 0, -1};
        const int lv = level[i];
        if (lv >= 0 && lv < nlevels) {                               // (PredictScale never leaves the table)
            float r = view_cos[i] > 0.998 ? 2.5 : 4.0;               // RadiusByViewingCos (:125-131)
            if (bFactor) r *= th;                                    // :59-60
            q.x = proj_xy[2 * i];
            q.y = proj_xy[2 * i + 1];
            q.r = r * scale_factors[lv];                             // :63, and the stereo bound of :88
            q.ur = u_right ? proj_xr[i] : 0.f;
            q.minLevel = lv - 1;
            q.maxLevel = lv;
            if (!(q.r >= 0.0f)) q.r = -1.0f;
        }
        queries[i] = q;
    }
    std::vector<int32_t> head;
    const uint32_t* cand = nullptr;
    int rc = run_window_search(m, "orbx_search_by_projection_points", queries, desc_mp, keys_un, desc, u_right, n,
                               cell_start, cell_items, min_x, min_y, max_x, max_y, head, &cand);
    if (rc) return rc;
    *nmatches = replay_search_by_projection_points(npts, head.data(), head.data() + npts, cand, has_observations,
                                                   keys_un, occupied, n, nnratio, matches);
    return ORBX_OK;
}

