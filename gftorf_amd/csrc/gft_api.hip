// gft_api.hip -- C ABI of libgftorf_rast.so (include/gftorf_rast.h): scratch layout,
// stage sequencing, stream handling, error strings, per-stage HIP-event timing.
// No torch headers; stateless between calls except for the opt-in profiler.
#include "gft_internal.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <cstdlib>
#include <mutex>
#include <vector>

// ---- errors ---------------------------------------------------------------
static thread_local char g_err[512] = "";

int gft_fail(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char* gft_last_error(void) { return g_err; }
extern "C" int gft_abi_version(void) { return GFT_ABI_VERSION; }
// ids a point_list sized for `binning_instances` holds: one head slot per tile + the pool (tile-pull binning), which
// is also what whole-frame binning needs (binning_instances ids)
static size_t point_list_slots(int64_t binning_instances, int32_t W, int32_t H)
{
    if (binning_instances <= 0) return 0;
    const size_t T = (size_t)((W + GFT_TILE_X - 1) / GFT_TILE_X) * (size_t)((H + GFT_TILE_Y - 1) / GFT_TILE_Y);
    return T * (size_t)GFT_HEAD_SLOT + (size_t)binning_instances;
}
extern "C" size_t gft_det_partials_bytes(int64_t binning_instances, int32_t W, int32_t H)
{
    return point_list_slots(binning_instances, W, H) * 4 * GFT_ACC_STRIDE * sizeof(float);
}
// inverse of gft_binning_bytes for capacities that are multiples of 64 (what a caller that only holds the buffer needs)
extern "C" int64_t gft_binning_capacity(size_t bytes, int32_t W, int32_t H)
{
    const size_t T = (size_t)((W + GFT_TILE_X - 1) / GFT_TILE_X) * (size_t)((H + GFT_TILE_Y - 1) / GFT_TILE_Y);
    const size_t fixed = T * (size_t)GFT_HEAD_SLOT * 4 + GFT_ALIGN;
    if (bytes <= fixed) return 0;
    return (int64_t)((bytes - fixed) / 12);
}

// ---- layout ------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + GFT_ALIGN - 1) & ~(size_t)(GFT_ALIGN - 1); }

void gft_compute_layout(int32_t P, int32_t W, int32_t H, int64_t R, gft_layout* L)
{
    memset(L, 0, sizeof(*L));
    const size_t p = (size_t)(P > 0 ? P : 0);
    size_t o = 0;
    L->geom_rec_a = o;    o = align_up(o + p * 32);
    L->geom_rec_b = o;    o = align_up(o + p * 32);
    L->geom_depth = o;    o = align_up(o + p * 4);
    L->geom_tiles = o;    o = align_up(o + p * 4);
    L->geom_rect = o;     o = align_up(o + p * 8);
    L->geom_dirgrad = o;  o = align_up(o + p * 64);
    L->geom_clamped = o;  o = align_up(o + p);
    L->geom_need = o;     o = align_up(o + p);
    L->geom_blockhist = o; o = align_up(o + ((p + BIN_CHUNK - 1) / BIN_CHUNK) * GFT_BLOCKHIST_TILES * 2);
    L->geom_total = o;

    const size_t n = (size_t)W * (size_t)H;
    const size_t T = (size_t)((W + GFT_TILE_X - 1) / GFT_TILE_X) * (size_t)((H + GFT_TILE_Y - 1) / GFT_TILE_Y);
    o = 0;
    L->img_pix_state = o;   o = align_up(o + n * 16);
    L->img_ranges = o;      o = align_up(o + T * 8);
    L->img_tile_max = o;    o = align_up(o + T * 4 * 4);   // one entry per 8x8 quadrant
    L->img_ctrl = o;        o += (GFT_CTRL_WORDS + GFT_TICKET_WORDS) * 4;   // ctrl words (+ ticket counters), tile counters, tile cuts and the supertile tables are
    L->img_tile_cnt = o;    o += T * 4;                       // contiguous: k_preprocess_fwd clears them in one sweep
    L->img_tile_cut = o;    o += T * 4;
    L->img_super_tab = o;   o = align_up(o + 4 * (size_t)GFT_SUPER_CELLS * 4);
    L->img_tile_cursor = o; o = align_up(o + T * 4);
    L->img_tile_order = o;  o = align_up(o + T * 4);
    L->img_front_len = o;   o = align_up(o + T * 4);
    L->img_unit_flag = o;   o = align_up(o + T * 16);
    L->img_resume_state = o; o = align_up(o + n * 64);
    L->img_pix_sums = o;    o = align_up(o + n * 32);
    L->img_snaps = o;       o = align_up(o + 4 * T * (size_t)(gft_bwd_segments(T) - 1) * GFT_SNAP_F4 * 64 * 16);
    L->img_total = o;

    const size_t r = (size_t)(R > 0 ? R : 0);
    o = 0;
    L->bin_keys = o;        o = align_up(o + r * 8);
    L->bin_point_list = o;  o = align_up(o + (r ? T * (size_t)GFT_HEAD_SLOT + r : 0) * 4);
    L->bin_total = o + GFT_ALIGN;
}

GeomView gft_geom_view(void* base, const gft_layout& L)
{
    char* b = (char*)base;
    GeomView g;
    g.rec_a = (float4*)(b + L.geom_rec_a);
    g.rec_b = (float4*)(b + L.geom_rec_b);
    g.depth = (float*)(b + L.geom_depth);
    g.tiles = (uint32_t*)(b + L.geom_tiles);
    g.rect = (ushort4*)(b + L.geom_rect);
    g.dirgrad = (float4*)(b + L.geom_dirgrad);
    g.clamped = (uint8_t*)(b + L.geom_clamped);
    g.need = (uint8_t*)(b + L.geom_need);
    g.blockhist = (uint16_t*)(b + L.geom_blockhist);
    return g;
}

ImgView gft_img_view(void* base, const gft_layout& L)
{
    char* b = (char*)base;
    ImgView v;
    v.pix_state = (float4*)(b + L.img_pix_state);
    v.ranges = (uint2*)(b + L.img_ranges);
    v.tile_max = (uint32_t*)(b + L.img_tile_max);
    v.ctrl = (uint32_t*)(b + L.img_ctrl);
    v.tile_cnt = (uint32_t*)(b + L.img_tile_cnt);
    v.tile_cut = (uint32_t*)(b + L.img_tile_cut);
    v.super_tab = (uint32_t*)(b + L.img_super_tab);
    v.tile_cursor = (uint32_t*)(b + L.img_tile_cursor);
    v.tile_order = (uint32_t*)(b + L.img_tile_order);
    v.front_len = (uint32_t*)(b + L.img_front_len);
    v.unit_flag = (uint32_t*)(b + L.img_unit_flag);
    v.resume_state = (float4*)(b + L.img_resume_state);
    v.pix_sums = (float4*)(b + L.img_pix_sums);
    v.snaps = (float4*)(b + L.img_snaps);
    return v;
}

BinView gft_bin_view(void* base, const gft_layout& L)
{
    char* b = (char*)base;
    BinView v;
    v.keys = (uint64_t*)(b + L.bin_keys);
    v.point_list = (uint32_t*)(b + L.bin_point_list);
    return v;
}

extern "C" size_t gft_geom_bytes(int32_t P)
{
    gft_layout L;
    gft_compute_layout(P, 16, 16, 0, &L);
    return L.geom_total + GFT_ALIGN;
}

extern "C" size_t gft_image_bytes(int32_t W, int32_t H)
{
    gft_layout L;
    gft_compute_layout(0, W, H, 0, &L);
    return L.img_total + GFT_ALIGN;
}

extern "C" size_t gft_cell_sched_words(int32_t W, int32_t H)
{
    gft_config c;
    memset(&c, 0, sizeof(c));
    c.W = W; c.H = H; c.near_n = 0.01f; c.far_n = 100.f;
    return (W > 0 && H > 0 && gft_tile_pull_ok(c)) ? gft_cell_sched_words_of(c) : 0;
}

extern "C" size_t gft_binning_bytes(int64_t R, int32_t W, int32_t H)
{
    gft_layout L;
    gft_compute_layout(0, W, H, R, &L);
    return L.bin_total;
}

extern "C" size_t gft_acc_bytes(int32_t P)
{
    const size_t p = (size_t)(P > 0 ? P : 0);
    return (p * GFT_ACC_STRIDE + 2 * ((p + 63) / 64) + 64) * sizeof(float);
}

extern "C" int gft_get_layout(int32_t P, int32_t W, int32_t H, int64_t R, gft_layout* out)
{
    if (!out) return gft_fail("gft_get_layout: out is NULL");
    gft_compute_layout(P, W, H, R, out);
    return 0;
}

// ---- profiler -----------------------------------------------------------------
enum Stage { ST_PRE_FWD, ST_TILE_COUNT, ST_TILE_SCATTER, ST_TILE_SORT, ST_RENDER_FWD, ST_RENDER_BWD, ST_PRE_BWD, ST_MEMSET, ST_COUNT };

struct Profiler {
    std::mutex mu;
    bool on = false;
    struct Rec { int stage; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double ms[ST_COUNT] = {0};
    int64_t fwd = 0, bwd = 0;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void drain()
    {
        for (auto& r : pending) {
            float t = 0.f;
            if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms[r.stage] += t;
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        pending.clear();
    }
};
static Profiler g_prof;

struct StageTimer {
    hipStream_t s; int stage; hipEvent_t a = nullptr, b = nullptr; bool on;
    StageTimer(hipStream_t s_, int st) : s(s_), stage(st), on(g_prof.on)
    {
        if (on) {
            std::lock_guard<std::mutex> lk(g_prof.mu);
            a = g_prof.get(); b = g_prof.get();
            if (a && b) (void)hipEventRecord(a, s); else on = false;
        }
    }
    ~StageTimer()
    {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof.mu);
            g_prof.pending.push_back({stage, a, b});
        }
    }
};

extern "C" int gft_profile_enable(int on) { std::lock_guard<std::mutex> lk(g_prof.mu); g_prof.on = on != 0; return 0; }

extern "C" int gft_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    memset(g_prof.ms, 0, sizeof(g_prof.ms));
    g_prof.fwd = g_prof.bwd = 0;
    return 0;
}

extern "C" int gft_profile_read(gft_profile* out)
{
    if (!out) return gft_fail("gft_profile_read: out is NULL");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.drain();
    out->preprocess_fwd_ms = g_prof.ms[ST_PRE_FWD];
    out->tile_count_ms = g_prof.ms[ST_TILE_COUNT];
    out->tile_scatter_ms = g_prof.ms[ST_TILE_SCATTER];
    out->tile_sort_ms = g_prof.ms[ST_TILE_SORT];
    out->render_fwd_ms = g_prof.ms[ST_RENDER_FWD];
    out->render_bwd_ms = g_prof.ms[ST_RENDER_BWD];
    out->preprocess_bwd_ms = g_prof.ms[ST_PRE_BWD];
    out->memset_ms = g_prof.ms[ST_MEMSET];
    out->forward_calls = g_prof.fwd;
    out->backward_calls = g_prof.bwd;
    return 0;
}

// ---- side stream for the gradient zero fill --------------------------------------------------
// One non-blocking stream and two events per device.  The forward records `after_main` behind its binning kernels and
// makes the side stream wait for it (everything queued on the caller's stream so far is then done: the buffer cannot
// still be in use by earlier work), fills, and records `filled`; the backward makes the caller's stream wait for
// `filled`.  Re-recording an event does not disturb waits already queued on its earlier record; the latest fill on
// the in-order side stream implies all earlier ones, so one event per device serves any number of forwards in flight.
struct SideFill {
    hipStream_t stream = nullptr;
    hipEvent_t after_main = nullptr, filled = nullptr;
    bool pending = false;
};
static std::mutex g_side_mu;
static SideFill g_side[64];

static SideFill* side_of_current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideFill& f = g_side[dev];
    if (!f.stream) {
        if (hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking) != hipSuccess) { f.stream = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&f.after_main, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f.filled, hipEventDisableTiming) != hipSuccess)
            return nullptr;
    }
    return &f;
}

// ---- helpers -----------------------------------------------------------------
#define GFT_STAGE(stream, cfg, name, call)                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return gft_fail("%s: launch failed: %s", name, hipGetErrorString(e_)); \
        if ((cfg)->debug) {                                                                  \
            e_ = hipStreamSynchronize(stream);                                               \
            if (e_ != hipSuccess) return gft_fail("%s: %s", name, hipGetErrorString(e_));    \
        }                                                                                    \
    } while (0)

// Zero fill as a KERNEL, not hipMemsetAsync.  On this platform (ROCm 7.0 / torch 2.10) a hipMemsetAsync captured into a HIP
// graph fills with the right value only the first time the graph is replayed; from the second replay on the buffer holds
// garbage (profiles/experiments/graph_memset_node_repro.py: fill with ones -> memset 0 -> sum, 1 MiB and 160 MiB, pure
// torch + HIP).  The deterministic backward's clear of its partial rows fell to it (wrong sums "from the second replay
// on", parked in round 5; tests/test_gpu_graph.py::test_two_deterministic_backwards_in_one_graph), and every small clear of
// a ticket or status word on a captured path was exposed to it.  The library issues no memset nodes.
__global__ __launch_bounds__(256) void k_zero_fill(uint4* __restrict__ p16, size_t n16, uint8_t* __restrict__ tail, size_t ntail)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

__global__ void k_post_empty(uint32_t* status) { status[GFT_CTRL_SEQ] = 1u; }      // gft_forward_enqueue with P == 0: "posted", R = 0

hipError_t gft_zero_async(void* ptr, size_t bytes, hipStream_t s)
{
    if (!ptr || !bytes) return hipSuccess;
    uint8_t* p = static_cast<uint8_t*>(ptr);
    // (head up to a 16-byte boundary and tail byte-wise by the first workgroup; every caller's buffers are at least 4-byte
    // aligned and mostly 256)
    const size_t head = (16 - ((uintptr_t)p & 15)) & 15;
    if (head && head < bytes) {
        hipLaunchKernelGGL(k_zero_fill, dim3(1), dim3(256), 0, s, (uint4*)nullptr, (size_t)0, p, head);
        p += head; bytes -= head;
    } else if (head) {
        hipLaunchKernelGGL(k_zero_fill, dim3(1), dim3(256), 0, s, (uint4*)nullptr, (size_t)0, p, bytes);
        return hipGetLastError();
    }
    const size_t n16 = bytes / 16, ntail = bytes % 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n16, p + n16 * 16, ntail);
    return hipGetLastError();
}

static int check_config(const gft_config* c)
{
    if (!c) return gft_fail("config is NULL");
    if (c->P < 0 || c->W <= 0 || c->H <= 0) return gft_fail("bad sizes P=%d W=%d H=%d", c->P, c->W, c->H);
    if (c->D < 0 || c->D > 3) return gft_fail("sh_degree %d not in 0..3", c->D);
    const int need = (c->D + 1) * (c->D + 1);
    if (c->M != 0 && c->M < need) return gft_fail("shs holds %d coefficients, sh_degree %d needs %d", c->M, c->D, need);
    if (c->M_p != 0 && c->M_p < need) return gft_fail("shs_p holds %d coefficients, sh_degree %d needs %d", c->M_p, c->D, need);
    return 0;
}

// ---- host mailbox ---------------------------------------------------------------
// The instance count R, the "prefiltered point culled" flag and the longest tile list are the
// only values the host needs from the device during a forward (reference: blocking cudaMemcpy,
// rasterizer_impl.cu:311).  The scan workgroup stores them into pinned host memory and the host
// polls the sequence word, so no copy kernel and no interrupt-driven wake-up sit on the path,
// and with gft_forward() the stage-2 kernels are already queued behind the scan.
struct Mailbox {
    std::mutex mu;
    uint32_t* host = nullptr;      // GFT_MAIL_SLOTS x GFT_CTRL_WORDS words, pinned + mapped
    uint32_t* dev = nullptr;
    uint32_t next = 0;
};
#define GFT_MAIL_SLOTS 256
static Mailbox g_mail;

static int mailbox_acquire(uint32_t** dev_slot, volatile uint32_t** host_slot, uint32_t* seq)
{
    std::lock_guard<std::mutex> lk(g_mail.mu);
    if (!g_mail.host) {
        void* p = nullptr;
        const size_t bytes = (size_t)GFT_MAIL_SLOTS * GFT_CTRL_WORDS * sizeof(uint32_t);
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped);
        }
        if (e != hipSuccess) return gft_fail("mailbox: hipHostMalloc failed: %s", hipGetErrorString(e));
        memset(p, 0, bytes);
        void* d = nullptr;
        e = hipHostGetDevicePointer(&d, p, 0);
        if (e != hipSuccess) { (void)hipHostFree(p); return gft_fail("mailbox: no device pointer: %s", hipGetErrorString(e)); }
        g_mail.host = (uint32_t*)p;
        g_mail.dev = (uint32_t*)d;
    }
    g_mail.next++;
    if (g_mail.next == 0) g_mail.next = 1;          // 0 marks an empty slot
    const uint32_t slot = g_mail.next % GFT_MAIL_SLOTS;
    *seq = g_mail.next;
    *dev_slot = g_mail.dev + (size_t)slot * GFT_CTRL_WORDS;
    *host_slot = g_mail.host + (size_t)slot * GFT_CTRL_WORDS;
    g_mail.host[(size_t)slot * GFT_CTRL_WORDS + GFT_CTRL_FLAGS] = 0u;     // the preprocess kernel ORs into it
    return 0;
}

// Spin on the sequence word; every few microseconds ask the stream whether it is idle or
// broken, so a failed launch can never turn into an endless wait.
static int mailbox_wait(hipStream_t s, volatile uint32_t* host_slot, uint32_t seq, uint32_t out[GFT_CTRL_WORDS])
{
    for (;;) {
        for (int i = 0; i < 4096; i++) {
            if (__atomic_load_n(&host_slot[GFT_CTRL_SEQ], __ATOMIC_ACQUIRE) == seq) goto ready;
            __builtin_ia32_pause();
        }
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) {
            if (__atomic_load_n(&host_slot[GFT_CTRL_SEQ], __ATOMIC_ACQUIRE) == seq) goto ready;
            return gft_fail("forward: the stream drained but the tile scan never reported");
        }
        if (q != hipErrorNotReady) return gft_fail("forward: %s", hipGetErrorString(q));
    }
ready:
    for (int i = 0; i < GFT_CTRL_WORDS; i++) out[i] = host_slot[i];
    return 0;
}

static int check_stage1(const gft_config* cfg, const gft_forward_io* io, const char* who)
{
    if (!io->means3D || !io->opacities || !io->viewmatrix || !io->projmatrix || !io->campos || !io->geom ||
        !io->radii || !io->pixels)
        return gft_fail("%s: required pointer is NULL", who);
    if ((io->shs == nullptr) == (io->colors_precomp == nullptr))
        return gft_fail("Please provide excatly one of either SHs or precomputed colors!");
    if ((io->scales == nullptr || io->rotations == nullptr) == (io->cov3D_precomp == nullptr))
        return gft_fail("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if ((io->shs != nullptr) != (cfg->M > 0) || (io->shs_p != nullptr) != (cfg->M_p > 0))
        return gft_fail("M / M_p do not match the presence of shs / shs_p");
    if (!io->img) return gft_fail("%s: img buffer is NULL", who);
    return 0;
}

static int check_stage2(const gft_forward_io* io, const char* who)
{
    if (!io->img || !io->bg || !io->out_color || !io->out_phasor || !io->out_depth || !io->out_normal ||
        !io->out_acc || !io->out_entropy || !io->out_depth_distortion || !io->out_amp_distortion ||
        !io->out_distribution)
        return gft_fail("%s: required pointer is NULL", who);
    return 0;
}

// GFT_LAZY_SORT=0 in the environment sorts every tile list whole (k_tile_sort_small / _big) instead of
// head first, tail on demand; results are identical.  (Whole-frame binning only: tile-pull binning needs the flag /
// resume protocol of the lazy sort.)
static bool lazy_sort_enabled()
{
    static const bool on = [] { const char* e = getenv("GFT_LAZY_SORT"); return e ? atoi(e) != 0 : true; }();
    return on;
}

extern "C" int gft_lazy_sort(void) { return lazy_sort_enabled() ? 1 : 0; }

// Binning mode: 1 = tile pull (k_pull.hip, the default), 0 = whole frame (k_binning.hip: every instance counted,
// scattered and sorted, the structure of the reference).  Results are identical; GFT_LAZY_BIN=0 in the environment or
// gft_set_binning_mode(0) select the whole-frame path (tests compare the two bit for bit).
static std::atomic<int> g_binning_mode{-1};
extern "C" int gft_set_binning_mode(int mode)
{
    g_binning_mode.store(mode < 0 ? -1 : (mode ? 1 : 0));      // (< 0: back to the environment's choice)
    return 0;
}
static bool pull_enabled(const gft_config* cfg)
{
    int m = g_binning_mode.load();
    if (m < 0) {
        const char* e = getenv("GFT_LAZY_BIN");
        m = e ? (atoi(e) != 0) : 1;
        g_binning_mode.store(m);
    }
    return m != 0 && lazy_sort_enabled() && gft_tile_pull_ok(*cfg);
}
extern "C" int gft_binning_mode(const gft_config* cfg) { return (cfg && pull_enabled(cfg)) ? 1 : 0; }

// Forward blend kernel: -1 = default (segment-parallel on frames that leave the chip under-filled), 0 = one wave per
// quadrant always, 1 = segments wherever more than one wave per quadrant fits; GFT_FWD_SEG in the environment.
static std::atomic<int> g_render_mode{-2};
extern "C" int gft_set_render_mode(int mode)
{
    g_render_mode.store(mode < 0 ? -2 : (mode ? 1 : 0));       // (< 0: back to the environment's choice)
    return 0;
}
int gft_render_mode()
{
    int m = g_render_mode.load();
    if (m == -2) {
        const char* e = getenv("GFT_FWD_SEG");
        m = e ? (atoi(e) != 0 ? 1 : 0) : -1;
        g_render_mode.store(m);
    }
    return m;
}

// preprocess + instance counting; the totals arrive in the mailbox slot
static int enqueue_count(hipStream_t s, const gft_config* cfg, const gft_forward_io* io, const GeomView& g, const ImgView& im,
                         uint32_t* mail_dev, uint32_t seq, bool pull, int64_t status_cap = -1);

// `count` = false: geometry only; the scatter pass of stage 2 bins by the caller's list schedule and posts the totals
static int enqueue_stage1(hipStream_t s, const gft_config* cfg, const gft_forward_io* io, const GeomView& g,
                          const ImgView& im, uint32_t* mail_dev, uint32_t seq, bool pull, bool count = true, int64_t status_cap = -1)
{
    {
        // (the preprocess kernel also zeroes the ctrl words, the tile counters and the supertile tables for the binning
        // kernels behind it: no fill launch)
        StageTimer t(s, ST_PRE_FWD);
        GFT_STAGE(s, cfg, "preprocess_fwd", gft_launch_preprocess_fwd(s, *cfg, *io, g, im, mail_dev, pull));
    }
    if (io->grads_zero && io->grads_zero_bytes) {
        // Optional (gft_forward_io.grads_zero; the Python operator does not use it: the fill costs beside the other kernels what
        // it saves the backward -- HBM time is conserved --, measured in round 2): zero fill of the backward's gradient
        // tensors on the side stream, behind the preprocess kernel.  HAZARD, which is why it stays a switch: only
        // gft_backward makes the caller's stream wait for the fill; a caller that drops the graph without running the
        // backward may hand the buffer to other work while the fill is still pending.
        SideFill* f = side_of_current_device();
        if (!f) return gft_fail("forward: no side stream for the gradient fill");
        GFT_CHECK_HIP(hipEventRecord(f->after_main, s));
        GFT_CHECK_HIP(hipStreamWaitEvent(f->stream, f->after_main, 0));
        GFT_CHECK_HIP(gft_zero_async(io->grads_zero, io->grads_zero_bytes, f->stream));
        GFT_CHECK_HIP(hipEventRecord(f->filled, f->stream));
        f->pending = true;
    }
    return count ? enqueue_count(s, cfg, io, g, im, mail_dev, seq, pull, status_cap) : 0;
}

static int enqueue_count(hipStream_t s, const gft_config* cfg, const gft_forward_io* io, const GeomView& g, const ImgView& im,
                         uint32_t* mail_dev, uint32_t seq, bool pull, int64_t status_cap)
{
    StageTimer t(s, ST_TILE_COUNT);
    if (pull) {
        BinView none;
        none.keys = nullptr; none.point_list = nullptr;
        const int T = ((cfg->W + GFT_TILE_X - 1) / GFT_TILE_X) * ((cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y);
        GFT_STAGE(s, cfg, "super_count", gft_launch_super_bin(s, *cfg, g, im, none, mail_dev, seq, 0, 0u,
                                                              gft_fwd_segmented(T) ? nullptr : io->tile_hints, io->cell_sched, status_cap));
    } else {
        GFT_STAGE(s, cfg, "tile_count", gft_launch_tile_count(s, *cfg, g, im, mail_dev, seq, status_cap));
    }
    return 0;
}

// binning + render; `cap` = instances the binning buffer holds.  With check_cap the kernels compare the device-side
// count against it and do nothing on overflow.
static int enqueue_stage2(hipStream_t s, const gft_config* cfg, const gft_forward_io* io, const GeomView& g,
                          const ImgView& im, const BinView& b, bool binned, int64_t max_tile_list, bool check_cap,
                          uint32_t cap, bool pull, bool whole_lists, uint32_t* sched_mail = nullptr, uint32_t sched_seq = 0u)
{
    // The backward's accumulator clear (64 B per Gaussian of pure HBM writes) rides along with the
    // per-tile sort kernels, whose workgroups are bound by LDS and VALU: each writes a slice of zeros first.
    float* clear = (io->acc && cfg->want_backward && cfg->P > 0) ? io->acc : nullptr;
    const size_t clear_bytes = (size_t)cfg->P * GFT_ACC_STRIDE * sizeof(float);
    const bool lazy = binned && lazy_sort_enabled();
    pull = pull && binned;
    if (pull) {
        // tile-pull binning: ids to supertiles, every tile collects, sorts and writes the head of its own list, the
        // Gaussians in a head get their appearance
        {
            StageTimer t(s, ST_TILE_SCATTER);
            if (sched_mail) {
                // no count pass ran: the entries go where the caller's list schedule puts them, this pass posts the totals
                const int T0 = ((cfg->W + GFT_TILE_X - 1) / GFT_TILE_X) * ((cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y);
                GFT_STAGE(s, cfg, "super_append", gft_launch_super_bin(s, *cfg, g, im, b, sched_mail, sched_seq, 2, cap,
                                                                       gft_fwd_segmented(T0) ? nullptr : io->tile_hints, io->cell_sched));
            } else {
                GFT_STAGE(s, cfg, "super_scatter", gft_launch_super_bin(s, *cfg, g, im, b, nullptr, 0u, 1, cap));
            }
        }
        {
            StageTimer t(s, ST_TILE_SORT);
            // (no schedule on frames whose forward blend is segment-parallel: see gft_launch_render_fwd)
            const int T = ((cfg->W + GFT_TILE_X - 1) / GFT_TILE_X) * ((cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y);
            const uint32_t* th = gft_fwd_segmented(T) ? nullptr : io->tile_hints;
            GFT_STAGE(s, cfg, "tile_pull", gft_launch_tile_pull(s, *cfg, g, im, b, cap, clear, clear_bytes, th, whole_lists && th));
        }
        {
            StageTimer t(s, ST_PRE_FWD);
            GFT_STAGE(s, cfg, "appearance", gft_launch_appearance(s, *cfg, *io, g, im, cap));
        }
    } else if (binned) {
        {
            StageTimer t(s, ST_TILE_SCATTER);
            GFT_STAGE(s, cfg, "tile_scatter", gft_launch_tile_scatter(s, *cfg, g, im, b, cap, (int64_t)cap));
        }
        {
            StageTimer t(s, ST_TILE_SORT);
            if (lazy)
                GFT_STAGE(s, cfg, "tile_front", gft_launch_tile_front(s, *cfg, im, b, cap, clear, clear_bytes));
            else
                GFT_STAGE(s, cfg, "tile_sort", gft_launch_tile_sort(s, *cfg, max_tile_list, im, b, cap, clear, clear_bytes));
        }
    } else if (clear) {
        GFT_CHECK_HIP(gft_zero_async(clear, clear_bytes, s));
    }
    {
        StageTimer t(s, ST_RENDER_FWD);
        GFT_STAGE(s, cfg, "render_fwd", gft_launch_render_fwd(s, *cfg, *io, g, im, b, check_cap, cap, lazy ? 1 : 0, pull));
    }
    if (lazy) {
        // quadrants that used up the sorted head of their list with unsaturated pixels: complete those lists, continue
        // those quadrants (both launches leave at once when the first pass raised no flag)
        {
            StageTimer t(s, ST_TILE_SORT);
            if (pull)
                GFT_STAGE(s, cfg, "tail_build", gft_launch_tail_build(s, *cfg, *io, g, im, b, cap, cfg->want_backward != 0));
            else
                GFT_STAGE(s, cfg, "tile_tail", gft_launch_tile_tail(s, *cfg, g, im, b, cap, cfg->want_backward != 0));
        }
        if (!(pull && gft_tail_resumes())) {      // (tile-pull binning: the tail builder's workgroups resume their quadrants themselves)
            StageTimer t(s, ST_RENDER_FWD);
            GFT_STAGE(s, cfg, "render_resume", gft_launch_render_fwd(s, *cfg, *io, g, im, b, check_cap, cap, 2, pull));
        }
    }
    if (g_prof.on) { std::lock_guard<std::mutex> lk(g_prof.mu); g_prof.fwd++; }
    return 0;
}

// ---- forward, stage 1 -----------------------------------------------------------
extern "C" int gft_forward_preprocess(void* hip_stream, const gft_config* cfg, const gft_forward_io* io,
                                      int64_t* num_rendered, int64_t* max_tile_list)
{
    if (max_tile_list) *max_tile_list = 0;
    if (check_config(cfg)) return 1;
    if (!io || !num_rendered) return gft_fail("gft_forward_preprocess: NULL argument");
    *num_rendered = 0;
    if (cfg->P == 0) return 0;
    if (check_stage1(cfg, io, "gft_forward_preprocess")) return 1;
    hipStream_t s = (hipStream_t)hip_stream;
    gft_layout L;
    gft_compute_layout(cfg->P, cfg->W, cfg->H, 0, &L);
    GeomView g = gft_geom_view(io->geom, L);
    ImgView im = gft_img_view(io->img, L);
    uint32_t* mail_dev; volatile uint32_t* mail_host; uint32_t seq;
    if (mailbox_acquire(&mail_dev, &mail_host, &seq)) return 1;
    if (enqueue_stage1(s, cfg, io, g, im, mail_dev, seq, pull_enabled(cfg))) return 1;
    // the one blocking read of the forward (reference rasterizer_impl.cu:311)
    uint32_t host[GFT_CTRL_WORDS];
    if (mailbox_wait(s, mail_host, seq, host)) return 1;
    if (host[GFT_CTRL_FLAGS] & 1u)
        return gft_fail("Point is filtered although prefiltered is set. This shouldn't happen!");
    *num_rendered = (int64_t)host[GFT_CTRL_TOTAL];
    if (max_tile_list) *max_tile_list = (int64_t)host[GFT_CTRL_MAXCNT];
    return 0;
}

// ---- forward, stage 2 -----------------------------------------------------------
extern "C" int gft_forward_render(void* hip_stream, const gft_config* cfg, const gft_forward_io* io,
                                  int64_t binning_instances, int64_t max_tile_list)
{
    if (check_config(cfg)) return 1;
    if (!io) return gft_fail("gft_forward_render: io is NULL");
    if (check_stage2(io, "gft_forward_render")) return 1;
    if (binning_instances < 0 || binning_instances > 0xffffffffll) return gft_fail("gft_forward_render: bad instance count");
    if (binning_instances > 0 && !io->binning) return gft_fail("gft_forward_render: binning buffer is NULL");
    if (cfg->P > 0 && (!io->geom || !io->radii || !io->pixels)) return gft_fail("gft_forward_render: geom/radii/pixels NULL");
    hipStream_t s = (hipStream_t)hip_stream;
    const int64_t R = cfg->P == 0 ? 0 : binning_instances;
    gft_layout L;
    gft_compute_layout(cfg->P, cfg->W, cfg->H, R, &L);
    GeomView g = gft_geom_view(io->geom, L);
    ImgView im = gft_img_view(io->img, L);
    BinView b = gft_bin_view(io->binning, L);
    const bool pull = cfg->P > 0 && pull_enabled(cfg);
    if (cfg->P == 0 || (pull && R == 0)) {
        // no stage 1 ran (or its tile-pull count pass, which leaves the per-tile tables to the pull kernel): every tile
        // list is empty, all totals are zero
        const int gx = (cfg->W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y;
        GFT_CHECK_HIP(gft_zero_async(im.ranges, (size_t)gx * gy * sizeof(uint2), s));
        if (cfg->P == 0) GFT_CHECK_HIP(gft_zero_async(im.ctrl, GFT_CTRL_WORDS * sizeof(uint32_t), s));
    }
    // (two stages: the first frame of a shape, a frame that outgrew its buffer, a caller that drives the C ABI itself -- the
    // build of k_tile_pull that honours a schedule whenever one is handed over)
    return enqueue_stage2(s, cfg, io, g, im, b, R > 0, max_tile_list, cfg->P > 0, (uint32_t)R, pull, io->tile_hints != nullptr);
}

// ---- forward, one call ------------------------------------------------------------
extern "C" int gft_forward(void* hip_stream, const gft_config* cfg, const gft_forward_io* io,
                           const gft_forward_hints* hints, gft_forward_report* report)
{
    if (report) memset(report, 0, sizeof(*report));
    if (check_config(cfg)) return 1;
    if (!io || !hints || !report) return gft_fail("gft_forward: NULL argument");
    if (check_stage2(io, "gft_forward")) return 1;
    const int64_t binning_instances = hints->binning_instances;
    if (binning_instances < 0 || binning_instances > 0xffffffffll) return gft_fail("gft_forward: bad instance count");
    if (cfg->P == 0) return gft_forward_render(hip_stream, cfg, io, 0, 0);
    if (check_stage1(cfg, io, "gft_forward")) return 1;
    if (binning_instances > 0 && !io->binning) return gft_fail("gft_forward: binning buffer is NULL");
    hipStream_t s = (hipStream_t)hip_stream;
    gft_layout L;
    gft_compute_layout(cfg->P, cfg->W, cfg->H, binning_instances, &L);
    GeomView g = gft_geom_view(io->geom, L);
    ImgView im = gft_img_view(io->img, L);
    BinView b = gft_bin_view(io->binning, L);
    uint32_t* mail_dev; volatile uint32_t* mail_host; uint32_t seq;
    if (mailbox_acquire(&mail_dev, &mail_host, &seq)) return 1;
    const bool pull = pull_enabled(cfg);
    // with a list schedule of this camera's earlier frame: no count pass, the scatter pass appends by it and posts the totals
    const bool by_sched = pull && binning_instances > 0 && io->cell_sched != nullptr && hints->use_cell_sched != 0;
    if (enqueue_stage1(s, cfg, io, g, im, mail_dev, seq, pull, !by_sched)) return 1;
    if (pull && binning_instances == 0) {
        // (no buffer: stage 2 renders empty lists, which is right only if R turns out to be 0 -- else the caller re-runs it)
        const int gx = (cfg->W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y;
        GFT_CHECK_HIP(gft_zero_async(im.ranges, (size_t)gx * gy * sizeof(uint2), s));
    }
    // stage 2 is queued before R is known; its kernels check R against the buffer themselves
    const uint32_t cap = (uint32_t)binning_instances;
    if (enqueue_stage2(s, cfg, io, g, im, b, binning_instances > 0, hints->max_tile_list, true, cap, pull, hints->whole_lists != 0,
                       by_sched ? mail_dev : nullptr, seq))
        return 1;
    uint32_t host[GFT_CTRL_WORDS];
    if (mailbox_wait(s, mail_host, seq, host)) return 1;
    if (by_sched && ((host[GFT_CTRL_FLAGS] & 2u) || host[GFT_CTRL_TOTAL] > cap)) {
        // A list outgrew the schedule's capacity (or the words were no schedule): the device has binned and rendered nothing
        // (ctrl[TOTAL] = 0xffffffff stopped every kernel behind the scatter pass).  The counted flow, in this call: clear what
        // the front end accumulates into, count (which also leaves the next frame's schedule), stage 2 again.
        const size_t T = (size_t)((cfg->W + GFT_TILE_X - 1) / GFT_TILE_X) * (size_t)((cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y);
        GFT_CHECK_HIP(gft_zero_async(im.ctrl, (GFT_CTRL_WORDS + GFT_TICKET_WORDS + 2 * T + GFT_SUPER_CELLS) * sizeof(uint32_t), s));
        if (mailbox_acquire(&mail_dev, &mail_host, &seq)) return 1;
        mail_host[GFT_CTRL_FLAGS] = host[GFT_CTRL_FLAGS] & 1u;
        // (... or the frame has more instances than the binning buffer holds: the caller's gft_forward_render with a larger one
        // needs the counted lists' tables as well)
        const bool render_again = host[GFT_CTRL_TOTAL] <= cap;
        if (enqueue_count(s, cfg, io, g, im, mail_dev, seq, pull)) return 1;
        if (render_again &&
            enqueue_stage2(s, cfg, io, g, im, b, true, hints->max_tile_list, true, cap, pull, hints->whole_lists != 0))
            return 1;
        if (mailbox_wait(s, mail_host, seq, host)) return 1;
        report->sched_misses = 1;
    }
    if (host[GFT_CTRL_FLAGS] & 1u)
        return gft_fail("Point is filtered although prefiltered is set. This shouldn't happen!");
    report->num_rendered = (int64_t)host[GFT_CTRL_TOTAL];
    report->max_tile_list = (int64_t)host[GFT_CTRL_MAXCNT];
    report->list_entries = pull ? (int64_t)host[GFT_CTRL_ENTRIES] : (int64_t)host[GFT_CTRL_TOTAL];
    report->hinted_tiles = pull ? (int64_t)host[GFT_CTRL_NHINT] : 0;
    // The hint said "no tile list longer than the short-sort limit" and the frame has one: its
    // tiles were rendered unsorted.  Sort them and render again (the contributing-pixel counters
    // are the only accumulated output).
    if (!lazy_sort_enabled() && hints->max_tile_list > 0 && hints->max_tile_list <= GFT_SHORT_LIST_MAX &&
        host[GFT_CTRL_MAXCNT] > GFT_SHORT_LIST_MAX && host[GFT_CTRL_TOTAL] <= cap && binning_instances > 0) {
        GFT_CHECK_HIP(gft_zero_async(io->pixels, (size_t)cfg->P * sizeof(float), s));
        GFT_STAGE(s, cfg, "tile_sort_long", gft_launch_tile_sort_long(s, *cfg, im, b, cap));
        GFT_STAGE(s, cfg, "render_fwd", gft_launch_render_fwd(s, *cfg, *io, g, im, b, true, cap, 0, false));
    }
    return 0;
}

// ---- forward, queued only ------------------------------------------------------------
// Both stages back to back and no read of anything the device computes: the call is pure launch work, so a caller may
// capture it in a HIP graph (torch.cuda.graphs) or simply not stall.  What the blocking flows learn from the mailbox the
// device posts into `status` instead (GFT_STATUS_WORDS words of pinned host or device memory owned by the caller, its first
// GFT_STATUS_STICKY words cleared here on the stream): status[0] = R, status[1] bit 0 = "prefiltered point culled", status[3] = 1 once stage 1 has
// posted.  The stage-2 kernels compare R with the buffer themselves and do nothing when it does not fit: the caller reads
// `status` whenever the stream has passed -- typically in front of its next call -- and re-renders with a larger buffer.
static_assert(GFT_STATUS_WORDS == GFT_CTRL_WORDS, "the status block is a mailbox slot");
extern "C" int gft_forward_enqueue(void* hip_stream, const gft_config* cfg, const gft_forward_io* io,
                                   const gft_forward_hints* hints, uint32_t* status)
{
    if (check_config(cfg)) return 1;
    if (!io || !hints) return gft_fail("gft_forward_enqueue: NULL argument");
    if (cfg->debug) return gft_fail("gft_forward_enqueue: cfg.debug synchronises after every stage; use gft_forward");
    if (check_stage2(io, "gft_forward_enqueue")) return 1;
    const int64_t binning_instances = hints->binning_instances;
    if (binning_instances < 0 || binning_instances > 0xffffffffll) return gft_fail("gft_forward_enqueue: bad instance count");
    hipStream_t s = (hipStream_t)hip_stream;
    // (the words from GFT_STATUS_STICKY on are the owner's to clear: overflow count, largest overflowing R -- a caller that runs
    // ahead of the device finds them whichever frame's posting the first words hold by then)
    if (status) GFT_CHECK_HIP(gft_zero_async(status, GFT_STATUS_STICKY * sizeof(uint32_t), s));
    if (cfg->P == 0) {
        // (no kernel posts anything: R = 0 fits every buffer)
        if (status) hipLaunchKernelGGL(k_post_empty, dim3(1), dim3(1), 0, s, status);
        return gft_forward_render(hip_stream, cfg, io, 0, 0);
    }
    if (check_stage1(cfg, io, "gft_forward_enqueue")) return 1;
    if (binning_instances > 0 && !io->binning) return gft_fail("gft_forward_enqueue: binning buffer is NULL");
    gft_layout L;
    gft_compute_layout(cfg->P, cfg->W, cfg->H, binning_instances, &L);
    GeomView g = gft_geom_view(io->geom, L);
    ImgView im = gft_img_view(io->img, L);
    BinView b = gft_bin_view(io->binning, L);
    const bool pull = pull_enabled(cfg);
    if (enqueue_stage1(s, cfg, io, g, im, status, 1u, pull, true, status ? binning_instances : -1)) return 1;
    if (pull && binning_instances == 0) {
        const int gx = (cfg->W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (cfg->H + GFT_TILE_Y - 1) / GFT_TILE_Y;
        GFT_CHECK_HIP(gft_zero_async(im.ranges, (size_t)gx * gy * sizeof(uint2), s));
    }
    return enqueue_stage2(s, cfg, io, g, im, b, binning_instances > 0, hints->max_tile_list, true, (uint32_t)binning_instances, pull,
                          hints->whole_lists != 0);
}

// ---- backward -----------------------------------------------------------------
extern "C" int gft_backward(void* hip_stream, const gft_config* cfg, const gft_backward_io* io, int64_t num_rendered)
{
    if (check_config(cfg)) return 1;
    if (!io) return gft_fail("gft_backward: io is NULL");
    hipStream_t s = (hipStream_t)hip_stream;
    // k_offset_reduce overwrites both scalars whenever shs_p is given
    if (cfg->P == 0 || io->shs_p == nullptr) {
        if (io->dL_dphase_offset) GFT_CHECK_HIP(gft_zero_async(io->dL_dphase_offset, sizeof(float), s));
        if (io->dL_ddc_offset) GFT_CHECK_HIP(gft_zero_async(io->dL_ddc_offset, sizeof(float), s));
    }
    if (cfg->P == 0) return 0;
    if (!io->means3D || !io->radii || !io->viewmatrix || !io->projmatrix || !io->campos || !io->geom || !io->img ||
        !io->bg || !io->acc || !io->dL_dmeans3D || !io->dL_dmeans2D || !io->dL_dopacity)
        return gft_fail("gft_backward: required pointer is NULL");
    if ((io->dL_dphase_offset == nullptr) != (io->dL_ddc_offset == nullptr))
        return gft_fail("gft_backward: dL_dphase_offset and dL_ddc_offset are wanted together or not at all");
    if (num_rendered < 0 || num_rendered > 0xffffffffll) return gft_fail("gft_backward: bad instance count");
    if (num_rendered > 0 && !io->binning) return gft_fail("gft_backward: binning buffer is NULL");
    if ((io->shs != nullptr) != (cfg->M > 0) || (io->shs_p != nullptr) != (cfg->M_p > 0))
        return gft_fail("M / M_p do not match the presence of shs / shs_p");
    if (io->shs && !io->dL_dsh) return gft_fail("gft_backward: dL_dsh is NULL");
    if (io->shs_p && !io->dL_dsh_p) return gft_fail("gft_backward: dL_dsh_p is NULL");
    if ((io->scales == nullptr || io->rotations == nullptr) == (io->cov3D_precomp == nullptr))
        return gft_fail("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (io->scales && (!io->dL_dscales || !io->dL_drotations)) return gft_fail("gft_backward: dL_dscales/dL_drotations NULL");
    if (cfg->grads_accumulate && !cfg->want_backward)
        return gft_fail("gft_backward: grads_accumulate needs the forward's direction-gradient records (want_backward)");

    gft_layout L;
    gft_compute_layout(cfg->P, cfg->W, cfg->H, num_rendered, &L);
    GeomView g = gft_geom_view(const_cast<void*>(io->geom), L);
    ImgView im = gft_img_view(const_cast<void*>(io->img), L);
    BinView b = gft_bin_view(const_cast<void*>(io->binning), L);

    if ((cfg->grads_zeroed == 2 || cfg->grads_zeroed == 3) && (!io->dirty_rows || !io->pixels || !cfg->want_backward))
        return gft_fail("gft_backward: grads_zeroed = 2 / 3 needs dirty_rows, the forward's pixels and its direction-gradient records");
    if (cfg->grads_zeroed == 1) {
        SideFill* f = side_of_current_device();
        if (!f || !f->pending) return gft_fail("gft_backward: grads_zeroed without a forward that filled them");
        GFT_CHECK_HIP(hipStreamWaitEvent(s, f->filled, 0));
    }
    if (cfg->acc_zeroed < 0 || cfg->acc_zeroed > 2) return gft_fail("gft_backward: cfg.acc_zeroed must be 0, 1 or 2");
    if (!cfg->acc_zeroed) {
        StageTimer t(s, ST_MEMSET);
        GFT_CHECK_HIP(gft_zero_async(io->acc, (size_t)cfg->P * GFT_ACC_STRIDE * sizeof(float), s));
    }
    if (num_rendered > 0 && io->det_partials) {
        // deterministic mode: slots of (entry, quadrant) pairs that store no row must read as zero
        StageTimer t(s, ST_MEMSET);
        GFT_CHECK_HIP(gft_zero_async(io->det_partials, gft_det_partials_bytes(num_rendered, cfg->W, cfg->H), s));
    }
    if (num_rendered > 0) {
        StageTimer t(s, ST_RENDER_BWD);
        GFT_STAGE(s, cfg, "render_bwd", gft_launch_render_bwd(s, *cfg, *io, g, im, b, lazy_sort_enabled(), (uint32_t)num_rendered));
    }
    {
        StageTimer t(s, ST_PRE_BWD);
        GFT_STAGE(s, cfg, "preprocess_bwd", gft_launch_preprocess_bwd(s, *cfg, *io, g));
    }
    if (g_prof.on) { std::lock_guard<std::mutex> lk(g_prof.mu); g_prof.bwd++; }
    return 0;
}

extern "C" int gft_grads_rezero(void* hip_stream, const gft_config* cfg, const gft_backward_io* io)
{
    if (check_config(cfg)) return 1;
    if (!io) return gft_fail("gft_grads_rezero: io is NULL");
    if (cfg->P == 0) return 0;
    if (!io->dirty_rows || !io->dL_dmeans3D || !io->dL_dmeans2D || !io->dL_dopacity)
        return gft_fail("gft_grads_rezero: required pointer is NULL");
    hipStream_t s = (hipStream_t)hip_stream;
    StageTimer t(s, ST_MEMSET);
    const hipError_t e = gft_launch_grads_rezero(s, *cfg, *io);
    if (e != hipSuccess) return gft_fail("grads_rezero: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int gft_mark_visible(void* hip_stream, int32_t P, const float* means3D, const float* viewmatrix,
                                const float* projmatrix, float near_n, float far_n, uint8_t* present)
{
    (void)projmatrix;  // the reference passes it but only tests view-space z (auxiliary.h:167-169)
    if (P < 0) return gft_fail("gft_mark_visible: P < 0");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return gft_fail("gft_mark_visible: NULL pointer");
    hipError_t e = gft_launch_mark_visible((hipStream_t)hip_stream, P, means3D, viewmatrix, near_n, far_n, present);
    if (e != hipSuccess) return gft_fail("mark_visible: %s", hipGetErrorString(e));
    return 0;
}
