// debig_hip.hip -- C-ABI shim (include/debig_hip.h) over the gfx950 kernels.
// Built by debigulator_amd/build.py:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/debig_hip.h"

#include "inflate_kernel.inc"
#include "inflate_mw_kernel.inc"
#include "inflate_split_kernel.inc"
#include "inflate_strand_kernel.inc"
#include "inflate_chunk_kernel.inc"
#include "png_kernel.inc"
#include "png_fused_kernel.inc"
#include "checksum_kernel.inc"

// BTYPE 1 tables, built once per device by a tiny kernel and then only copied into LDS.  Two
// images: the single-wavefront kernel and the multi-wavefront kernels use different direct
// table widths (TabCfg<NW>).
struct FixedTabs {
    uint32_t *one, *mw, *scan; /* scan: the scan kernel's 16-bit format */
};
static FixedTabs g_fixed_tabs[64];
static std::mutex g_init_mutex; /* the lazy per-device initialisations below: the drop-in API allows 10 concurrent thread ids */
// the device the work is launched on: the stream's (a caller may pass a stream of another
// device than the thread's current one), else the current device
static int launch_device(hipStream_t s)
{
    int dev = 0;
    if (s != nullptr && hipStreamGetDevice(s, &dev) == hipSuccess) return dev;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}
struct DeviceGuard { /* allocations and table kernels happen on the launch device */
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (d >= 0 && d != prev) (void)hipSetDevice(d);
    }
    ~DeviceGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};
static const FixedTabs *fixed_tables(hipStream_t s)
{
    const int dev = launch_device(s);
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_init_mutex);
    DeviceGuard guard(dev);
    FixedTabs *f = &g_fixed_tabs[dev];
    if (f->one && f->mw && f->scan) return f;
    uint32_t *a = nullptr, *b = nullptr, *c = nullptr;
    if (hipMalloc(&a, sizeof(decltype(WaveLdsT<1>::t))) != hipSuccess) return nullptr;
    if (hipMalloc(&b, sizeof(decltype(WaveLdsT<2>::t))) != hipSuccess) { (void)hipFree(a); return nullptr; }
    if (hipMalloc(&c, sizeof(decltype(ScanLds::t))) != hipSuccess) { (void)hipFree(a); (void)hipFree(b); return nullptr; }
    hipLaunchKernelGGL(debig_fixed_tables_kernel<1>, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(debig_fixed_tables_kernel<2>, dim3(1), dim3(64), 0, s, b);
    hipLaunchKernelGGL(debig_scan_fixed_tables_kernel, dim3(1), dim3(64), 0, s, c);
    // later launches may use other streams: make the tables globally visible first
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); return nullptr; }
    f->one = a;
    f->mw = b;
    f->scan = c;
    return f;
}

static CkTables *g_ck_tabs[64];
static CkTables *checksum_tables(hipStream_t s)
{
    const int dev = launch_device(s);
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_init_mutex);
    DeviceGuard guard(dev);
    if (g_ck_tabs[dev]) return g_ck_tabs[dev];
    CkTables *p = nullptr;
    if (hipMalloc(&p, sizeof(CkTables)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(debig_checksum_tables_kernel, dim3(1), dim3(CK_THREADS), 0, s, p);
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_ck_tabs[dev] = p;
    return p;
}

extern "C" {

int debig_hip_checksum_batch(const void *d_arena, const debig_span *d_spans, uint32_t *d_out,
                             uint32_t n, uint32_t kind, void *hip_stream)
{
    if (n == 0) return 0;
    DeviceGuard launch_guard(launch_device((hipStream_t)hip_stream)); /* kernels go to the stream's device */
    CkTables *t = checksum_tables((hipStream_t)hip_stream);
    if (!t) return (int)hipErrorOutOfMemory;
    hipLaunchKernelGGL(debig_checksum_kernel, dim3(n), dim3(CK_THREADS), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_arena, d_spans, d_out, n, kind, t);
    return (int)hipGetLastError();
}

int debig_hip_gather(const void *d_src_arena, void *d_dst_arena, const debig_copy *d_copies, uint32_t n,
                     void *hip_stream)
{
    if (n == 0) return 0;
    DeviceGuard launch_guard(launch_device((hipStream_t)hip_stream)); /* kernels go to the stream's device */
    hipLaunchKernelGGL(debig_gather_kernel, dim3(n), dim3(CK_THREADS), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, d_copies, n);
    return (int)hipGetLastError();
}

// Wavefronts per stream when the caller leaves the choice to the library.  The chip is full
// at about 2048 resident decode wavefronts (256 CUs x 4 SIMDs x 2); with fewer streams than
// that, several wavefronts share one stream (debig_inflate_mw_kernel).  Measured crossovers
// (profiles/r01_mw_sweep.txt): n <= 256 -> 8, n <= 512 -> 4, n <= 1024 -> 2, else 1.  The mixed modes are
// never picked here: they only pay when FEW streams of a big batch are large, and stream
// sizes live in device memory -- callers that know them ask for a mixed mode themselves
// (csrc/host/debig_ctx.c: debig_pick_waves).
static uint32_t auto_waves_per_stream(uint32_t n)
{
    // DEBIG_WAVES_PER_STREAM=1|2|4|0x41|0x42 overrides the choice (measurements, bisecting)
    static std::once_flag env_once; /* the drop-in API allows concurrent callers */
    static uint32_t env_val = 0;
    std::call_once(env_once, [] {
        const char *e = getenv("DEBIG_WAVES_PER_STREAM");
        if (e && *e) env_val = (uint32_t)strtoul(e, nullptr, 0);
    });
    if (env_val) return env_val;
    if (n <= 256u) return 8u;
    if (n <= 512u) return 4u;
    if (n <= DEBIG_STRAND_MIN_STREAMS) return 2u;
    // 769..3072 streams: fewer wavefronts than the chip holds (under 3 per SIMD): every wavefront is bound by its
    // own chain of dependent look-ups, and the long-segment scan decodes every symbol once (profiles/r04_width_grid.txt:
    // 1024 streams of 64 KiB / 1 MiB: 2-wide 36 / 36, pair 62 / 63, strands 66 / 80 GB/s on image rows; 2048 streams:
    // pair 110 / 109, strands 116 / 133).  Beyond that the chip is full, both paths are bound by their VALU
    // instructions and the 68-byte scan's shorter tail wins by 3..12 %.
    // up to 2048 streams a second wavefront per stream still finds an empty slot: scan and LZ77 half side by side
    // (debig_strand_pipe_kernel; 1024 x 1 MiB text 166 -> 244 GB/s, 2048: 277 -> 351, 3072: 348 -> 308;
    // single-window streams -- 64 KiB -- have nothing to overlap and run as fast either way)
    if (n <= DEBIG_STRAND_PIPE_MAX_STREAMS) return DEBIG_WAVES_STRAND_PIPE;
    if (n <= DEBIG_STRAND_MAX_STREAMS) return DEBIG_WAVES_STRAND;
    return DEBIG_WAVES_SPLIT;
}

// ---- the scan / LZ77 kernel pair (inflate_split_kernel.inc)
#define SPLIT_GROUP 16384u /* streams per plan + scan + lz + retry group: they share the workspace */
static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// default workspace of callers that bring none (debig_hip_inflate_batch / _ex): one cached
// allocation per device, DEBIG_WORKSPACE_MB (default 1024) MiB
struct DefaultWs {
    void *ptr;
    uint64_t bytes;
    // ONE buffer per device shared by every caller that brings no workspace: a call's group of
    // launches holds `use` while it is enqueued, waits for the event of the group before it and
    // records the event behind itself -- two host threads or two HIP streams never interleave their
    // plan / scan / lz launches on the buffer (their groups run one after the other on the device).
    std::mutex use;
    hipEvent_t done;
    int has_done;
};
static DefaultWs g_default_ws[64];
struct SharedWsUse { /* one call's launches on the default workspace (nullptr: caller-owned workspace) */
    DefaultWs *w;
    hipStream_t s;
    int err; /* hipError_t of the wait in front of the launches (0: the group before this one is ordered before us) */
    bool finished;
    SharedWsUse(DefaultWs *w_, hipStream_t s_) : w(w_), s(s_), err(0), finished(false)
    {
        if (!w) return;
        w->use.lock();
        if (w->has_done) err = (int)hipStreamWaitEvent(s, w->done, 0);
    }
    // records the event the next group waits for; returns its hipError_t.  A stream that is being captured into
    // a graph cannot carry this chain (the event would belong to the capture): callers that capture bring their
    // own workspace (include/debig_hip.h), and a failure here is REPORTED instead of silently dropping the
    // serialisation of the shared buffer
    int finish()
    {
        if (!w || finished) return 0;
        finished = true;
        int rc = 0;
        if (!w->has_done) {
            rc = (int)hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
            if (rc == 0) w->has_done = 1;
        }
        if (w->has_done) {
            const int r2 = (int)hipEventRecord(w->done, s);
            if (rc == 0) rc = r2;
        }
        w->use.unlock();
        return rc;
    }
    ~SharedWsUse() { (void)finish(); }
};
static DefaultWs *default_workspace(hipStream_t s)
{
    const int dev = launch_device(s);
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_init_mutex);
    DeviceGuard guard(dev);
    DefaultWs *w = &g_default_ws[dev];
    if (w->ptr) return w;
    uint64_t mb = 1024;
    const char *e = getenv("DEBIG_WORKSPACE_MB");
    if (e && *e) mb = strtoull(e, nullptr, 0);
    if (mb == 0) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, mb << 20) != hipSuccess) return nullptr;
    w->ptr = p;
    w->bytes = mb << 20;
    return w;
}

// one group of at most SPLIT_GROUP streams: plan, scan, lz, and debig_inflate_kernel for what the
// pair handed back.  Returns 0, a hipError_t, or -1 when the workspace is too small to try.
// what = 1: carve the workspace only (debig_split_plan_kernel), 2: scan + lz + hand-back over a
// workspace already carved for exactly these descriptors, 3: both (one call does everything)
// workgroups of debig_scanlz_queue_kernel the current device holds at once (per device, asked once)
static uint32_t scanlz_resident_workgroups()
{
    static uint32_t cap[64];
    static std::mutex cap_mutex; /* concurrent callers (the drop-in API allows 10 thread ids) */
    const char *e = getenv("DEBIG_SPLIT_WORKGROUPS"); /* tests and experiments: read at every call */
    const uint32_t forced = e && *e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
    if (forced) return forced;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 4096u;
    std::lock_guard<std::mutex> lock(cap_mutex);
    if (cap[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, debig_scanlz_queue_kernel, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cap[dev] = (uint32_t)per_cu * (uint32_t)cus;
    }
    return cap[dev];
}

// streams up to which the strand pipeline runs with the large LZ77 tile: what the device holds at once of that instantiation
// (DEBIG_PIPE_BIG_TILE_STREAMS overrides: tests, measurements; 0 = never)
static uint32_t pipe_big_tile_streams()
{
    const char *e = getenv("DEBIG_PIPE_BIG_TILE_STREAMS");
    if (e && *e) return (uint32_t)strtoul(e, nullptr, 0);
    static uint32_t cap[64];
    static std::mutex cap_mutex;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1280u;
    std::lock_guard<std::mutex> lock(cap_mutex);
    if (cap[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, debig_strand_pipe_kernel<LzLdsBig>, 128, 0) != hipSuccess || per_cu <= 0) per_cu = 5;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cap[dev] = (uint32_t)per_cu * (uint32_t)cus;
    }
    return cap[dev];
}

static int launch_split_group(hipStream_t s, const void *d_in, void *d_out, const debig_stream *d_streams,
                              debig_result *d_results, uint32_t n, const FixedTabs *tabs, void *ws, uint64_t ws_bytes,
                              int what = 3, int queued = 0)
{
    const uint64_t slots_bytes = align_up((uint64_t)n * sizeof(debig_ws_slot) + 4u * SPLIT_QUEUE_WORDS + 4u * (uint64_t)n, 256); /* + the work queue + the dispatch order */
    if (ws_bytes < slots_bytes + (uint64_t)n * 1024u) return -1;
    const uint64_t rest = ws_bytes - slots_bytes;
    const uint64_t total_recs = rest / 16u / sizeof(debig_ws_rec);
    const uint64_t recs_bytes = align_up(total_recs * sizeof(debig_ws_rec), 256);
    const uint64_t total_rows = (rest - recs_bytes) / 256u;
    debig_ws_slot *slots = (debig_ws_slot *)ws;
    debig_ws_rec *recs = (debig_ws_rec *)((uint8_t *)ws + slots_bytes);
    uint32_t *rows = (uint32_t *)((uint8_t *)ws + slots_bytes + recs_bytes);
    if (what & 1)
        hipLaunchKernelGGL(debig_split_plan_kernel, dim3(1), dim3(1024), 0, s, d_streams, n, slots, total_rows, total_recs);
    if (!(what & 2)) return (int)hipGetLastError();
#ifndef DEBIG_SPLIT_FUSED
#define DEBIG_SPLIT_FUSED 1
#endif
#if DEBIG_SPLIT_FUSED
    const uint32_t cap = queued == 1 ? scanlz_resident_workgroups() : 0u;
    if (queued == 3) /* DEBIG_WAVES_STRAND_PIPE: the same two halves on two wavefronts of a workgroup, record by record */
    {   /* a 12 KB LZ77 tile while the device holds every stream at once with it (29 KB of LDS: 5 workgroups per CU) */
        if (n <= pipe_big_tile_streams())
            hipLaunchKernelGGL(debig_strand_pipe_kernel<LzLdsBig>, dim3(n), dim3(128), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out, d_streams, n,
                               tabs->scan, slots, recs, rows, d_results);
        else
            hipLaunchKernelGGL(debig_strand_pipe_kernel<LzLds>, dim3(n), dim3(128), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out, d_streams, n,
                               tabs->scan, slots, recs, rows, d_results);
    }
    else if (queued == 2) /* DEBIG_WAVES_STRAND: the long-segment scan in front of the same LZ77 half */
        hipLaunchKernelGGL(debig_strand_kernel, dim3(n), dim3(64), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out, d_streams, n,
                           tabs->scan, slots, recs, rows, d_results);
    else if (queued && n > cap) /* persistent workgroups: as many as the device holds at once, streams from a queue */
        hipLaunchKernelGGL(debig_scanlz_queue_kernel, dim3(cap), dim3(64), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out,
                           d_streams, n, tabs->scan, slots, recs, rows, d_results);
    else
        hipLaunchKernelGGL(debig_scanlz_kernel, dim3(n), dim3(64), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out, d_streams, n,
                           tabs->scan, slots, recs, rows, d_results);
#else
    hipLaunchKernelGGL(debig_scan_kernel, dim3(n), dim3(64), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out, d_streams, n,
                       tabs->scan, slots, recs, rows, d_results);
    hipLaunchKernelGGL(debig_lz_kernel, dim3(n), dim3(64), 0, s, (uint8_t *)d_out, d_streams, d_results, n,
                       (const debig_ws_slot *)slots, (const debig_ws_rec *)recs, (const uint32_t *)rows);
#endif
    hipLaunchKernelGGL(debig_inflate_kernel, dim3(n), dim3(64), 0, s, (const uint8_t *)d_in, (uint8_t *)d_out, d_streams,
                       d_results, n, tabs->one, DEBIG_CLASS_RETRY);
    return (int)hipGetLastError();
}

static int launch_split(hipStream_t s, const void *d_in, void *d_out, const debig_stream *d_streams,
                        debig_result *d_results, uint32_t n, const FixedTabs *tabs, void *ws, uint64_t ws_bytes, int queued = 0)
{
    for (uint32_t first = 0; first < n; first += SPLIT_GROUP) {
        const uint32_t cnt = n - first < SPLIT_GROUP ? n - first : SPLIT_GROUP;
        int rc = launch_split_group(s, d_in, d_out, d_streams + first, d_results + first, cnt, tabs, ws, ws_bytes, 3, queued);
        if (rc) return rc;
    }
    return 0;
}

// ---- the chunk-parallel path for large streams (inflate_chunk_kernel.inc): fourteen small launches,
// no host synchronisation; sizes live in device memory, so every grid is laid out for the number of
// chunk tasks the workspace could hold and surplus workgroups leave at once.  Returns 0, a
// hipError_t, or -1 when the workspace is too small to try.
static int launch_inflate(uint32_t width, uint32_t cls, hipStream_t s, const void *d_in, void *d_out,
                          const debig_stream *d_streams, debig_result *d_results, uint32_t n, const FixedTabs *tabs);
static uint32_t chunk_bytes_override()
{
    // DEBIG_CHUNK_BYTES: compressed bytes per chunk task (default: by batch size, 32..256 KiB).  Read
    // at every call: tests cut small streams into many tasks with it.
    const char *e = getenv("DEBIG_CHUNK_BYTES");
    return e && *e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
}
static int launch_chunked(hipStream_t s, const void *d_in, void *d_out, const debig_stream *d_streams,
                          debig_result *d_results, uint32_t n, const FixedTabs *tabs, void *ws_, uint64_t ws_bytes)
{
    const uint32_t mt = ck_max_tasks(ws_bytes, n);
    if (mt == 0) return -1;
    uint8_t *ws = (uint8_t *)ws_;
    const uint8_t *in = (const uint8_t *)d_in;
    uint8_t *out = (uint8_t *)d_out;
    // what is handed back: one workgroup per stream, as wide as the batch size allows
    const uint32_t retry_w = n <= 256u ? 8u : n <= 512u ? 4u : n <= 1024u ? 2u : 1u;
    hipLaunchKernelGGL(debig_ck_plan_kernel, dim3(1), dim3(1024), 0, s, d_streams, n, ws, ws_bytes, mt, chunk_bytes_override());
    hipLaunchKernelGGL(debig_ck_find_kernel, dim3(mt), dim3(64), 0, s, in, d_streams, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_bounds_kernel, dim3((n + 63u) / 64u), dim3(64), 0, s, d_streams, n, ws, retry_w > 1u ? 1u : 0u);
    hipLaunchKernelGGL(debig_ck_carve_kernel, dim3(1), dim3(1024), 0, s, d_streams, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_scan_kernel, dim3(mt), dim3(64), 0, s, in, d_streams, n, (const uint32_t *)tabs->scan, ws, mt, 0u);
    hipLaunchKernelGGL(debig_ck_repair_kernel, dim3((n + 63u) / 64u), dim3(64), 0, s, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_scan_kernel, dim3(mt), dim3(64), 0, s, in, d_streams, n, (const uint32_t *)tabs->scan, ws, mt, 1u);
    hipLaunchKernelGGL(debig_ck_chain_kernel, dim3((n + 63u) / 64u), dim3(64), 0, s, d_streams, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_place_kernel, dim3(1), dim3(1024), 0, s, n, ws);
    hipLaunchKernelGGL(debig_ck_lz_kernel, dim3(mt), dim3(64), 0, s, in, out, d_streams, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_window_kernel, dim3(n), dim3(CK_WIN_THREADS), 0, s, (const uint8_t *)out, d_streams, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_translate_kernel, dim3(mt * CK_TR_PARTS), dim3(CK_TR_THREADS), 0, s, out, d_streams, n, ws, mt);
    hipLaunchKernelGGL(debig_ck_finish_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, n, (const uint8_t *)ws, d_results);
    return launch_inflate(retry_w, DEBIG_CLASS_RETRY, s, d_in, d_out, d_streams, d_results, n, tabs);
}

static int launch_inflate(uint32_t width, uint32_t cls, hipStream_t s, const void *d_in, void *d_out,
                          const debig_stream *d_streams, debig_result *d_results, uint32_t n, const FixedTabs *tabs)
{
    const uint32_t grid = n; /* one workgroup per stream: the hardware scheduler balances lengths */
    if (width == 1)
        hipLaunchKernelGGL(debig_inflate_kernel, dim3(grid), dim3(64), 0, s, (const uint8_t *)d_in,
                           (uint8_t *)d_out, d_streams, d_results, n, tabs->one, cls);
    else if (width == 2)
        hipLaunchKernelGGL(debig_inflate_mw_kernel<2>, dim3(grid), dim3(128), 0, s, (const uint8_t *)d_in,
                           (uint8_t *)d_out, d_streams, d_results, n, tabs->mw, cls);
    else if (width == 4)
        hipLaunchKernelGGL(debig_inflate_mw_kernel<4>, dim3(grid), dim3(256), 0, s, (const uint8_t *)d_in,
                           (uint8_t *)d_out, d_streams, d_results, n, tabs->mw, cls);
    else
        hipLaunchKernelGGL(debig_inflate_mw_kernel<8>, dim3(grid), dim3(512), 0, s, (const uint8_t *)d_in,
                           (uint8_t *)d_out, d_streams, d_results, n, tabs->mw, cls);
    return (int)hipGetLastError();
}

// side stream + fork/join events for the two concurrent launches of a mixed-width batch
struct SideLane {
    hipStream_t stream;
    hipEvent_t fork, join;
    int ready;
};
static SideLane g_side[64];
static SideLane *side_lane(hipStream_t s)
{
    const int dev = launch_device(s);
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_init_mutex);
    DeviceGuard guard(dev);
    SideLane *l = &g_side[dev];
    if (l->ready) return l;
    if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&l->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&l->join, hipEventDisableTiming) != hipSuccess) return nullptr;
    l->ready = 1;
    return l;
}

uint64_t debig_hip_inflate_workspace_bytes(uint64_t total_in_bytes, uint32_t n)
{
    // token rows: about 4-5 x the compressed bytes for text-like data (one 256-byte row per symbol
    // index of a 64-lane window) + per-stream slack for partial windows; records and slots on top.
    // 12 x since round 4 (was 9): the long-segment scan (DEBIG_WAVES_STRAND) stores a 64-byte unit per lane and
    // phase until its slowest lane is done, and literal-heavy data with short codes (noisy image rows: 1.4
    // symbols per compressed byte) needed 10-11 x -- with 9 x a quarter of such streams went to the one-kernel
    // path (4096 x 64 KiB: 3.2 ms instead of 1.7)
    const uint32_t group = n < SPLIT_GROUP ? n : SPLIT_GROUP;
    const uint64_t per_group_in = n ? (total_in_bytes + n - 1) / n * group : 0; /* average streams */
    return align_up((uint64_t)group * (sizeof(debig_ws_slot) + 24576u) + per_group_in * 12u, 4096);
}
uint64_t debig_hip_inflate_workspace_bytes_io(uint64_t total_in_bytes, uint64_t total_out_cap, uint32_t n)
{
    // the plan step's second weight term (PlanStreamWeight: min(out_cap / 64, 4 in_len) per stream), 12 x as the first
    const uint32_t group = n < SPLIT_GROUP ? n : SPLIT_GROUP;
    const uint64_t out_term = total_out_cap / 64u < 4u * total_in_bytes ? total_out_cap / 64u : 4u * total_in_bytes;
    const uint64_t per_group = n ? (out_term + n - 1) / n * group : 0;
    return debig_hip_inflate_workspace_bytes(total_in_bytes, n) + align_up(per_group * 12u, 4096);
}

uint64_t debig_hip_inflate_chunked_workspace_bytes(uint64_t total_in_bytes, uint64_t total_out_bytes, uint32_t n)
{
    // tables + tokens (about 9 x the compressed bytes) + two planes of the output + per chunk task
    // (32 KiB of input at the smallest chunk size) a window, two synthetic histories and token slack
    const uint64_t tasks = total_in_bytes / CK_MIN_CHUNK + n;
    return align_up((uint64_t)n * 4224u + tasks * (128u + 24576u + 98304u) + total_in_bytes * 10u + total_out_bytes * 2u +
                        (total_out_bytes >> 6) + (1u << 20), 4096);
}

int debig_hip_inflate_batch_ws(const void *d_in, void *d_out, const debig_stream *d_streams,
                               debig_result *d_results, uint32_t n, uint32_t waves_per_stream,
                               void *d_workspace, uint64_t workspace_bytes, void *hip_stream)
{
    if (n == 0) return 0;
    DeviceGuard launch_guard(launch_device((hipStream_t)hip_stream)); /* kernels go to the stream's device */
    if (waves_per_stream == 0) waves_per_stream = auto_waves_per_stream(n);
    const int mixed = waves_per_stream == DEBIG_WAVES_LARGE4_SMALL1 || waves_per_stream == DEBIG_WAVES_LARGE4_SMALL2;
    if (!mixed && waves_per_stream != 1 && waves_per_stream != 2 && waves_per_stream != 4 && waves_per_stream != 8 &&
        waves_per_stream != DEBIG_WAVES_SPLIT && waves_per_stream != DEBIG_WAVES_SPLIT_QUEUED &&
        waves_per_stream != DEBIG_WAVES_STRAND && waves_per_stream != DEBIG_WAVES_STRAND_PIPE &&
        waves_per_stream != DEBIG_WAVES_CHUNKED)
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)hip_stream;
    const FixedTabs *ft = fixed_tables(s);
    if (!ft) return (int)hipErrorOutOfMemory;
    if (waves_per_stream == DEBIG_WAVES_CHUNKED) {
        DefaultWs *shared = nullptr;
        if (!d_workspace) {
            shared = default_workspace(s);
            if (shared) { d_workspace = shared->ptr; workspace_bytes = shared->bytes; }
        }
        int rc = -1;
        if (d_workspace) {
            SharedWsUse hold(shared, s);
            if (hold.err) return hold.err;
            rc = launch_chunked(s, d_in, d_out, d_streams, d_results, n, ft, d_workspace, workspace_bytes);
            const int frc = hold.finish();
            if (rc == 0 && frc) return frc;
        }
        if (rc >= 0) return rc;
        d_workspace = nullptr;
        waves_per_stream = n <= 256u ? 8u : n <= 512u ? 4u : n <= 1024u ? 2u : 1u; /* no usable workspace */
    }
    if (waves_per_stream == DEBIG_WAVES_SPLIT || waves_per_stream == DEBIG_WAVES_SPLIT_QUEUED ||
        waves_per_stream == DEBIG_WAVES_STRAND || waves_per_stream == DEBIG_WAVES_STRAND_PIPE) {
        DefaultWs *shared = nullptr;
        if (!d_workspace) {
            shared = default_workspace(s);
            if (shared) { d_workspace = shared->ptr; workspace_bytes = shared->bytes; }
        }
        int rc = -1;
        if (d_workspace) {
            SharedWsUse hold(shared, s);
            if (hold.err) return hold.err;
            rc = launch_split(s, d_in, d_out, d_streams, d_results, n, ft, d_workspace, workspace_bytes,
                              waves_per_stream == DEBIG_WAVES_SPLIT_QUEUED ? 1 : waves_per_stream == DEBIG_WAVES_STRAND ? 2 :
                              waves_per_stream == DEBIG_WAVES_STRAND_PIPE ? 3 : 0);
            const int frc = hold.finish();
            if (rc == 0 && frc) return frc;
        }
        if (rc >= 0) return rc;
        waves_per_stream = 1; /* no usable workspace: the one-kernel path */
    }
    if (!mixed) return launch_inflate(waves_per_stream, DEBIG_CLASS_ALL, s, d_in, d_out, d_streams, d_results, n, ft);

    // large streams 4-wide on the side stream, small ones beside them on the caller's stream;
    // the caller's stream continues only when both are done
    SideLane *l = side_lane(s);
    if (!l) return (int)hipErrorOutOfMemory;
    hipError_t e;
    if ((e = hipEventRecord(l->fork, s)) != hipSuccess) return (int)e;
    if ((e = hipStreamWaitEvent(l->stream, l->fork, 0)) != hipSuccess) return (int)e;
    int rc = launch_inflate(4, DEBIG_CLASS_LARGE, l->stream, d_in, d_out, d_streams, d_results, n, ft);
    if (rc) return rc;
    if ((e = hipEventRecord(l->join, l->stream)) != hipSuccess) return (int)e;
    rc = launch_inflate(waves_per_stream & 15u, DEBIG_CLASS_SMALL, s, d_in, d_out, d_streams, d_results, n, ft);
    if (rc) return rc;
    if ((e = hipStreamWaitEvent(s, l->join, 0)) != hipSuccess) return (int)e;
    return 0;
}

static DefaultWs *png_gsync(hipStream_t s, uint32_t n);
// ---- plan once, execute many times (DEBIG_WAVES_SPLIT, one group of streams)
int debig_hip_inflate_plan_ws(const debig_stream *d_streams, uint32_t n, void *d_workspace, uint64_t workspace_bytes,
                              void *hip_stream)
{
    if (n == 0) return 0;
    if (n > SPLIT_GROUP || !d_workspace) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)hip_stream;
    DeviceGuard launch_guard(launch_device(s));
    int rc = launch_split_group(s, nullptr, nullptr, d_streams, nullptr, n, nullptr, d_workspace, workspace_bytes, 1);
    return rc < 0 ? (int)hipErrorInvalidValue : rc;
}

int debig_hip_inflate_planned_ws_ex(const void *d_in, void *d_out, const debig_stream *d_streams, debig_result *d_results,
                                    uint32_t n, uint32_t waves_per_stream, void *d_workspace, uint64_t workspace_bytes,
                                    void *hip_stream)
{
    if (n == 0) return 0;
    if (n > SPLIT_GROUP || !d_workspace) return (int)hipErrorInvalidValue;
    if (waves_per_stream != DEBIG_WAVES_SPLIT && waves_per_stream != DEBIG_WAVES_SPLIT_QUEUED &&
        waves_per_stream != DEBIG_WAVES_STRAND && waves_per_stream != DEBIG_WAVES_STRAND_PIPE)
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)hip_stream;
    DeviceGuard launch_guard(launch_device(s));
    const FixedTabs *ft = fixed_tables(s);
    if (!ft) return (int)hipErrorOutOfMemory;
    int rc = launch_split_group(s, d_in, d_out, d_streams, d_results, n, ft, d_workspace, workspace_bytes, 2,
                                waves_per_stream == DEBIG_WAVES_SPLIT_QUEUED ? 1 : waves_per_stream == DEBIG_WAVES_STRAND ? 2 :
                              waves_per_stream == DEBIG_WAVES_STRAND_PIPE ? 3 : 0);
    return rc < 0 ? (int)hipErrorInvalidValue : rc;
}

int debig_hip_inflate_planned_ws(const void *d_in, void *d_out, const debig_stream *d_streams, debig_result *d_results,
                                 uint32_t n, void *d_workspace, uint64_t workspace_bytes, void *hip_stream)
{
    return debig_hip_inflate_planned_ws_ex(d_in, d_out, d_streams, d_results, n, DEBIG_WAVES_SPLIT, d_workspace, workspace_bytes,
                                           hip_stream);
}

int debig_hip_init(void *hip_stream)
{
    // everything a first call would allocate or build lazily on the stream's device: the BTYPE 1
    // table images (three allocations, three tiny kernels, one synchronisation).  After it a call
    // with a caller-owned workspace enqueues kernels and nothing else: capturable into a hipGraph.
    hipStream_t s = (hipStream_t)hip_stream;
    DeviceGuard launch_guard(launch_device(s));
    if (!fixed_tables(s)) return (int)hipErrorOutOfMemory;
    (void)png_gsync(s, 1); /* the de-filter's progress counters (few-images mode) */
    return 0;
}

int debig_hip_inflate_batch_ex(const void *d_in, void *d_out, const debig_stream *d_streams,
                               debig_result *d_results, uint32_t n, uint32_t waves_per_stream,
                               void *hip_stream)
{
    return debig_hip_inflate_batch_ws(d_in, d_out, d_streams, d_results, n, waves_per_stream, nullptr, 0, hip_stream);
}

int debig_hip_inflate_batch(const void *d_in, void *d_out, const debig_stream *d_streams,
                            debig_result *d_results, uint32_t n, void *hip_stream)
{
    return debig_hip_inflate_batch_ex(d_in, d_out, d_streams, d_results, n, 0, hip_stream);
}

// progress counters of the de-filter's several-workgroups-per-image mode: one cached allocation per
// device (256 images at most take this path)
static DefaultWs g_png_gsync[64]; /* shared by the callers of a device: uses are serialised (SharedWsUse) */
static DefaultWs *png_gsync(hipStream_t s, uint32_t n)
{
    const int dev = launch_device(s);
    if (dev < 0 || dev >= 64 || n > 256u) return nullptr;
    std::lock_guard<std::mutex> lock(g_init_mutex);
    DeviceGuard guard(dev);
    DefaultWs *w = &g_png_gsync[dev];
    if (!w->ptr) {
        void *p = nullptr;
        if (hipMalloc(&p, 256u * PNG_GSYNC_STRIDE * sizeof(uint32_t)) != hipSuccess) return nullptr;
        w->ptr = p;
        w->bytes = 256u * PNG_GSYNC_STRIDE * sizeof(uint32_t);
    }
    return w;
}

// Wavefronts per image for the de-filter (png_kernel.inc): few images -> several wavefronts each
// (bands pipelined through the workgroup), many images -> one each.  DEBIG_DEFILTER_WAVES=1|2|4|8
// overrides (measurements).
static uint32_t defilter_waves(uint32_t n)
{
    static std::once_flag env_once;
    static uint32_t env_val = 0;
    std::call_once(env_once, [] {
        const char *e = getenv("DEBIG_DEFILTER_WAVES");
        if (e && *e) env_val = (uint32_t)strtoul(e, nullptr, 0);
    });
    if (env_val == 1 || env_val == 2 || env_val == 4 || env_val == 8 || env_val == 16) return env_val;
    if (n <= 64u) return 16u; /* 32 images of 8192 x 8192: 52.5 -> 46.2 ms; from 128 images on 8 is as good */
    if (n <= 256u) return 8u;
    if (n <= 512u) return 4u;
    if (n <= 1024u) return 2u;
    return 1u;
}

// workgroups of the several-workgroups-per-image de-filter (wpw wavefronts each) that device `dev` holds at once
static uint32_t mwg_resident_workgroups(int dev, uint32_t wpw)
{
    const char *e = getenv("DEBIG_DEFILTER_RESIDENT"); /* tests: read at every call */
    if (e && *e) return (uint32_t)strtoul(e, nullptr, 0);
    if (dev < 0 || dev >= 64) return 0u;
    static std::mutex m;
    static uint32_t cap[64][3]; /* wpw = 2, 4, 8 */
    const int k = wpw == 2u ? 0 : wpw == 4u ? 1 : 2;
    std::lock_guard<std::mutex> lock(m);
    if (cap[dev][k] == 0) {
        DeviceGuard guard(dev);
        int per_cu = 0, cus = 0;
        hipError_t rc = k == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, debig_png_defilter_kernel<2, 16, true>, 128, 0)
                        : k == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, debig_png_defilter_kernel<4, 16, true>, 256, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, debig_png_defilter_kernel<8, 16, true>, 512, 0);
        if (rc != hipSuccess || per_cu <= 0) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 0;
        // one workgroup per CU is what the measured shapes assume (and what 8 wavefronts' LDS allows); a device
        // that would hold two is still asked for one
        cap[dev][k] = per_cu > 0 ? (uint32_t)cus : 1u; /* never 0: asked once */
        if (per_cu == 0 || cus == 0) cap[dev][k] = 1u;
    }
    return cap[dev][k];
}

int debig_hip_png_defilter_batch(const void *d_streams_arena, void *d_rgba_arena,
                                 const debig_png_image *d_images, debig_png_result *d_results,
                                 uint32_t n, void *hip_stream)
{
    if (n == 0) return 0;
    DeviceGuard launch_guard(launch_device((hipStream_t)hip_stream)); /* kernels go to the stream's device */
    const uint32_t nwd = defilter_waves(n);
    hipStream_t s = (hipStream_t)hip_stream;
    // few images: an image on several workgroups (CUs).  G workgroups of 4 wavefronts per image; all of
    // them must be resident together (87 KB of LDS: one per CU), so n * G stays within the CU count.
    // DEBIG_DEFILTER_WGS = 1 turns it off, 2 / 4 / 8 force G (measurements).
    {
        static std::once_flag once;
        static uint32_t env_g = 0, env_w = 0, env_blk = 0, env_px = 1;
        std::call_once(once, [] {
            const char *ep = getenv("DEBIG_DEFILTER_PXSKEW"); /* 0: the group-skew step (measurements) */
            if (ep && *ep) env_px = (uint32_t)strtoul(ep, nullptr, 0);
            const char *eb = getenv("DEBIG_DEFILTER_BLK");
            if (eb && *eb) env_blk = (uint32_t)strtoul(eb, nullptr, 0);
            const char *e = getenv("DEBIG_DEFILTER_WGS");
            if (e && *e) env_g = (uint32_t)strtoul(e, nullptr, 0);
            e = getenv("DEBIG_DEFILTER_WG_WAVES"); /* 2 | 4 | 8 wavefronts per workgroup in that mode (measurements) */
            if (e && *e) env_w = (uint32_t)strtoul(e, nullptr, 0);
        });
        // measured, 8192 x 8192 images (profiles/r03_defilter_wgs.txt): 32 images 43.6 -> 17.0 ms with 8 x 4
        // wavefronts, 64 images 46.6 -> 25.4 with 4 x 4, 128 images 54.6 -> 36.2 with 2 x 8
        // with the pixel-skew step (round 4, profiles/r04_defilter_pixel_skew.txt: a band follows the one above 32 macro-steps
        // behind, the image's wavefronts run back to back) more wavefronts per image pay again: 16 images 12.2 -> 8.4 ms with
        // 16 x 4, 32 images 12.9 -> 12.1 with 8 x 8, 64 images 22.2 -> 19.0 with 4 x 8
        // (and with the predictor of a row selected by masks in the 4-wavefront kernel: 16 images 7.4 ms, 32 images 11.2 ms
        // with 8 x 4 -- two wavefronts per SIMD prefer the branches and lose to it: 13.0)
        uint32_t g = n <= 16u ? 16u : n <= 32u ? 8u : n <= 64u ? 4u : n <= 128u ? 2u : 1u;
        uint32_t wpw = n <= 32u ? 4u : 8u;
        if (env_g) g = env_g;
        if (env_w == 2u || env_w == 4u || env_w == 8u) wpw = env_w;
        if (g > 16u) g = 16u;
        if (g * wpw > PNG_GSYNC_STRIDE - 16u) g = (PNG_GSYNC_STRIDE - 16u) / wpw; /* progress words per image */
        // all n * G workgroups must be resident TOGETHER (the bands of an image are a ring of dependencies): the limit
        // is what THIS device holds of THIS instantiation (its CU count x the runtime's occupancy answer), not a
        // constant -- a partitioned device, a CU mask or a smaller part holds fewer.  G is halved until the grid
        // fits; when even G = 2 does not, the one-workgroup-per-image launch below takes the batch.
        // DEBIG_DEFILTER_RESIDENT overrides the limit (tests).
        const uint32_t resident = mwg_resident_workgroups(launch_device(s), wpw);
        while (g > 1u && (uint64_t)n * g > resident) g >>= 1;
        DefaultWs *gw = g > 1u ? png_gsync(s, n) : nullptr;
        if (gw) {
            SharedWsUse hold(gw, s);
            if (hold.err) return hold.err;
            uint32_t *gsync = (uint32_t *)gw->ptr;
            hipError_t e = hipMemsetAsync(gsync, 0, (size_t)n * PNG_GSYNC_STRIDE * sizeof(uint32_t), s);
            if (e != hipSuccess) return (int)e;
            if (wpw == 8u && env_px != 0u)
                hipLaunchKernelGGL((debig_png_defilter_kernel<8, 16, true, true>), dim3(n * g), dim3(512), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            else if (wpw == 8u)
                hipLaunchKernelGGL((debig_png_defilter_kernel<8, 16, true>), dim3(n * g), dim3(512), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            else if (wpw == 2u)
                hipLaunchKernelGGL((debig_png_defilter_kernel<2, 16, true>), dim3(n * g), dim3(128), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            else if (env_blk == 32u) /* measurements: macro-steps per load phase (the share of the load phases in a band) */
                hipLaunchKernelGGL((debig_png_defilter_kernel<4, 32, true>), dim3(n * g), dim3(256), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            else if (env_blk == 8u)
                hipLaunchKernelGGL((debig_png_defilter_kernel<4, 8, true>), dim3(n * g), dim3(256), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            else if (env_px != 0u) /* lanes one PIXEL behind the row above (png_kernel.inc "PX"): the band below follows 32 macro-steps behind, not 79 */
                hipLaunchKernelGGL((debig_png_defilter_kernel<4, 16, true, true>), dim3(n * g), dim3(256), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            else
                hipLaunchKernelGGL((debig_png_defilter_kernel<4, 16, true>), dim3(n * g), dim3(256), 0, s,
                                   (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n, g, gsync, 0u);
            // Residency is the runtime's promise, not a guarantee (another stream's kernels may hold LDS when this
            // grid starts): a workgroup that waited in vain gave its image up as REDO after some tens of milliseconds,
            // and this launch -- one workgroup per image, nothing to wait for across workgroups -- decodes exactly
            // those images again.  Normally every workgroup finds nothing to do and leaves at once (a few us).
            if (n <= 64u)
                hipLaunchKernelGGL((debig_png_defilter_kernel<16, 6>), dim3(n), dim3(1024), 0, s, (const uint8_t *)d_streams_arena,
                                   (uint8_t *)d_rgba_arena, d_images, d_results, n, 1u, (uint32_t *)nullptr, 1u);
            else
                hipLaunchKernelGGL((debig_png_defilter_kernel<8>), dim3(n), dim3(512), 0, s, (const uint8_t *)d_streams_arena,
                                   (uint8_t *)d_rgba_arena, d_images, d_results, n, 1u, (uint32_t *)nullptr, 1u);
            hipLaunchKernelGGL(debig_png_p3_kernel, dim3(n), dim3(PNG_P3_THREADS), 0, s,
                               (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images, d_results, n);
            const int lrc = (int)hipGetLastError();
            const int frc = hold.finish();
            return lrc ? lrc : frc;
        }
    }
#define DEFILTER_LAUNCH(W)                                                                              \
    hipLaunchKernelGGL(debig_png_defilter_kernel<W>, dim3(n), dim3(64 * W), 0, s, (const uint8_t *)d_streams_arena, \
                       (uint8_t *)d_rgba_arena, d_images, d_results, n)
    if (nwd == 16)
        hipLaunchKernelGGL((debig_png_defilter_kernel<16, 6>), dim3(n), dim3(1024), 0, s, (const uint8_t *)d_streams_arena,
                           (uint8_t *)d_rgba_arena, d_images, d_results, n);
    else if (nwd == 8) DEFILTER_LAUNCH(8);
    else if (nwd == 4) DEFILTER_LAUNCH(4);
    else if (nwd == 2) DEFILTER_LAUNCH(2);
    else DEFILTER_LAUNCH(1);
#undef DEFILTER_LAUNCH
    // colour type 2 images that ask for the reference's exact (P3) output; a no-op otherwise
    hipLaunchKernelGGL(debig_png_p3_kernel, dim3(n), dim3(PNG_P3_THREADS), 0, s,
                       (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images,
                       d_results, n);
    return (int)hipGetLastError();
}

// inflate -> de-filter in one kernel (png_fused_kernel.inc).  Per group of at most SPLIT_GROUP
// images: plan, the fused kernel, debig_inflate_kernel for the streams the scan handed back; then, over the whole
// batch, the one-workgroup de-filter for the images of those streams (PNG_ROW_REDO) and the P3 kernel.
#define PNG_FUSED_NWD 2
#define PNG_FUSED_BLK 6
int debig_hip_png_decode_fused_batch(const void *d_in, void *d_streams_arena, const debig_stream *d_streams,
                                     debig_result *d_results, void *d_rgba_arena, const debig_png_image *d_images,
                                     debig_png_result *d_png_results, uint32_t n, void *d_workspace,
                                     uint64_t workspace_bytes, void *hip_stream)
{
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)hip_stream;
    DeviceGuard launch_guard(launch_device(s));
    const FixedTabs *ft = fixed_tables(s);
    if (!ft) return (int)hipErrorOutOfMemory;
    DefaultWs *shared = nullptr;
    if (!d_workspace) {
        shared = default_workspace(s);
        if (!shared) return (int)hipErrorOutOfMemory;
        d_workspace = shared->ptr;
        workspace_bytes = shared->bytes;
    }
    // register budget of the kernel: DEBIG_FUSED_WPE = 2 | 3 (measurements); default: three workgroups per CU once the
    // batch needs them
    const char *we = getenv("DEBIG_FUSED_WPE");
    const int wpe = we && *we ? (int)strtol(we, nullptr, 0) : (n > 512u ? 3 : 2);
    const char *fe = getenv("DEBIG_FUSED_FLAGS"); /* measurements: 0 = fixed roles, default priority; & 2: no hand-back launches */
    const uint32_t kflags = fe && *fe ? (uint32_t)strtoul(fe, nullptr, 0) : 5u; /* 1: roles rotate, 4: raised wavefront priority */
    int rc = 0, no_ws = 0;
    {
        SharedWsUse hold(shared, s);
        if (hold.err) return hold.err;
        for (uint32_t first = 0; first < n && rc == 0; first += SPLIT_GROUP) {
            const uint32_t cnt = n - first < SPLIT_GROUP ? n - first : SPLIT_GROUP;
            const uint64_t slots_bytes = align_up((uint64_t)cnt * sizeof(debig_ws_slot) + 4u * SPLIT_QUEUE_WORDS + 4u * (uint64_t)cnt, 256);
            if (workspace_bytes < slots_bytes + (uint64_t)cnt * 1024u) {
                /* no usable workspace (as debig_hip_inflate_batch_ws): a workgroup per stream, and every image of the group
                 * goes through the one-workgroup de-filter below */
                const hipError_t me = hipMemsetAsync(d_png_results + first, 0, (size_t)cnt * sizeof(debig_png_result), s);
                if (me != hipSuccess) { rc = (int)me; break; }
                rc = launch_inflate(n <= 256u ? 8u : n <= 512u ? 4u : n <= 1024u ? 2u : 1u, 0u, s, d_in, d_streams_arena,
                                    d_streams + first, d_results + first, cnt, ft);
                no_ws = 1;
                continue;
            }
            const uint64_t rest = workspace_bytes - slots_bytes;
            const uint64_t total_recs = rest / 16u / sizeof(debig_ws_rec);
            const uint64_t recs_bytes = align_up(total_recs * sizeof(debig_ws_rec), 256);
            const uint64_t total_rows = (rest - recs_bytes) / 256u;
            debig_ws_slot *slots = (debig_ws_slot *)d_workspace;
            debig_ws_rec *recs = (debig_ws_rec *)((uint8_t *)d_workspace + slots_bytes);
            uint32_t *rows = (uint32_t *)((uint8_t *)d_workspace + slots_bytes + recs_bytes);
            hipLaunchKernelGGL(debig_split_plan_kernel, dim3(1), dim3(1024), 0, s, d_streams + first, cnt, slots, total_rows, total_recs);
            if (wpe == 3)
                hipLaunchKernelGGL((debig_png_fused_kernel<PNG_FUSED_NWD, PNG_FUSED_BLK, 3, LzLds>), dim3(cnt), dim3(64 * (2 + PNG_FUSED_NWD)), 0, s,
                                   (const uint8_t *)d_in, (uint8_t *)d_streams_arena, d_streams + first, cnt, ft->scan, slots, recs, rows,
                                   d_results + first, (uint8_t *)d_rgba_arena, d_images + first, d_png_results + first, kflags);
            else /* two workgroups per CU by registers anyway: the 12 KB LZ77 tile fits beside them (61 KB of LDS each) */
                hipLaunchKernelGGL((debig_png_fused_kernel<PNG_FUSED_NWD, PNG_FUSED_BLK, 2, LzLdsBig>), dim3(cnt), dim3(64 * (2 + PNG_FUSED_NWD)), 0, s,
                                   (const uint8_t *)d_in, (uint8_t *)d_streams_arena, d_streams + first, cnt, ft->scan, slots, recs, rows,
                                   d_results + first, (uint8_t *)d_rgba_arena, d_images + first, d_png_results + first, kflags);
            // what the scan handed back: one workgroup per stream, as wide as the batch size allows
            if (kflags & 2u) { rc = (int)hipGetLastError(); continue; } /* DEBIG_FUSED_FLAGS & 2 (diagnostic): leave them as DEBIG_E_RETRY */
            rc = launch_inflate(n <= 256u ? 8u : n <= 512u ? 4u : n <= 1024u ? 2u : 1u, DEBIG_CLASS_RETRY, s, d_in, d_streams_arena,
                                d_streams + first, d_results + first, cnt, ft);
        }
        const int frc = hold.finish();
        if (rc == 0 && frc) rc = frc;
    }
    if (rc) return rc;
    if (kflags & 2u) return 0;
    if (no_ws) /* some group had no workspace: the ordinary de-filter launch takes the whole batch (images the fused kernel finished
                  are simply decoded again) */
        return debig_hip_png_defilter_batch(d_streams_arena, d_rgba_arena, d_images, d_png_results, n, hip_stream);
    hipLaunchKernelGGL(debig_png_defilter_kernel<8>, dim3(n), dim3(512), 0, s, (const uint8_t *)d_streams_arena,
                       (uint8_t *)d_rgba_arena, d_images, d_png_results, n, 1u, (uint32_t *)nullptr, 1u);
    hipLaunchKernelGGL(debig_png_p3_kernel, dim3(n), dim3(PNG_P3_THREADS), 0, s, (const uint8_t *)d_streams_arena,
                       (uint8_t *)d_rgba_arena, d_images, d_png_results, n);
    return (int)hipGetLastError();
}

int debig_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int debig_hip_set_device(int dev) { return (int)hipSetDevice(dev); }
int debig_hip_get_device(void)
{
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
uint64_t debig_hip_mem_free(void)
{
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
    return (uint64_t)fr;
}
void *debig_hip_malloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    return p;
}
void debig_hip_free(void *p) { (void)hipFree(p); }
int debig_hip_memcpy_h2d(void *d, const void *h, uint64_t bytes, void *s)
{
    return (int)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s);
}
int debig_hip_memcpy_d2h(void *h, const void *d, uint64_t bytes, void *s)
{
    return (int)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s);
}
int debig_hip_memset(void *d, int v, uint64_t bytes, void *s)
{
    return (int)hipMemsetAsync(d, v, bytes, (hipStream_t)s);
}
int debig_hip_stream_sync(void *s) { return (int)hipStreamSynchronize((hipStream_t)s); }
void *debig_hip_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void debig_hip_host_free(void *p) { (void)hipHostFree(p); }
int debig_hip_event_sync(void *ev) { return (int)hipEventSynchronize((hipEvent_t)ev); }
const char *debig_hip_error_string(int err) { return hipGetErrorString((hipError_t)err); }
void *debig_hip_event_create(void)
{
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void *)e;
}
int debig_hip_event_record(void *ev, void *s) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)s); }
float debig_hip_event_elapsed_ms(void *a, void *b)
{
    float ms = -1.f;
    (void)hipEventSynchronize((hipEvent_t)b);
    (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
    return ms;
}
void debig_hip_event_destroy(void *ev) { (void)hipEventDestroy((hipEvent_t)ev); }

} // extern "C"
