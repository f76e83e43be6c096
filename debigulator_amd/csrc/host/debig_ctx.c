#include "debig_ctx.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static debig_ctx g_ctx[DEBIG_MAX_THREADS];

debig_ctx *debig_ctx_get(uint32_t thread_id)
{
    if (thread_id >= DEBIG_MAX_THREADS) return NULL;
    return &g_ctx[thread_id];
}

int debig_devbuf_reserve(debig_devbuf *b, uint64_t bytes)
{
    if (b->cap >= bytes && b->ptr) return 0;
    if (b->ptr) debig_hip_free(b->ptr);
    uint64_t cap = bytes + bytes / 4 + 4096; /* grow geometrically: the buffers are reused */
    b->ptr = debig_hip_malloc(cap);
    b->cap = b->ptr ? cap : 0;
    return b->ptr ? 0 : 2; /* hipErrorOutOfMemory */
}

static void buf_free(debig_devbuf *b)
{
    if (b->ptr) debig_hip_free(b->ptr);
    b->ptr = NULL;
    b->cap = 0;
}

void debig_ctx_release(uint32_t thread_id) { debig_ctx_release_ptr(debig_ctx_get(thread_id)); }

void debig_ctx_release_ptr(debig_ctx *c)
{
    if (!c) return;
    buf_free(&c->in);
    buf_free(&c->out);
    buf_free(&c->desc);
    buf_free(&c->res);
    buf_free(&c->rgba);
    buf_free(&c->img);
    buf_free(&c->imgres);
    buf_free(&c->files);
    buf_free(&c->spans);
    buf_free(&c->crcs);
    buf_free(&c->copies);
    buf_free(&c->ws);
    buf_free(&c->dense);
    buf_free(&c->dense_list);
    if (c->pin_in.ptr) debig_hip_host_free(c->pin_in.ptr);
    if (c->pin_out.ptr) debig_hip_host_free(c->pin_out.ptr);
    c->pin_in.ptr = c->pin_out.ptr = NULL;
    c->pin_in.cap = c->pin_out.cap = 0;
    for (int k = 0; k < DEBIG_STAGE_CHUNKS; k++) {
        if (c->ev[k]) debig_hip_event_destroy(c->ev[k]);
        c->ev[k] = NULL;
    }
}

/* ---- page-locked staging */
/* The arenas are reused from call to call, but not hoarded: one that is more than 4 x what the calls
 * need and larger than 256 MiB is given back (10 thread ids x up to 16 device contexts each keep a pair
 * of them) -- but only after DEBIG_PIN_SMALL_CALLS such calls IN A ROW: a caller that alternates large and
 * small batches must not free and page-lock hundreds of MiB again on every call of the drop-in API
 * (hipHostMalloc costs milliseconds per GiB). */
#define DEBIG_PIN_KEEP (256ull << 20)
#define DEBIG_PIN_SMALL_CALLS 8u
static int pin_reserve(debig_devbuf *b, uint64_t bytes)
{
    if (b->ptr && b->cap > DEBIG_PIN_KEEP && b->cap / 4u > bytes) {
        if (++b->small_calls >= DEBIG_PIN_SMALL_CALLS) {
            debig_hip_host_free(b->ptr);
            b->ptr = NULL;
            b->cap = 0;
            b->small_calls = 0;
        }
    } else {
        b->small_calls = 0;
    }
    if (b->cap >= bytes && b->ptr) return 0;
    if (b->ptr) debig_hip_host_free(b->ptr);
    uint64_t cap = bytes + bytes / 4 + 4096;
    b->ptr = debig_hip_host_alloc(cap);
    b->cap = b->ptr ? cap : 0;
    b->small_calls = 0;
    return b->ptr ? 0 : 2;
}

#define DEBIG_STAGE_THREADS 12
typedef struct copy_job {
    uint8_t *base;              /* pinned arena */
    uint8_t *const *dsts;       /* download: destinations */
    const uint8_t *const *srcs; /* upload: sources */
    const uint64_t *sizes, *offs;
    uint32_t lo, hi;            /* streams [lo, hi) */
    uint64_t arena_lo, arena_hi; /* download: only streams whose last byte lies in [arena_lo, arena_hi) */
} copy_job;

static void *copy_worker(void *arg)
{
    const copy_job *j = (const copy_job *)arg;
    for (uint32_t i = j->lo; i < j->hi; i++) {
        if (!j->sizes[i]) continue;
        if (j->srcs) {
            if (j->srcs[i]) memcpy(j->base + j->offs[i], j->srcs[i], (size_t)j->sizes[i]);
        } else if (j->dsts[i]) {
            const uint64_t last = j->offs[i] + j->sizes[i] - 1u; /* the piece its LAST byte arrives in */
            if (last >= j->arena_lo && last < j->arena_hi) memcpy(j->dsts[i], j->base + j->offs[i], (size_t)j->sizes[i]);
        }
    }
    return NULL;
}

static uint32_t stage_threads(uint64_t bytes)
{
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    uint32_t t = nc > DEBIG_STAGE_THREADS ? (uint32_t)DEBIG_STAGE_THREADS : (nc < 1 ? 1u : (uint32_t)nc);
    if (bytes < (1u << 20)) t = 1; /* not worth a thread */
    return t;
}

/* run the job over streams [0, n) on several threads, split by stream count */
static void run_copy(copy_job proto, uint32_t n, uint64_t bytes)
{
    const uint32_t nt = stage_threads(bytes);
    pthread_t th[DEBIG_STAGE_THREADS];
    copy_job jobs[DEBIG_STAGE_THREADS];
    uint32_t started = 0;
    for (uint32_t t = 0; t < nt; t++) {
        jobs[t] = proto;
        jobs[t].lo = (uint32_t)((uint64_t)n * t / nt);
        jobs[t].hi = (uint32_t)((uint64_t)n * (t + 1) / nt);
        if (t + 1 == nt || pthread_create(&th[t], NULL, copy_worker, &jobs[t]) != 0) {
            copy_worker(&jobs[t]); /* the calling thread takes the last share (and any that failed to start) */
            continue;
        }
        started |= 1u << t;
    }
    for (uint32_t t = 0; t < nt; t++)
        if (started & (1u << t)) pthread_join(th[t], NULL);
}

int debig_upload_packed(debig_ctx *c, void *d_arena, const uint8_t *const *srcs, const uint64_t *sizes,
                        const uint64_t *offs, uint32_t n, uint64_t total)
{
    if (total == 0 || n == 0) return 0;
    /* the previous call's H2D out of this arena has completed: every batch call synchronises */
    if (pin_reserve(&c->pin_in, total)) return 2;
    copy_job j;
    memset(&j, 0, sizeof j);
    j.base = (uint8_t *)c->pin_in.ptr;
    j.srcs = srcs;
    j.sizes = sizes;
    j.offs = offs;
    uint64_t bytes = 0;
    int ascending = 1;
    for (uint32_t i = 0; i < n; i++) {
        bytes += sizes[i];
        if (i && offs[i] < offs[i - 1]) ascending = 0;
    }
    /* two halves (callers lay streams out in ascending order): the first half is on the wire while
     * the threads pack the second */
    uint32_t m = 0;
    while (ascending && m < n && offs[m] < total / 2) m++;
    if (!ascending || m == 0 || m == n || bytes < (8u << 20)) {
        run_copy(j, n, bytes);
        return debig_hip_memcpy_h2d(d_arena, c->pin_in.ptr, total, NULL);
    }
    copy_job a = j, b = j;
    b.srcs = srcs + m; b.sizes = sizes + m; b.offs = offs + m;
    run_copy(a, m, bytes / 2);
    int rc = debig_hip_memcpy_h2d(d_arena, c->pin_in.ptr, offs[m], NULL);
    run_copy(b, n - m, bytes / 2);
    if (!rc) rc = debig_hip_memcpy_h2d((uint8_t *)d_arena + offs[m], (uint8_t *)c->pin_in.ptr + offs[m], total - offs[m], NULL);
    return rc;
}

/* last resort when no page-locked memory can be had: one synchronous copy per stream, straight into
 * the caller's (pageable) buffers */
static int download_direct(const void *d_arena, uint8_t *const *dsts, const uint64_t *sizes, const uint64_t *offs, uint32_t n)
{
    int rc = debig_hip_stream_sync(NULL);
    for (uint32_t i = 0; i < n && !rc; i++)
        if (sizes[i] && dsts[i]) rc = debig_hip_memcpy_d2h(dsts[i], (const uint8_t *)d_arena + offs[i], sizes[i], NULL);
    if (!rc) rc = debig_hip_stream_sync(NULL);
    return rc;
}

/* [0, total) of the pinned arena <- the same range of a device buffer, in DEBIG_STAGE_CHUNKS pieces;
 * every piece is unpacked (memcpy to the callers' buffers, several threads) while the next ones are
 * still on the wire.  offs: where each stream lies inside that range. */
static int download_pieces(debig_ctx *c, const void *d_src, uint8_t *const *dsts, const uint64_t *sizes,
                           const uint64_t *offs, uint32_t n, uint64_t total, uint64_t bytes)
{
    /* pieces of equal size; a stream is unpacked with the piece that holds its LAST byte (the
     * copies are issued in order, so everything before it has landed as well) */
    const uint64_t step = (total + DEBIG_STAGE_CHUNKS - 1) / DEBIG_STAGE_CHUNKS;
    int rc = 0;
    for (int k = 0; k < DEBIG_STAGE_CHUNKS && !rc; k++) {
        const uint64_t lo = step * (uint64_t)k, hi = lo + step < total ? lo + step : total;
        if (lo < hi) rc = debig_hip_memcpy_d2h((uint8_t *)c->pin_out.ptr + lo, (const uint8_t *)d_src + lo, hi - lo, NULL);
        if (!rc) rc = debig_hip_event_record(c->ev[k], NULL);
    }
    copy_job j;
    memset(&j, 0, sizeof j);
    j.base = (uint8_t *)c->pin_out.ptr;
    j.dsts = dsts;
    j.sizes = sizes;
    j.offs = offs;
    for (int k = 0; k < DEBIG_STAGE_CHUNKS && !rc; k++) {
        rc = debig_hip_event_sync(c->ev[k]);
        j.arena_lo = step * (uint64_t)k;
        j.arena_hi = j.arena_lo + step;
        if (!rc) run_copy(j, n, bytes / DEBIG_STAGE_CHUNKS);
    }
    return rc;
}

/* Bring n decoded ranges [offs[i], offs[i] + sizes[i]) of a device arena of `total` bytes to the
 * callers' buffers.  The arena is laid out by recipient CAPACITIES, the ranges are what was decoded:
 *   dense arena  (the ranges fill at least half of it): the span goes over the wire as it is;
 *   sparse arena (2048 streams with 1 MiB recipients that decode to 64 KiB: 2 GiB of span for
 *                 128 MiB of data): the ranges are first packed on the device (debig_hip_gather, a
 *                 copy list) and only the packed bytes cross PCIe and take pinned memory.
 * No page-locked memory: one copy per stream into the pageable buffers instead of failing the batch. */
int debig_download_unpack(debig_ctx *c, const void *d_arena, uint8_t *const *dsts, const uint64_t *sizes,
                          const uint64_t *offs, uint32_t n, uint64_t total)
{
    if (total == 0 || n == 0) return debig_hip_stream_sync(NULL);
    for (int k = 0; k < DEBIG_STAGE_CHUNKS; k++)
        if (!c->ev[k] && !(c->ev[k] = debig_hip_event_create())) return 2;
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n; i++) bytes += dsts[i] ? sizes[i] : 0;
    if (bytes == 0) return debig_hip_stream_sync(NULL);
    if (bytes >= total / 2u) { /* dense */
        if (pin_reserve(&c->pin_out, total)) return download_direct(d_arena, dsts, sizes, offs, n);
        return download_pieces(c, d_arena, dsts, sizes, offs, n, total, bytes);
    }
    /* sparse: pack on the device */
    debig_copy *list = (debig_copy *)malloc((size_t)n * sizeof(debig_copy));
    uint64_t *dense_offs = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
    int rc = (list && dense_offs) ? 0 : 2;
    uint64_t at = 0;
    uint32_t m = 0;
    if (!rc) {
        for (uint32_t i = 0; i < n; i++) {
            dense_offs[i] = at;
            if (!dsts[i] || !sizes[i]) continue;
            list[m].src_off = offs[i];
            list[m].dst_off = at;
            list[m].len = sizes[i];
            m++;
            at += (sizes[i] + 15u) & ~(uint64_t)15; /* 16-byte aligned starts: aligned stores in the copy kernel */
        }
        if (debig_devbuf_reserve(&c->dense, at) || debig_devbuf_reserve(&c->dense_list, (uint64_t)m * sizeof(debig_copy)) ||
            pin_reserve(&c->pin_out, at))
            rc = 2;
    }
    if (!rc) rc = debig_hip_memcpy_h2d(c->dense_list.ptr, list, (uint64_t)m * sizeof(debig_copy), NULL);
    if (!rc) rc = debig_hip_gather(d_arena, c->dense.ptr, (const debig_copy *)c->dense_list.ptr, m, NULL);
    if (!rc) rc = debig_hip_stream_sync(NULL); /* `list` is pageable: the upload must be over before it is freed */
    if (!rc) rc = download_pieces(c, c->dense.ptr, dsts, sizes, dense_offs, n, at, bytes);
    free(list);
    free(dense_offs);
    if (rc == 2) return download_direct(d_arena, dsts, sizes, offs, n); /* out of (pinned / device) memory */
    return rc;
}

int debig_launch_inflate_planned(debig_ctx *c, const void *d_in_arena, const debig_stream *desc,
                                 debig_result *res, uint32_t n)
{
    if (n == 0) return 0;
    int rc = 0, permuted = 0;
    uint32_t *order = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    const uint32_t waves = debig_plan_batch(desc, n, order, &permuted);
    debig_stream *sorted = NULL;
    debig_result *tmp = NULL;
    const debig_stream *up = desc;
    debig_result *down = res;
    if (permuted) {
        sorted = (debig_stream *)malloc((size_t)n * sizeof(debig_stream));
        tmp = (debig_result *)malloc((size_t)n * sizeof(debig_result));
        if (sorted && tmp) {
            for (uint32_t k = 0; k < n; k++) sorted[k] = desc[order[k]];
            up = sorted;
            down = tmp;
        } else {
            permuted = 0; /* out of host memory for the plan: launch in the caller's order */
        }
    }
    const uint32_t w = permuted ? waves : debig_pick_waves(desc, n);
    uint64_t ws_bytes = 0;
    if (w == DEBIG_WAVES_SPLIT || w == DEBIG_WAVES_STRAND || w == DEBIG_WAVES_STRAND_PIPE) { /* the throughput paths want a token workspace sized from the input */
        uint64_t total_in = 0, total_out = 0;
        for (uint32_t i = 0; i < n; i++) { total_in += desc[i].in_len; total_out += desc[i].out_cap; }
        ws_bytes = debig_hip_inflate_workspace_bytes_io(total_in, total_out, n);
        if (debig_devbuf_reserve(&c->ws, ws_bytes)) ws_bytes = 0; /* no memory: the library's own fallback */
    }
    if ((rc = debig_devbuf_reserve(&c->desc, (uint64_t)n * sizeof(debig_stream))) ||
        (rc = debig_devbuf_reserve(&c->res, (uint64_t)n * sizeof(debig_result))) ||
        (rc = debig_hip_memcpy_h2d(c->desc.ptr, up, (uint64_t)n * sizeof(debig_stream), NULL))) {
        /* fall through to the cleanup */
    } else if (w == DEBIG_WAVES_CHUNKED) {
        /* chunk tasks need about 5 bytes of workspace per decoded byte: the batch goes through in
         * groups of streams, launched back to back.  FEW, LARGE groups: every group pays the serial
         * walk of the window kernel and a dozen launch tails (config 4, 32 images: two groups 75.7 ms,
         * one group 63.9 ms; profiles/r03_chunk_workspace_groups.txt), so a group may take what the
         * device has free (less an eighth, at least 40 GiB asked for), DEBIG_CHUNKED_WS_MB overrides;
         * the groups are evened out (as many streams in each as the fullest needs) */
        uint64_t cap = 40960ull << 20;
        const char *e = getenv("DEBIG_CHUNKED_WS_MB");
        if (e && *e) cap = strtoull(e, NULL, 0) << 20;
        else {
            const uint64_t fr = debig_hip_mem_free() + c->ws.cap; /* what the context's workspace holds is ours to reuse */
            if (fr - fr / 8u > cap) cap = fr - fr / 8u;
        }
        /* how many streams per group when the groups are even */
        uint32_t per_group = n;
        {
            uint32_t groups = 0, first = 0;
            while (first < n) {
                uint64_t tin = 0, tout = 0;
                uint32_t cnt = 0;
                while (first + cnt < n) {
                    const uint64_t a = up[first + cnt].in_len, b = up[first + cnt].out_cap;
                    if (cnt && debig_hip_inflate_chunked_workspace_bytes(tin + a, tout + b, cnt + 1u) > cap) break;
                    tin += a; tout += b; cnt++;
                }
                first += cnt;
                groups++;
            }
            per_group = (n + groups - 1u) / groups;
        }
        /* pass 0 sizes the largest group (one reservation: growing the buffer between launches would
         * wait for the device), pass 1 launches */
        uint64_t biggest = 0;
        void *ws = NULL;
        for (int pass = 0; pass < 2 && !rc; pass++) {
            uint32_t first = 0;
            while (first < n && !rc) {
                uint64_t tin = 0, tout = 0, need = 0;
                uint32_t cnt = 0;
                while (first + cnt < n) {
                    const uint64_t a = up[first + cnt].in_len, b = up[first + cnt].out_cap;
                    const uint64_t nd = debig_hip_inflate_chunked_workspace_bytes(tin + a, tout + b, cnt + 1u);
                    if (cnt && (nd > cap || cnt >= per_group)) break;
                    tin += a; tout += b; need = nd; cnt++;
                }
                if (need > cap) need = cap; /* one stream larger than the cap: the chunk path hands it back */
                if (pass == 0) { if (need > biggest) biggest = need; }
                else /* no workspace: the call falls back to whole workgroups per stream */
                    rc = debig_hip_inflate_batch_ws(d_in_arena, c->out.ptr, (const debig_stream *)c->desc.ptr + first,
                                                    (debig_result *)c->res.ptr + first, cnt, DEBIG_WAVES_CHUNKED, ws,
                                                    ws ? need : 0, NULL);
                first += cnt;
            }
            if (pass == 0) ws = debig_devbuf_reserve(&c->ws, biggest) ? NULL : c->ws.ptr;
        }
    } else {
        rc = debig_hip_inflate_batch_ws(d_in_arena, c->out.ptr, (const debig_stream *)c->desc.ptr,
                                        (debig_result *)c->res.ptr, n, w, ws_bytes ? c->ws.ptr : NULL, ws_bytes, NULL);
    }
    if (!rc) rc = debig_hip_memcpy_d2h(down, c->res.ptr, (uint64_t)n * sizeof(debig_result), NULL);
    if (!rc) rc = debig_hip_stream_sync(NULL);
    if (!rc && permuted) {
        for (uint32_t k = 0; k < n; k++) res[order[k]] = tmp[k];
    }
    free(order);
    free(sorted);
    free(tmp);
    return rc;
}
