/* Per-thread_id device context shared by the drop-in entry points (host side, plain C). */
#ifndef DEBIG_CTX_H
#define DEBIG_CTX_H
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "debig_hip.h"

#define DEBIG_MAX_THREADS 10 /* reference: INFLATE_MAX_THREADS / PNG_DECODER_MAX_THREADS */
#define DEBIG_API __attribute__((visibility("default")))

#define DEBIG_STAGE_CHUNKS 4
typedef struct debig_devbuf {
    void *ptr;
    uint64_t cap;
    uint32_t small_calls; /* page-locked arenas: consecutive calls that needed less than a quarter of it */
} debig_devbuf;

typedef struct debig_ctx {
    debig_devbuf in, out, desc, res, rgba, img, imgres;
    debig_devbuf files, spans, crcs, copies; /* PNG: whole files, chunk spans, their CRCs, IDAT gather list */
    debig_devbuf ws; /* token workspace of the scan / LZ77 kernel pair (DEBIG_WAVES_SPLIT) */
    debig_devbuf pin_in, pin_out; /* page-locked staging arenas (host memory) */
    debig_devbuf dense, dense_list; /* sparse downloads: the decoded ranges packed on the device first */
    void *ev[DEBIG_STAGE_CHUNKS]; /* download pipeline events */
} debig_ctx;

debig_ctx *debig_ctx_get(uint32_t thread_id);
/* make sure b holds at least `bytes` (contents are not preserved); 0 on success */
int debig_devbuf_reserve(debig_devbuf *b, uint64_t bytes);
void debig_ctx_release(uint32_t thread_id);
void debig_ctx_release_ptr(debig_ctx *c); /* every device buffer, pinned arena and event of one context */

static inline uint64_t debig_align16(uint64_t x) { return (x + 15u) & ~(uint64_t)15u; }

/* Host buffers <-> device arena through page-locked staging, the way PCIe likes it (one big
 * transfer instead of one small pageable copy per stream: 11 us each, profiles/r01_host_api.txt).
 *   debig_upload_packed    srcs[i] (sizes[i] bytes, may be NULL / 0) -> d_arena + offs[i]: packed
 *                          into the pinned arena by several host threads, then ONE H2D of
 *                          [0, total).  Asynchronous on the default stream.
 *   debig_download_unpack  d_arena + offs[i] -> dsts[i] (sizes[i] bytes): the arena comes down in
 *                          DEBIG_STAGE_CHUNKS pieces; while piece k+1 is on the wire the host threads
 *                          copy piece k out to the callers' buffers.  Synchronises.
 * Both return 0 or an error code. */
int debig_upload_packed(debig_ctx *c, void *d_arena, const uint8_t *const *srcs, const uint64_t *sizes,
                        const uint64_t *offs, uint32_t n, uint64_t total);
int debig_download_unpack(debig_ctx *c, const void *d_arena, uint8_t *const *dsts, const uint64_t *sizes,
                          const uint64_t *offs, uint32_t n, uint64_t total);

/* waves_per_stream for debig_hip_inflate_batch_ex from what the host knows about the batch
 * (the shim itself only sees n: descriptors live in device memory).  Few streams: several
 * wavefronts each.  A big batch in which only a few streams are large: those 4-wide beside
 * the small ones 1-wide, so the longest stream does not set the run time (measured:
 * profiles/r01_mw_sweep.txt). */
/* Few streams, large on average (a batch of big PNG images): cut them into chunk tasks
 * (DEBIG_WAVES_CHUNKED).  Measured (profiles/r02_chunked.txt): 64 streams of 1.6 MB, 13.3 ms with
 * 8 wavefronts per stream, 5.5 ms in chunk tasks; config 3's 1024 sample PNGs (0.43 MB on
 * average) are 40 % slower in chunk tasks.  The line is drawn at 1 MiB of input per stream. */
#define DEBIG_CHUNKED_MEAN_IN_BYTES (1u << 20)
#define DEBIG_CHUNKED_LONGEST_IN_BYTES (4u << 20) /* n > 1024: one stream this long is enough */
#ifndef DEBIG_CHUNKED_ROWS_MIN_IN_BYTES
#define DEBIG_CHUNKED_ROWS_MIN_IN_BYTES (256u << 10) /* image rows (DEBIG_STREAM_IMAGE_ROWS): every stream at least this long */
#endif
static inline uint32_t debig_pick_waves(const debig_stream *desc, uint32_t n)
{
    if (n <= 1024u) {
        uint64_t total_in = 0, shortest = ~0ull;
        uint32_t rows = 0;
        for (uint32_t i = 0; i < n; i++) {
            total_in += desc[i].in_len;
            if (desc[i].in_len < shortest) shortest = desc[i].in_len;
            rows += (desc[i].flags & DEBIG_STREAM_IMAGE_ROWS) != 0;
        }
        if (total_in >= (uint64_t)n * DEBIG_CHUNKED_MEAN_IN_BYTES) return DEBIG_WAVES_CHUNKED;
        /* filtered image rows (include/debig_hip.h: DEBIG_STREAM_IMAGE_ROWS), every stream of the batch long: chunk tasks
         * from a quarter of that size on (profiles/r04_single_stream.txt: one 1 MB sample PNG 28.7 -> 4.9 ms, 73 copies
         * 25 -> 6 ms, 365 copies 26 -> 12..19 ms; streams of few blocks -- 100 KB that decode to 4 MB -- gain nothing,
         * hence "every stream") */
        if (rows == n && n <= 512u && shortest >= DEBIG_CHUNKED_ROWS_MIN_IN_BYTES) return DEBIG_WAVES_CHUNKED;
    }
    if (n <= 256u) return 8u;
    if (n <= DEBIG_STRAND_MIN_STREAMS) {
        /* long streams (several windows each): scan and LZ77 half side by side beat a workgroup per stream
         * (profiles/r04_pipe.txt: 384 x 1 MiB image rows 37 -> 49 GB/s, 512 x 1 MiB text 112 -> 132; 512 x 64 KiB:
         * 86 4-wide, 65 as a pipeline -- one window per stream, nothing to overlap) */
        uint64_t total_in = 0;
        for (uint32_t i = 0; i < n; i++) total_in += desc[i].in_len;
        if (total_in >= (uint64_t)n * DEBIG_STRAND_PIPE_MEAN_IN_BYTES) return DEBIG_WAVES_STRAND_PIPE;
        return n <= 512u ? 4u : 2u;
    }
    if (n <= 1024u) return DEBIG_WAVES_STRAND_PIPE; /* (debig_hip.hip: auto_waves_per_stream has the measurements) */
    uint32_t n_large = 0;
    uint64_t longest = 0;
    for (uint32_t i = 0; i < n; i++) {
        n_large += desc[i].in_len >= DEBIG_LARGE_IN_BYTES || desc[i].out_cap >= DEBIG_LARGE_OUT_BYTES;
        if (desc[i].in_len > longest) longest = desc[i].in_len;
    }
    /* thousands of streams and a very large one among them: chunk tasks for everything (a small
     * stream is one task: the scan / LZ77 bodies of the throughput path) */
    if (longest >= DEBIG_CHUNKED_LONGEST_IN_BYTES && n <= 16384u) return DEBIG_WAVES_CHUNKED;
    if (n_large != 0 && n_large <= 256u) return DEBIG_WAVES_LARGE4_SMALL1;
    if (n <= DEBIG_STRAND_PIPE_MAX_STREAMS) return DEBIG_WAVES_STRAND_PIPE;
    return n <= DEBIG_STRAND_MAX_STREAMS ? DEBIG_WAVES_STRAND : DEBIG_WAVES_SPLIT;
}

/* Dispatch plan for one inflate launch.  Workgroups start in descriptor order, so for a batch
 * of 513..1024 streams whose sizes are strongly skewed (the largest quarter of the streams
 * holds at least half of the input bytes) the descriptors are launched 4 wavefronts wide, THE
 * STREAMS THAT TOUCH THE MOST BYTES (in_len + out_cap) FIRST: only 512 such workgroups are
 * resident at a time (2 per CU), the long streams start at once and the short ones fill in
 * behind them (config 3, 1024 sample PNGs: 49.1 ms uniform 2-wide -> 40.3 ms longest input
 * first -> 34.1 ms by bytes touched: a 30 KB stream that decodes to 4 MB runs 20 ms).  Uniform batches of that
 * size stay 2-wide in their own order (4-wide would lose a third).  order[k] = index of the
 * descriptor to launch k-th (only written when *permuted is set). */
typedef struct debig_len_idx {
    uint64_t len;
    uint32_t idx;
} debig_len_idx;
static int debig_len_idx_desc(const void *a, const void *b)
{
    const debig_len_idx *x = (const debig_len_idx *)a, *y = (const debig_len_idx *)b;
    if (x->len != y->len) return x->len < y->len ? 1 : -1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static inline uint32_t debig_plan_batch(const debig_stream *desc, uint32_t n, uint32_t *order, int *permuted)
{
    *permuted = 0;
    const uint32_t waves = debig_pick_waves(desc, n);
    if (waves == DEBIG_WAVES_CHUNKED) return waves; /* chunk tasks are balanced by construction: no reordering */
    if (n <= 512u || n > 1024u || !order) return waves;
    debig_len_idx *v = (debig_len_idx *)malloc((size_t)n * sizeof(debig_len_idx));
    if (!v) return waves;
    uint64_t total = 0, top = 0;
    for (uint32_t i = 0; i < n; i++) {
        v[i].len = desc[i].in_len;
        v[i].idx = i;
        total += desc[i].in_len;
    }
    qsort(v, n, sizeof(debig_len_idx), debig_len_idx_desc);
    for (uint32_t k = 0; k < n / 4u; k++) top += v[k].len;
    const int skewed = total != 0 && top * 2u >= total;
    if (skewed) {
        /* the order itself goes by the bytes a stream touches: a small input that decodes to
         * megabytes runs as long as a large one */
        for (uint32_t i = 0; i < n; i++) {
            v[i].len = desc[i].in_len + desc[i].out_cap;
            v[i].idx = i;
        }
        qsort(v, n, sizeof(debig_len_idx), debig_len_idx_desc);
        for (uint32_t k = 0; k < n; k++) order[k] = v[k].idx;
    }
    free(v);
    *permuted = skewed;
    return skewed ? 4u : waves;
}

/* One planned inflate launch (debig_plan_batch): uploads the descriptors (longest first when the
 * plan says so), launches on the default stream over in_arena -> c->out, brings the results back
 * in the CALLER'S order and synchronises.  c->desc / c->res are (re)sized here.  0 or an error. */
int debig_launch_inflate_planned(debig_ctx *c, const void *d_in_arena, const debig_stream *desc,
                                 debig_result *res, uint32_t n);

#endif
