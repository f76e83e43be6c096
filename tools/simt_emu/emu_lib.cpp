/* DEVELOPMENT / TEST TOOLING: the product's HIP kernels compiled for the CPU lock-step
 * emulator (hip_emu.h).  Exposes the same batch entry points as libdebigulator_hip.so
 * (include/debig_hip.h) with an emu_ prefix and HOST pointers. */
#include <stdio.h>
#include <stdlib.h>
#include "hip_emu.h"
/* the single-workgroup planning kernels take any power-of-two block size: 64 fibers instead of 1024 */
#define EMU_PLAN_THREADS 64
#include "../../include/debig_hip.h"
#include "../../debigulator_amd/csrc/inflate_kernel.inc"
#include "../../debigulator_amd/csrc/inflate_mw_kernel.inc"
#include "../../debigulator_amd/csrc/inflate_split_kernel.inc"
#include "../../debigulator_amd/csrc/inflate_strand_kernel.inc"
#include "../../debigulator_amd/csrc/inflate_chunk_kernel.inc"
#include "../../debigulator_amd/csrc/png_kernel.inc"
#include "../../debigulator_amd/csrc/png_fused_kernel.inc"
#include "../../debigulator_amd/csrc/checksum_kernel.inc"

/* cls: DEBIG_CLASS_ALL / _SMALL / _LARGE (streams outside the class are left untouched) */
extern "C" int emu_inflate_batch_cls(const void *in, void *out, const debig_stream *streams,
                                     debig_result *results, uint32_t n, uint32_t grid, uint32_t nw, uint32_t cls);

extern "C" int emu_inflate_batch(const void *in, void *out, const debig_stream *streams,
                                 debig_result *results, uint32_t n, uint32_t grid)
{
    return emu_inflate_batch_cls(in, out, streams, results, n, grid, 1, 0);
}

extern "C" int emu_inflate_mw_batch(const void *in, void *out, const debig_stream *streams,
                                    debig_result *results, uint32_t n, uint32_t grid, uint32_t nw)
{
    return emu_inflate_batch_cls(in, out, streams, results, n, grid, nw, 0);
}

static int emu_inflate_1(const void *in, void *out, const debig_stream *streams,
                                 debig_result *results, uint32_t n, uint32_t grid, uint32_t cls)
{
    if (grid == 0 || grid > n) grid = n;
    static uint32_t *ft = nullptr;
    if (!ft) {
        ft = (uint32_t *)calloc(1, sizeof(decltype(WaveLdsT<1>::t)));
        EMU_LAUNCH(debig_fixed_tables_kernel<1>, 1, 64, ft);
    }
    EMU_LAUNCH(debig_inflate_kernel, grid, 64, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft, cls);
    return 0;
}

/* the multi-wavefront-per-stream kernel: nw = 2 or 4 wavefronts per workgroup */
extern "C" int emu_inflate_batch_cls(const void *in, void *out, const debig_stream *streams,
                                     debig_result *results, uint32_t n, uint32_t grid, uint32_t nw, uint32_t cls)
{
    if (nw == 1) return emu_inflate_1(in, out, streams, results, n, grid, cls);
    if (grid == 0 || grid > n) grid = n;
    static uint32_t *ft = nullptr;
    if (!ft) {
        ft = (uint32_t *)calloc(1, sizeof(decltype(WaveLdsT<2>::t)));
        EMU_LAUNCH(debig_fixed_tables_kernel<2>, 1, 64, ft);
    }
    if (nw == 2)
        EMU_LAUNCH(debig_inflate_mw_kernel<2>, grid, 128, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft, cls);
    else if (nw == 4)
        EMU_LAUNCH(debig_inflate_mw_kernel<4>, grid, 256, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft, cls);
    else if (nw == 8)
        EMU_LAUNCH(debig_inflate_mw_kernel<8>, grid, 512, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft, cls);
    else
        return -1;
    return 0;
}

/* the scan / LZ77 kernel pair (inflate_split_kernel.inc) with a workspace of ws_bytes, and the
 * one-kernel path for the streams the pair hands back; *n_retried = how many those were */
extern "C" int emu_inflate_split_batch(const void *in, void *out, const debig_stream *streams, debig_result *results,
                                       uint32_t n, uint64_t ws_bytes, uint32_t *n_retried)
{
    static uint32_t *ft = nullptr;
    if (!ft) {
        ft = (uint32_t *)calloc(1, sizeof(decltype(WaveLdsT<1>::t)));
        EMU_LAUNCH(debig_fixed_tables_kernel<1>, 1, 64, ft);
    }
    static uint32_t *fts = nullptr;
    if (!fts) {
        fts = (uint32_t *)calloc(1, sizeof(decltype(ScanLds::t)));
        EMU_LAUNCH(debig_scan_fixed_tables_kernel, 1, 64, fts);
    }
    const uint64_t slots_bytes = ((uint64_t)n * sizeof(debig_ws_slot) + 4u * SPLIT_QUEUE_WORDS + 4u * (uint64_t)n + 255) / 256 * 256;
    if (ws_bytes < slots_bytes + (uint64_t)n * 1024u) return -1;
    uint8_t *ws = (uint8_t *)malloc(ws_bytes);
    memset(ws, 0xEE, ws_bytes); /* poison: nothing may be read before it is written */
    const uint64_t rest = ws_bytes - slots_bytes;
    const uint64_t total_recs = rest / 16u / sizeof(debig_ws_rec);
    const uint64_t recs_bytes = (total_recs * sizeof(debig_ws_rec) + 255) / 256 * 256;
    const uint64_t total_rows = (rest - recs_bytes) / 256u;
    debig_ws_slot *slots = (debig_ws_slot *)ws;
    debig_ws_rec *recs = (debig_ws_rec *)(ws + slots_bytes);
    uint32_t *rows = (uint32_t *)(ws + slots_bytes + recs_bytes);
    EMU_LAUNCH(debig_split_plan_kernel, 1, EMU_PLAN_THREADS, streams, n, slots, total_rows, total_recs);
    { /* the dispatch order must be a permutation: streams that look expensive first, each class in the caller's order */
        const uint32_t *perm = split_perm(slots, n);
        std::vector<uint8_t> seen(n, 0);
        int in_cheap = 0;
        uint32_t last_e = 0, last_c = 0;
        for (uint32_t b = 0; b < n; b++) {
            const uint32_t i = perm[b];
            if (i >= n || seen[i]) return -3;
            seen[i] = 1;
            const int cheap = split_stream_is_cheap(streams[i]);
            if (cheap) { if (in_cheap && i < last_c) return -3; in_cheap = 1; last_c = i; }
            else { if (in_cheap || (b && i < last_e)) return -3; last_e = i; }
        }
    }
    if (getenv("DEBIG_EMU_TWO_KERNELS")) { /* the two halves as separate kernels (the chunk path's shape) */
        EMU_LAUNCH(debig_scan_kernel, n, 64, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
        EMU_LAUNCH(debig_lz_kernel, n, 64, (uint8_t *)out, streams, results, n, (const debig_ws_slot *)slots,
                   (const debig_ws_rec *)recs, (const uint32_t *)rows);
    } else {
        if (getenv("DEBIG_EMU_SPLIT_QUEUED")) {
            // DEBIG_WAVES_SPLIT_QUEUED: fewer workgroups than streams, and twice (the queue counter carries on)
            const uint32_t grid = n > 3u ? 3u : n;
            const uint64_t *q = reinterpret_cast<const uint64_t *>(split_queue(slots, n));
            EMU_LAUNCH(debig_scanlz_queue_kernel, grid, 64, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
            if (*q != (uint64_t)n) return -2; /* a launch adds exactly n to the queue counter */
            EMU_LAUNCH(debig_scanlz_queue_kernel, grid, 64, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
            if (*q != 2u * (uint64_t)n) return -2;
        } else if (getenv("DEBIG_EMU_STRAND_PIPE")) { /* DEBIG_WAVES_STRAND_PIPE: scan and LZ77 wavefronts side by side */
            if (getenv("DEBIG_EMU_PIPE_BIG_TILE")) /* the 12 KB LZ77 tile (what the shim launches while the device holds every stream) */
                EMU_LAUNCH(debig_strand_pipe_kernel<LzLdsBig>, n, 128, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
            else
                EMU_LAUNCH(debig_strand_pipe_kernel<LzLds>, n, 128, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
        } else if (getenv("DEBIG_EMU_STRAND")) { /* DEBIG_WAVES_STRAND: the long-segment scan */
            EMU_LAUNCH(debig_strand_kernel, n, 64, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
        } else {
            EMU_LAUNCH(debig_scanlz_kernel, n, 64, (const uint8_t *)in, (uint8_t *)out, streams, n, fts, slots, recs, rows, results);
        }
    }
    uint32_t retried = 0;
    for (uint32_t i = 0; i < n; i++) retried += results[i].status == DEBIG_E_RETRY;
    if (n_retried) *n_retried = retried;
    EMU_LAUNCH(debig_inflate_kernel, n, 64, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft, DEBIG_CLASS_RETRY);
    free(ws);
    return 0;
}

// the chunk-parallel path for large streams (inflate_chunk_kernel.inc), kernel by kernel as the
// shim launches them; n_retried: streams handed to debig_inflate_kernel
extern "C" int emu_inflate_chunked_batch(const void *in, void *out, const debig_stream *streams, debig_result *results,
                                         uint32_t n, uint64_t ws_bytes, uint32_t chunk_bytes, uint32_t retry_width,
                                         uint32_t *n_retried)
{
    static uint32_t *ft = nullptr;
    if (!ft) {
        ft = (uint32_t *)calloc(1, sizeof(decltype(WaveLdsT<1>::t)));
        EMU_LAUNCH(debig_fixed_tables_kernel<1>, 1, 64, ft);
    }
    static uint32_t *fts = nullptr;
    if (!fts) {
        fts = (uint32_t *)calloc(1, sizeof(decltype(ScanLds::t)));
        EMU_LAUNCH(debig_scan_fixed_tables_kernel, 1, 64, fts);
    }
    const uint32_t mt = ck_max_tasks(ws_bytes, n);
    if (mt == 0) return -1;
    uint8_t *ws = (uint8_t *)aligned_alloc(256, (ws_bytes + 255) / 256 * 256);
    memset(ws, 0xEE, ws_bytes); /* poison: nothing may be read before it is written */
    EMU_LAUNCH(debig_ck_plan_kernel, 1, EMU_PLAN_THREADS, streams, n, ws, ws_bytes, mt, chunk_bytes);
    EMU_LAUNCH(debig_ck_find_kernel, mt, 64, (const uint8_t *)in, streams, n, ws, mt);
    EMU_LAUNCH(debig_ck_bounds_kernel, (n + 63) / 64, 64, streams, n, ws, retry_width > 1 ? 1u : 0u);
    EMU_LAUNCH(debig_ck_carve_kernel, 1, EMU_PLAN_THREADS, streams, n, ws, mt);
    EMU_LAUNCH(debig_ck_scan_kernel, mt, 64, (const uint8_t *)in, streams, n, (const uint32_t *)fts, ws, mt, 0u);
    EMU_LAUNCH(debig_ck_repair_kernel, (n + 63) / 64, 64, n, ws, mt);
    EMU_LAUNCH(debig_ck_scan_kernel, mt, 64, (const uint8_t *)in, streams, n, (const uint32_t *)fts, ws, mt, 1u);
    EMU_LAUNCH(debig_ck_chain_kernel, (n + 63) / 64, 64, streams, n, ws, mt);
    EMU_LAUNCH(debig_ck_place_kernel, 1, EMU_PLAN_THREADS, n, ws);
    EMU_LAUNCH(debig_ck_lz_kernel, mt, 64, (const uint8_t *)in, (uint8_t *)out, streams, n, ws, mt);
    EMU_LAUNCH(debig_ck_window_kernel, n, CK_WIN_THREADS, (const uint8_t *)out, streams, n, ws, mt);
    EMU_LAUNCH(debig_ck_translate_kernel, mt * CK_TR_PARTS, CK_TR_THREADS, (uint8_t *)out, streams, n, ws, mt);
    EMU_LAUNCH(debig_ck_finish_kernel, (n + 255) / 256, 256, n, (const uint8_t *)ws, results);
    uint32_t retried = 0;
    for (uint32_t i = 0; i < n; i++) retried += results[i].status == DEBIG_E_RETRY;
    if (n_retried) *n_retried = retried;
    if (getenv("DEBIG_EMU_CK_DEBUG")) {
        const debig_ck_hdr *h = (const debig_ck_hdr *)ws;
        fprintf(stderr, "ck: tasks %u of %u, C %u, rows %llu recs %llu planes %llu\n", h->n_tasks, h->max_tasks, h->chunk_bytes,
                (unsigned long long)h->total_rows, (unsigned long long)h->total_recs, (unsigned long long)h->planes_bytes);
        for (uint32_t i = 0; i < n; i++) {
            const debig_ck_stream &c = ck_streams(ws)[i];
            fprintf(stderr, " stream %u: tasks %u+%u state %u final %u total %llu bad %u need %llu\n", i, c.first_task, c.n_tasks,
                    c.state, c.final_task, (unsigned long long)c.total_out, c.bad, (unsigned long long)c.plane_need);
            for (uint32_t k = 0; k < c.n_tasks; k++) {
                const debig_ck_task &t = ck_tasks(ws, n)[c.first_task + k];
                const debig_ws_slot &sl = ck_slots(ws, n, mt)[c.first_task + k];
                fprintf(stderr, "   task %u: found %lld start %lld stop %lld live %u | slot state %u flags %u out %llu end %lld rows %u recs %u\n",
                        k, (long long)t.found_bit, (long long)t.start_bit, (long long)t.stop_bit, t.live, sl.state, sl.flags,
                        (unsigned long long)sl.out_total, (long long)sl.end_bit, sl.rows, sl.recs);
            }
        }
    }
    if (retry_width > 1) {
        static uint32_t *ftm = nullptr;
        if (!ftm) {
            ftm = (uint32_t *)calloc(1, sizeof(decltype(WaveLdsT<2>::t)));
            EMU_LAUNCH(debig_fixed_tables_kernel<2>, 1, 64, ftm);
        }
        EMU_LAUNCH(debig_inflate_mw_kernel<4>, n, 256, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ftm, DEBIG_CLASS_RETRY);
    } else
    EMU_LAUNCH(debig_inflate_kernel, n, 64, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft, DEBIG_CLASS_RETRY);
    free(ws);
    return 0;
}

extern "C" int emu_png_defilter_batch_w(const void *streams_arena, void *rgba_arena, const debig_png_image *images,
                                        debig_png_result *results, uint32_t n, uint32_t nwd)
{
    if (nwd == 16)
        EMU_LAUNCH((debig_png_defilter_kernel<16, 6>), n, 1024, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n);
    else if (nwd == 8)
        EMU_LAUNCH(debig_png_defilter_kernel<8>, n, 512, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n);
    else if (nwd == 4)
        EMU_LAUNCH(debig_png_defilter_kernel<4>, n, 256, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n);
    else if (nwd == 2)
        EMU_LAUNCH(debig_png_defilter_kernel<2>, n, 128, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n);
    else
        EMU_LAUNCH(debig_png_defilter_kernel<1>, n, 64, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n);
    EMU_LAUNCH(debig_png_p3_kernel, n, PNG_P3_THREADS, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images,
               results, n);
    return 0;
}
/* the several-workgroups-per-image mode as the shim launches it: G workgroups of 4 wavefronts per image, then the
 * one-workgroup pass over the images given up as REDO.  The emulator runs ONE workgroup at a time, i.e. the
 * workgroups of an image are never resident together: every cross-workgroup wait fails, which is exactly the
 * situation the REDO pass exists for.  *n_redo = images the first launch gave up. */
extern "C" int emu_png_defilter_mwg(const void *streams_arena, void *rgba_arena, const debig_png_image *images,
                                    debig_png_result *results, uint32_t n, uint32_t g, uint32_t *n_redo)
{
    uint32_t *gsync = (uint32_t *)calloc((size_t)n * PNG_GSYNC_STRIDE, sizeof(uint32_t));
    EMU_LAUNCH((debig_png_defilter_kernel<4, 16, true, true>), n * g, 256, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images,
               results, n, g, gsync, 0u);
    uint32_t redo = 0;
    for (uint32_t i = 0; i < n; i++) redo += results[i].good == 0 && results[i].bad_row == PNG_ROW_REDO;
    if (n_redo) *n_redo = redo;
    EMU_LAUNCH((debig_png_defilter_kernel<16, 6>), n, 1024, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n,
               1u, (uint32_t *)nullptr, 1u);
    EMU_LAUNCH(debig_png_p3_kernel, n, PNG_P3_THREADS, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, results, n);
    free(gsync);
    return 0;
}
/* SURVEY.md 8(f) row 1, as debig_hip_png_decode_fused_batch launches it: plan, the fused kernel (scan | LZ77 | two de-filter
 * wavefronts per image), the one-kernel inflate for what the scan handed back, the one-workgroup de-filter for the images
 * of those streams, the P3 kernel.  *n_retried = streams handed back. */
extern "C" int emu_png_fused_batch(const void *in, void *streams_arena, const debig_stream *streams, debig_result *results,
                                   void *rgba_arena, const debig_png_image *images, debig_png_result *png_results, uint32_t n,
                                   uint64_t ws_bytes, uint32_t *n_retried)
{
    static uint32_t *ft = nullptr;
    if (!ft) {
        ft = (uint32_t *)calloc(1, sizeof(decltype(WaveLdsT<1>::t)));
        EMU_LAUNCH(debig_fixed_tables_kernel<1>, 1, 64, ft);
    }
    static uint32_t *fts = nullptr;
    if (!fts) {
        fts = (uint32_t *)calloc(1, sizeof(decltype(ScanLds::t)));
        EMU_LAUNCH(debig_scan_fixed_tables_kernel, 1, 64, fts);
    }
    const uint64_t slots_bytes = ((uint64_t)n * sizeof(debig_ws_slot) + 4u * SPLIT_QUEUE_WORDS + 4u * (uint64_t)n + 255) / 256 * 256;
    if (ws_bytes < slots_bytes + (uint64_t)n * 1024u) return -1;
    uint8_t *ws = (uint8_t *)malloc(ws_bytes);
    memset(ws, 0xEE, ws_bytes); /* poison: nothing may be read before it is written */
    const uint64_t rest = ws_bytes - slots_bytes;
    const uint64_t total_recs = rest / 16u / sizeof(debig_ws_rec);
    const uint64_t recs_bytes = (total_recs * sizeof(debig_ws_rec) + 255) / 256 * 256;
    const uint64_t total_rows = (rest - recs_bytes) / 256u;
    debig_ws_slot *slots = (debig_ws_slot *)ws;
    debig_ws_rec *recs = (debig_ws_rec *)(ws + slots_bytes);
    uint32_t *rows = (uint32_t *)(ws + slots_bytes + recs_bytes);
    EMU_LAUNCH(debig_split_plan_kernel, 1, EMU_PLAN_THREADS, streams, n, slots, total_rows, total_recs);
    EMU_LAUNCH((debig_png_fused_kernel<2, 6, 2, LzLdsBig>), n, 256, (const uint8_t *)in, (uint8_t *)streams_arena, streams, n, fts, slots, recs, rows,
               results, (uint8_t *)rgba_arena, images, png_results, 1u);
    uint32_t retried = 0;
    for (uint32_t i = 0; i < n; i++) retried += results[i].status == DEBIG_E_RETRY;
    if (n_retried) *n_retried = retried;
    EMU_LAUNCH(debig_inflate_kernel, n, 64, (const uint8_t *)in, (uint8_t *)streams_arena, streams, results, n, ft, DEBIG_CLASS_RETRY);
    EMU_LAUNCH(debig_png_defilter_kernel<8>, n, 512, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, png_results, n,
               1u, (uint32_t *)nullptr, 1u);
    EMU_LAUNCH(debig_png_p3_kernel, n, PNG_P3_THREADS, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images, png_results, n);
    free(ws);
    return 0;
}
extern "C" int emu_png_defilter_batch(const void *streams_arena, void *rgba_arena, const debig_png_image *images,
                                      debig_png_result *results, uint32_t n)
{
    return emu_png_defilter_batch_w(streams_arena, rgba_arena, images, results, n, 1);
}

extern "C" int emu_checksum_batch(const void *arena, const debig_span *spans, uint32_t *out, uint32_t n, uint32_t kind)
{
    static CkTables *t = nullptr;
    if (!t) {
        t = (CkTables *)calloc(1, sizeof(CkTables));
        EMU_LAUNCH(debig_checksum_tables_kernel, 1, CK_THREADS, t);
    }
    EMU_LAUNCH(debig_checksum_kernel, n, CK_THREADS, (const uint8_t *)arena, spans, out, n, kind, t);
    return 0;
}
