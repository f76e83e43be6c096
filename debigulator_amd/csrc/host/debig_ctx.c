#include "debig_ctx.h"

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

void debig_ctx_release(uint32_t thread_id)
{
    debig_ctx *c = debig_ctx_get(thread_id);
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
    if (w == DEBIG_WAVES_SPLIT) { /* the throughput path wants a token workspace sized from the input */
        uint64_t total_in = 0;
        for (uint32_t i = 0; i < n; i++) total_in += desc[i].in_len;
        ws_bytes = debig_hip_inflate_workspace_bytes(total_in, n);
        if (debig_devbuf_reserve(&c->ws, ws_bytes)) ws_bytes = 0; /* no memory: the library's own fallback */
    }
    if ((rc = debig_devbuf_reserve(&c->desc, (uint64_t)n * sizeof(debig_stream))) ||
        (rc = debig_devbuf_reserve(&c->res, (uint64_t)n * sizeof(debig_result))) ||
        (rc = debig_hip_memcpy_h2d(c->desc.ptr, up, (uint64_t)n * sizeof(debig_stream), NULL)) ||
        (rc = debig_hip_inflate_batch_ws(d_in_arena, c->out.ptr, (const debig_stream *)c->desc.ptr,
                                         (debig_result *)c->res.ptr, n, w, ws_bytes ? c->ws.ptr : NULL, ws_bytes, NULL)) ||
        (rc = debig_hip_memcpy_d2h(down, c->res.ptr, (uint64_t)n * sizeof(debig_result), NULL)) ||
        (rc = debig_hip_stream_sync(NULL))) {
        /* fall through to the cleanup */
    } else if (permuted) {
        for (uint32_t k = 0; k < n; k++) res[order[k]] = tmp[k];
    }
    free(order);
    free(sorted);
    free(tmp);
    return rc;
}
