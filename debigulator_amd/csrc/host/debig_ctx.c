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
}
