/*
 * Drop-in inflate_init / inflate_destroy / inflate (reference src/inflate.h:22-60,
 * src/inflate.c:40-64, :786-1965) on top of the batched GPU path.  Host side, plain C:
 * argument gates, staging through device arenas, one kernel launch, results copied back
 * into the caller's buffers.  There is no CPU decode path in this library.
 */
#include <stdlib.h>
#include <string.h>
#include "inflate.h"
#include "debig_ctx.h"

static int g_inited[DEBIG_MAX_THREADS];

DEBIG_API void inflate_init(void *(*malloc_funcptr)(uint64_t), void *(*arg_memset_func)(void *, int, uint64_t),
                            void *(*arg_memcpy_func)(void *, const void *, uint64_t), const uint32_t thread_id)
{
    /* the reference allocates an InflateState per thread id and stores memset/memcpy in
     * process globals (src/inflate.c:40-56); nothing of that is needed here */
    (void)malloc_funcptr;
    (void)arg_memset_func;
    (void)arg_memcpy_func;
    if (thread_id < DEBIG_MAX_THREADS) g_inited[thread_id] = 1;
}

DEBIG_API void inflate_destroy(void (*free_funcptr)(void *), const uint32_t thread_id)
{
    (void)free_funcptr; /* reference impl frees interior pointers (a bug, src/inflate.c:58-64) */
    if (thread_id < DEBIG_MAX_THREADS) {
        g_inited[thread_id] = 0;
        debig_ctx_release(thread_id);
    }
}

/* shared with debig_gz.c: same as debig_inflate_batch, and reports where each recipient lives
 * in the context's device output arena (for checks that run on the device afterwards) */
int debig_inflate_batch_impl(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, const uint32_t thread_id, uint64_t *dev_out_offs);

DEBIG_API int debig_inflate_batch(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                                  const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                                  uint32_t n, const uint32_t thread_id)
{
    return debig_inflate_batch_impl(outs, out_caps, finals, ins, in_sizes, goods, n, thread_id, NULL);
}

int debig_inflate_batch_impl(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, const uint32_t thread_id, uint64_t *dev_out_offs)
{
    debig_ctx *c = debig_ctx_get(thread_id);
    for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    if (!c) return 1;
    if (n == 0) return 0;
    debig_stream *desc = (debig_stream *)calloc(n, sizeof(debig_stream));
    debig_result *res = (debig_result *)calloc(n, sizeof(debig_result));
    uint8_t *skip = (uint8_t *)calloc(n, 1);
    if (!desc || !res || !skip) { free(desc); free(res); free(skip); return 2; }
    uint64_t in_total = 0, out_total = 0;
    for (uint32_t i = 0; i < n; i++) {
        /* NULL gates, in the reference's order (src/inflate.c:797-824): nothing is written */
        if (!outs[i] || !finals || !ins[i]) { skip[i] = 1; continue; }
        desc[i].in_off = in_total;
        desc[i].in_len = in_sizes[i];
        desc[i].out_off = out_total;
        if (dev_out_offs) dev_out_offs[i] = out_total;
        desc[i].out_cap = out_caps[i];
        /* a stream that fails the size gates is never read or written by the kernel */
        int gated = out_caps[i] < in_sizes[i] || in_sizes[i] < 5;
        in_total += debig_align16(gated ? 0 : in_sizes[i]) + 16;
        out_total += debig_align16(gated ? 0 : out_caps[i]) + 16;
    }
    int rc = 0;
    if ((rc = debig_devbuf_reserve(&c->in, in_total + 64)) || (rc = debig_devbuf_reserve(&c->out, out_total + 64)) ||
        (rc = debig_devbuf_reserve(&c->desc, (uint64_t)n * sizeof(debig_stream))) ||
        (rc = debig_devbuf_reserve(&c->res, (uint64_t)n * sizeof(debig_result))))
        goto done;
    for (uint32_t i = 0; i < n && !rc; i++) {
        if (skip[i] || out_caps[i] < in_sizes[i] || in_sizes[i] < 5) continue;
        rc = debig_hip_memcpy_h2d((uint8_t *)c->in.ptr + desc[i].in_off, ins[i], in_sizes[i], NULL);
    }
    if (rc) goto done;
    /* streams with NULL arguments are still launched as zero-length (they fail the gates) */
    for (uint32_t i = 0; i < n; i++)
        if (skip[i]) { desc[i].in_len = 0; desc[i].out_cap = 0; }
    if ((rc = debig_launch_inflate_planned(c, c->in.ptr, desc, res, n))) goto done;
    for (uint32_t i = 0; i < n && !rc; i++) {
        if (skip[i]) continue;
        if (res[i].final_set) {
            finals[i] = res[i].final_size;
            uint64_t nb = res[i].final_size < out_caps[i] ? res[i].final_size : out_caps[i];
            if (nb) rc = debig_hip_memcpy_d2h(outs[i], (uint8_t *)c->out.ptr + desc[i].out_off, nb, NULL);
        }
        goods[i] = res[i].good;
    }
    if (!rc) rc = debig_hip_stream_sync(NULL);
done:
    if (rc)
        for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    free(desc);
    free(res);
    free(skip);
    return rc;
}

DEBIG_API void debig_inflate(uint8_t const *recipient, const uint64_t recipient_size, uint64_t *final_recipient_size,
                             uint8_t *temp_working_memory, const uint64_t temp_working_memory_size,
                             uint8_t const *compressed_input, const uint64_t compressed_input_size,
                             uint32_t *out_good, const uint32_t thread_id)
{
    (void)temp_working_memory;
    (void)temp_working_memory_size;
    /* reference gate order: recipient, final_recipient_size, compressed_input (src/inflate.c:797-824) */
    if (recipient == NULL || final_recipient_size == NULL || compressed_input == NULL) {
        *out_good = 0;
        return;
    }
    uint8_t *out = (uint8_t *)recipient;
    uint64_t fin = *final_recipient_size;
    uint32_t good = 0;
    const uint8_t *in = compressed_input;
    int rc = debig_inflate_batch(&out, &recipient_size, &fin, &in, &compressed_input_size, &good, 1, thread_id);
    if (rc) good = 0;
    *final_recipient_size = fin;
    *out_good = good;
}
