/* Per-thread_id device context shared by the drop-in entry points (host side, plain C). */
#ifndef DEBIG_CTX_H
#define DEBIG_CTX_H
#include <stdint.h>
#include <stddef.h>
#include "debig_hip.h"

#define DEBIG_MAX_THREADS 10 /* reference: INFLATE_MAX_THREADS / PNG_DECODER_MAX_THREADS */
#define DEBIG_API __attribute__((visibility("default")))

typedef struct debig_devbuf {
    void *ptr;
    uint64_t cap;
} debig_devbuf;

typedef struct debig_ctx {
    debig_devbuf in, out, desc, res, rgba, img, imgres;
    debig_devbuf files, spans, crcs, copies; /* PNG: whole files, chunk spans, their CRCs, IDAT gather list */
} debig_ctx;

debig_ctx *debig_ctx_get(uint32_t thread_id);
/* make sure b holds at least `bytes` (contents are not preserved); 0 on success */
int debig_devbuf_reserve(debig_devbuf *b, uint64_t bytes);
void debig_ctx_release(uint32_t thread_id);

static inline uint64_t debig_align16(uint64_t x) { return (x + 15u) & ~(uint64_t)15u; }

/* waves_per_stream for debig_hip_inflate_batch_ex from what the host knows about the batch
 * (the shim itself only sees n: descriptors live in device memory).  Few streams: several
 * wavefronts each.  A big batch in which only a few streams are large: those 4-wide beside
 * the small ones 1-wide, so the longest stream does not set the run time (measured:
 * profiles/r01_mw_sweep.txt). */
static inline uint32_t debig_pick_waves(const debig_stream *desc, uint32_t n)
{
    if (n <= 256u) return 8u;
    if (n <= 512u) return 4u;
    if (n <= 1024u) return 2u;
    uint32_t n_large = 0;
    for (uint32_t i = 0; i < n; i++)
        n_large += desc[i].in_len >= DEBIG_LARGE_IN_BYTES || desc[i].out_cap >= DEBIG_LARGE_OUT_BYTES;
    return (n_large != 0 && n_large <= 256u) ? DEBIG_WAVES_LARGE4_SMALL1 : 1u;
}

#endif
