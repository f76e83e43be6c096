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

#endif
