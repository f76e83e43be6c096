/*
 * Drop-in init_decode_gz / decode_gz (reference src/decode_gz.h:23-38,
 * src/decode_gz.c:8-301): gzip header walk on the host, inflate on the GPU.
 * Same accept/reject rules as the reference's DECODE_GZ_SILENCE build:
 * magic 31,139 + CM 8; FNAME skipped; FCOMMENT / FEXTRA / FHCRC not skipped (G1);
 * first member only; the 8 trailer bytes are cut off, never verified.
 */
#include <stdlib.h>
#include <string.h>
#include "decode_gz.h"
#include "debig_ctx.h"

static void *(*g_malloc)(size_t);

DEBIG_API void init_decode_gz(void *(*malloc_funcptr)(size_t), void *(*arg_memset_func)(void *, int, size_t),
                              void *(*arg_memcpy_func)(void *, const void *, size_t))
{
    (void)arg_memset_func;
    (void)arg_memcpy_func;
    g_malloc = malloc_funcptr;
}

/* where the DEFLATE payload starts and the size decode_gz hands to inflate (src/decode_gz.c:123-270) */
static int gz_locate(const uint8_t *in, uint32_t in_size, uint32_t *off, uint32_t *len)
{
    if (in == NULL || in_size < 10) return 0;
    if (in[0] != 31 || in[1] != 139) return 0;
    if (in[2] != 8) return 0;
    uint32_t o = 10, left = in_size - 10;
    if ((in[3] >> 3) & 1u) { /* FNAME */
        uint32_t n = 0;
        while (o + n < in_size && in[o + n] != 0 && n < left) n++;
        o += n + 1;
        left -= n + 1;
    }
    *off = o;
    *len = left - 8u; /* uint32 arithmetic, as the reference */
    return 1;
}

int debig_inflate_batch_impl(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, const uint32_t thread_id, uint64_t *dev_out_offs);

DEBIG_API int debig_decode_gz_batch(const uint8_t *const *inputs, const uint32_t *input_sizes,
                                    uint8_t *const *outs, const uint64_t *out_caps, uint64_t *out_sizes,
                                    uint32_t *goods, uint32_t n)
{
    return debig_decode_gz_batch_ex(inputs, input_sizes, outs, out_caps, out_sizes, goods, NULL, n);
}

DEBIG_API int debig_decode_gz_batch_ex(const uint8_t *const *inputs, const uint32_t *input_sizes,
                                       uint8_t *const *outs, const uint64_t *out_caps, uint64_t *out_sizes,
                                       uint32_t *goods, uint32_t *trailer_ok, uint32_t n)
{
    const uint8_t **ins = (const uint8_t **)calloc(n ? n : 1, sizeof(*ins));
    uint64_t *offs = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    debig_span *spans = (debig_span *)calloc(n ? n : 1, sizeof(debig_span));
    uint32_t *crcs = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    uint64_t *lens = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    uint64_t *fin = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    uint64_t *caps = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    uint8_t **dst = (uint8_t **)calloc(n ? n : 1, sizeof(*dst));
    int rc = 2;
    if (!ins || !lens || !fin || !caps || !dst || !offs || !spans || !crcs) goto done;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t off = 0, len = 0;
        goods[i] = 0;
        out_sizes[i] = 0;
        if (gz_locate(inputs[i], input_sizes[i], &off, &len) && off <= input_sizes[i] &&
            (uint64_t)off + len <= input_sizes[i]) {
            ins[i] = inputs[i] + off;
            lens[i] = len;
            caps[i] = out_caps[i];
            dst[i] = outs[i];
        } else {
            ins[i] = NULL; /* NULL input: the batch call reports good = 0 for it */
            dst[i] = outs[i];
        }
    }
    rc = debig_inflate_batch_impl(dst, caps, fin, ins, lens, goods, n, 0, offs);
    for (uint32_t i = 0; i < n; i++) out_sizes[i] = goods[i] ? fin[i] : 0;
    if (!rc && trailer_ok) {
        /* what the reference reads and ignores (src/decode_gz.c:281-297): CRC-32 and ISIZE of the
         * member, checked against the decompressed bytes while they are still in HBM */
        debig_ctx *c = debig_ctx_get(0);
        for (uint32_t i = 0; i < n; i++) {
            spans[i].off = offs[i];
            spans[i].len = goods[i] ? fin[i] : 0;
            trailer_ok[i] = 0;
        }
        if ((rc = debig_devbuf_reserve(&c->spans, (uint64_t)n * sizeof(debig_span))) == 0 &&
            (rc = debig_devbuf_reserve(&c->crcs, (uint64_t)n * sizeof(uint32_t))) == 0 &&
            (rc = debig_hip_memcpy_h2d(c->spans.ptr, spans, (uint64_t)n * sizeof(debig_span), NULL)) == 0 &&
            (rc = debig_hip_checksum_batch(c->out.ptr, (const debig_span *)c->spans.ptr, (uint32_t *)c->crcs.ptr,
                                           n, 0, NULL)) == 0 &&
            (rc = debig_hip_memcpy_d2h(crcs, c->crcs.ptr, (uint64_t)n * sizeof(uint32_t), NULL)) == 0 &&
            (rc = debig_hip_stream_sync(NULL)) == 0) {
            for (uint32_t i = 0; i < n; i++) {
                if (!goods[i] || !ins[i]) continue;
                const uint8_t *t = ins[i] + lens[i]; /* the 8 trailer bytes follow the payload */
                uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                uint32_t isz = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                trailer_ok[i] = (crc == crcs[i]) && (isz == (uint32_t)fin[i]);
            }
        }
    }
done:
    free(offs);
    free(spans);
    free(crcs);
    free(ins);
    free(lens);
    free(fin);
    free(caps);
    free(dst);
    return rc;
}

DEBIG_API DecodedData *decode_gz(uint8_t *compressed_bytes, uint32_t compressed_bytes_size)
{
    if (g_malloc == NULL) return NULL; /* src/decode_gz.c:105-113 */
    DecodedData *r = (DecodedData *)g_malloc(sizeof(DecodedData));
    if (!r) return NULL;
    r->data = NULL;
    r->data_size = 0;
    r->good = 0;
    uint32_t off = 0, len = 0;
    if (!gz_locate(compressed_bytes, compressed_bytes_size, &off, &len)) return r;
    /* the reference's own sizing of the output buffer (src/decode_gz.c:245), uint32 wrap included;
     * it is also the recipient_size the inflate gates see */
    uint32_t left = len + 8u;
    uint32_t guess = left * 35u + 1000000u;
    if (off > compressed_bytes_size || (uint64_t)off + len > compressed_bytes_size) return r;
    uint8_t *recipient = (uint8_t *)g_malloc(guess);
    if (!recipient) return r;
    uint64_t fin = 0;
    uint32_t good = 0;
    debig_inflate(recipient, guess, &fin, NULL, 0, compressed_bytes + off, len, &good, 0);
    if (!good) return r; /* the reference leaks `recipient` here too */
    r->data = (char *)recipient;
    r->data_size = (uint32_t)fin;
    r->good = 1;
    return r;
}
