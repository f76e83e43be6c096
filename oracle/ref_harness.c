/*
 * TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * Thin C harness around the *unmodified* reference sources, which the
 * oracle/Makefile compiles in place from /root/reference/src (nothing is
 * copied into this repo).  It exists so tests can drive the reference through
 * ctypes with a buffer layout that keeps it out of undefined behaviour
 * (SURVEY.md Appendix A):
 *
 *   arena = [ recipient (recipient_size) | 1 KiB slack | scratch 4 MiB ]
 *
 *   - recipient sits BELOW the scratch (reference src/inflate.c:1877, quirk Q11)
 *   - >= 774 bytes of slack behind the recipient (4x over-copy, inflate.c:1862)
 *   - input is copied into a buffer padded with zero bytes (peek_bits reads 4
 *     bytes at the cursor, inflate.c:252-256)
 *
 * The prototypes below are the reference's own (src/inflate.h:22-60,
 * src/decode_png.h:43-103, src/decode_gz.h:23-38).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* reference API (src/inflate.h) */
void inflate_init(void *(*m)(uint64_t), void *(*ms)(void *, int, uint64_t),
                  void *(*mc)(void *, const void *, uint64_t), const uint32_t thread_id);
void inflate(uint8_t const *recipient, const uint64_t recipient_size,
             uint64_t *final_recipient_size, uint8_t *temp_working_memory,
             const uint64_t temp_working_memory_size, uint8_t const *compressed_input,
             const uint64_t compressed_input_size, uint32_t *out_good,
             const uint32_t thread_id);
/* reference API (src/decode_png.h) */
void decode_png_init(void *(*m)(uint64_t), void (*f)(void *),
                     void *(*ms)(void *, int, uint64_t),
                     void *(*mc)(void *, const void *, uint64_t),
                     const uint32_t wm_size, const uint32_t thread_id);
void decode_png_deinit(const uint32_t thread_id);
void decode_png_get_width_height(const uint8_t *in, const uint64_t in_size, uint32_t *w,
                                 uint32_t *h, uint8_t *good);
void decode_png(const uint8_t *in, const uint64_t in_size, const uint8_t *out_rgba,
                const uint64_t rgba_size, const uint32_t thread_id, uint8_t *out_good);

/* reference API (src/decode_gz.h), built through gz_shim.h (old 3-/8-arg inflate calls) */
typedef struct DecodedData {
    char *data;
    uint32_t data_size;
    uint32_t good;
} DecodedData;
void init_decode_gz(void *(*m)(size_t), void *(*ms)(void *, int, size_t),
                    void *(*mc)(void *, const void *, size_t));
DecodedData *decode_gz(uint8_t *compressed_bytes, uint32_t compressed_bytes_size);

static void *m64(uint64_t n) { return malloc((size_t)n); }
static void *ms64(void *p, int c, uint64_t n) { return memset(p, c, (size_t)n); }
static void *mc64(void *d, const void *s, uint64_t n) { return memcpy(d, s, (size_t)n); }

static int g_inflate_inited[10];
int refh_inflate_inited(uint32_t tid) { return g_inflate_inited[tid]; }
void refh_mark_inflate_inited(uint32_t tid) { g_inflate_inited[tid] = 1; }

static void ensure_inflate_init(uint32_t tid)
{
    if (!g_inflate_inited[tid]) {
        inflate_init(m64, ms64, mc64, tid);
        g_inflate_inited[tid] = 1;
    }
}

#define REFH_SLACK 1024u
#define REFH_SCRATCH (4u << 20)

/* Raw inflate() through the safe arena.  out must hold recipient_size bytes.
 * Returns good; *final gets final_recipient_size (0xFFFFFFFFFFFFFFFF if the
 * reference left it untouched, i.e. a gate failure). */
uint32_t refh_inflate(const uint8_t *in, uint64_t in_size, uint8_t *out,
                      uint64_t recipient_size, uint64_t *final, uint64_t scratch_size)
{
    ensure_inflate_init(0);
    if (scratch_size == 0) scratch_size = REFH_SCRATCH;
    uint8_t *arena = (uint8_t *)malloc(recipient_size + REFH_SLACK + scratch_size + 64);
    uint8_t *inbuf = (uint8_t *)calloc(in_size + 16, 1);
    memcpy(inbuf, in, in_size);
    memset(arena, 0, recipient_size + REFH_SLACK);
    uint8_t *scratch = arena + recipient_size + REFH_SLACK;
    uint32_t good = 0xdeadbeef;
    uint64_t fin = ~(uint64_t)0;
    inflate(arena, recipient_size, &fin, scratch, scratch_size, inbuf, in_size, &good, 0);
    if (fin != ~(uint64_t)0) {
        uint64_t n = fin < recipient_size ? fin : recipient_size;
        memcpy(out, arena, n);
    }
    *final = fin;
    free(arena);
    free(inbuf);
    return good;
}

static uint32_t g_png_wm[10];

/* decode_png() with a per-call private copy of the input (the reference packs
 * the IDAT payloads to the front of the caller's buffer, decode_png.c:1285). */
uint32_t refh_decode_png(const uint8_t *in, uint64_t in_size, uint8_t *out_rgba,
                         uint64_t rgba_size, uint32_t wm_size, uint32_t tid)
{
    if (g_png_wm[tid] == 0) {
        /* decode_png_init calls inflate_init(tid), which asserts on re-init in
         * asserts-on builds (inflate.c:47): one init per thread id, ever. */
        if (g_inflate_inited[tid]) return 9;
        decode_png_init(m64, free, ms64, mc64, wm_size, tid);
        g_inflate_inited[tid] = 1;
        g_png_wm[tid] = wm_size;
    } else if (g_png_wm[tid] != wm_size) {
        return 9; /* harness: pick another thread id for another scratch size */
    }
    uint8_t *inbuf = (uint8_t *)calloc(in_size + 64, 1);
    memcpy(inbuf, in, in_size);
    uint8_t good = 7;
    decode_png(inbuf, in_size, out_rgba, rgba_size, tid, &good);
    free(inbuf);
    return good;
}

void refh_png_wh(const uint8_t *in, uint64_t in_size, uint32_t *w, uint32_t *h, uint8_t *good)
{
    decode_png_get_width_height(in, in_size, w, h, good);
}

static int g_gz_inited;
static void *msz(size_t n) { return calloc(n, 1); } /* DecodedData.good is left unset on failure (decode_gz.c:277) */

/* decode_gz(): the reference never sets data_size (decode_gz.c:299-300), so the
 * harness cannot learn the length from it; callers pass the expected maximum and
 * compare a prefix.  Returns good (0/1) or 2 when the reference returned NULL;
 * copies up to out_cap bytes of DecodedData.data into out. The 50 MB scratch and
 * the output buffer the reference mallocs are leaked by the reference itself;
 * the harness frees data. */
uint32_t refh_decode_gz(const uint8_t *in, uint32_t in_size, uint8_t *out, uint64_t out_cap)
{
    if (!g_gz_inited) {
        init_decode_gz(msz, (void *(*)(void *, int, size_t))memset,
                       (void *(*)(void *, const void *, size_t))memcpy);
        g_gz_inited = 1;
        g_inflate_inited[0] = 1;
    }
    uint8_t *inbuf = (uint8_t *)calloc((size_t)in_size + 64, 1);
    memcpy(inbuf, in, in_size);
    DecodedData *dd = decode_gz(inbuf, in_size);
    if (!dd) { free(inbuf); return 2; }
    uint32_t good = dd->good;
    if (good == 1 && dd->data) {
        /* guess_decompressed_size in the reference: left*35 + 1,000,000 */
        memcpy(out, dd->data, out_cap);
        free(dd->data);
    }
    free(dd);
    free(inbuf);
    return good;
}
