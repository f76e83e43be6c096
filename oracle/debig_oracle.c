/*
 * TEST INFRASTRUCTURE ONLY -- see debig_oracle.h.
 *
 * CPU restatement of the reference's observable behaviour, written from
 * SURVEY.md Appendix E and the reference sources as cited per function.  It is
 * NOT a copy: the reference decodes with a 792 KB hashed table and a 7-bit bit
 * buffer; this file uses a 64-bit bit buffer and a direct 15-bit lookup table,
 * and reproduces the reference's accept/reject rules and quirks on top.
 */
#include "debig_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ bits */

typedef struct {
    const uint8_t *in;
    uint64_t in_size;
    uint64_t bitpos; /* bits consumed so far (reference: data cursor + bit_buffer) */
} bitrd;

/* bytes past in_size read as zero (reference peek_bits reads 4 bytes at the
 * cursor unconditionally, src/inflate.c:252-256) */
static inline uint64_t rd_bytes8(const bitrd *b, uint64_t byte)
{
    uint64_t v = 0;
    if (byte + 8 <= b->in_size) {
        memcpy(&v, b->in + byte, 8);
        return v;
    }
    for (int i = 0; i < 8; i++)
        if (byte + (uint64_t)i < b->in_size) v |= (uint64_t)b->in[byte + i] << (8 * i);
    return v;
}

/* LSB-first, up to 32 bits (src/inflate.c:225-278) */
static inline uint32_t peek(const bitrd *b, uint32_t n)
{
    uint64_t v = rd_bytes8(b, b->bitpos >> 3) >> (b->bitpos & 7);
    return (uint32_t)(v & ((n >= 32) ? 0xffffffffull : ((1ull << n) - 1)));
}
static inline void drop(bitrd *b, uint32_t n) { b->bitpos += n; }
static inline uint32_t take(bitrd *b, uint32_t n)
{
    uint32_t v = peek(b, n);
    drop(b, n);
    return v;
}

static inline uint32_t bitrev(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

/* -------------------------------------------------------------- Huffman */

#define ORC_MAXBITS 15
typedef struct {
    uint16_t tab[1 << ORC_MAXBITS]; /* (symbol << 4) | length; 0 = no code */
    uint32_t qmin, qmax;            /* the reference's probe range (Q7)    */
    uint32_t long_slots;            /* linear-list slots in use (Q8)       */
} orc_code;

/* Restates unpack_huffman (src/inflate.c:565-706) + huffman_to_hashmap
 * (:494-557) + the probe loop of hashed_huffman_decode (:421-474):
 *   - any length >= n  => failure (Q6, :599-602)
 *   - canonical codes per RFC 1951 3.2.2 for every symbol whose length is
 *     >= the smallest non-zero length (:685)
 *   - probe range [qmin,qmax] follows the reference's if / else-if update
 *     (:528-539), so a code that sets a new minimum never raises the maximum
 *   - lookup = the SHORTEST length in [qmin,qmax] whose low bits equal a stored
 *     (length, reversed code); same (length,code) stored twice => last wins.
 * Returns 0 on the Q6 failure. */
static int build_code(const uint32_t *lens, uint32_t n, orc_code *c, uint32_t *ub)
{
    uint32_t bl_count[19];
    uint32_t next[19];
    uint32_t code_of[320];
    uint8_t used[320];
    uint32_t minlen = 123454321u, maxlen = 0;
    memset(bl_count, 0, sizeof bl_count);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t l = lens[i];
        if (l >= n) return 0;
        if (l > maxlen) maxlen = l;
        if (l < minlen && l > 0) minlen = l;
        bl_count[l]++;
    }
    uint32_t code = 0;
    bl_count[0] = 0;
    memset(next, 0, sizeof next);
    for (uint32_t bits = 1; bits <= maxlen; bits++) {
        code = (code + bl_count[bits - 1]) << 1;
        next[bits] = code;
        if (code >= (1u << bits) && bl_count[bits] > 0) *ub |= ORC_UB_OVERSUBSCRIBED;
    }
    for (uint32_t i = 0; i < n; i++) {
        uint32_t l = lens[i];
        used[i] = 0;
        if (l >= minlen) {
            code_of[i] = next[l]++;
            used[i] = 1;
            if (code_of[i] >= (1u << l)) *ub |= ORC_UB_OVERSUBSCRIBED;
        }
    }
    c->qmin = 9999;
    c->qmax = 1;
    c->long_slots = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!used[i]) continue;
        if (lens[i] < c->qmin) c->qmin = lens[i];
        else if (lens[i] > c->qmax) c->qmax = lens[i];
        if (lens[i] > 12) c->long_slots++;
    }
    memset(c->tab, 0, sizeof c->tab);
    /* longest first so that shorter codes overwrite: shortest match wins */
    for (uint32_t l = ORC_MAXBITS; l >= 1; l--) {
        if (l < c->qmin || l > c->qmax) continue;
        for (uint32_t i = 0; i < n; i++) {
            if (!used[i] || lens[i] != l) continue;
            uint32_t r = bitrev(code_of[i] & ((1u << l) - 1), l);
            for (uint32_t k = r; k < (1u << ORC_MAXBITS); k += (1u << l))
                c->tab[k] = (uint16_t)((i << 4) | l);
        }
    }
    return 1;
}

/* returns symbol, or -1 when no code matches (reference: *good = 0) */
static inline int decode_sym(const orc_code *c, bitrd *b, uint32_t *ub_long_probe)
{
    uint32_t e = c->tab[peek(b, ORC_MAXBITS)];
    uint32_t l = e & 15u;
    if (l == 0) return -1;
    if (l > 13 && ub_long_probe) {
        uint32_t lo = c->qmin > 13 ? c->qmin : 13;
        if (l > lo) *ub_long_probe += l - lo; /* failed 13/14-bit probes append slots (Q8) */
    }
    drop(b, l);
    return (int)(e >> 4);
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35,
                                      43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                      3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                       33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                       1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6,
                                       6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* --------------------------------------------------------------- inflate */

uint32_t orc_inflate_ex(const uint8_t *in, uint64_t in_size, uint8_t *out,
                        uint64_t recipient_size, uint64_t *final, uint32_t *final_set,
                        orc_header_hook hook, void *hook_ctx, orc_stats *stats)
{
    orc_stats st_local;
    orc_stats *st = stats ? stats : &st_local;
    memset(st, 0, sizeof *st);
    if (final_set) *final_set = 0;
    /* argument gates, in the reference's order (src/inflate.c:797-844, Q1) */
    if (out == NULL) return 0;
    if (final == NULL) return 0;
    if (in == NULL) return 0;
    if (recipient_size < in_size) return 0;
    if (in_size < 5) return 0;
    *final = 0; /* :852 */
    if (final_set) *final_set = 1;

    orc_code *litlen = (orc_code *)malloc(sizeof(orc_code));
    orc_code *dist = (orc_code *)malloc(sizeof(orc_code));
    orc_code *clc = (orc_code *)malloc(sizeof(orc_code));
    uint32_t good = 0;
    uint64_t pos = 0;
    bitrd b = {in, in_size, 0};
    int more = 1;

    while (more) { /* :870 */
        uint32_t bfinal = take(&b, 1);
        if (bfinal) more = 0;
        uint32_t btype = take(&b, 2);
        st->n_blocks++;
        if (btype == 0) { /* stored, :919-989 */
            st->n_stored++;
            b.bitpos = (b.bitpos + 7) & ~7ull;
            uint32_t len = take(&b, 16);
            uint32_t nlen = take(&b, 16);
            if ((uint16_t)len != (uint16_t)~nlen) goto done; /* Q4 */
            if (pos + len > recipient_size) goto done;       /* clean failure instead of overflow */
            uint64_t byte = b.bitpos >> 3;
            for (uint32_t i = 0; i < len; i++)
                out[pos + i] = (byte + i < in_size) ? in[byte + i] : 0;
            if (byte + len > in_size) st->ub_flags |= ORC_UB_INPUT_OVERRUN;
            pos += len;
            *final = pos;
            b.bitpos += 8ull * len;
            continue;
        }
        if (btype == 3) { /* :990-998: asserts-off build skips the block (Q5) */
            st->ub_flags |= ORC_UB_BTYPE3;
            continue;
        }
        int have_dist = 0;
        if (btype == 1) { /* fixed, :1018-1181 */
            st->n_fixed++;
            uint32_t lens[288];
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            if (hook) hook(hook_ctx, 1, pos, NULL, out);
            if (!build_code(lens, 288, litlen, &st->ub_flags)) goto done;
        } else { /* dynamic, :1182-1667 */
            st->n_dynamic++;
            uint32_t hlit = take(&b, 5) + 257;
            uint32_t hdist = take(&b, 5) + 1;
            uint32_t hclen = take(&b, 4) + 4;
            if (hlit > 286) st->ub_flags |= ORC_UB_HLIT_GT_286;
            uint32_t cl_lens[19];
            memset(cl_lens, 0, sizeof cl_lens);
            for (uint32_t i = 0; i < hclen; i++) cl_lens[CL_ORDER[i]] = take(&b, 3);
            if (hook) hook(hook_ctx, 2, pos, cl_lens, out);
            if (!build_code(cl_lens, 19, clc, &st->ub_flags)) goto done;
            uint32_t lens[320 + 140];
            uint32_t total = hlit + hdist, i = 0;
            memset(lens, 0, sizeof lens);
            while (i < total) { /* :1416-1520 */
                int s = decode_sym(clc, &b, NULL);
                if (s < 0) goto done;
                if (s <= 15) {
                    lens[i++] = (uint32_t)s;
                } else if (s == 16) {
                    uint32_t rep = take(&b, 2) + 3;
                    uint32_t prev = 0;
                    if (i == 0) st->ub_flags |= ORC_UB_CL16_AT_0;
                    else prev = lens[i - 1];
                    for (uint32_t k = 0; k < rep; k++) lens[i + k] = prev;
                    i += rep;
                } else if (s == 17) {
                    uint32_t rep = take(&b, 3) + 3;
                    for (uint32_t k = 0; k < rep; k++) lens[i + k] = 0;
                    i += rep;
                } else {
                    uint32_t rep = take(&b, 7) + 11;
                    for (uint32_t k = 0; k < rep; k++) lens[i + k] = 0;
                    i += rep;
                }
            }
            if (i > total) st->ub_flags |= ORC_UB_CL_OVERSHOOT;
            if (!build_code(lens, hlit, litlen, &st->ub_flags)) goto done;       /* :1543 */
            if (!build_code(lens + hlit, hdist, dist, &st->ub_flags)) goto done; /* :1615, Q6 */
            have_dist = 1;
        }
        uint32_t long_probes_ll = litlen->long_slots, long_probes_d = have_dist ? dist->long_slots : 0;
        for (;;) { /* symbol loop, :1697-1909 */
            /* Q2 tail gate (:1702-1717): (data - input) >= size, data = ceil(bits/8) */
            if (((b.bitpos + 7) >> 3) >= in_size) {
                more = 0;
                st->tail_gate_fired = 1;
                break;
            }
            int s = decode_sym(litlen, &b, &long_probes_ll);
            if (s < 0) goto done;
            st->n_symbols++;
            if (s < 256) {
                if (pos + 1 > recipient_size) goto done;
                out[pos++] = (uint8_t)s;
                *final = pos;
            } else if (s == 256) {
                break;
            } else {
                if (s > 285) { /* Q9: the reference indexes past its table */
                    st->ub_flags |= ORC_UB_SYM_286_287;
                    goto done;
                }
                uint32_t li = (uint32_t)s - 257;
                uint32_t len = LEN_BASE[li] + (LEN_EXTRA[li] ? take(&b, LEN_EXTRA[li]) : 0);
                uint32_t ds;
                if (!have_dist) {
                    ds = bitrev(take(&b, 5), 5); /* :1783-1788 */
                } else {
                    int d = decode_sym(dist, &b, &long_probes_d);
                    if (d < 0) goto done;
                    ds = (uint32_t)d;
                }
                if (ds > 29) goto done; /* :1809 */
                uint32_t dd = DIST_BASE[ds] + (DIST_EXTRA[ds] ? take(&b, DIST_EXTRA[ds]) : 0);
                if (dd > pos) goto done; /* :1843, final stays at the partial count (Q10) */
                if (pos + len > recipient_size) goto done;
                for (uint32_t k = 0; k < len; k++) out[pos + k] = out[pos + k - dd];
                pos += len;
                *final = pos;
                st->n_matches++;
            }
        }
        if (long_probes_ll >= 500 || long_probes_d >= 500) st->ub_flags |= ORC_UB_LONGCODE_500;
    }
    good = 1;
    if (((b.bitpos + 7) >> 3) > in_size) st->ub_flags |= ORC_UB_INPUT_OVERRUN;
done:
    st->bits_consumed = b.bitpos;
    free(litlen);
    free(dist);
    free(clc);
    return good;
}

uint32_t orc_inflate(const uint8_t *in, uint64_t in_size, uint8_t *out,
                     uint64_t recipient_size, uint64_t *final)
{
    uint32_t fs;
    return orc_inflate_ex(in, in_size, out, recipient_size, final, &fs, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------ CRC */

static uint32_t g_crc_table[256];
static int g_crc_ready;
static void crc_init(void)
{
    for (uint32_t n = 0; n < 256; n++) { /* same generator the reference checks its table with,
                                            src/decode_png.c:289-305 */
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        g_crc_table[n] = c;
    }
    g_crc_ready = 1;
}
uint32_t orc_crc32(uint32_t crc, const uint8_t *buf, uint64_t len)
{
    if (!g_crc_ready) crc_init();
    for (uint64_t i = 0; i < len; i++) crc = g_crc_table[(crc ^ buf[i]) & 0xff] ^ (crc >> 8);
    return crc;
}

/* ------------------------------------------------------------------ PNG */

static inline uint32_t be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

void orc_png_get_width_height(const uint8_t *in, uint64_t in_size, uint32_t *w, uint32_t *h,
                              uint8_t *good)
{ /* src/decode_png.c:620-681 */
    *w = 0;
    *h = 0;
    *good = 0;
    if (in_size < 28) return;
    if (in[1] != 'P' || in[2] != 'N' || in[3] != 'G') return;
    *w = be32(in + 16);
    *h = be32(in + 20);
    *good = 1;
}

static struct {
    uint8_t red[256], green[256], blue[256];
    uint32_t size;
} g_palette; /* PNGDecoderThreadState.palette persists across calls, decode_png.c:543-557 */

void orc_png_reset_palette(void) { memset(&g_palette, 0, sizeof g_palette); }

static inline uint8_t paeth(int32_t a, int32_t b, int32_t c)
{ /* src/decode_png.c:441-487 */
    int32_t p = a + b - c;
    int32_t pa = p > a ? p - a : a - p;
    int32_t pb = p > b ? p - b : b - p;
    int32_t pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return (uint8_t)a;
    if (pb <= pc) return (uint8_t)b;
    return (uint8_t)c;
}

typedef struct {
    uint64_t est;  /* recipient_size handed to inflate = 4wh + h + 1 */
    int64_t s0;    /* stream byte aliased by the first table byte (may be < 0 for tiny images) */
    int enabled;
} p2_ctx;

/* P2: recipient = wm + 772 with size est, scratch = wm +
 * est, so stream byte s and scratch byte t alias when s = est - 772 + t.  Each
 * Huffman block header writes its first table at the 16-aligned scratch start
 * over bytes already decoded.  Entry layout restated from src/inflate.c:96-101:
 * {u32 code; u32 symbol; u32 used; u16 length; 2 bytes untouched}. */
static void p2_hook(void *vctx, uint32_t btype, uint64_t out_pos, const uint32_t *cl_lens,
                    uint8_t *out)
{
    p2_ctx *c = (p2_ctx *)vctx;
    if (!c->enabled || (int64_t)out_pos <= c->s0) return;
    uint64_t end = out_pos < c->est ? out_pos : c->est;
    uint32_t lens[288], codes[288], used[288], n;
    if (btype == 1) {
        n = 288;
        for (uint32_t i = 0; i < 288; i++) {
            used[i] = 1;
            if (i < 144) { lens[i] = 8; codes[i] = 48 + i; }
            else if (i < 256) { lens[i] = 9; codes[i] = 400 + (i - 144); }
            else if (i < 280) { lens[i] = 7; codes[i] = i - 256; }
            else { lens[i] = 8; codes[i] = 192 + (i - 280); }
        }
    } else {
        n = 19;
        uint32_t bl[19], next[19], minlen = 123454321u, maxlen = 0;
        memset(bl, 0, sizeof bl);
        memset(next, 0, sizeof next);
        for (uint32_t i = 0; i < 19; i++) {
            lens[i] = cl_lens[i];
            if (lens[i] > maxlen) maxlen = lens[i];
            if (lens[i] && lens[i] < minlen) minlen = lens[i];
            bl[lens[i]]++;
        }
        bl[0] = 0;
        uint32_t code = 0;
        for (uint32_t bits = 1; bits <= maxlen; bits++) {
            code = (code + bl[bits - 1]) << 1;
            next[bits] = code;
        }
        for (uint32_t i = 0; i < 19; i++) {
            if (lens[i] >= minlen) { codes[i] = next[lens[i]]++; used[i] = 1; }
            else { codes[i] = 1234543u; used[i] = 0; } /* src/inflate.c:578-583 */
        }
    }
    for (uint64_t s = c->s0 > 0 ? (uint64_t)c->s0 : 0; s < end; s++) {
        uint64_t t = (uint64_t)((int64_t)s - c->s0);
        uint64_t e = t >> 4, k = t & 15;
        if (e >= n) { out[s] = 0; continue; } /* memset of the following HashedHuffman, :483 */
        if (k >= 14) continue;                /* struct padding: never written */
        uint32_t word = k < 4 ? codes[e] : k < 8 ? (uint32_t)e : k < 12 ? used[e] : lens[e];
        out[s] = (uint8_t)(word >> (8 * (k & 3)));
    }
}

uint32_t orc_decode_png(const uint8_t *in_const, uint64_t in_size, uint8_t *out_rgba,
                        uint64_t rgba_size, uint32_t wm_size, uint32_t flags)
{ /* src/decode_png.c:683-1567 */
    if (in_size < 8) return 0;
    uint8_t *in = (uint8_t *)calloc(in_size + 64, 1);
    memcpy(in, in_const, in_size);
    uint8_t *stream = NULL;
    uint32_t good = 0;
    uint64_t left = in_size;
    uint64_t at = 0;    /* read cursor */
    uint64_t packed = 0; /* IDAT payload packed to the front of the buffer (P1) */
    uint32_t found_idat = 0, ran_inflate = 0, found_ihdr = 0, found_iend = 0;
    uint32_t w = 0, h = 0, ct = 0;
    uint64_t est = 0, actual = 0;

    if (in[1] != 'P' || in[2] != 'N' || in[3] != 'G') goto out;
    at = 8;
    left -= 8;
    while (left >= 8 && !found_iend) {
        if (at + 8 > in_size) goto out; /* true bounds (reference: UB) */
        uint32_t len = be32(in + at);
        const uint8_t *type = in + at + 4;
        at += 8;
        left -= 8;
        int is_idat = !memcmp(type, "IDAT", 4);
        if (!is_idat && found_idat) { /* :775-860 */
            uint64_t csize = (uint64_t)(uint32_t)((uint32_t)packed - 4u);
            free(stream);
            stream = (uint8_t *)calloc(est + 1024, 1);
            p2_ctx pc;
            pc.est = est;
            pc.enabled = !(flags & ORC_PNG_STRICT);
            /* tables start at the first 16-aligned scratch byte; wm itself is 16-aligned */
            pc.s0 = (int64_t)est - 772 + (int64_t)((16 - (est & 15)) & 15);
            uint32_t fs = 0;
            uint32_t ok = orc_inflate_ex(in, csize, stream, est, &actual, &fs, p2_hook, &pc, NULL);
            ran_inflate = 1;
            if (!ok) goto out;
            if (stream[0] > 4) goto out; /* :847-858 */
        }
        if (at + (uint64_t)len + 4 > in_size || (uint64_t)len >= left) goto out; /* :886-898 */
        uint32_t crc = 0;
        if (!(flags & ORC_PNG_NO_CRC)) {
            crc = orc_crc32(0xffffffffu, type, 4);
            if (len) crc = orc_crc32(crc, in + at, len);
            crc ^= 0xffffffffu;
        }
        if (!memcmp(type, "PLTE", 4)) { /* :900-950 */
            if (!found_ihdr) goto out;
            if (ct == 0 && !(flags & ORC_PNG_ASSERTS_OFF)) goto out;
            if (len % 3 != 0) goto out;
            g_palette.size = len / 3;
            for (uint32_t i = 0; i < g_palette.size; i++) {
                if (i < 256) { /* the reference overflows its 256-entry arrays beyond this */
                    g_palette.red[i] = in[at];
                    g_palette.green[i] = in[at + 1];
                    g_palette.blue[i] = in[at + 2];
                }
                at += 3;
            }
        } else if (!memcmp(type, "IHDR", 4)) { /* :951-1138 */
            found_ihdr = 1;
            w = be32(in + at);
            h = be32(in + at + 4);
            uint8_t depth = in[at + 8];
            ct = in[at + 9];
            uint8_t filter_method = in[at + 11];
            at += 13;
            est = (uint64_t)(uint32_t)(w * h * 4u + h + 1u);
            if ((uint64_t)(uint32_t)(w * h * 4u) != rgba_size) goto out;
            if (ct != 2 && ct != 3 && ct != 6) goto out;
            if (w < 1 || h < 1) goto out;
            if ((uint64_t)(uint32_t)(w * h * 4u + h + 1u + 3000000u) > wm_size) goto out;
            if (depth != 8) goto out;
            if (filter_method != 0) goto out;
            if (left < 4) goto out;
        } else if (is_idat) { /* :1139-1292 */
            if (!found_ihdr) goto out;
            uint32_t dlen = len;
            if (!found_idat) {
                found_idat = 1;
                uint8_t cmf = in[at], flg = in[at + 1];
                at += 2;
                dlen -= 2;
                if ((cmf & 15) != 8) goto out;
                uint32_t chk = (uint16_t)(flg | (uint16_t)(cmf << 8));
                if (chk == 0 || chk % 31 != 0) goto out;
                if ((flg >> 5) & 1) goto out; /* FDICT */
            }
            if (at + (uint64_t)dlen > in_size) goto out;
            memmove(in + packed, in + at, dlen);
            packed += dlen;
            at += dlen;
            left -= dlen;
        } else if (!memcmp(type, "IEND", 4)) {
            found_iend = 1;
        } else if ((char)type[0] > 'Z') {
            at += len;
            left -= len;
        } else {
            goto out;
        }
        if (left < 4) goto out;
        if (at + 4 > in_size) goto out;
        uint32_t file_crc = be32(in + at);
        at += 4;
        left -= 4;
        if (!(flags & ORC_PNG_NO_CRC) && crc != file_crc) goto out;
    }
    if (!ran_inflate) goto out; /* P6, :1358-1367 */
    {
        uint32_t bpp = ct == 2 ? 3 : ct == 3 ? 1 : 4;
        uint32_t pixels = w * h;
        uint64_t o = 0, s = 0;
        uint64_t rowb = (uint64_t)w * bpp;
        for (uint32_t y = 0; y < h; y++) { /* :1430-1507 */
            uint8_t ft = stream[s++];
            if (ft > 4 && !(flags & ORC_PNG_ASSERTS_OFF)) goto out; /* P5 */
            for (uint32_t x = 0; x < w; x++) {
                for (uint32_t k = 0; k < bpp; k++) {
                    uint8_t a = x > 0 ? out_rgba[o - bpp] : 0;
                    uint8_t bb = y > 0 ? out_rgba[o - rowb] : 0;
                    uint8_t cc = (y > 0 && x > 0) ? out_rgba[o - rowb - bpp] : 0;
                    if (o >= rgba_size) goto out;
                    uint8_t v = stream[s++];
                    uint8_t r;
                    switch (ft) {
                    case 0: r = v; break;
                    case 1: r = (uint8_t)(v + a); break;
                    case 2: r = (uint8_t)(v + bb); break;
                    case 3: r = (uint8_t)(v + (uint8_t)(((uint32_t)a + bb) / 2)); break;
                    case 4: r = (uint8_t)(v + paeth(a, bb, cc)); break;
                    default: r = 0; break;
                    }
                    out_rgba[o++] = r;
                }
            }
            if (bpp == 3) { /* P3: whole-buffer re-expansion inside the row loop, :1512-1535 */
                uint64_t wr = (uint64_t)pixels * 4 - 1, rd = (uint64_t)pixels * 3 - 1;
                for (uint32_t p = 0; p < pixels; p++) {
                    out_rgba[wr--] = 255;
                    out_rgba[wr--] = out_rgba[rd--];
                    out_rgba[wr--] = out_rgba[rd--];
                    out_rgba[wr--] = out_rgba[rd--];
                }
            }
        }
        if (ct == 3) { /* :1538-1564 (tRNS ignored, alpha 255) */
            uint8_t *idx = (uint8_t *)malloc(pixels ? pixels : 1);
            memcpy(idx, out_rgba, pixels);
            for (uint32_t p = 0; p < pixels; p++) {
                out_rgba[4ull * p + 0] = g_palette.red[idx[p]];
                out_rgba[4ull * p + 1] = g_palette.green[idx[p]];
                out_rgba[4ull * p + 2] = g_palette.blue[idx[p]];
                out_rgba[4ull * p + 3] = 255;
            }
            free(idx);
        }
        good = 1;
    }
out:
    free(stream);
    free(in);
    return good;
}

/* ------------------------------------------------------------------- gz */

uint32_t orc_gz_locate(const uint8_t *in, uint32_t in_size, uint32_t *payload_off,
                       uint32_t *payload_len)
{ /* src/decode_gz.c:101-246, DECODE_GZ_SILENCE build (FCOMMENT not skipped, G1) */
    if (in == NULL || in_size < 10) return 0;
    if (in[0] != 31 || in[1] != 139) return 0;
    if (in[2] != 8) return 0;
    uint32_t off = 10, left = in_size - 10;
    if ((in[3] >> 3) & 1) {
        uint32_t n = 0;
        while (off + n < in_size && in[off + n] != 0 && n < left) n++;
        off += n + 1;
        left -= n + 1;
    }
    *payload_off = off;
    *payload_len = left - 8u; /* uint32 arithmetic as in the reference (:270) */
    return 1;
}

uint32_t orc_decode_gz(const uint8_t *in, uint32_t in_size, uint8_t *out, uint64_t out_cap,
                       uint64_t *out_size)
{
    uint32_t off, len;
    *out_size = 0;
    if (!orc_gz_locate(in, in_size, &off, &len)) return 0;
    uint32_t left = len + 8u;
    uint32_t guess = left * 35u + 1000000u; /* :245, uint32 wrap included */
    uint8_t *tmp = (uint8_t *)malloc((size_t)guess + 1024);
    uint64_t fin = 0;
    uint32_t fs = 0;
    if (off > in_size) { free(tmp); return 0; }
    uint32_t ok = orc_inflate_ex(in + off, len, tmp, guess, &fin, &fs, NULL, NULL, NULL);
    if (ok) {
        uint64_t n = fin < out_cap ? fin : out_cap;
        memcpy(out, tmp, n);
        *out_size = fin;
    }
    free(tmp);
    return ok;
}
