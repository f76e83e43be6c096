/*
 * Deterministic synthetic-workload generator (BASELINE.json configs).
 *
 * Self-contained on purpose: payloads come from splitmix64 and the DEFLATE
 * streams from the small encoders below, so the very same bytes regenerate on
 * any machine (no dependency on the local zlib version).  This is workload
 * tooling for tests/ and bench.py -- it is not part of the decode path.
 *
 *   payloads : random bytes | text-like (3000-word vocabulary, words 3..9 bytes)
 *              | smooth RGBA image rows (gradient + noise) for PNG synthesis
 *   encoders : stored blocks | fixed-Huffman (greedy hash-chain LZ77, 32 KiB
 *              window, min match 3) | dynamic-Huffman (same matcher, length-
 *              limited canonical codes, optional "EOB >= 8 bits" rule)
 *   wrappers : gzip member (10-byte header, CRC32 + ISIZE trailer), zlib + PNG
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void sg_payload_random(uint64_t seed, uint8_t *out, uint64_t n)
{
    uint64_t s = seed;
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t v = splitmix64(&s);
        memcpy(out + i, &v, 8);
    }
    if (i < n) {
        uint64_t v = splitmix64(&s);
        memcpy(out + i, &v, n - i);
    }
}

/* text-like: vocabulary of 3000 words (3..9 random bytes each) drawn with a
 * skewed distribution, separated by spaces */
void sg_payload_text(uint64_t seed, uint8_t *out, uint64_t n)
{
    enum { NW = 3000 };
    static uint8_t words[NW][9];
    static uint8_t wlen[NW];
    uint64_t vs = 0xC0FFEE; /* the vocabulary is global, the word order is per-seed */
    for (int i = 0; i < NW; i++) {
        uint64_t v = splitmix64(&vs);
        wlen[i] = (uint8_t)(3 + v % 7);
        uint64_t b = splitmix64(&vs), c = splitmix64(&vs);
        for (int k = 0; k < 9; k++) {
            uint64_t src = k < 8 ? (b >> (8 * k)) : c;
            words[i][k] = (uint8_t)('a' + (src & 0xff) % 26);
        }
    }
    uint64_t s = seed, o = 0;
    while (o < n) {
        uint64_t v = splitmix64(&s);
        /* squared uniform => frequent words are reused often */
        uint64_t u = v % NW, w = (v >> 32) % NW;
        uint32_t idx = (uint32_t)((u * w) / NW);
        for (uint32_t k = 0; k < wlen[idx] && o < n; k++) out[o++] = words[idx][k];
        if (o < n) out[o++] = ((v >> 60) == 0) ? '\n' : ' ';
    }
}

/* ------------------------------------------------------------ bit writer */
typedef struct {
    uint8_t *dst;
    uint64_t cap, pos;
    uint64_t acc;
    uint32_t nacc;
    int overflow;
} bitw;

static inline void bw_put(bitw *w, uint32_t v, uint32_t n)
{
    w->acc |= (uint64_t)v << w->nacc;
    w->nacc += n;
    while (w->nacc >= 8) {
        if (w->pos < w->cap) w->dst[w->pos] = (uint8_t)w->acc;
        else w->overflow = 1;
        w->pos++;
        w->acc >>= 8;
        w->nacc -= 8;
    }
}
static inline void bw_align(bitw *w)
{
    if (w->nacc) bw_put(w, 0, 8 - w->nacc);
}
static inline uint32_t rev(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

/* --------------------------------------------------------------- stored */
uint64_t sg_enc_stored(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, uint32_t blk)
{
    if (blk == 0 || blk > 65535) blk = 65535;
    uint64_t o = 0, i = 0;
    do {
        uint32_t len = (uint32_t)((n - i) < blk ? (n - i) : blk);
        int fin = (i + len == n);
        if (o + 5 + len > cap) return 0;
        dst[o++] = (uint8_t)fin;
        dst[o++] = (uint8_t)len;
        dst[o++] = (uint8_t)(len >> 8);
        dst[o++] = (uint8_t)~len;
        dst[o++] = (uint8_t)(~len >> 8);
        memcpy(dst + o, src + i, len);
        o += len;
        i += len;
    } while (i < n);
    return o;
}

/* ----------------------------------------------------------------- LZ77 */
typedef struct {
    uint16_t len;  /* 0 => literal */
    uint16_t dist; /* or the literal byte */
} token;

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35,
                                   43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                   3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                   33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                   1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6,
                                   6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t len_sym(uint32_t len)
{
    uint32_t s = 28;
    while (LBASE[s] > len) s--;
    if (len == 258) s = 28;
    else if (s == 28) s = 27;
    return s;
}
static inline uint32_t dist_sym(uint32_t d)
{
    uint32_t s = 29;
    while (DBASE[s] > d) s--;
    return s;
}

/* greedy hash-chain matcher; window 32 KiB, min match 3, chain depth `depth` */
static uint64_t lz77(const uint8_t *src, uint64_t n, token *tok, uint32_t depth)
{
    enum { HB = 15, HS = 1 << HB, WS = 32768 };
    int32_t *head = (int32_t *)malloc(sizeof(int32_t) * HS);
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * WS);
    for (int i = 0; i < HS; i++) head[i] = -1;
    uint64_t nt = 0, i = 0;
    while (i < n) {
        uint32_t bl = 0, bd = 0;
        if (i + 3 <= n) {
            uint32_t h = ((uint32_t)src[i] << 16 | (uint32_t)src[i + 1] << 8 | src[i + 2]);
            h = (h * 2654435761u) >> (32 - HB);
            int32_t c = head[h];
            uint32_t d = depth;
            uint32_t maxl = (uint32_t)((n - i) < 258 ? (n - i) : 258);
            while (c >= 0 && (uint64_t)c + WS > i && d--) {
                /* prev[] is a ring: an entry is valid only while within the window */
                uint32_t l = 0;
                while (l < maxl && src[c + l] == src[i + l]) l++;
                if (l > bl) {
                    bl = l;
                    bd = (uint32_t)(i - (uint64_t)c);
                    if (l == maxl) break;
                }
                int32_t p = prev[c & (WS - 1)];
                if (p >= c) break;
                c = p;
            }
        }
        uint32_t adv = 1;
        if (bl >= 3) {
            tok[nt].len = (uint16_t)bl;
            tok[nt].dist = (uint16_t)bd;
            adv = bl;
        } else {
            tok[nt].len = 0;
            tok[nt].dist = src[i];
        }
        nt++;
        for (uint32_t k = 0; k < adv; k++) {
            uint64_t p = i + k;
            if (p + 3 <= n) {
                uint32_t h = ((uint32_t)src[p] << 16 | (uint32_t)src[p + 1] << 8 | src[p + 2]);
                h = (h * 2654435761u) >> (32 - HB);
                prev[p & (WS - 1)] = head[h];
                head[h] = (int32_t)p;
            }
        }
        i += adv;
    }
    free(head);
    free(prev);
    return nt;
}

/* ---------------------------------------------------------------- fixed */
static inline void put_fixed_litlen(bitw *w, uint32_t s)
{
    if (s < 144) bw_put(w, rev(48 + s, 8), 8);
    else if (s < 256) bw_put(w, rev(400 + (s - 144), 9), 9);
    else if (s < 280) bw_put(w, rev(s - 256, 7), 7);
    else bw_put(w, rev(192 + (s - 280), 8), 8);
}

static void put_tokens_fixed(bitw *w, const token *tok, uint64_t a, uint64_t b)
{
    for (uint64_t t = a; t < b; t++) {
        if (tok[t].len == 0) {
            put_fixed_litlen(w, tok[t].dist);
        } else {
            uint32_t ls = len_sym(tok[t].len), ds = dist_sym(tok[t].dist);
            put_fixed_litlen(w, 257 + ls);
            if (LEXTRA[ls]) bw_put(w, tok[t].len - LBASE[ls], LEXTRA[ls]);
            bw_put(w, rev(ds, 5), 5);
            if (DEXTRA[ds]) bw_put(w, tok[t].dist - DBASE[ds], DEXTRA[ds]);
        }
    }
}

/* tokens_per_block = 0 => one BFINAL block.  depth: hash-chain search depth */
uint64_t sg_enc_fixed(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap,
                      uint32_t tokens_per_block, uint32_t depth)
{
    token *tok = (token *)malloc(sizeof(token) * (n ? n : 1));
    uint64_t nt = lz77(src, n, tok, depth ? depth : 8);
    bitw w = {dst, cap, 0, 0, 0, 0};
    uint64_t t = 0;
    do {
        uint64_t e = tokens_per_block ? t + tokens_per_block : nt;
        if (e > nt) e = nt;
        bw_put(&w, e == nt ? 1 : 0, 1);
        bw_put(&w, 1, 2);
        put_tokens_fixed(&w, tok, t, e);
        put_fixed_litlen(&w, 256);
        t = e;
    } while (t < nt);
    bw_align(&w);
    free(tok);
    return w.overflow ? 0 : w.pos;
}

/* -------------------------------------------------------------- dynamic */
/* length-limited Huffman lengths: build an ordinary Huffman tree, then repair
 * overlong codes with the classic "kraft sum" adjustment. */
static void huff_lengths(const uint32_t *freq, uint32_t n, uint32_t maxbits, uint8_t *lens)
{
    typedef struct { uint64_t w; int32_t l, r; } node;
    node *nd = (node *)malloc(sizeof(node) * (2 * n + 2));
    int32_t *alive = (int32_t *)malloc(sizeof(int32_t) * (2 * n + 2));
    uint32_t na = 0, nn = 0;
    memset(lens, 0, n);
    for (uint32_t i = 0; i < n; i++)
        if (freq[i]) {
            nd[nn].w = freq[i]; nd[nn].l = -1; nd[nn].r = (int32_t)i;
            alive[na++] = (int32_t)nn++;
        }
    if (na == 0) { free(nd); free(alive); return; }
    if (na == 1) { lens[nd[alive[0]].r] = 1; free(nd); free(alive); return; }
    while (na > 1) {
        uint32_t a = 0, b = 1;
        if (nd[alive[b]].w < nd[alive[a]].w) { a = 1; b = 0; }
        for (uint32_t i = 2; i < na; i++) {
            if (nd[alive[i]].w < nd[alive[a]].w) { b = a; a = i; }
            else if (nd[alive[i]].w < nd[alive[b]].w) b = i;
        }
        nd[nn].w = nd[alive[a]].w + nd[alive[b]].w;
        nd[nn].l = alive[a]; nd[nn].r = alive[b];
        uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        alive[lo] = (int32_t)nn++;
        alive[hi] = alive[--na];
    }
    /* depths by DFS */
    int32_t *stk = (int32_t *)malloc(sizeof(int32_t) * (2 * n + 2));
    uint8_t *dep = (uint8_t *)calloc(2 * n + 2, 1);
    int sp = 0;
    stk[sp++] = alive[0];
    while (sp) {
        int32_t x = stk[--sp];
        if (nd[x].l < 0) { lens[nd[x].r] = dep[x] ? dep[x] : 1; continue; }
        dep[nd[x].l] = dep[nd[x].r] = (uint8_t)(dep[x] + 1);
        stk[sp++] = nd[x].l; stk[sp++] = nd[x].r;
    }
    /* limit */
    int over = 0;
    for (uint32_t i = 0; i < n; i++) if (lens[i] > maxbits) { lens[i] = (uint8_t)maxbits; over = 1; }
    if (over) {
        uint64_t kraft = 0, one = 1ull << maxbits;
        for (uint32_t i = 0; i < n; i++) if (lens[i]) kraft += one >> lens[i];
        while (kraft > one) { /* lengthen the longest code shorter than maxbits */
            uint32_t best = n; uint8_t bl = 0;
            for (uint32_t i = 0; i < n; i++)
                if (lens[i] && lens[i] < maxbits && lens[i] > bl) { bl = lens[i]; best = i; }
            if (best == n) break;
            kraft -= one >> lens[best];
            lens[best]++;
            kraft += one >> lens[best];
        }
    }
    free(stk); free(dep); free(nd); free(alive);
}

static void canon_codes(const uint8_t *lens, uint32_t n, uint16_t *codes)
{
    uint32_t bl[16] = {0}, next[16] = {0};
    for (uint32_t i = 0; i < n; i++) bl[lens[i]]++;
    bl[0] = 0;
    uint32_t code = 0;
    for (uint32_t b = 1; b <= 15; b++) { code = (code + bl[b - 1]) << 1; next[b] = code; }
    for (uint32_t i = 0; i < n; i++) codes[i] = lens[i] ? (uint16_t)rev(next[lens[i]]++, lens[i]) : 0;
}

/* one dynamic block for tokens [a,b); eob_min_bits: force the EOB code to at
 * least that many bits (8 => the reference's tail rule Q2 can never truncate) */
static void put_block_dynamic(bitw *w, const token *tok, uint64_t a, uint64_t b, int final,
                              uint32_t eob_min_bits)
{
    uint32_t lf[286] = {0}, df[30] = {0};
    for (uint64_t t = a; t < b; t++) {
        if (tok[t].len == 0) lf[tok[t].dist]++;
        else { lf[257 + len_sym(tok[t].len)]++; df[dist_sym(tok[t].dist)]++; }
    }
    lf[256] = 1;
    int nd = 0;
    for (int i = 0; i < 30; i++) nd += df[i] != 0;
    /* the reference rejects a distance table whose code length >= HDIST (Q6) and
     * cannot decode a lone code longer than 1 bit (Q7): keep >= 2 codes */
    if (nd == 0) { df[0] = 1; df[1] = 1; }
    else if (nd == 1) { df[df[0] ? 1 : 0] = 1; }
    uint8_t ll[286], dl[30];
    huff_lengths(lf, 286, 15, ll);
    huff_lengths(df, 30, 15, dl);
    if (eob_min_bits && ll[256] < eob_min_bits) {
        /* demote EOB: give it eob_min_bits and rebuild the rest under the freed budget by
         * scaling frequencies: simplest robust way is to make EOB very rare and re-run with a
         * virtual sibling weight */
        uint32_t saved = lf[256];
        (void)saved;
        for (int tries = 0; tries < 16 && ll[256] < eob_min_bits; tries++) {
            /* halve every other weight's advantage by boosting them */
            for (int i = 0; i < 286; i++) if (i != 256 && lf[i] && lf[i] < (1u << 29)) lf[i] = lf[i] * 2 + 1;
            huff_lengths(lf, 286, 15, ll);
        }
    }
    uint32_t hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) hlit--;
    while (hdist > 1 && dl[hdist - 1] == 0) hdist--;
    /* Q6: every distance code length must be < HDIST */
    uint8_t maxdl = 0;
    for (uint32_t i = 0; i < hdist; i++) if (dl[i] > maxdl) maxdl = dl[i];
    if (hdist <= maxdl) hdist = (uint32_t)maxdl + 1;
    uint16_t lc[286], dc[30];
    canon_codes(ll, 286, lc);
    canon_codes(dl, 30, dc);
    /* code-length alphabet with simple RLE (16/17/18) */
    uint8_t seq[320];
    uint32_t ns = 0;
    for (uint32_t i = 0; i < hlit; i++) seq[ns++] = ll[i];
    for (uint32_t i = 0; i < hdist; i++) seq[ns++] = dl[i];
    uint8_t cls[320], clx[320];
    uint32_t ncl = 0, cf[19] = {0};
    for (uint32_t i = 0; i < ns;) {
        uint32_t j = i;
        while (j < ns && seq[j] == seq[i]) j++;
        uint32_t run = j - i;
        if (seq[i] == 0 && run >= 3) {
            while (run >= 3) {
                uint32_t r = run > 138 ? 138 : run;
                if (r >= 11) { cls[ncl] = 18; clx[ncl++] = (uint8_t)(r - 11); }
                else { cls[ncl] = 17; clx[ncl++] = (uint8_t)(r - 3); }
                run -= r;
            }
            while (run--) { cls[ncl] = 0; clx[ncl++] = 0; }
        } else if (run >= 4) {
            cls[ncl] = seq[i]; clx[ncl++] = 0;
            run--;
            while (run >= 3) {
                uint32_t r = run > 6 ? 6 : run;
                cls[ncl] = 16; clx[ncl++] = (uint8_t)(r - 3);
                run -= r;
            }
            while (run--) { cls[ncl] = seq[i]; clx[ncl++] = 0; }
        } else {
            while (run--) { cls[ncl] = seq[i]; clx[ncl++] = 0; }
        }
        i = j;
    }
    for (uint32_t i = 0; i < ncl; i++) cf[cls[i]]++;
    uint8_t cl[19];
    uint16_t cc[19];
    huff_lengths(cf, 19, 7, cl);
    { /* a lone code-length code longer than 1 bit is undecodable by the reference (Q7) */
        int used = 0;
        for (int i = 0; i < 19; i++) used += cl[i] != 0;
        if (used == 1) for (int i = 0; i < 19; i++) if (!cl[i]) { cl[i] = 1; break; }
    }
    canon_codes(cl, 19, cc);
    static const uint8_t ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t hclen = 19;
    while (hclen > 4 && cl[ORD[hclen - 1]] == 0) hclen--;
    bw_put(w, final ? 1 : 0, 1);
    bw_put(w, 2, 2);
    bw_put(w, hlit - 257, 5);
    bw_put(w, hdist - 1, 5);
    bw_put(w, hclen - 4, 4);
    for (uint32_t i = 0; i < hclen; i++) bw_put(w, cl[ORD[i]], 3);
    for (uint32_t i = 0; i < ncl; i++) {
        bw_put(w, cc[cls[i]], cl[cls[i]]);
        if (cls[i] == 16) bw_put(w, clx[i], 2);
        else if (cls[i] == 17) bw_put(w, clx[i], 3);
        else if (cls[i] == 18) bw_put(w, clx[i], 7);
    }
    for (uint64_t t = a; t < b; t++) {
        if (tok[t].len == 0) {
            bw_put(w, lc[tok[t].dist], ll[tok[t].dist]);
        } else {
            uint32_t ls = len_sym(tok[t].len), ds = dist_sym(tok[t].dist);
            bw_put(w, lc[257 + ls], ll[257 + ls]);
            if (LEXTRA[ls]) bw_put(w, tok[t].len - LBASE[ls], LEXTRA[ls]);
            bw_put(w, dc[ds], dl[ds]);
            if (DEXTRA[ds]) bw_put(w, tok[t].dist - DBASE[ds], DEXTRA[ds]);
        }
    }
    bw_put(w, lc[256], ll[256]);
}

uint64_t sg_enc_dynamic(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap,
                        uint32_t tokens_per_block, uint32_t depth, uint32_t eob_min_bits)
{
    token *tok = (token *)malloc(sizeof(token) * (n ? n : 1));
    uint64_t nt = lz77(src, n, tok, depth ? depth : 8);
    bitw w = {dst, cap, 0, 0, 0, 0};
    if (tokens_per_block == 0) tokens_per_block = 16384;
    uint64_t t = 0;
    do {
        uint64_t e = t + tokens_per_block;
        if (e > nt || nt - e < tokens_per_block / 4) e = nt;
        put_block_dynamic(&w, tok, t, e, e == nt, eob_min_bits);
        t = e;
    } while (t < nt);
    bw_align(&w);
    free(tok);
    return w.overflow ? 0 : w.pos;
}

/* ------------------------------------------------------------- wrappers */
static uint32_t crc_tab[256];
static void crc_init(void)
{
    if (crc_tab[1]) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_tab[i] = c;
    }
}
uint32_t sg_crc32(uint32_t crc, const uint8_t *p, uint64_t n)
{
    crc_init();
    crc = ~crc;
    for (uint64_t i = 0; i < n; i++) crc = crc_tab[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
uint32_t sg_adler32(const uint8_t *p, uint64_t n)
{
    uint32_t a = 1, b = 0;
    for (uint64_t i = 0; i < n; i++) { a = (a + p[i]) % 65521; b = (b + a) % 65521; }
    return (b << 16) | a;
}

/* wrap a raw DEFLATE stream as one gzip member: 10-byte header, no optional fields */
uint64_t sg_wrap_gzip(const uint8_t *raw, uint64_t nraw, const uint8_t *plain, uint64_t nplain,
                      uint8_t *dst, uint64_t cap)
{
    if (nraw + 18 > cap) return 0;
    static const uint8_t H[10] = {31, 139, 8, 0, 0, 0, 0, 0, 0, 3};
    memcpy(dst, H, 10);
    memcpy(dst + 10, raw, nraw);
    uint32_t c = sg_crc32(0, plain, nplain), isz = (uint32_t)nplain;
    memcpy(dst + 10 + nraw, &c, 4);
    memcpy(dst + 14 + nraw, &isz, 4);
    return nraw + 18;
}

static void be32w(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
static uint64_t png_chunk(uint8_t *dst, const char *type, const uint8_t *data, uint32_t len)
{
    be32w(dst, len);
    memcpy(dst + 4, type, 4);
    if (len) memcpy(dst + 8, data, len);
    be32w(dst + 8 + len, sg_crc32(0, dst + 4, len + 4));
    return 12ull + len;
}

/* PNG file around a raw DEFLATE stream of the filtered scanlines.
 * ct: 6 (RGBA) | 2 (RGB) | 3 (palette; plte = 768 bytes or NULL).  idat_chunk: payload bytes
 * per IDAT chunk (0 = single chunk). */
uint64_t sg_wrap_png(const uint8_t *raw, uint64_t nraw, const uint8_t *filtered, uint64_t nfilt,
                     uint32_t w, uint32_t h, uint32_t ct, const uint8_t *plte, uint32_t nplte,
                     uint32_t idat_chunk, uint8_t *dst, uint64_t cap)
{
    uint64_t need = 8 + 25 + (plte ? 12 + 3ull * nplte : 0) + nraw + 6 + 12 +
                    12ull * (idat_chunk ? (nraw + 6) / idat_chunk + 2 : 1);
    if (need > cap) return 0;
    static const uint8_t SIG[8] = {137, 'P', 'N', 'G', 13, 10, 26, 10};
    uint64_t o = 0;
    memcpy(dst, SIG, 8);
    o = 8;
    uint8_t ihdr[13];
    be32w(ihdr, w);
    be32w(ihdr + 4, h);
    ihdr[8] = 8; ihdr[9] = (uint8_t)ct; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    o += png_chunk(dst + o, "IHDR", ihdr, 13);
    if (plte) o += png_chunk(dst + o, "PLTE", plte, 3 * nplte);
    uint64_t nz = nraw + 6;
    uint8_t *z = (uint8_t *)malloc(nz);
    z[0] = 0x78; z[1] = 0x9c;
    memcpy(z + 2, raw, nraw);
    be32w(z + 2 + nraw, sg_adler32(filtered, nfilt));
    uint64_t i = 0;
    if (idat_chunk == 0) idat_chunk = 0x7fffffff;
    while (i < nz) {
        uint32_t len = (uint32_t)((nz - i) < idat_chunk ? (nz - i) : idat_chunk);
        o += png_chunk(dst + o, "IDAT", z + i, len);
        i += len;
    }
    free(z);
    o += png_chunk(dst + o, "IEND", NULL, 0);
    return o;
}

/* PNG-filter an image (forward filters; type per row: 0..4, or 5 = cycle 0..4 by row) */
void sg_png_filter(const uint8_t *pix, uint32_t w, uint32_t h, uint32_t bpp, uint32_t ftype,
                   uint8_t *out)
{
    uint64_t rowb = (uint64_t)w * bpp;
    for (uint32_t y = 0; y < h; y++) {
        uint32_t ft = ftype == 5 ? y % 5 : ftype;
        uint8_t *o = out + (uint64_t)y * (rowb + 1);
        const uint8_t *cur = pix + (uint64_t)y * rowb;
        const uint8_t *up = y ? cur - rowb : NULL;
        o[0] = (uint8_t)ft;
        for (uint64_t x = 0; x < rowb; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pr = 0;
            switch (ft) {
            case 1: pr = a; break;
            case 2: pr = b; break;
            case 3: pr = (a + b) >> 1; break;
            case 4: {
                int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } break;
            default: pr = 0;
            }
            o[1 + x] = (uint8_t)(cur[x] - pr);
        }
    }
}

/* smooth RGBA image: 2-D gradient + noise of the given amplitude (0..255) */
void sg_payload_image(uint64_t seed, uint32_t w, uint32_t h, uint32_t bpp, uint32_t noise,
                      uint8_t *out)
{
    uint64_t s = seed;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint64_t v = splitmix64(&s);
            for (uint32_t k = 0; k < bpp; k++) {
                uint32_t g = (x * (k + 1) / 4 + y * (4 - k) / 4 + 16 * k);
                uint32_t nz = noise ? (uint32_t)((v >> (8 * k)) & 0xff) % (noise + 1) : 0;
                out[((uint64_t)y * w + x) * bpp + k] = (uint8_t)(g + nz);
            }
        }
}

/* Like sg_payload_image, with a finer noise knob: every sample draws its amplitude from
 * {noise_lo, noise_hi}, noise_hi with probability hi_per_256 / 256.  Used to land config 4's
 * synthetic PNGs at the compression ratio BASELINE.json asks for (about 3:1). */
void sg_payload_image_mix(uint64_t seed, uint32_t w, uint32_t h, uint32_t bpp, uint32_t noise_lo,
                          uint32_t noise_hi, uint32_t hi_per_256, uint8_t *out)
{
    uint64_t s = seed;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint64_t v = splitmix64(&s);
            for (uint32_t k = 0; k < bpp; k++) {
                uint32_t g = (x * (k + 1) / 4 + y * (4 - k) / 4 + 16 * k);
                uint32_t r = (uint32_t)((v >> (8 * k)) & 0xff);
                uint32_t pick = (uint32_t)((v >> (32 + 8 * (k & 3))) & 0xff);
                uint32_t amp = pick < hi_per_256 ? noise_hi : noise_lo;
                out[((uint64_t)y * w + x) * bpp + k] = (uint8_t)(g + (amp ? r % (amp + 1) : 0));
            }
        }
}
