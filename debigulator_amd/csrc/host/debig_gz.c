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

/* ---------------------------------------------------------------------------------------
 * debig_gunzip_batch: RFC 1952 complete (include/decode_gz.h).  Not a reference function.
 * ------------------------------------------------------------------------------------- */
static uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static uint32_t le32(const uint8_t *p) { return le16(p) | (le16(p + 2) << 16); }

/* One member header at p (avail bytes left in the file).  Returns DEBIG_GZ_OK and the header
 * length; *bsize = total member size from a BGZF "BC" subfield, 0 when there is none. */
static uint32_t gz_member_header(const uint8_t *p, uint64_t avail, uint64_t *hdr_len, uint64_t *bsize)
{
    *bsize = 0;
    if (avail < 10) return avail >= 2 && (p[0] != 31 || p[1] != 139) ? DEBIG_GZ_E_HEADER : DEBIG_GZ_E_TRUNCATED;
    if (p[0] != 31 || p[1] != 139 || p[2] != 8 || (p[3] & 0xE0u)) return DEBIG_GZ_E_HEADER;
    const uint32_t flg = p[3];
    uint64_t o = 10;
    if (flg & 4u) { /* FEXTRA */
        if (o + 2 > avail) return DEBIG_GZ_E_TRUNCATED;
        uint64_t xlen = le16(p + o);
        o += 2;
        if (o + xlen > avail) return DEBIG_GZ_E_TRUNCATED;
        for (uint64_t q = o; q + 4 <= o + xlen;) { /* subfields: SI1 SI2 LEN data */
            uint64_t sl = le16(p + q + 2);
            if (q + 4 + sl > o + xlen) break;
            if (p[q] == 'B' && p[q + 1] == 'C' && sl == 2) *bsize = (uint64_t)le16(p + q + 4) + 1u;
            q += 4 + sl;
        }
        o += xlen;
    }
    for (uint32_t bit = 8u; bit <= 16u; bit <<= 1) /* FNAME, FCOMMENT: zero-terminated */
        if (flg & bit) {
            while (o < avail && p[o] != 0) o++;
            if (o >= avail) return DEBIG_GZ_E_TRUNCATED;
            o++;
        }
    if (flg & 2u) { /* FHCRC */
        if (o + 2 > avail) return DEBIG_GZ_E_TRUNCATED;
        o += 2;
    }
    *hdr_len = o;
    return DEBIG_GZ_OK;
}

DEBIG_API uint32_t debig_gz_parse_header(const uint8_t *p, uint64_t avail, uint64_t *header_len, uint64_t *member_size)
{
    uint64_t hl = 0, bs = 0;
    if (!p) return DEBIG_GZ_E_HEADER;
    uint32_t st = gz_member_header(p, avail, &hl, &bs);
    if (header_len) *header_len = st == DEBIG_GZ_OK ? hl : 0;
    if (member_size) *member_size = st == DEBIG_GZ_OK ? bs : 0;
    return st;
}

typedef struct gz_item { /* one member in flight */
    uint32_t file;
    uint64_t payload;    /* file offset of its DEFLATE data            */
    uint64_t known_end;  /* file offset just past its trailer, or 0    */
    uint64_t out_rel;    /* where its output starts in the file's output */
} gz_item;

typedef struct gz_file {
    uint64_t pos;      /* file offset of the next member header            */
    uint64_t used;     /* output bytes of the members verified so far      */
    uint64_t dev_in, dev_out;
    uint32_t status, members;
    uint32_t active;   /* has a member to decode in the next pass          */
    uint32_t failed;   /* an error ended this file                         */
} gz_file;

static int gz_all_zero(const uint8_t *p, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
        if (p[i]) return 0;
    return 1;
}

DEBIG_API int debig_gunzip_batch(const uint8_t *const *inputs, const uint64_t *input_sizes,
                                 uint8_t *const *outs, const uint64_t *out_caps, uint64_t *out_sizes,
                                 uint32_t *status, uint32_t *n_members, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        out_sizes[i] = 0;
        status[i] = DEBIG_GZ_E_HEADER;
        if (n_members) n_members[i] = 0;
    }
    if (n == 0) return 0;
    debig_ctx *c = debig_ctx_get(0);
    if (!c) return 1;
    gz_file *F = (gz_file *)calloc(n, sizeof(gz_file));
    gz_item *items = NULL;
    debig_stream *desc = NULL;
    debig_result *res = NULL;
    debig_span *spans = NULL;
    uint32_t *crcs = NULL;
    uint64_t cap_items = 0;
    int rc = 2;
    if (!F) goto done;
    /* ---- whole files and the output regions go to / live in HBM once */
    uint64_t in_total = 0, out_total = 0;
    for (uint32_t i = 0; i < n; i++) {
        F[i].dev_in = in_total;
        F[i].dev_out = out_total;
        F[i].active = inputs[i] != NULL && outs[i] != NULL;
        F[i].status = F[i].active ? DEBIG_GZ_OK : DEBIG_GZ_E_HEADER;
        F[i].failed = !F[i].active;
        if (!F[i].active) continue;
        in_total += debig_align16(input_sizes[i]) + 16;
        out_total += debig_align16(out_caps[i]) + 16;
    }
    if ((rc = debig_devbuf_reserve(&c->files, in_total + 64)) || (rc = debig_devbuf_reserve(&c->out, out_total + 64)))
        goto done;
    {
        /* whole files up in one transfer through the page-locked arena */
        uint64_t *up_sizes = (uint64_t *)calloc(n, sizeof(uint64_t));
        uint64_t *up_offs = (uint64_t *)calloc(n, sizeof(uint64_t));
        if (!up_sizes || !up_offs) { free(up_sizes); free(up_offs); rc = 2; goto done; }
        for (uint32_t i = 0; i < n; i++) {
            up_sizes[i] = F[i].active ? input_sizes[i] : 0;
            up_offs[i] = F[i].dev_in;
        }
        rc = debig_upload_packed(c, c->files.ptr, inputs, up_sizes, up_offs, n, in_total);
        free(up_sizes);
        free(up_offs);
    }
    if (rc) goto done;

    for (;;) {
        /* ---- this pass: the next member of every active file; every member of a BGZF file */
        uint64_t n_items = 0;
        for (int fill = 0; fill < 2; fill++) { /* 0: count, 1: fill */
            n_items = 0;
            for (uint32_t i = 0; i < n; i++) {
                if (!F[i].active) continue;
                uint64_t pos = F[i].pos, used = F[i].used;
                uint32_t queued = 0; /* members of this file in this pass */
                for (;;) {
                    uint64_t hl = 0, bs = 0;
                    uint32_t st = gz_member_header(inputs[i] + pos, input_sizes[i] - pos, &hl, &bs);
                    if (st != DEBIG_GZ_OK) {
                        /* behind queued members the next pass meets this header again, first */
                        if (fill && !queued) { F[i].status = st; F[i].failed = 1; F[i].active = 0; }
                        break;
                    }
                    /* a BGZF member: its end and (from ISIZE) its output size are known now */
                    int bg = bs >= hl + 8 && pos + bs <= input_sizes[i];
                    if (fill) {
                        items[n_items].file = i;
                        items[n_items].payload = pos + hl;
                        items[n_items].known_end = bg ? pos + bs : 0;
                        items[n_items].out_rel = used;
                    }
                    n_items++;
                    queued++;
                    if (!bg) break;
                    used += le32(inputs[i] + pos + bs - 4);
                    pos += bs;
                    if (input_sizes[i] - pos < 10 || inputs[i][pos] != 31 || inputs[i][pos + 1] != 139) break;
                }
            }
            if (!fill) {
                if (n_items == 0) break;
                if (n_items > cap_items) {
                    free(items); free(desc); free(res); free(spans); free(crcs);
                    cap_items = n_items + n_items / 2 + 16;
                    items = (gz_item *)calloc(cap_items, sizeof(gz_item));
                    desc = (debig_stream *)calloc(cap_items, sizeof(debig_stream));
                    res = (debig_result *)calloc(cap_items, sizeof(debig_result));
                    spans = (debig_span *)calloc(cap_items, sizeof(debig_span));
                    crcs = (uint32_t *)calloc(cap_items, sizeof(uint32_t));
                    if (!items || !desc || !res || !spans || !crcs) { rc = 2; goto done; }
                }
            }
        }
        if (n_items == 0) break;
        if (n_items > 0xffffffffull) { rc = 2; goto done; }
        const uint32_t m = (uint32_t)n_items;
        memset(desc, 0, (size_t)m * sizeof(debig_stream));
        for (uint32_t k = 0; k < m; k++) {
            const gz_item *it = &items[k];
            const uint32_t i = it->file;
            desc[k].in_off = F[i].dev_in + it->payload;
            /* the trailer (and whatever follows) stays part of the input span: the end-of-input
             * rule of the reference's inflate (Q2) never sees the stream's last byte */
            desc[k].in_len = (it->known_end ? it->known_end : input_sizes[i]) - it->payload;
            desc[k].out_off = F[i].dev_out + it->out_rel;
            desc[k].out_cap = it->out_rel <= out_caps[i] ? out_caps[i] - it->out_rel : 0;
            desc[k].flags = DEBIG_STREAM_NO_REF_GATES;
        }
        if ((rc = debig_devbuf_reserve(&c->spans, (uint64_t)m * sizeof(debig_span))) ||
            (rc = debig_devbuf_reserve(&c->crcs, (uint64_t)m * sizeof(uint32_t))) ||
            (rc = debig_launch_inflate_planned(c, c->files.ptr, desc, res, m)))
            goto done;
        for (uint32_t k = 0; k < m; k++) {
            spans[k].off = desc[k].out_off;
            spans[k].len = res[k].good ? res[k].final_size : 0;
        }
        if ((rc = debig_hip_memcpy_h2d(c->spans.ptr, spans, (uint64_t)m * sizeof(debig_span), NULL)) ||
            (rc = debig_hip_checksum_batch(c->out.ptr, (const debig_span *)c->spans.ptr, (uint32_t *)c->crcs.ptr, m, 0, NULL)) ||
            (rc = debig_hip_memcpy_d2h(crcs, c->crcs.ptr, (uint64_t)m * sizeof(uint32_t), NULL)) ||
            (rc = debig_hip_stream_sync(NULL)))
            goto done;
        /* ---- verdicts, in member order per file */
        for (uint32_t k = 0; k < m; k++) {
            const gz_item *it = &items[k];
            const uint32_t i = it->file;
            gz_file *f = &F[i];
            if (f->failed) continue; /* an earlier member of this file failed */
            if (!res[k].good) {
                f->status = res[k].status == DEBIG_E_OUTPUT_FULL ? DEBIG_GZ_E_OUTPUT_FULL : DEBIG_GZ_E_INFLATE;
                f->failed = 1;
                continue;
            }
            const uint64_t end = it->payload + (res[k].in_end_bits + 7u) / 8u; /* the trailer starts here */
            if (end + 8 > input_sizes[i]) { f->status = DEBIG_GZ_E_TRUNCATED; f->failed = 1; continue; }
            if (it->known_end && end + 8 != it->known_end) { /* the BC size disagrees with the stream */
                f->status = DEBIG_GZ_E_INFLATE;
                f->failed = 1;
                continue;
            }
            if (it->out_rel != f->used) { /* the ISIZE of an earlier BGZF member was wrong */
                f->status = DEBIG_GZ_E_ISIZE;
                f->failed = 1;
                continue;
            }
            const uint8_t *t = inputs[i] + end;
            f->used += res[k].final_size;
            f->members++;
            f->pos = end + 8;
            if (le32(t) != crcs[k]) { f->status = DEBIG_GZ_E_CRC; f->failed = 1; continue; }
            if (le32(t + 4) != (uint32_t)res[k].final_size) { f->status = DEBIG_GZ_E_ISIZE; f->failed = 1; continue; }
        }
        /* ---- which files go on */
        for (uint32_t i = 0; i < n; i++) {
            gz_file *f = &F[i];
            if (!f->active) continue;
            if (f->failed) { f->active = 0; continue; }
            const uint64_t left = input_sizes[i] - f->pos;
            if (left == 0 || gz_all_zero(inputs[i] + f->pos, left)) { f->active = 0; continue; }
            if (left < 10 || inputs[i][f->pos] != 31 || inputs[i][f->pos + 1] != 139) {
                f->status = DEBIG_GZ_E_TRAILING;
                f->active = 0;
            }
        }
    }
    /* ---- results back to the host */
    rc = 0;
    {
        uint64_t *dn_sizes = (uint64_t *)calloc(n, sizeof(uint64_t));
        uint64_t *dn_offs = (uint64_t *)calloc(n, sizeof(uint64_t));
        if (!dn_sizes || !dn_offs) { free(dn_sizes); free(dn_offs); rc = 2; goto done; }
        uint64_t last_end = 0;
        for (uint32_t i = 0; i < n; i++) {
            status[i] = F[i].status;
            if (n_members) n_members[i] = F[i].members;
            out_sizes[i] = F[i].used;
            dn_sizes[i] = outs[i] ? F[i].used : 0;
            dn_offs[i] = F[i].dev_out;
            if (dn_sizes[i] && dn_offs[i] + dn_sizes[i] > last_end) last_end = dn_offs[i] + dn_sizes[i];
        }
        rc = debig_download_unpack(c, c->out.ptr, outs, dn_sizes, dn_offs, n, last_end);
        free(dn_sizes);
        free(dn_offs);
    }
done:
    free(F);
    free(items);
    free(desc);
    free(res);
    free(spans);
    free(crcs);
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
