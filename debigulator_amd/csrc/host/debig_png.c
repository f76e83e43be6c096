/*
 * Drop-in decode_png_init / decode_png_deinit / decode_png_get_width_height / decode_png
 * (reference src/decode_png.h:43-103, src/decode_png.c:562-1567) and their legacy names.
 *
 * Host side (plain C): only the METADATA walk -- signature, chunk headers, IHDR/PLTE/IDAT
 * validation -- following the reference's accept/reject rules.  Everything that touches the
 * bulk bytes runs on the GPU: chunk CRC-32 (debig_hip_checksum_batch, replaces update_crc
 * src/decode_png.c:313-333), IDAT concatenation (debig_hip_gather, replaces :1285-1291),
 * inflate (debig_hip_inflate_batch, replaces src/inflate.c) and the de-filter / palette
 * loops (debig_hip_png_defilter_batch, replaces :1381-1564).
 */
#include <stdlib.h>
#include <string.h>
#include "decode_png.h"
#include "debig_ctx.h"

#define INFLATE_HASHMAPS_SIZE 3000000u /* reference src/decode_png.c:16 */

typedef struct png_state {
    uint8_t palette[768]; /* R[256] G[256] B[256]; persists between calls like the reference's */
    uint32_t palette_size;
    uint32_t wm_size;
    int initialized;
    void *(*malloc_fn)(uint64_t);
    void (*free_fn)(void *);
} png_state;

static png_state g_png[DEBIG_MAX_THREADS];

static uint32_t be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

DEBIG_API void decode_png_init(void *(*malloc_funcptr)(uint64_t), void (*arg_free_funcptr)(void *),
                               void *(*arg_memset_funcptr)(void *, int, uint64_t),
                               void *(*arg_memcpy_func)(void *, const void *, uint64_t),
                               const uint32_t dpng_working_memory_size, const uint32_t thread_id)
{
    if (thread_id >= DEBIG_MAX_THREADS) return;
    png_state *s = &g_png[thread_id];
    if (s->initialized) return; /* reference: second init of a thread id is ignored (:580-593) */
    memset(s, 0, sizeof *s);
    s->malloc_fn = malloc_funcptr;
    s->free_fn = arg_free_funcptr;
    s->wm_size = dpng_working_memory_size; /* only its size matters: no host scratch is needed */
    s->initialized = 1;
    inflate_init(malloc_funcptr, arg_memset_funcptr, arg_memcpy_func, thread_id);
}

DEBIG_API void decode_png_deinit(const uint32_t thread_id)
{
    if (thread_id >= DEBIG_MAX_THREADS) return;
    g_png[thread_id].initialized = 0;
    debig_ctx_release(thread_id);
}

DEBIG_API void decode_png_get_width_height(const uint8_t *in, const uint64_t in_size, uint32_t *w,
                                           uint32_t *h, uint8_t *good)
{ /* src/decode_png.c:620-681: needs 28 bytes, only checks "PNG" */
    *w = 0;
    *h = 0;
    *good = 0;
    if (in == NULL || in_size < 28) return;
    if (in[1] != 'P' || in[2] != 'N' || in[3] != 'G') return;
    *w = be32(in + 16);
    *h = be32(in + 20);
    *good = 1;
}

/* result of the host-side container walk for one file */
typedef struct png_parsed {
    int ok;            /* container accepted; inflate + de-filter still to run */
    uint32_t w, h, ct;
    uint64_t est;      /* recipient_size the reference hands to inflate: 4wh + h + 1 */
    uint64_t zsize;    /* compressed_input_size handed to inflate (payload - 4 Adler bytes) */
    uint64_t packed;   /* total IDAT payload bytes (minus the 2-byte zlib header) */
    /* chunk spans for the CRC kernel (type + data) with the CRC stored in the file, and the
     * IDAT payload pieces for the gather kernel; offsets are relative to the file start */
    uint32_t n_chunks, cap_chunks, n_idat, cap_idat;
    struct png_chunk { uint64_t off, len; uint32_t crc; } *chunks;
    struct png_piece { uint64_t off, len; } *idat;
} png_parsed;

static int push_chunk(png_parsed *r, uint64_t off, uint64_t len, uint32_t crc)
{
    if (r->n_chunks == r->cap_chunks) {
        uint32_t cap = r->cap_chunks ? 2 * r->cap_chunks : 16;
        void *p = realloc(r->chunks, cap * sizeof *r->chunks);
        if (!p) return 0;
        r->chunks = (struct png_chunk *)p;
        r->cap_chunks = cap;
    }
    r->chunks[r->n_chunks].off = off;
    r->chunks[r->n_chunks].len = len;
    r->chunks[r->n_chunks].crc = crc;
    r->n_chunks++;
    return 1;
}
static int push_piece(png_parsed *r, uint64_t off, uint64_t len)
{
    if (r->n_idat == r->cap_idat) {
        uint32_t cap = r->cap_idat ? 2 * r->cap_idat : 16;
        void *p = realloc(r->idat, cap * sizeof *r->idat);
        if (!p) return 0;
        r->idat = (struct png_piece *)p;
        r->cap_idat = cap;
    }
    r->idat[r->n_idat].off = off;
    r->idat[r->n_idat].len = len;
    r->n_idat++;
    return 1;
}
static void parsed_free(png_parsed *r)
{
    free(r->chunks);
    free(r->idat);
    r->chunks = NULL;
    r->idat = NULL;
}

/* The container walk restated (src/decode_png.c:730-1367): returns ok = 0 wherever the
 * reference sets *out_good = 0 before inflate runs.  size_left bookkeeping mirrors the
 * reference's (it does not subtract the IHDR / PLTE bodies or the zlib header). */
static void png_walk(png_state *st, const uint8_t *in, uint64_t in_size, uint64_t rgba_size, png_parsed *r)
{
    memset(r, 0, sizeof *r);
    if (in == NULL || in_size < 8) return;
    if (in[1] != 'P' || in[2] != 'N' || in[3] != 'G') return;
    uint64_t at = 8, left = in_size - 8, packed = 0;
    int found_ihdr = 0, found_idat = 0, found_iend = 0, ready = 0;
    while (left >= 8 && !found_iend) {
        if (at + 8 > in_size) goto fail;
        uint32_t len = be32(in + at);
        const uint8_t *type = in + at + 4;
        at += 8;
        left -= 8;
        int is_idat = !memcmp(type, "IDAT", 4);
        if (!is_idat && found_idat) ready = 1; /* the reference runs inflate here (:775-860) */
        if ((uint64_t)len >= left || at + (uint64_t)len + 4 > in_size) goto fail; /* :886-898 */
        const uint64_t chunk_at = at - 4; /* the CRC covers type + data (:862-874) */
        if (!memcmp(type, "PLTE", 4)) { /* :900-950 */
            if (!found_ihdr) goto fail;
            if (r->ct == 0) goto fail;
            if (len % 3 != 0) goto fail;
            st->palette_size = len / 3;
            for (uint32_t i = 0; i < st->palette_size; i++) {
                if (i < 256) {
                    st->palette[i] = in[at];
                    st->palette[256 + i] = in[at + 1];
                    st->palette[512 + i] = in[at + 2];
                }
                at += 3;
            }
        } else if (!memcmp(type, "IHDR", 4)) { /* :951-1138 */
            found_ihdr = 1;
            if (at + 13 > in_size) goto fail;
            r->w = be32(in + at);
            r->h = be32(in + at + 4);
            uint8_t depth = in[at + 8];
            r->ct = in[at + 9];
            uint8_t filter_method = in[at + 11];
            at += 13;
            r->est = (uint64_t)(uint32_t)(r->w * r->h * 4u + r->h + 1u);
            if ((uint64_t)(uint32_t)(r->w * r->h * 4u) != rgba_size) goto fail;
            /* The reference does this arithmetic in uint32 (:965-985), so dimensions whose
             * 4wh + h + 1 wraps can pass the size check with a small buffer; its de-filter
             * loop then runs into the end of that buffer and reports out_good = 0 (:1459).
             * The kernels iterate the real w and h: such a file must never reach them. */
            if ((uint64_t)r->w * (uint64_t)r->h * 4u + (uint64_t)r->h + 1u != r->est) goto fail;
            if (r->ct != 2 && r->ct != 3 && r->ct != 6) goto fail;
            if (r->w < 1 || r->h < 1) goto fail;
            if ((uint64_t)(uint32_t)(r->w * r->h * 4u + r->h + 1u + INFLATE_HASHMAPS_SIZE) > st->wm_size) goto fail;
            if (depth != 8) goto fail;
            if (filter_method != 0) goto fail;
            if (left < 4) goto fail;
        } else if (is_idat) { /* :1139-1292 */
            if (!found_ihdr) goto fail;
            uint32_t dlen = len;
            if (!found_idat) {
                found_idat = 1;
                if (at + 2 > in_size) goto fail;
                uint8_t cmf = in[at], flg = in[at + 1];
                at += 2;
                dlen -= 2;
                if ((cmf & 15u) != 8) goto fail;
                uint32_t chk = (uint16_t)(flg | (uint16_t)(cmf << 8));
                if (chk == 0 || chk % 31u != 0) goto fail;
                if ((flg >> 5) & 1u) goto fail; /* FDICT */
            }
            if (at + (uint64_t)dlen > in_size) goto fail;
            if (dlen && !push_piece(r, at, dlen)) goto fail;
            packed += dlen;
            at += dlen;
            left -= dlen;
        } else if (!memcmp(type, "IEND", 4)) {
            found_iend = 1;
        } else if ((char)type[0] > 'Z') {
            at += len;
            left -= len;
        } else {
            goto fail; /* unknown critical chunk */
        }
        if (left < 4 || at + 4 > in_size) goto fail;
        uint32_t file_crc = be32(in + at);
        at += 4;
        left -= 4;
        /* verified on the GPU together with every other chunk of the batch (:1333-1354) */
        if (!push_chunk(r, chunk_at, (uint64_t)len + 4, file_crc)) goto fail;
    }
    if (!ready) goto fail; /* P6: inflate only runs when a non-IDAT chunk follows the IDATs */
    r->packed = packed;
    r->zsize = (uint64_t)(uint32_t)((uint32_t)packed - 4u); /* uint32 arithmetic as in :816 */
    /* inflate()'s own argument gates (src/inflate.c:826-844) fail these before a byte is read:
     * do not reserve arena space for them (an IDAT payload under 4 bytes wraps zsize to 4 GiB) */
    if (packed > 0xffffffffull || r->zsize < 5 || r->zsize > r->est) goto fail;
    r->ok = 1;
    return;
fail:
    parsed_free(r);
    r->ok = 0;
}

/* Host-only view of the container walk: would decode_png hand this file to inflate, and with
 * which sizes?  (0 wherever the reference returns out_good = 0 before inflate runs.) */
DEBIG_API int debig_png_probe(const uint8_t *in, const uint64_t in_size, const uint64_t rgba_values_size,
                              const uint32_t dpng_working_memory_size, uint32_t *out_width, uint32_t *out_height,
                              uint64_t *out_recipient_size, uint64_t *out_zlib_size)
{
    png_state st;
    png_parsed r;
    memset(&st, 0, sizeof st);
    st.wm_size = dpng_working_memory_size;
    png_walk(&st, in, in_size, rgba_values_size, &r);
    if (out_width) *out_width = r.ok ? r.w : 0;
    if (out_height) *out_height = r.ok ? r.h : 0;
    if (out_recipient_size) *out_recipient_size = r.ok ? r.est : 0;
    if (out_zlib_size) *out_zlib_size = r.ok ? r.zsize : 0;
    const int ok = r.ok;
    parsed_free(&r);
    return ok;
}

static int strict_mode(void)
{
    const char *e = getenv("DEBIG_STRICT");
    return e && e[0] == '1';
}

DEBIG_API int debig_decode_png_batch(const uint8_t *const *inputs, const uint64_t *input_sizes,
                                     uint8_t *const *outs, const uint64_t *out_sizes, uint8_t *goods,
                                     uint32_t n, const uint32_t thread_id)
{
    for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    if (thread_id >= DEBIG_MAX_THREADS || !g_png[thread_id].initialized) return 0; /* :691-700 */
    png_state *st = &g_png[thread_id];
    debig_ctx *c = debig_ctx_get(thread_id);
    if (n == 0) return 0;
    png_parsed *P = (png_parsed *)calloc(n, sizeof(png_parsed));
    debig_stream *desc = (debig_stream *)calloc(n, sizeof(debig_stream));
    debig_result *res = (debig_result *)calloc(n, sizeof(debig_result));
    debig_png_image *img = (debig_png_image *)calloc(n, sizeof(debig_png_image));
    debig_png_result *ires = (debig_png_result *)calloc(n, sizeof(debig_png_result));
    uint8_t *pals = (uint8_t *)calloc(n, 768);
    int rc = 2;
    if (!P || !desc || !res || !img || !ires || !pals) goto done;
    rc = 0;
    const int strict = strict_mode();
    uint64_t in_total = 0, out_total = 0, rgba_total = 0;
    for (uint32_t i = 0; i < n; i++) {
        png_walk(st, inputs[i], input_sizes[i], out_sizes[i], &P[i]);
        /* palettes are per-thread state in the reference: snapshot what this file sees */
        memcpy(pals + 768u * i, st->palette, 768);
        desc[i].in_off = in_total;
        desc[i].out_off = out_total;
        if (P[i].ok) {
            desc[i].in_len = P[i].zsize;
            desc[i].out_cap = P[i].est;
            /* P2: recipient = wm + 772, scratch = wm + est; the first
             * table sits at the first 16-aligned scratch byte (wm itself 16-aligned) */
            desc[i].p2_on = strict ? 0u : 1u;
            desc[i].flags = DEBIG_STREAM_IMAGE_ROWS; /* a hint for debig_pick_waves: long IDAT streams go as chunk tasks */
            desc[i].p2_est = P[i].est;
            desc[i].p2_s0 = (int64_t)P[i].est - 772 + (int64_t)((16u - (P[i].est & 15u)) & 15u);
            in_total += debig_align16(P[i].zsize) + 16;
            out_total += debig_align16(P[i].est) + 16 + 768; /* + room for the palette */
            rgba_total += debig_align16(out_sizes[i]) + 16;
            if (P[i].ct == 2 && !strict) rgba_total += debig_align16(out_sizes[i]) + 16; /* P3 replay: second buffer */
            if (P[i].ct == 3 && P[i].w > 16384u) rgba_total += debig_align16(P[i].w) + 32; /* wide palette rows: index-row scratch */
        }
    }
    {
        uint32_t n_ok = 0;
        for (uint32_t i = 0; i < n; i++) n_ok += P[i].ok != 0;
        if (n_ok == 0) goto done; /* nothing passed the container walk: no device work at all */
    }
    /* ---- bulk bytes go to HBM once: whole files; chunk CRCs and the IDAT concatenation
     *      happen there (the host has only looked at chunk headers) */
    uint64_t files_total = 0;
    uint32_t n_chunks = 0, n_pieces = 0;
    uint64_t *file_off = (uint64_t *)calloc(n, sizeof(uint64_t));
    if (!file_off) { rc = 2; goto done; }
    for (uint32_t i = 0; i < n; i++) {
        file_off[i] = files_total;
        if (!P[i].ok) continue;
        files_total += debig_align16(input_sizes[i]) + 16;
        n_chunks += P[i].n_chunks;
        n_pieces += P[i].n_idat;
    }
    debig_span *spans = (debig_span *)calloc(n_chunks ? n_chunks : 1, sizeof(debig_span));
    uint32_t *crcs = (uint32_t *)calloc(n_chunks ? n_chunks : 1, sizeof(uint32_t));
    debig_copy *copies = (debig_copy *)calloc(n_pieces ? n_pieces : 1, sizeof(debig_copy));
    if (!spans || !crcs || !copies) { rc = 2; goto done_bulk; }
    if ((rc = debig_devbuf_reserve(&c->files, files_total + 64)) ||
        (rc = debig_devbuf_reserve(&c->spans, (uint64_t)(n_chunks + 1) * sizeof(debig_span))) ||
        (rc = debig_devbuf_reserve(&c->crcs, (uint64_t)(n_chunks + 1) * sizeof(uint32_t))) ||
        (rc = debig_devbuf_reserve(&c->copies, (uint64_t)(n_pieces + 1) * sizeof(debig_copy))) ||
        (rc = debig_devbuf_reserve(&c->in, in_total + 64)) || (rc = debig_devbuf_reserve(&c->out, out_total + 64)) ||
        (rc = debig_devbuf_reserve(&c->rgba, rgba_total + 64)) ||
        (rc = debig_devbuf_reserve(&c->desc, (uint64_t)n * sizeof(debig_stream))) ||
        (rc = debig_devbuf_reserve(&c->res, (uint64_t)n * sizeof(debig_result))) ||
        (rc = debig_devbuf_reserve(&c->img, (uint64_t)n * sizeof(debig_png_image))) ||
        (rc = debig_devbuf_reserve(&c->imgres, (uint64_t)n * sizeof(debig_png_result))))
        goto done_bulk;
    {
        uint32_t ci = 0, pi = 0;
        {
            /* whole files up in one transfer through the page-locked arena */
            uint64_t *up_sizes = (uint64_t *)calloc(n, sizeof(uint64_t));
            if (!up_sizes) { rc = 2; goto done_bulk; }
            for (uint32_t i = 0; i < n; i++) up_sizes[i] = P[i].ok ? input_sizes[i] : 0;
            rc = debig_upload_packed(c, c->files.ptr, inputs, up_sizes, file_off, n, files_total);
            free(up_sizes);
        }
        for (uint32_t i = 0; i < n && !rc; i++) {
            if (!P[i].ok) continue;
            for (uint32_t k = 0; k < P[i].n_chunks; k++, ci++) {
                spans[ci].off = file_off[i] + P[i].chunks[k].off;
                spans[ci].len = P[i].chunks[k].len;
            }
            uint64_t pos = desc[i].in_off;
            for (uint32_t k = 0; k < P[i].n_idat; k++, pi++) {
                copies[pi].src_off = file_off[i] + P[i].idat[k].off;
                copies[pi].dst_off = pos;
                copies[pi].len = P[i].idat[k].len;
                pos += P[i].idat[k].len;
            }
        }
        if (!rc && n_chunks) {
            rc = debig_hip_memcpy_h2d(c->spans.ptr, spans, (uint64_t)n_chunks * sizeof(debig_span), NULL);
            if (!rc) rc = debig_hip_checksum_batch(c->files.ptr, (const debig_span *)c->spans.ptr,
                                                   (uint32_t *)c->crcs.ptr, n_chunks, 0, NULL);
            if (!rc) rc = debig_hip_memcpy_d2h(crcs, c->crcs.ptr, (uint64_t)n_chunks * sizeof(uint32_t), NULL);
        }
        if (!rc && n_pieces) {
            rc = debig_hip_memcpy_h2d(c->copies.ptr, copies, (uint64_t)n_pieces * sizeof(debig_copy), NULL);
            if (!rc) rc = debig_hip_gather(c->files.ptr, c->in.ptr, (const debig_copy *)c->copies.ptr, n_pieces, NULL);
        }
        if (!rc) rc = debig_hip_stream_sync(NULL);
        /* a chunk whose CRC does not match fails its file (src/decode_png.c:1333-1354) */
        ci = 0;
        for (uint32_t i = 0; i < n && !rc; i++) {
            if (!P[i].ok) continue;
            for (uint32_t k = 0; k < P[i].n_chunks; k++, ci++)
                if (crcs[ci] != P[i].chunks[k].crc) P[i].ok = 0;
            if (!P[i].ok) { desc[i].in_len = 0; desc[i].out_cap = 0; } /* fails the inflate gates */
        }
    }
done_bulk:
    free(spans);
    free(crcs);
    free(copies);
    free(file_off);
    if (rc) goto done;
    if ((rc = debig_launch_inflate_planned(c, c->in.ptr, desc, res, n))) goto done;
    /* de-filter the images whose stream inflated (the first filter byte is checked on the
     * device together with every other row's; the reference checks it first, :847-858) */
    uint32_t nimg = 0;
    uint32_t *map = (uint32_t *)calloc(n, sizeof(uint32_t));
    if (!map) { rc = 2; goto done; }
    uint64_t rgba_off = 0;
    for (uint32_t i = 0; i < n && !rc; i++) {
        if (!P[i].ok || !res[i].good) continue;
        debig_png_image *im = &img[nimg];
        im->stream_off = desc[i].out_off;
        im->rgba_off = rgba_off;
        im->pal_off = desc[i].out_off + debig_align16(P[i].est) + 16;
        im->width = P[i].w;
        im->height = P[i].h;
        im->color_type = P[i].ct;
        im->asserts_off = 0;
        im->tmp_off = 0;
        im->replay_p3 = 0;
        if (P[i].ct == 3) rc = debig_hip_memcpy_h2d((uint8_t *)c->out.ptr + im->pal_off, pals + 768u * i, 768, NULL);
        rgba_off += debig_align16(out_sizes[i]) + 16;
        if (P[i].ct == 2 && !strict) {
            /* the reference's RGB output depends on the caller's prior buffer contents (P3):
             * upload them, and give the kernel its second buffer */
            im->replay_p3 = 1;
            im->tmp_off = rgba_off;
            rgba_off += debig_align16(out_sizes[i]) + 16;
            if (!rc) rc = debig_hip_memcpy_h2d((uint8_t *)c->rgba.ptr + im->rgba_off, outs[i], out_sizes[i], NULL);
        }
        if (P[i].ct == 3 && P[i].w > 16384u) { /* include/debig_hip.h: debig_png_image.tmp_off for wide palette rows */
            im->tmp_off = rgba_off;
            rgba_off += debig_align16(P[i].w) + 32;
        }
        map[nimg++] = i;
    }
    if (!rc && nimg) {
        rc = debig_hip_memcpy_h2d(c->img.ptr, img, (uint64_t)nimg * sizeof(debig_png_image), NULL);
        if (!rc)
            rc = debig_hip_png_defilter_batch(c->out.ptr, c->rgba.ptr, (const debig_png_image *)c->img.ptr,
                                              (debig_png_result *)c->imgres.ptr, nimg, NULL);
        if (!rc) rc = debig_hip_memcpy_d2h(ires, c->imgres.ptr, (uint64_t)nimg * sizeof(debig_png_result), NULL);
        if (!rc) rc = debig_hip_stream_sync(NULL);
        if (!rc) {
            /* RGBA down in pieces through the page-locked arena, unpacked behind the wire */
            uint8_t **dn_dsts = (uint8_t **)calloc(nimg, sizeof(uint8_t *));
            uint64_t *dn_sizes = (uint64_t *)calloc(nimg, sizeof(uint64_t));
            uint64_t *dn_offs = (uint64_t *)calloc(nimg, sizeof(uint64_t));
            if (!dn_dsts || !dn_sizes || !dn_offs) rc = 2;
            uint64_t last_end = 0;
            for (uint32_t k = 0; k < nimg && !rc; k++) {
                uint32_t i = map[k];
                dn_offs[k] = img[k].rgba_off;
                if (!ires[k].good) continue;
                dn_dsts[k] = outs[i];
                dn_sizes[k] = out_sizes[i];
                if (dn_offs[k] + dn_sizes[k] > last_end) last_end = dn_offs[k] + dn_sizes[k];
                goods[i] = 1;
            }
            if (!rc) rc = debig_download_unpack(c, c->rgba.ptr, dn_dsts, dn_sizes, dn_offs, nimg, last_end);
            free(dn_dsts);
            free(dn_sizes);
            free(dn_offs);
        }
    }
    free(map);
done:
    if (rc)
        for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    if (P)
        for (uint32_t i = 0; i < n; i++) parsed_free(&P[i]);
    free(P);
    free(desc);
    free(res);
    free(img);
    free(ires);
    free(pals);
    return rc;
}

DEBIG_API void decode_png(const uint8_t *compressed_input, const uint64_t compressed_input_size,
                          const uint8_t *out_rgba_values, const uint64_t rgba_values_size,
                          const uint32_t thread_id, uint8_t *out_good)
{
    uint8_t good = 0;
    uint8_t *out = (uint8_t *)out_rgba_values;
    const uint8_t *in = compressed_input;
    *out_good = 0;
    if (!out || !in) return;
    debig_decode_png_batch(&in, &compressed_input_size, &out, &rgba_values_size, &good, 1, thread_id);
    *out_good = good;
}

/* ---- legacy generation (src/hellopng.c:154-200; thread_id 0, good is a uint32_t there) */
static void *legacy_malloc64(uint64_t n) { return malloc((size_t)n); }
static void *legacy_memset64(void *p, int c, uint64_t n) { return memset(p, c, (size_t)n); }
static void *legacy_memcpy64(void *d, const void *s, uint64_t n) { return memcpy(d, s, (size_t)n); }

DEBIG_API void init_PNG_decoder(void *(*malloc_funcptr)(size_t))
{
    (void)malloc_funcptr;
    decode_png_init(legacy_malloc64, free, legacy_memset64, legacy_memcpy64, 120000000u, 0);
}
DEBIG_API void get_PNG_width_height(const uint8_t *in, const uint64_t in_size, uint32_t *w, uint32_t *h,
                                    uint32_t *out_good)
{
    uint8_t g = 0;
    decode_png_get_width_height(in, in_size, w, h, &g);
    *out_good = g;
}
DEBIG_API void decode_PNG(const uint8_t *in, const uint64_t in_size, const uint8_t *out_rgba_values,
                          const uint64_t rgba_values_size, uint32_t *out_good)
{
    uint8_t g = 0;
    decode_png(in, in_size, out_rgba_values, rgba_values_size, 0, &g);
    *out_good = g;
}
