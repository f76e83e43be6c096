/*
 * TEST INFRASTRUCTURE ONLY -- the parity oracle.
 *
 * A clean-room CPU restatement of the OBSERVABLE behaviour of the reference's
 * inflate() / decode_png() / decode_gz() (ArtOfBBQ/debigulator), quirks
 * included.  Nothing here ships in the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this
 * restatement against the compiled, unmodified reference (oracle/_ref, built by
 * oracle/Makefile from /root/reference/src in place) on the reference's own
 * resources/ files and on thousands of synthetic streams, and
 * tests/golden/ (json files) pins sha256 digests of reference outputs for the GPU box
 * where the reference source does not exist.
 */
#ifndef DEBIG_ORACLE_H
#define DEBIG_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* flags noted while decoding; tests use them to drop inputs on which the
 * reference itself is in undefined behaviour. */
#define ORC_UB_BTYPE3        0x01u /* reserved block type seen (ref: assert / skipped)        */
#define ORC_UB_CL16_AT_0     0x02u /* repeat-previous at position 0 (ref reads table[-1])     */
#define ORC_UB_CL_OVERSHOOT  0x04u /* code-length run past HLIT+HDIST (ref: assert only)      */
#define ORC_UB_SYM_286_287   0x08u /* litlen symbol 286/287 (ref indexes past its table)      */
#define ORC_UB_OVERSUBSCRIBED 0x10u/* over-subscribed code lengths (ref: assert only)         */
#define ORC_UB_INPUT_OVERRUN 0x20u /* bits consumed beyond compressed_input_size              */
#define ORC_UB_LONGCODE_500  0x40u /* >500 failed 13/14-bit probes in one table (ref asserts) */
#define ORC_UB_HLIT_GT_286   0x80u /* HLIT 287/288 (ref: assert only)                         */

typedef struct orc_stats {
    uint32_t ub_flags;
    uint32_t n_blocks, n_stored, n_fixed, n_dynamic;
    uint64_t n_symbols;      /* litlen symbols decoded (incl. EOB) */
    uint64_t n_matches;
    uint64_t bits_consumed;
    uint32_t tail_gate_fired; /* Q2: decoding stopped by the input-size gate */
} orc_stats;

/* called when a Huffman block header has been parsed (before its tables are
 * "built"): btype 1 or 2, bytes produced so far, and for btype 2 the 19
 * code-length-code lengths.  `out` is the recipient; decode_png's aliasing
 * replay (P2) patches it from here. */
typedef void (*orc_header_hook)(void *ctx, uint32_t btype, uint64_t out_pos,
                                const uint32_t *cl_lens, uint8_t *out);

/* Reference inflate() restated (src/inflate.c:786-1965).
 *   returns good (0/1).  *final_set tells whether the reference would have
 *   written *final_recipient_size at all (it does not on the argument gates).
 *   Input bytes at index >= in_size read as zero (the reference reads whatever
 *   follows in the caller's memory; only corrupt streams can observe this).
 *   Output never exceeds recipient_size: where the reference would overflow
 *   (asserts off) or abort (asserts on) this returns good = 0. */
uint32_t orc_inflate_ex(const uint8_t *in, uint64_t in_size, uint8_t *out,
                        uint64_t recipient_size, uint64_t *final, uint32_t *final_set,
                        orc_header_hook hook, void *hook_ctx, orc_stats *stats);

uint32_t orc_inflate(const uint8_t *in, uint64_t in_size, uint8_t *out,
                     uint64_t recipient_size, uint64_t *final);

/* decode_png flags */
#define ORC_PNG_STRICT      0x1u /* no P2 aliasing replay (spec-conforming tail)          */
#define ORC_PNG_NO_CRC      0x2u /* DECODE_PNG_IGNORE_CRC_CHECKS build                    */
#define ORC_PNG_ASSERTS_OFF 0x4u /* oracle-B behaviour for filter byte > 4 (zeros, good)  */

void orc_png_get_width_height(const uint8_t *in, uint64_t in_size, uint32_t *w, uint32_t *h,
                              uint8_t *good);

/* Reference decode_png() restated (src/decode_png.c:683-1567).  `in` is not
 * modified (the reference packs IDAT payloads to the front of it; a private
 * copy is used here).  out_rgba must be rgba_size bytes; its PRIOR contents are
 * an input for colour type 2 (reference bug P3).  wm_size is the
 * dpng_working_memory_size given to decode_png_init. */
uint32_t orc_decode_png(const uint8_t *in, uint64_t in_size, uint8_t *out_rgba,
                        uint64_t rgba_size, uint32_t wm_size, uint32_t flags);

/* palette state persists between calls in the reference (per thread_id) */
void orc_png_reset_palette(void);

/* Reference decode_gz() restated (src/decode_gz.c:101-301).  Returns good; the
 * decompressed bytes go to out (cap out_cap), *out_size gets their count.  */
uint32_t orc_decode_gz(const uint8_t *in, uint32_t in_size, uint8_t *out, uint64_t out_cap,
                       uint64_t *out_size);
/* helper: where the DEFLATE payload starts and how many bytes inflate() gets */
uint32_t orc_gz_locate(const uint8_t *in, uint32_t in_size, uint32_t *payload_off,
                       uint32_t *payload_len);

uint32_t orc_crc32(uint32_t crc, const uint8_t *buf, uint64_t len);

#ifdef __cplusplus
}
#endif
#endif
