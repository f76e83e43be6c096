/*
 * decode_gz.h -- drop-in for debigulator's src/decode_gz.h (src/decode_gz.h:23-38).
 * Superset of the reference: DecodedData.data_size is filled in (the reference never
 * assigns it, src/decode_gz.c:299-300) and good is 0 on failure (left unset there).
 */
#ifndef DEBIG_DECODE_GZ_H
#define DEBIG_DECODE_GZ_H
#include <stdint.h>
#include <stddef.h>
#include "inflate.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct DecodedData {
    char *data;
    uint32_t data_size;
    uint32_t good;
} DecodedData;

void init_decode_gz(void *(*malloc_funcptr)(size_t __size),
                    void *(*arg_memset_func)(void *str, int c, size_t n),
                    void *(*arg_memcpy_func)(void *dest, const void *src, size_t n));

/* returns a DecodedData obtained from the caller's malloc (NULL if init_decode_gz was
 * never called, as in the reference); data is obtained from the same malloc */
DecodedData *decode_gz(uint8_t *compressed_bytes, uint32_t compressed_bytes_size);

/* Extension: n gzip members in one launch; outs[i] (capacity out_caps[i]) receive the
 * decompressed bytes.  Returns 0 or a HIP error code. */
int debig_decode_gz_batch(const uint8_t *const *inputs, const uint32_t *input_sizes,
                          uint8_t *const *outs, const uint64_t *out_caps, uint64_t *out_sizes,
                          uint32_t *goods, uint32_t n);

/* Same, plus trailer_ok[i] = 1 when the member's CRC-32 and ISIZE trailer (which the
 * reference reads and ignores, src/decode_gz.c:281-297) match the decompressed bytes; the
 * CRC is computed on the GPU.  trailer_ok may be NULL. */
int debig_decode_gz_batch_ex(const uint8_t *const *inputs, const uint32_t *input_sizes,
                             uint8_t *const *outs, const uint64_t *out_caps, uint64_t *out_sizes,
                             uint32_t *goods, uint32_t *trailer_ok, uint32_t n);

/* ---- beyond the reference: complete RFC 1952 handling -----------------
 * decode_gz() above keeps the reference's header rules (only FNAME is skipped, first member
 * only, trailer ignored).  debig_gunzip_batch is what a real-world .gz corpus needs:
 *   - FEXTRA, FNAME, FCOMMENT and FHCRC are all skipped correctly; reserved FLG bits reject;
 *   - every member of a multi-member file is decoded and the outputs are concatenated;
 *   - the CRC-32 and ISIZE of every member are verified (CRC computed on the GPU);
 *   - files whose members carry their size in a "BC" extra subfield (BGZF) are split on the
 *     host, so all their members are inflated by ONE launch; other multi-member files take
 *     one launch per member rank (member k of every file together), because a member's end
 *     is only known once it has been inflated;
 *   - inflate runs without the reference's size gates and without its end-of-input quirk
 *     (the trailer bytes follow the stream), i.e. plain RFC 1951 output.
 * Zero bytes after the last member are accepted (tape padding); anything else that is not
 * a gzip header ends the file with DEBIG_GZ_E_TRAILING.
 * status[i]: DEBIG_GZ_*; out_sizes[i]: bytes produced (also on error: what was decoded before
 * it); n_members[i]: members decoded (may be NULL).  Returns 0 or a device error code. */
enum {
    DEBIG_GZ_OK = 0,
    DEBIG_GZ_E_HEADER = 1,      /* not a gzip header / CM != 8 / reserved flag bits       */
    DEBIG_GZ_E_TRUNCATED = 2,   /* header, stream or trailer runs past the end of the file */
    DEBIG_GZ_E_INFLATE = 3,     /* the DEFLATE stream is damaged                           */
    DEBIG_GZ_E_OUTPUT_FULL = 4, /* out_caps[i] is too small                                */
    DEBIG_GZ_E_CRC = 5,         /* CRC-32 of a member does not match its trailer           */
    DEBIG_GZ_E_ISIZE = 6,       /* ISIZE of a member does not match                        */
    DEBIG_GZ_E_TRAILING = 7     /* garbage after the last member (output is complete)      */
};
/* Host-only helper (no GPU involved): parse ONE gzip member header at p (avail bytes left in
 * the file).  Returns DEBIG_GZ_OK, DEBIG_GZ_E_HEADER or DEBIG_GZ_E_TRUNCATED; on success
 * *header_len is where the DEFLATE data starts and *member_size the total member size taken
 * from a BGZF "BC" extra subfield (0 when there is none).  What debig_gunzip_batch uses. */
uint32_t debig_gz_parse_header(const uint8_t *p, uint64_t avail, uint64_t *header_len, uint64_t *member_size);

int debig_gunzip_batch(const uint8_t *const *inputs, const uint64_t *input_sizes,
                       uint8_t *const *outs, const uint64_t *out_caps, uint64_t *out_sizes,
                       uint32_t *status, uint32_t *n_members, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
