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

#ifdef __cplusplus
}
#endif
#endif
