/*
 * TEST INFRASTRUCTURE ONLY (ours, not reference content).
 *
 * The reference's src/decode_gz.c still calls the previous generation of the
 * inflate API (3-arg inflate_init at decode_gz.c:15, 8-arg inflate at
 * decode_gz.c:256) and therefore does not compile against the current
 * src/inflate.h.  Force-including this header (gcc -include) lets the
 * UNMODIFIED decode_gz.c build: it suppresses inflate.h and routes the two old
 * call shapes to forwarding functions in gz_shim.c (thread_id = 0).
 */
#ifndef DEBIG_GZ_SHIM_H
#define DEBIG_GZ_SHIM_H
#define INFLATE_H
#include <inttypes.h>
#include <stddef.h>
void inflate_init_v1(void *(*m)(size_t), void *(*ms)(void *, int, size_t),
                     void *(*mc)(void *, const void *, size_t));
void inflate_v1(uint8_t const *recipient, const uint64_t recipient_size,
                uint64_t *final_recipient_size, uint8_t *temp_working_memory,
                const uint64_t temp_working_memory_size, uint8_t const *compressed_input,
                const uint64_t compressed_input_size, uint32_t *out_good);
#define inflate_init inflate_init_v1
#define inflate inflate_v1
#endif
