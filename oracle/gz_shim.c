/* TEST INFRASTRUCTURE ONLY: forwards the old-generation inflate calls made by the
 * reference's decode_gz.c (see gz_shim.h) to the current API with thread_id 0. */
#include <inttypes.h>
#include <stddef.h>
void inflate_init(void *(*m)(uint64_t), void *(*ms)(void *, int, uint64_t),
                  void *(*mc)(void *, const void *, uint64_t), const uint32_t thread_id);
void inflate(uint8_t const *recipient, const uint64_t recipient_size,
             uint64_t *final_recipient_size, uint8_t *temp_working_memory,
             const uint64_t temp_working_memory_size, uint8_t const *compressed_input,
             const uint64_t compressed_input_size, uint32_t *out_good,
             const uint32_t thread_id);
int refh_inflate_inited(uint32_t tid);
void refh_mark_inflate_inited(uint32_t tid);

void inflate_init_v1(void *(*m)(size_t), void *(*ms)(void *, int, size_t),
                     void *(*mc)(void *, const void *, size_t))
{
    /* the reference asserts if thread 0 is initialised twice (inflate.c:47) */
    if (refh_inflate_inited(0)) return;
    inflate_init((void *(*)(uint64_t))m, (void *(*)(void *, int, uint64_t))ms,
                 (void *(*)(void *, const void *, uint64_t))mc, 0);
    refh_mark_inflate_inited(0);
}

void inflate_v1(uint8_t const *recipient, const uint64_t recipient_size,
                uint64_t *final_recipient_size, uint8_t *temp_working_memory,
                const uint64_t temp_working_memory_size, uint8_t const *compressed_input,
                const uint64_t compressed_input_size, uint32_t *out_good)
{
    inflate(recipient, recipient_size, final_recipient_size, temp_working_memory,
            temp_working_memory_size, compressed_input, compressed_input_size, out_good, 0);
}
