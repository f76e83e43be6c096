/*
 * libdebig_compat.a -- the literal `inflate` symbol for callers that cannot recompile against
 * include/inflate.h (SURVEY.md 8b rule 3).  The reference's entry point is named `inflate`
 * (src/inflate.h:51-60), the same unversioned global symbol zlib exports, and libamdhip64 /
 * librccl / python all load zlib: the shared library therefore exports debig_inflate and the
 * header maps `inflate` to it.  This archive is the opt-in for pre-built objects that reference
 * `inflate` directly: link it STATICALLY into the executable (never into a shared object, never
 * LD_PRELOAD it), and only when nothing else in that executable calls zlib's inflate.
 */
#include <stdint.h>

void debig_inflate(uint8_t const *recipient, const uint64_t recipient_size, uint64_t *final_recipient_size,
                   uint8_t *temp_working_memory, const uint64_t temp_working_memory_size,
                   uint8_t const *compressed_input, const uint64_t compressed_input_size, uint32_t *out_good,
                   const uint32_t thread_id);

void inflate(uint8_t const *recipient, const uint64_t recipient_size, uint64_t *final_recipient_size,
             uint8_t *temp_working_memory, const uint64_t temp_working_memory_size,
             uint8_t const *compressed_input, const uint64_t compressed_input_size, uint32_t *out_good,
             const uint32_t thread_id)
{
    debig_inflate(recipient, recipient_size, final_recipient_size, temp_working_memory, temp_working_memory_size,
                  compressed_input, compressed_input_size, out_good, thread_id);
}
