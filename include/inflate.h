/*
 * inflate.h -- drop-in for debigulator's src/inflate.h (ArtOfBBQ/debigulator), served by
 * the MI355X batched inflate path (libdebigulator_hip.so).
 *
 * Same three prototypes as the reference header (src/inflate.h:22-60).  One difference is
 * forced by the platform: the reference's entry point is literally named `inflate`, the
 * same global symbol zlib exports, and libamdhip64 / librccl / python all load zlib.  The
 * exported symbol is therefore `debig_inflate` and this header maps the reference's name
 * onto it at source level (the reference is consumed as source, README.md:127-137).
 * Define DEBIG_NO_INFLATE_RENAME to opt out and call debig_inflate() explicitly.
 */
#ifndef DEBIG_INFLATE_H
#define DEBIG_INFLATE_H
#include <inttypes.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/inflate.h:22-26.  The function pointers are accepted for source
 * compatibility; results are produced on the GPU and copied into the caller's buffers. */
void inflate_init(void *(*malloc_funcptr)(uint64_t __size),
                  void *(*arg_memset_func)(void *str, int c, uint64_t n),
                  void *(*arg_memcpy_func)(void *dest, const void *src, uint64_t n),
                  const uint32_t thread_id);

/* reference: src/inflate.h:28-30 */
void inflate_destroy(void (*free_funcptr)(void *to_free), const uint32_t thread_id);

/* reference: src/inflate.h:51-60.  Raw DEFLATE (no zlib/gzip wrapper) -> recipient.
 *   - *out_good = 1 on success, 0 on failure; *final_recipient_size = bytes produced
 *     (left untouched when an argument gate fails, exactly like the reference)
 *   - recipient_size must be >= compressed_input_size and compressed_input_size >= 5
 *     (reference gates, src/inflate.c:826-844)
 *   - temp_working_memory is not used (the reference needs >= ~3.2 MB of it and zeroes it
 *     on every call); it may be NULL
 *   - thread_id < 10 selects an independent context, as in the reference */
void debig_inflate(uint8_t const *recipient, const uint64_t recipient_size,
                   uint64_t *final_recipient_size, uint8_t *temp_working_memory,
                   const uint64_t temp_working_memory_size, uint8_t const *compressed_input,
                   const uint64_t compressed_input_size, uint32_t *out_good,
                   const uint32_t thread_id);
#ifndef DEBIG_NO_INFLATE_RENAME
#define inflate debig_inflate
#endif

/* Extension: N independent inflate() calls in one GPU launch (host buffers).
 * Per stream i: ins[i]/in_sizes[i] -> outs[i] (capacity out_caps[i]); results in
 * finals[i] / goods[i].  Returns 0, or a HIP error code if the device path failed
 * (then every goods[i] is 0). */
int debig_inflate_batch(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                        const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                        uint32_t n, const uint32_t thread_id);

/* Extension: the same batch over several GPUs of one host process (streams are
 * independent; stream i goes to GPU i mod n_devices, as BASELINE config 5 asks; no payload ever
 * crosses GPUs, and inside one process there is no shard map to broadcast: every worker derives
 * its share from i mod n_devices -- the RCCL broadcast of debigulator_amd/shard.py is for ranks
 * that are separate processes).  One host thread and one device context per GPU; staging as in
 * debig_inflate_batch.  n_devices = 0: every visible GPU.  Returns 0 or a HIP error code
 * (hipErrorInvalidDevice when fewer GPUs are visible).
 * debig_shard_round_robin lists the streams of one device (idx_out may be NULL): host only. */
uint32_t debig_shard_round_robin(uint32_t n, uint32_t n_devices, uint32_t device, uint32_t *idx_out);
int debig_inflate_batch_multi(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                              const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                              uint32_t n, uint32_t n_devices);
/* One call at a time (concurrent calls are serialised inside); the calling thread's current device is
 * left as it was; a NULL array returns hipErrorInvalidValue.  The per-device buffers are kept between
 * calls: debig_inflate_batch_multi_release() gives them back. */
void debig_inflate_batch_multi_release(void);

#ifdef __cplusplus
}
#endif
#endif
