/*
 * Drop-in inflate_init / inflate_destroy / inflate (reference src/inflate.h:22-60,
 * src/inflate.c:40-64, :786-1965) on top of the batched GPU path.  Host side, plain C:
 * argument gates, staging through device arenas, one kernel launch, results copied back
 * into the caller's buffers.  There is no CPU decode path in this library.
 */
#include <stdlib.h>
#include <string.h>
#include "inflate.h"
#include "debig_ctx.h"

static int g_inited[DEBIG_MAX_THREADS];

DEBIG_API void inflate_init(void *(*malloc_funcptr)(uint64_t), void *(*arg_memset_func)(void *, int, uint64_t),
                            void *(*arg_memcpy_func)(void *, const void *, uint64_t), const uint32_t thread_id)
{
    /* the reference allocates an InflateState per thread id and stores memset/memcpy in
     * process globals (src/inflate.c:40-56); nothing of that is needed here */
    (void)malloc_funcptr;
    (void)arg_memset_func;
    (void)arg_memcpy_func;
    if (thread_id < DEBIG_MAX_THREADS) g_inited[thread_id] = 1;
}

DEBIG_API void inflate_destroy(void (*free_funcptr)(void *), const uint32_t thread_id)
{
    (void)free_funcptr; /* reference impl frees interior pointers (a bug, src/inflate.c:58-64) */
    if (thread_id < DEBIG_MAX_THREADS) {
        g_inited[thread_id] = 0;
        debig_ctx_release(thread_id);
    }
}

/* shared with debig_gz.c: same as debig_inflate_batch, and reports where each recipient lives
 * in the context's device output arena (for checks that run on the device afterwards) */
int debig_inflate_batch_impl(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, const uint32_t thread_id, uint64_t *dev_out_offs);

DEBIG_API int debig_inflate_batch(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                                  const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                                  uint32_t n, const uint32_t thread_id)
{
    return debig_inflate_batch_impl(outs, out_caps, finals, ins, in_sizes, goods, n, thread_id, NULL);
}

static int inflate_batch_ctx(debig_ctx *c, uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, uint64_t *dev_out_offs);

int debig_inflate_batch_impl(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, const uint32_t thread_id, uint64_t *dev_out_offs)
{
    return inflate_batch_ctx(debig_ctx_get(thread_id), outs, out_caps, finals, ins, in_sizes, goods, n, dev_out_offs);
}

static int inflate_batch_ctx(debig_ctx *c, uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                             const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                             uint32_t n, uint64_t *dev_out_offs)
{
    for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    if (!c) return 1;
    if (n == 0) return 0;
    debig_stream *desc = (debig_stream *)calloc(n, sizeof(debig_stream));
    debig_result *res = (debig_result *)calloc(n, sizeof(debig_result));
    uint8_t *skip = (uint8_t *)calloc(n, 1);
    if (!desc || !res || !skip) { free(desc); free(res); free(skip); return 2; }
    uint64_t in_total = 0, out_total = 0;
    for (uint32_t i = 0; i < n; i++) {
        /* NULL gates, in the reference's order (src/inflate.c:797-824): nothing is written */
        if (!outs[i] || !finals || !ins[i]) { skip[i] = 1; desc[i].in_off = in_total; desc[i].out_off = out_total; continue; }
        desc[i].in_off = in_total;
        desc[i].in_len = in_sizes[i];
        desc[i].out_off = out_total;
        if (dev_out_offs) dev_out_offs[i] = out_total;
        desc[i].out_cap = out_caps[i];
        /* a stream that fails the size gates is never read or written by the kernel */
        int gated = out_caps[i] < in_sizes[i] || in_sizes[i] < 5;
        in_total += debig_align16(gated ? 0 : in_sizes[i]) + 16;
        out_total += debig_align16(gated ? 0 : out_caps[i]) + 16;
    }
    int rc = 0;
    if ((rc = debig_devbuf_reserve(&c->in, in_total + 64)) || (rc = debig_devbuf_reserve(&c->out, out_total + 64)) ||
        (rc = debig_devbuf_reserve(&c->desc, (uint64_t)n * sizeof(debig_stream))) ||
        (rc = debig_devbuf_reserve(&c->res, (uint64_t)n * sizeof(debig_result))))
        goto done;
    {
        /* inputs: packed into the page-locked arena by several threads, ONE host-to-device copy */
        uint64_t *up_sizes = (uint64_t *)calloc(n, sizeof(uint64_t));
        uint64_t *up_offs = (uint64_t *)calloc(n, sizeof(uint64_t));
        if (!up_sizes || !up_offs) { free(up_sizes); free(up_offs); rc = 2; goto done; }
        for (uint32_t i = 0; i < n; i++) {
            const int gated = skip[i] || out_caps[i] < in_sizes[i] || in_sizes[i] < 5;
            up_sizes[i] = gated ? 0 : in_sizes[i];
            up_offs[i] = desc[i].in_off;
        }
        rc = debig_upload_packed(c, c->in.ptr, ins, up_sizes, up_offs, n, in_total);
        free(up_sizes);
        free(up_offs);
    }
    if (rc) goto done;
    /* streams with NULL arguments are still launched as zero-length (they fail the gates) */
    for (uint32_t i = 0; i < n; i++)
        if (skip[i]) { desc[i].in_len = 0; desc[i].out_cap = 0; }
    if ((rc = debig_launch_inflate_planned(c, c->in.ptr, desc, res, n))) goto done;
    {
        /* outputs: the arena comes down in pieces, unpacked by several threads behind the wire */
        uint64_t *dn_sizes = (uint64_t *)calloc(n, sizeof(uint64_t));
        uint64_t *dn_offs = (uint64_t *)calloc(n, sizeof(uint64_t));
        if (!dn_sizes || !dn_offs) { free(dn_sizes); free(dn_offs); rc = 2; goto done; }
        uint64_t last_end = 0;
        for (uint32_t i = 0; i < n; i++) {
            dn_offs[i] = desc[i].out_off;
            if (skip[i]) continue;
            if (res[i].final_set) {
                finals[i] = res[i].final_size;
                dn_sizes[i] = res[i].final_size < out_caps[i] ? res[i].final_size : out_caps[i];
                if (dn_sizes[i] && dn_offs[i] + dn_sizes[i] > last_end) last_end = dn_offs[i] + dn_sizes[i];
            }
            goods[i] = res[i].good;
        }
        rc = debig_download_unpack(c, c->out.ptr, outs, dn_sizes, dn_offs, n, last_end);
        free(dn_sizes);
        free(dn_offs);
    }
done:
    if (rc)
        for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    free(desc);
    free(res);
    free(skip);
    return rc;
}

/* ---- one host process, several GPUs (streams are independent, member i -> GPU
 * i mod n, no payload collective; within one process there is no shard map to broadcast either:
 * every worker thread derives its share from i mod n_devices) */
DEBIG_API uint32_t debig_shard_round_robin(uint32_t n, uint32_t n_devices, uint32_t device, uint32_t *idx_out)
{
    uint32_t k = 0;
    if (n_devices == 0 || device >= n_devices) return 0;
    for (uint32_t i = device; i < n; i += n_devices) {
        if (idx_out) idx_out[k] = i;
        k++;
    }
    return k;
}

#include <pthread.h>
#define DEBIG_MAX_DEVICES 16
static debig_ctx g_dev_ctx[DEBIG_MAX_DEVICES]; /* one context per device for the multi-device call */
static pthread_mutex_t g_multi_lock = PTHREAD_MUTEX_INITIALIZER; /* the contexts serve one multi call at a time */
/* DEBIG_MULTI_ONE_DEVICE=1 (rehearsal on a one-GPU machine): every worker thread and its context use
 * device 0, so n_devices may exceed the GPUs present; the sharding and the threading are the real ones */
static int multi_one_device(void)
{
    const char *e = getenv("DEBIG_MULTI_ONE_DEVICE");
    return e && *e && *e != '0';
}

typedef struct multi_job {
    uint32_t device, n_devices, n;
    uint8_t *const *outs;
    const uint64_t *out_caps;
    uint64_t *finals;
    const uint8_t *const *ins;
    const uint64_t *in_sizes;
    uint32_t *goods;
    int rc;
} multi_job;

static void *multi_worker(void *arg)
{
    multi_job *j = (multi_job *)arg;
    const uint32_t m = debig_shard_round_robin(j->n, j->n_devices, j->device, NULL);
    j->rc = 0;
    if (m == 0) return NULL;
    /* this thread's shard as dense arrays */
    uint8_t **outs = (uint8_t **)malloc(m * sizeof(*outs));
    const uint8_t **ins = (const uint8_t **)malloc(m * sizeof(*ins));
    uint64_t *caps = (uint64_t *)malloc(m * sizeof(uint64_t)), *sizes = (uint64_t *)malloc(m * sizeof(uint64_t));
    uint64_t *finals = (uint64_t *)malloc(m * sizeof(uint64_t));
    uint32_t *goods = (uint32_t *)malloc(m * sizeof(uint32_t));
    if (!outs || !ins || !caps || !sizes || !finals || !goods) j->rc = 2;
    for (uint32_t k = 0, i = j->device; !j->rc && k < m; k++, i += j->n_devices) {
        outs[k] = j->outs[i];
        ins[k] = j->ins[i];
        caps[k] = j->out_caps[i];
        sizes[k] = j->in_sizes[i];
        finals[k] = j->finals[i];
    }
    if (!j->rc) j->rc = debig_hip_set_device(multi_one_device() ? 0 : (int)j->device); /* per-thread current device */
    if (!j->rc) j->rc = inflate_batch_ctx(&g_dev_ctx[j->device], outs, caps, finals, ins, sizes, goods, m, NULL);
    for (uint32_t k = 0, i = j->device; k < m; k++, i += j->n_devices) {
        j->goods[i] = j->rc ? 0u : goods[k];
        if (!j->rc) j->finals[i] = finals[k];
    }
    free(outs); free(ins); free(caps); free(sizes); free(finals); free(goods);
    return NULL;
}

DEBIG_API int debig_inflate_batch_multi(uint8_t *const *outs, const uint64_t *out_caps, uint64_t *finals,
                                        const uint8_t *const *ins, const uint64_t *in_sizes, uint32_t *goods,
                                        uint32_t n, uint32_t n_devices)
{
    if (!goods) return 1; /* hipErrorInvalidValue */
    for (uint32_t i = 0; i < n; i++) goods[i] = 0;
    if (n == 0) return 0;
    if (!outs || !out_caps || !finals || !ins || !in_sizes) return 1; /* the single-device call's NULL gates, for the arrays */
    const int have = debig_hip_device_count();
    if (n_devices == 0) n_devices = have > 0 ? (uint32_t)have : 0;
    if (n_devices == 0 || n_devices > DEBIG_MAX_DEVICES || ((int)n_devices > have && !multi_one_device())) return 101; /* hipErrorInvalidDevice */
    const int caller_dev = debig_hip_get_device();
    pthread_mutex_lock(&g_multi_lock);
    multi_job jobs[DEBIG_MAX_DEVICES];
    pthread_t th[DEBIG_MAX_DEVICES];
    uint32_t started = 0;
    for (uint32_t d = 0; d < n_devices; d++) {
        multi_job *j = &jobs[d];
        j->device = d; j->n_devices = n_devices; j->n = n;
        j->outs = outs; j->out_caps = out_caps; j->finals = finals; j->ins = ins; j->in_sizes = in_sizes; j->goods = goods;
        j->rc = 0;
        if (d + 1 == n_devices || pthread_create(&th[d], NULL, multi_worker, j) != 0) multi_worker(j);
        else started |= 1u << d;
    }
    int rc = 0;
    for (uint32_t d = 0; d < n_devices; d++) {
        if (started & (1u << d)) pthread_join(th[d], NULL);
        if (jobs[d].rc && !rc) rc = jobs[d].rc;
    }
    pthread_mutex_unlock(&g_multi_lock);
    if (caller_dev >= 0) (void)debig_hip_set_device(caller_dev); /* the last share ran on the calling thread */
    return rc;
}

/* give back what debig_inflate_batch_multi keeps between calls: per device the device buffers, the
 * token workspace, the page-locked arenas */
DEBIG_API void debig_inflate_batch_multi_release(void)
{
    const int caller_dev = debig_hip_get_device();
    pthread_mutex_lock(&g_multi_lock);
    const int have = debig_hip_device_count();
    for (int d = 0; d < DEBIG_MAX_DEVICES; d++) {
        if (d < have && !multi_one_device()) (void)debig_hip_set_device(d);
        debig_ctx_release_ptr(&g_dev_ctx[d]);
    }
    pthread_mutex_unlock(&g_multi_lock);
    if (caller_dev >= 0) (void)debig_hip_set_device(caller_dev);
}

DEBIG_API void debig_inflate(uint8_t const *recipient, const uint64_t recipient_size, uint64_t *final_recipient_size,
                             uint8_t *temp_working_memory, const uint64_t temp_working_memory_size,
                             uint8_t const *compressed_input, const uint64_t compressed_input_size,
                             uint32_t *out_good, const uint32_t thread_id)
{
    (void)temp_working_memory;
    (void)temp_working_memory_size;
    /* reference gate order: recipient, final_recipient_size, compressed_input (src/inflate.c:797-824) */
    if (recipient == NULL || final_recipient_size == NULL || compressed_input == NULL) {
        *out_good = 0;
        return;
    }
    uint8_t *out = (uint8_t *)recipient;
    uint64_t fin = *final_recipient_size;
    uint32_t good = 0;
    const uint8_t *in = compressed_input;
    int rc = debig_inflate_batch(&out, &recipient_size, &fin, &in, &compressed_input_size, &good, 1, thread_id);
    if (rc) good = 0;
    *final_recipient_size = fin;
    *out_good = good;
}
