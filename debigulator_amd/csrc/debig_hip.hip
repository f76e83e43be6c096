// debig_hip.hip -- C-ABI shim (include/debig_hip.h) over the gfx950 kernels.
// Built by debigulator_amd/build.py:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/debig_hip.h"

#include "inflate_kernel.inc"
#include "png_kernel.inc"
#include "checksum_kernel.inc"

// one wavefront per workgroup; enough workgroups in flight to fill 256 CUs x (LDS-limited)
// resident waves, the rest grid-strides
static inline uint32_t pick_grid(uint32_t n, uint32_t per_cu)
{
    int dev = 0;
    hipDeviceProp_t p;
    uint32_t cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
        cus = (uint32_t)p.multiProcessorCount;
    uint32_t cap = cus * per_cu * 4u; /* several waves of workgroups: streams differ in length */
    return n < cap ? n : cap;
}

// BTYPE 1 tables, built once per device by a tiny kernel and then only copied into LDS
static CodeTabs *g_fixed_tabs[64];
static CodeTabs *fixed_tables(hipStream_t s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (g_fixed_tabs[dev]) return g_fixed_tabs[dev];
    CodeTabs *p = nullptr;
    if (hipMalloc(&p, sizeof(CodeTabs)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(debig_fixed_tables_kernel, dim3(1), dim3(64), 0, s, p);
    // later launches may use other streams: make the tables globally visible first
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    g_fixed_tabs[dev] = p;
    return p;
}

static CkTables *g_ck_tabs[64];
static CkTables *checksum_tables(hipStream_t s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (g_ck_tabs[dev]) return g_ck_tabs[dev];
    CkTables *p = nullptr;
    if (hipMalloc(&p, sizeof(CkTables)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(debig_checksum_tables_kernel, dim3(1), dim3(CK_THREADS), 0, s, p);
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    g_ck_tabs[dev] = p;
    return p;
}

extern "C" {

int debig_hip_checksum_batch(const void *d_arena, const debig_span *d_spans, uint32_t *d_out,
                             uint32_t n, uint32_t kind, void *hip_stream)
{
    if (n == 0) return 0;
    CkTables *t = checksum_tables((hipStream_t)hip_stream);
    if (!t) return (int)hipErrorOutOfMemory;
    hipLaunchKernelGGL(debig_checksum_kernel, dim3(n), dim3(CK_THREADS), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_arena, d_spans, d_out, n, kind, t);
    return (int)hipGetLastError();
}

int debig_hip_gather(const void *d_src_arena, void *d_dst_arena, const debig_copy *d_copies, uint32_t n,
                     void *hip_stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(debig_gather_kernel, dim3(n), dim3(CK_THREADS), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_src_arena, (uint8_t *)d_dst_arena, d_copies, n);
    return (int)hipGetLastError();
}

int debig_hip_inflate_batch(const void *d_in, void *d_out, const debig_stream *d_streams,
                            debig_result *d_results, uint32_t n, void *hip_stream)
{
    if (n == 0) return 0;
    CodeTabs *ft = fixed_tables((hipStream_t)hip_stream);
    if (!ft) return (int)hipErrorOutOfMemory;
    uint32_t grid = n; /* one workgroup per stream: the hardware scheduler balances lengths */
    hipLaunchKernelGGL(debig_inflate_kernel, dim3(grid), dim3(64), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_in, (uint8_t *)d_out, d_streams, d_results, n, ft);
    return (int)hipGetLastError();
}

int debig_hip_png_defilter_batch(const void *d_streams_arena, void *d_rgba_arena,
                                 const debig_png_image *d_images, debig_png_result *d_results,
                                 uint32_t n, void *hip_stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(debig_png_defilter_kernel, dim3(n), dim3(64), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images,
                       d_results, n);
    // colour type 2 images that ask for the reference's exact (P3) output; a no-op otherwise
    hipLaunchKernelGGL(debig_png_p3_kernel, dim3(n), dim3(64), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)d_streams_arena, (uint8_t *)d_rgba_arena, d_images,
                       d_results, n);
    return (int)hipGetLastError();
}

int debig_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int debig_hip_set_device(int dev) { return (int)hipSetDevice(dev); }
void *debig_hip_malloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    return p;
}
void debig_hip_free(void *p) { (void)hipFree(p); }
int debig_hip_memcpy_h2d(void *d, const void *h, uint64_t bytes, void *s)
{
    return (int)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)s);
}
int debig_hip_memcpy_d2h(void *h, const void *d, uint64_t bytes, void *s)
{
    return (int)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)s);
}
int debig_hip_memset(void *d, int v, uint64_t bytes, void *s)
{
    return (int)hipMemsetAsync(d, v, bytes, (hipStream_t)s);
}
int debig_hip_stream_sync(void *s) { return (int)hipStreamSynchronize((hipStream_t)s); }
const char *debig_hip_error_string(int err) { return hipGetErrorString((hipError_t)err); }
void *debig_hip_event_create(void)
{
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void *)e;
}
int debig_hip_event_record(void *ev, void *s) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)s); }
float debig_hip_event_elapsed_ms(void *a, void *b)
{
    float ms = -1.f;
    (void)hipEventSynchronize((hipEvent_t)b);
    (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
    return ms;
}
void debig_hip_event_destroy(void *ev) { (void)hipEventDestroy((hipEvent_t)ev); }

} // extern "C"
