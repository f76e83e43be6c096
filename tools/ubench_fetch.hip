// Diagnostic (not on the product path): what rocprofv3's FETCH_SIZE reports for the access shapes of
// the LZ77 kernel on gfx950 -- (a) coalesced 4 B/lane row reads (token rows, nt), (b) 16-byte sc1
// buffer-load gathers at random unaligned addresses (history reads of far matches), (c) wide
// 16 B/lane streaming reads (the shape MI355X_MICROARCH.md says is reported at 1/2).
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- ./ubench_fetch
// Each kernel moves a known number of bytes from a buffer far larger than L2 + Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void rows4(const uint32_t *__restrict__ src, uint32_t *out, uint64_t n_dw)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dw; i += (uint64_t)gridDim.x * blockDim.x)
        acc += __builtin_nontemporal_load(src + i);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void wide16(const uint4 *__restrict__ src, uint32_t *out, uint64_t n_q)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_q; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// per lane `per` gathers of 16 bytes at pseudo-random byte addresses (unaligned), sc1 buffer loads
__global__ void gather16(const uint8_t *src, uint32_t *out, uint64_t span, uint32_t per, uint32_t align)
{
    typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
    uint64_t x = 0x9e3779b97f4a7c15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1u);
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per; k++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        uint64_t a = (x % (span - 64u)) & ~(uint64_t)(align - 1u);
        u32x4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a, 0, 16 /* sc1 */);
        acc += q[0] ^ q[1] ^ q[2] ^ q[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const uint64_t SPAN = 2000ull << 20; /* 2000 MiB: offsets fit the 31-bit buffer range */
    uint8_t *buf; uint32_t *out;
    hipMalloc(&buf, SPAN); hipMalloc(&out, 64);
    hipMemset(buf, 1, SPAN);
    hipDeviceSynchronize();
    const uint64_t bytes_rows = 1024ull << 20, bytes_wide = 1024ull << 20;
    const uint32_t blocks = 4096, per = 64;
    rows4<<<blocks, 256>>>((const uint32_t *)buf, out, bytes_rows / 4);
    wide16<<<blocks, 256>>>((const uint4 *)(buf + bytes_rows / 2), out, bytes_wide / 16);
    gather16<<<blocks, 256>>>(buf, out, SPAN, per, 1);
    gather16<<<blocks, 256>>>(buf, out, SPAN, per, 64);
    hipDeviceSynchronize();
    printf("rows4   : %.1f MB read as 4 B/lane coalesced nt loads\n", bytes_rows / 1e6);
    printf("wide16  : %.1f MB read as 16 B/lane coalesced loads\n", bytes_wide / 1e6);
    printf("gather16: %u gathers of 16 B at random unaligned addresses (%.1f MB useful); 64-byte lines touched ~ %.1f MB, 128-byte ~ %.1f MB\n",
           blocks * 256 * per, blocks * 256.0 * per * 16 / 1e6, blocks * 256.0 * per * 64 * (1 + 15.0 / 64) / 1e6,
           blocks * 256.0 * per * 128 * (1 + 15.0 / 128) / 1e6);
    printf("gather16 (64-byte aligned): %u gathers (%.1f MB useful); 64-byte lines %.1f MB, 128-byte lines %.1f MB\n",
           blocks * 256 * per, blocks * 256.0 * per * 16 / 1e6, blocks * 256.0 * per * 64 / 1e6, blocks * 256.0 * per * 128 / 1e6);
    return 0;
}
