// Diagnostic microbenchmark (not on the product path): what ONE wavefront pays per instruction on
// gfx950 -- dependent / independent integer VALU, LDS pointer chases alone and interleaved, the
// symbol-decode recurrence of the scan kernel.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_latency
// tools/ubench_latency.hip ; run on the GPU box; prints shader cycles (s_memtime) per loop iteration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define N_IT 4096

__device__ __forceinline__ uint64_t now() { return __builtin_amdgcn_s_memtime(); }

// dependent chain of K VALU ops per iteration
template <int MODE> __global__ void k_valu(uint32_t *out, uint64_t *cyc, uint32_t seed)
{
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;
    uint64_t t0 = now();
    for (int i = 0; i < N_IT; i++) {
        if (MODE == 0) { /* 8 dependent adds/ands/shifts */
            a = (a + b) & 0xffffffu; a = (a << 1) ^ b; a = a + 3u; a = (a >> 2) + b;
            a = (a + b) & 0xffffffu; a = (a << 1) ^ b; a = a + 3u; a = (a >> 2) + b;
        } else if (MODE == 1) { /* 2 chains x 4 */
            a = (a + b) & 0xffffffu; c = (c + d) & 0xffffffu; a = (a << 1) ^ b; c = (c << 1) ^ d;
            a = a + 3u; c = c + 3u; a = (a >> 2) + b; c = (c >> 2) + d;
        } else { /* 8 dependent alignbit / bfe */
            a = __builtin_amdgcn_alignbit(a, b, a); a = __builtin_amdgcn_ubfe(a, 3, 20) + b;
            a = __builtin_amdgcn_alignbit(a, b, a); a = __builtin_amdgcn_ubfe(a, 3, 20) + b;
            a = __builtin_amdgcn_alignbit(a, b, a); a = __builtin_amdgcn_ubfe(a, 3, 20) + b;
            a = __builtin_amdgcn_alignbit(a, b, a); a = __builtin_amdgcn_ubfe(a, 3, 20) + b;
        }
    }
    uint64_t t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS pointer chase: CH independent chains per lane; table of 2048 u16 (random), like lit_tab
template <int CH> __global__ void k_lds(const uint16_t *tab, uint32_t *out, uint64_t *cyc)
{
    __shared__ uint16_t T[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) T[i] = tab[i];
    __syncthreads();
    uint32_t p[CH];
    for (int c = 0; c < CH; c++) p[c] = (threadIdx.x * 37u + c * 911u) & 2047u;
    uint64_t t0 = now();
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) p[c] = T[p[c]];
    }
    uint64_t t1 = now();
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s += p[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the scan kernel's recurrence: window read (3 dwords) -> table -> table -> advance; CH chains
template <int CH> __global__ void k_decode(const uint16_t *tab, const uint32_t *win, uint32_t *out, uint64_t *cyc)
{
    __shared__ uint16_t T[2048];
    __shared__ uint32_t W[1104];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) T[i] = tab[i];
    for (int i = threadIdx.x; i < 1104; i += blockDim.x) W[i] = win[i];
    __syncthreads();
    uint32_t lp[CH];
    for (int c = 0; c < CH; c++) lp[c] = ((threadIdx.x & 63u) * 544u + c * 1000u) % 30000u;
    uint64_t t0 = now();
    for (int i = 0; i < N_IT; i++) {
        uint32_t lo[CH], hi[CH], e[CH], n1[CH], de[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            uint32_t w = lp[c] >> 5;
            uint32_t d0 = W[w], d1 = W[w + 1], d2 = W[w + 2];
            lo[c] = __builtin_amdgcn_alignbit(d1, d0, lp[c]);
            hi[c] = __builtin_amdgcn_alignbit(d2, d1, lp[c]);
        }
#pragma unroll
        for (int c = 0; c < CH; c++) e[c] = T[lo[c] & 511u];
#pragma unroll
        for (int c = 0; c < CH; c++) { n1[c] = (e[c] & 15u) + ((e[c] >> 4) & 7u); de[c] = T[1024u + (__builtin_amdgcn_alignbit(hi[c], lo[c], n1[c]) & 31u)]; }
#pragma unroll
        for (int c = 0; c < CH; c++) { lp[c] += n1[c] + (de[c] & 31u) + 1u; lp[c] = lp[c] > 30000u ? lp[c] - 30000u : lp[c]; }
    }
    uint64_t t1 = now();
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s += lp[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same recurrence with the second table look-up replaced by arithmetic on the bits (a fixed-Huffman
// distance code: 5 bits, reversed; extra bits from the symbol)
template <int CH> __global__ void k_decode_fx(const uint16_t *tab, const uint32_t *win, uint32_t *out, uint64_t *cyc)
{
    __shared__ uint16_t T[2048];
    __shared__ uint32_t W[1104];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) T[i] = tab[i];
    for (int i = threadIdx.x; i < 1104; i += blockDim.x) W[i] = win[i];
    __syncthreads();
    uint32_t lp[CH];
    for (int c = 0; c < CH; c++) lp[c] = ((threadIdx.x & 63u) * 544u + c * 1000u) % 30000u;
    uint64_t t0 = now();
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            uint32_t w = lp[c] >> 5;
            uint32_t d0 = W[w], d1 = W[w + 1], d2 = W[w + 2];
            uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, lp[c]);
            uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, lp[c]);
            uint32_t e = T[lo & 511u];
            uint32_t n1 = (e & 15u) + ((e >> 4) & 7u);
            uint32_t b2 = __builtin_amdgcn_alignbit(hi, lo, n1);
            uint32_t D = __builtin_bitreverse32(b2) >> 27;
            uint32_t ex = (D >> 1) > 1u ? (D >> 1) - 1u : 0u;
            uint32_t y = (e & 128u) ? 5u + ex : 0u;
            lp[c] += n1 + y + 1u;
            lp[c] = lp[c] > 30000u ? lp[c] - 30000u : lp[c];
        }
    }
    uint64_t t1 = now();
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s += lp[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F> static void run(const char *name, F launch, int blocks, uint64_t *d_cyc, double per)
{
    launch();
    hipDeviceSynchronize();
    launch();
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; i++) s += (double)h[i];
    printf("%-44s %8.1f cycles / iteration   (%6.2f per unit)\n", name, s / blocks / N_IT, s / blocks / N_IT / per);
}

int main()
{
    uint32_t *d_out; uint64_t *d_cyc; uint16_t *d_tab; uint32_t *d_win;
    hipMalloc(&d_out, 1 << 24); hipMalloc(&d_cyc, 1 << 16); hipMalloc(&d_tab, 4096); hipMalloc(&d_win, 1104 * 4);
    std::vector<uint16_t> tab(2048); std::vector<uint32_t> win(1104);
    uint64_t x = 88172645463325252ull;
    for (auto &v : tab) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint16_t)(x & 2047u); }
    for (int i = 1024; i < 2048; i++) tab[i] &= 31;
    for (int i = 0; i < 1024; i++) tab[i] = (uint16_t)((tab[i] & 0xff80u) | (7 + (tab[i] & 1)) | ((tab[i] >> 3) & 0x30));
    for (auto &v : win) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x; }
    hipMemcpy(d_tab, tab.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(d_win, win.data(), 1104 * 4, hipMemcpyHostToDevice);
    for (int wpc : {1, 4, 16}) { /* wavefronts per CU: 1 = one wave alone on a SIMD, 4 = one per SIMD, 16 = 4 per SIMD */
        const int blocks = 256 * wpc; /* one 64-thread workgroup per wavefront */
        printf("---- %d wavefront(s) per CU (%d workgroups of 64)\n", wpc, blocks);
        run("VALU 8 dependent int ops", [&] { k_valu<0><<<blocks, 64>>>(d_out, d_cyc, 1); }, blocks, d_cyc, 8);
        run("VALU 2 chains x 4 ops", [&] { k_valu<1><<<blocks, 64>>>(d_out, d_cyc, 1); }, blocks, d_cyc, 8);
        run("VALU 8 dependent alignbit/bfe+add", [&] { k_valu<2><<<blocks, 64>>>(d_out, d_cyc, 1); }, blocks, d_cyc, 12);
        run("LDS u16 chase, 1 chain", [&] { k_lds<1><<<blocks, 64>>>(d_tab, d_out, d_cyc); }, blocks, d_cyc, 1);
        run("LDS u16 chase, 2 chains", [&] { k_lds<2><<<blocks, 64>>>(d_tab, d_out, d_cyc); }, blocks, d_cyc, 2);
        run("LDS u16 chase, 4 chains", [&] { k_lds<4><<<blocks, 64>>>(d_tab, d_out, d_cyc); }, blocks, d_cyc, 4);
        run("decode recurrence, 1 chain", [&] { k_decode<1><<<blocks, 64>>>(d_tab, d_win, d_out, d_cyc); }, blocks, d_cyc, 1);
        run("decode, distance by arithmetic, 1 chain", [&] { k_decode_fx<1><<<blocks, 64>>>(d_tab, d_win, d_out, d_cyc); }, blocks, d_cyc, 1);
        run("decode recurrence, 2 chains", [&] { k_decode<2><<<blocks, 64>>>(d_tab, d_win, d_out, d_cyc); }, blocks, d_cyc, 2);
        run("decode recurrence, 4 chains", [&] { k_decode<4><<<blocks, 64>>>(d_tab, d_win, d_out, d_cyc); }, blocks, d_cyc, 4);
    }
    return 0;
}
