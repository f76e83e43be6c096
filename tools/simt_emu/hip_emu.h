/*
 * DEVELOPMENT / TEST TOOLING -- not part of the product.
 *
 * A tiny lock-step SIMT emulator so the HIP kernels of debigulator_amd/csrc can be
 * compiled with g++ and run on the CPU (under ASan/UBSan) in a container that has
 * no GPU.  One workgroup runs at a time; each of its threads is a ucontext fiber;
 * wave collectives (__shfl*, __ballot, __syncthreads ...) are rendezvous points
 * at which every thread of the workgroup must arrive (the kernels are written so
 * that collectives are only reached in convergent code).
 *
 * Only the subset of HIP the kernels use is provided.  The emulator exists to
 * check INDEXING and ALGORITHM (bit-exactness against the oracle); it says
 * nothing about performance or about the memory model of the real machine.
 */
#ifndef DEBIG_HIP_EMU_H
#define DEBIG_HIP_EMU_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <ucontext.h>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#define EMU_ASAN 1
#else
#define EMU_ASAN 0
#endif

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define DEBIG_EMU 1

struct emu_dim3 { unsigned x, y, z; };

struct emu_fiber {
    ucontext_t ctx;
    void *sp; /* x86-64: saved stack pointer of the hand-written switch (no system calls) */
    char *stack;
    size_t stack_size;
    unsigned tid;
    int done;
    void *fake_stack;
};

struct emu_state {
    ucontext_t sched;
    void *sched_sp; /* x86-64: the scheduler's saved stack pointer */
    std::vector<emu_fiber> fibers;
    unsigned cur;
    unsigned nthreads;
    /* rendezvous: index 0..15 = the wavefronts of the workgroup (wave collectives), 16 = the
     * whole workgroup (__syncthreads) */
    uint64_t slot[1024];
    unsigned arrived[17];
    unsigned generation[17];
    unsigned waiting;
    unsigned long long progress;
    void (*entry)(void *);
    void *entry_arg;
    void *sched_fake_stack;
};

extern emu_state g_emu;
extern emu_dim3 g_emu_blockIdx, g_emu_blockDim, g_emu_gridDim;

struct emu_tid_proxy {
    struct X { operator unsigned() const { return g_emu.fibers[g_emu.cur].tid; } } x;
};
static emu_tid_proxy threadIdx __attribute__((unused));
#define blockIdx g_emu_blockIdx
#define blockDim g_emu_blockDim
#define gridDim g_emu_gridDim

/* Fiber switch.  swapcontext() saves and restores the signal mask with two system calls per switch,
 * which was most of the emulator's run time; on x86-64 a dozen instructions do (callee-saved
 * registers, MXCSR and the x87 control word travel on the fiber's own stack). */
#if defined(__x86_64__)
extern "C" void emu_ctx_switch(void **from_sp, void *to_sp);
static inline void emu_to_scheduler(emu_fiber &f) { emu_ctx_switch(&f.sp, g_emu.sched_sp); }
static inline void emu_to_fiber(emu_fiber &f) { emu_ctx_switch(&g_emu.sched_sp, f.sp); }
#else
static inline void emu_to_scheduler(emu_fiber &f) { swapcontext(&f.ctx, &g_emu.sched); }
static inline void emu_to_fiber(emu_fiber &f) { swapcontext(&g_emu.sched, &f.ctx); }
#endif

static inline void emu_yield()
{
    emu_fiber &f = g_emu.fibers[g_emu.cur];
#if EMU_ASAN
    __sanitizer_start_switch_fiber(&f.fake_stack, g_emu.sched.uc_stack.ss_sp, g_emu.sched.uc_stack.ss_size);
#endif
    emu_to_scheduler(f);
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(f.fake_stack, nullptr, nullptr);
#endif
}

/* rendezvous of one group of threads: group 16 = all threads of the workgroup, group w < 16 =
 * the 64 threads of wavefront w.  Each exchanges one 64-bit value through slot[tid]. */
static inline void emu_rendezvous_group(unsigned grp, unsigned nmembers, uint64_t mine)
{
    unsigned me = g_emu.fibers[g_emu.cur].tid;
    for (int phase = 0; phase < 2; phase++) {
        unsigned gen = g_emu.generation[grp];
        if (phase == 0) g_emu.slot[me] = mine;
        g_emu.arrived[grp]++;
        if (g_emu.arrived[grp] == nmembers) {
            g_emu.arrived[grp] = 0;
            g_emu.generation[grp]++;
            g_emu.progress++;
        } else {
            g_emu.waiting++;
            while (g_emu.generation[grp] == gen) emu_yield();
            g_emu.waiting--;
        }
        /* between the two phases every member reads what it needs from slot[]; the second
         * phase keeps anybody from overwriting slot[] before everyone has read it */
        if (phase == 0) return;
    }
}
static inline void emu_rendezvous_finish(unsigned grp, unsigned nmembers)
{
    unsigned gen = g_emu.generation[grp];
    g_emu.arrived[grp]++;
    if (g_emu.arrived[grp] == nmembers) {
        g_emu.arrived[grp] = 0;
        g_emu.generation[grp]++;
        g_emu.progress++;
    } else {
        g_emu.waiting++;
        while (g_emu.generation[grp] == gen) emu_yield();
        g_emu.waiting--;
    }
}

/* all threads of the WORKGROUP exchange one 64-bit value */
static inline void emu_rendezvous(uint64_t mine, uint64_t *all)
{
    emu_rendezvous_group(16, g_emu.nthreads, mine);
    for (unsigned i = 0; i < g_emu.nthreads; i++) all[i] = g_emu.slot[i];
    emu_rendezvous_finish(16, g_emu.nthreads);
}

/* the 64 threads of the calling thread's WAVEFRONT exchange one 64-bit value; all[0..63] */
static inline void emu_wave_rendezvous(uint64_t mine, uint64_t *all)
{
    unsigned me = g_emu.fibers[g_emu.cur].tid;
    unsigned w = me >> 6, base = me & ~63u;
    unsigned members = g_emu.nthreads - base < 64 ? g_emu.nthreads - base : 64;
    emu_rendezvous_group(w, members, mine);
    for (unsigned i = 0; i < 64; i++) all[i] = (base + i < g_emu.nthreads) ? g_emu.slot[base + i] : 0;
    emu_rendezvous_finish(w, members);
}

static inline void emu___syncthreads()
{
    uint64_t all[1024];
    emu_rendezvous(0, all);
}

template <typename T> static inline T emu_xchg(T v, int src_lane, bool clamp_self)
{
    uint64_t all[64];
    uint64_t bits = 0;
    static_assert(sizeof(T) <= 8, "shfl of <= 8 bytes");
    memcpy(&bits, &v, sizeof(T));
    emu_wave_rendezvous(bits, all);
    unsigned me = g_emu.fibers[g_emu.cur].tid;
    int lane = (int)(me & 63u);
    if (src_lane < 0 || src_lane > 63) src_lane = clamp_self ? lane : (src_lane & 63);
    T r;
    memcpy(&r, &all[(unsigned)src_lane], sizeof(T));
    return r;
}
template <typename T> static inline T emu___shfl(T v, int src) { return emu_xchg(v, src & 63, false); }
template <typename T> static inline T emu___shfl_up(T v, unsigned d)
{
    int lane = (int)(g_emu.fibers[g_emu.cur].tid & 63u);
    return emu_xchg(v, lane - (int)d, true);
}
template <typename T> static inline T emu___shfl_down(T v, unsigned d)
{
    int lane = (int)(g_emu.fibers[g_emu.cur].tid & 63u);
    return emu_xchg(v, lane + (int)d, true);
}
template <typename T> static inline T emu___shfl_xor(T v, int m)
{
    int lane = (int)(g_emu.fibers[g_emu.cur].tid & 63u);
    return emu_xchg(v, lane ^ m, true);
}
static inline unsigned long long emu___ballot(int pred)
{
    uint64_t all[64];
    emu_wave_rendezvous(pred ? 1 : 0, all);
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64; i++)
        if (all[i]) m |= 1ull << i;
    return m;
}
static inline int emu___any(int p) { return emu___ballot(p) != 0; }
static inline int emu___all(int p) { return emu___ballot(!p) == 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __brev(unsigned v)
{
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline int emu_readfirstlane(int v) { return emu___shfl(v, 0); }

template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }

struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };

/* intra-wavefront barrier (on the GPU: LDS ordering inside one wave) */
static inline void emu_wave_barrier()
{
    uint64_t all[64];
    emu_wave_rendezvous(0, all);
}
extern int g_emu_line[1024];
static inline void emu_mark(int line) { g_emu_line[g_emu.fibers[g_emu.cur].tid] = line; }
#define __syncthreads() (emu_mark(__LINE__), emu___syncthreads())
#define __wave_barrier() (emu_mark(__LINE__), emu_wave_barrier())
#define __shfl(v, s) (emu_mark(__LINE__), emu___shfl((v), (s)))
#define __shfl_up(v, d) (emu_mark(__LINE__), emu___shfl_up((v), (d)))
#define __shfl_down(v, d) (emu_mark(__LINE__), emu___shfl_down((v), (d)))
#define __shfl_xor(v, m) (emu_mark(__LINE__), emu___shfl_xor((v), (m)))
#define __ballot(p) (emu_mark(__LINE__), emu___ballot((p)))
#define __any(p) (emu_mark(__LINE__), emu___any((p)))
#define __all(p) (emu_mark(__LINE__), emu___all((p)))
#define __builtin_amdgcn_readfirstlane(v) (emu_mark(__LINE__), emu_readfirstlane((v)))

void emu_launch(emu_dim3 grid, emu_dim3 block, void (*entry)(void *), void *arg);

/* launch helper: EMU_LAUNCH(kernel, grid, block, args...) */
#define EMU_LAUNCH(kernel, grid_x, block_x, ...)                                      \
    do {                                                                              \
        auto emu_thunk = [&]() { kernel(__VA_ARGS__); };                              \
        using emu_thunk_t = decltype(emu_thunk);                                      \
        emu_launch(emu_dim3{(unsigned)(grid_x), 1, 1}, emu_dim3{(unsigned)(block_x), 1, 1}, \
                   [](void *p) { (*(emu_thunk_t *)p)(); }, &emu_thunk);              \
    } while (0)

#endif
