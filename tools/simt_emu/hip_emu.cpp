/* DEVELOPMENT / TEST TOOLING -- scheduler of the lock-step SIMT emulator (see hip_emu.h). */
#include "hip_emu.h"
#include <pthread.h>

emu_state g_emu;
emu_dim3 g_emu_blockIdx, g_emu_blockDim, g_emu_gridDim;
int g_emu_line[1024];

#if defined(__x86_64__)
asm(R"(
    .text
    .globl emu_ctx_switch
    .type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_ctx_switch, .-emu_ctx_switch
)");
#endif

static void fiber_main()
{
#if EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, nullptr, nullptr);
#endif
    g_emu.entry(g_emu.entry_arg);
    emu_fiber &f = g_emu.fibers[g_emu.cur];
    f.done = 1;
#if EMU_ASAN
    /* nullptr: this fiber's fake stack is destroyed, it never resumes */
    __sanitizer_start_switch_fiber(nullptr, g_emu.sched.uc_stack.ss_sp, g_emu.sched.uc_stack.ss_size);
#endif
    emu_to_scheduler(f);
}

void emu_launch(emu_dim3 grid, emu_dim3 block, void (*entry)(void *), void *arg)
{
    const size_t STK = 512 * 1024;
    g_emu_gridDim = grid;
    g_emu_blockDim = block;
    g_emu.entry = entry;
    g_emu.entry_arg = arg;
    g_emu.nthreads = block.x;
    g_emu.fibers.resize(block.x);
    for (unsigned t = 0; t < block.x; t++) {
        g_emu.fibers[t].stack = (char *)malloc(STK);
        g_emu.fibers[t].stack_size = STK;
    }
#if EMU_ASAN
    /* the scheduler runs on the caller's stack; ASan needs its bounds (asked for ONCE per launch:
     * for the main thread glibc answers this by parsing /proc/self/maps) */
    {
        pthread_attr_t attr;
        void *sb = nullptr;
        size_t ss = 0;
        pthread_getattr_np(pthread_self(), &attr);
        pthread_attr_getstack(&attr, &sb, &ss);
        pthread_attr_destroy(&attr);
        g_emu.sched.uc_stack.ss_sp = sb;
        g_emu.sched.uc_stack.ss_size = ss;
    }
#endif
    for (unsigned b = 0; b < grid.x; b++) {
        g_emu_blockIdx = emu_dim3{b, 0, 0};
        for (int g = 0; g < 17; g++) g_emu.arrived[g] = 0;
        g_emu.waiting = 0;
        for (unsigned t = 0; t < block.x; t++) {
            emu_fiber &f = g_emu.fibers[t];
            f.tid = t;
            f.done = 0;
            f.fake_stack = nullptr;
#if defined(__x86_64__)
            {
                /* a frame emu_ctx_switch can "return" into: [MXCSR | x87 CW] r15 r14 r13 r12 rbx rbp,
                 * then fiber_main as the return address (16-byte aligned slot) and a null return
                 * address above it, as if fiber_main had been called */
                uintptr_t top = ((uintptr_t)f.stack + f.stack_size) & ~(uintptr_t)15;
                uint64_t *sp = (uint64_t *)top;
                *--sp = 0;
                *--sp = (uint64_t)(uintptr_t)&fiber_main;
                for (int r = 0; r < 6; r++) *--sp = 0;
                *--sp = 0x037F00001F80ull; /* MXCSR 0x1F80 at +0, x87 control word 0x037F at +4 */
                f.sp = sp;
            }
#else
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = f.stack_size;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())fiber_main, 0);
#endif
        }
        unsigned live = block.x;
        unsigned spins = 0;
        while (live) {
            unsigned progressed = 0;
            unsigned long long prog0 = g_emu.progress;
            unsigned live0 = live;
            for (unsigned t = 0; t < block.x; t++) {
                emu_fiber &f = g_emu.fibers[t];
                if (f.done) continue;
                g_emu.cur = t;
#if EMU_ASAN
                __sanitizer_start_switch_fiber(&g_emu.sched_fake_stack, f.stack, f.stack_size);
#endif
                emu_to_fiber(f);
#if EMU_ASAN
                __sanitizer_finish_switch_fiber(g_emu.sched_fake_stack, nullptr, nullptr);
#endif
                progressed++;
                if (f.done) live--;
            }
            if (!progressed) break;
            if (live && live == live0 && prog0 == g_emu.progress && g_emu.waiting == live) {
                fprintf(stderr, "emu: workgroup %u deadlocked: %u threads exited, %u wait at a collective\n",
                        b, block.x - live, live);
                for (unsigned t = 0; t < block.x; t++)
                    fprintf(stderr, "  thread %u: %s, last collective at line %d\n", t,
                            g_emu.fibers[t].done ? "exited" : "waiting", g_emu_line[t]);
                abort();
            }
            if (++spins > 2000000000u) {
                fprintf(stderr, "emu: workgroup %u appears deadlocked (divergent collective?)\n", b);
                abort();
            }
        }
    }
    for (unsigned t = 0; t < block.x; t++) free(g_emu.fibers[t].stack);
}
