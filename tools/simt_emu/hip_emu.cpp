/* DEVELOPMENT / TEST TOOLING -- scheduler of the lock-step SIMT emulator (see hip_emu.h). */
#include "hip_emu.h"
#include <pthread.h>

emu_state g_emu;
emu_dim3 g_emu_blockIdx, g_emu_blockDim, g_emu_gridDim;
int g_emu_line[1024];

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
    swapcontext(&f.ctx, &g_emu.sched);
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
    /* the scheduler runs on the caller's stack; ASan needs its bounds */
    static char probe;
    (void)probe;
    for (unsigned b = 0; b < grid.x; b++) {
        g_emu_blockIdx = emu_dim3{b, 0, 0};
        for (int g = 0; g < 17; g++) g_emu.arrived[g] = 0;
        g_emu.waiting = 0;
        for (unsigned t = 0; t < block.x; t++) {
            emu_fiber &f = g_emu.fibers[t];
            f.tid = t;
            f.done = 0;
            f.fake_stack = nullptr;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = f.stack_size;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())fiber_main, 0);
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
                {
                    /* record the scheduler's own stack so fibers can switch back to it */
                    pthread_attr_t attr;
                    void *sb = nullptr;
                    size_t ss = 0;
                    pthread_getattr_np(pthread_self(), &attr);
                    pthread_attr_getstack(&attr, &sb, &ss);
                    pthread_attr_destroy(&attr);
                    g_emu.sched.uc_stack.ss_sp = sb;
                    g_emu.sched.uc_stack.ss_size = ss;
                    __sanitizer_start_switch_fiber(&g_emu.sched_fake_stack, f.stack, f.stack_size);
                }
#endif
                swapcontext(&g_emu.sched, &f.ctx);
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
