/* DEVELOPMENT / TEST TOOLING: the product's HIP kernels compiled for the CPU lock-step
 * emulator (hip_emu.h).  Exposes the same batch entry points as libdebigulator_hip.so
 * (include/debig_hip.h) with an emu_ prefix and HOST pointers. */
#include "hip_emu.h"
#include "../../include/debig_hip.h"
#include "../../debigulator_amd/csrc/inflate_kernel.inc"
#include "../../debigulator_amd/csrc/png_kernel.inc"
#include "../../debigulator_amd/csrc/checksum_kernel.inc"

extern "C" int emu_inflate_batch(const void *in, void *out, const debig_stream *streams,
                                 debig_result *results, uint32_t n, uint32_t grid)
{
    if (grid == 0 || grid > n) grid = n;
    static CodeTabs *ft = nullptr;
    if (!ft) {
        ft = (CodeTabs *)calloc(1, sizeof(CodeTabs));
        EMU_LAUNCH(debig_fixed_tables_kernel, 1, 64, ft);
    }
    EMU_LAUNCH(debig_inflate_kernel, grid, 64, (const uint8_t *)in, (uint8_t *)out, streams, results, n, ft);
    return 0;
}

extern "C" int emu_png_defilter_batch(const void *streams_arena, void *rgba_arena, const debig_png_image *images,
                                      debig_png_result *results, uint32_t n)
{
    EMU_LAUNCH(debig_png_defilter_kernel, n, 64, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images,
               results, n);
    EMU_LAUNCH(debig_png_p3_kernel, n, 64, (const uint8_t *)streams_arena, (uint8_t *)rgba_arena, images,
               results, n);
    return 0;
}

extern "C" int emu_checksum_batch(const void *arena, const debig_span *spans, uint32_t *out, uint32_t n, uint32_t kind)
{
    static CkTables *t = nullptr;
    if (!t) {
        t = (CkTables *)calloc(1, sizeof(CkTables));
        EMU_LAUNCH(debig_checksum_tables_kernel, 1, CK_THREADS, t);
    }
    EMU_LAUNCH(debig_checksum_kernel, n, CK_THREADS, (const uint8_t *)arena, spans, out, n, kind, t);
    return 0;
}
