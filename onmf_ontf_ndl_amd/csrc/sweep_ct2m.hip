// Probing tile sweep kernel with 128 columns per workgroup (two tiles per wave, four waves: one per SIMD),
// for batches too large for one 64-column workgroup per CU and too small to give every CU a 256-column
// one: between 64 and 128 columns per CU the default geometry would leave half of the chip idle with two
// waves on each SIMD of the other half (see sweep_tile.h).
#include "sweep_tile.h"

int launch_sweep_tile_ct2_medium(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                                 float alpha, int exact_cd, int k_first, int k_count, double* stats,
                                 int* n_groups, hipStream_t s, const ScreenDev* screen) {
    if (!screen) return onmf_fail(c, ONMF_ERR_ARG, "sweep: the 128-column kernel is the probing kernel only");
    switch ((r + 15) / 16) {
#define MEDIUM_CASE(NBV)                                                                                      \
    case NBV:                                                                                                 \
        return run_tile<NBV, (NBV >= 12), true, 2, false, 4>(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first,  \
                                                             k_count, stats, n_groups, *screen, s)
        MEDIUM_CASE(1);
        MEDIUM_CASE(2);
        MEDIUM_CASE(3);
        MEDIUM_CASE(4);
        MEDIUM_CASE(5);
        MEDIUM_CASE(6);
        MEDIUM_CASE(7);
        MEDIUM_CASE(8);
        MEDIUM_CASE(9);
        MEDIUM_CASE(10);
        MEDIUM_CASE(11);
        MEDIUM_CASE(12);
        MEDIUM_CASE(13);
        MEDIUM_CASE(14);
        MEDIUM_CASE(15);
        MEDIUM_CASE(16);
#undef MEDIUM_CASE
    }
    return onmf_fail(c, ONMF_ERR_ARG, "sweep: the tile kernel covers r <= 256 (r=%d)", r);
}
