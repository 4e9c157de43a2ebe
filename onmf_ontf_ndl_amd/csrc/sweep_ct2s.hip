// Probing tile sweep kernel with 2 tiles per wave and only 2 waves (64 columns) per workgroup, for
// small r and small batches: a batch of a few thousand columns then spreads over four times as many
// CUs, one wave per SIMD, instead of 8 waves taking turns on the SIMDs of a few CUs (see sweep_tile.h).
#include "sweep_tile.h"

bool sweep_tile_small_covers(int r) { return (r + 15) / 16 <= 8; }

int launch_sweep_tile_ct2_small(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                                float alpha, int exact_cd, int k_first, int k_count, double* stats,
                                int* n_groups, hipStream_t s, const ScreenDev* screen) {
    if (!screen) return onmf_fail(c, ONMF_ERR_ARG, "sweep: the small-workgroup kernel is the probing kernel only");
    switch ((r + 15) / 16) {
#define SMALL_CASE(NBV)                                                                                      \
    case NBV:                                                                                                \
        return run_tile<NBV, false, true, 2, false, 2>(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, \
                                                       stats, n_groups, *screen, s)
        SMALL_CASE(1);
        SMALL_CASE(2);
        SMALL_CASE(3);
        SMALL_CASE(4);
        SMALL_CASE(5);
        SMALL_CASE(6);
        SMALL_CASE(7);
        SMALL_CASE(8);
#undef SMALL_CASE
    }
    return onmf_fail(c, ONMF_ERR_ARG, "sweep: the small-workgroup kernel covers r <= 128 (r=%d)", r);
}
