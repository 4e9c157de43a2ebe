// Probing tile sweep kernel with 64 columns per workgroup, for small r and small batches: a batch of a few
// thousand columns then spreads over four times as many CUs instead of 8 waves taking turns on the SIMDs of
// a few CUs (see sweep_tile.h).  One tile per wave, four waves per workgroup - one wave per SIMD, every SIMD
// of the CU at work; with two tiles per wave (two waves per workgroup) half of the SIMDs idle and a wave has
// twice the arithmetic per sweep (measured: d=1200 r=100 b=8192 0.365 -> 0.311 ms per step, d=1024 r=64
// b=8192 0.26 -> 0.24).  At one wave per SIMD the LDS traffic that makes two tiles per wave pay at full
// occupancy is no concern.
#include "sweep_tile.h"

#ifndef SMALL_CT
#define SMALL_CT 1
#endif

bool sweep_tile_small_covers(int r) { return (r + 15) / 16 <= 16; }

int launch_sweep_tile_ct2_small(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                                float alpha, int exact_cd, int k_first, int k_count, double* stats,
                                int* n_groups, hipStream_t s, const ScreenDev* screen) {
    if (!screen) return onmf_fail(c, ONMF_ERR_ARG, "sweep: the small-workgroup kernel is the probing kernel only");
    switch ((r + 15) / 16) {
#define SMALL_CASE(NBV)                                                                                      \
    case NBV:                                                                                                \
        return run_tile<NBV, (NBV >= 12), true, SMALL_CT, false, 128 / (32 * SMALL_CT)>(                     \
            c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, stats, n_groups, *screen, s)
        SMALL_CASE(1);
        SMALL_CASE(2);
        SMALL_CASE(3);
        SMALL_CASE(4);
        SMALL_CASE(5);
        SMALL_CASE(6);
        SMALL_CASE(7);
        SMALL_CASE(8);
        SMALL_CASE(9);
        SMALL_CASE(10);
        SMALL_CASE(11);
        SMALL_CASE(12);
        SMALL_CASE(13);
        SMALL_CASE(14);
        SMALL_CASE(15);
        SMALL_CASE(16);
#undef SMALL_CASE
    }
    return onmf_fail(c, ONMF_ERR_ARG, "sweep: the small-workgroup kernel covers r <= 256 (r=%d)", r);
}
