// Tile sweep kernel, 1 column tile of 16 per wave (see sweep_tile.h).
#include "sweep_tile.h"

int launch_sweep_tile_ct1(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                          float alpha, int exact_cd, int k_first, int k_count, double* stats,
                          int* n_groups, hipStream_t s, const ScreenDev* screen) {
    return launch_sweep_tile<1>(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, stats, n_groups, s,
                                screen);
}
