// Tile sweep kernel, 1 column tile of 16 per wave (see sweep_tile.h).
#include "sweep_tile.h"

int launch_sweep_tile_ct1(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                          float* Hout, float alpha, float damp, int exact_cd, double* stats,
                          int* n_groups, hipStream_t s, const ScreenDev* screen) {
    return launch_sweep_tile<1>(c, r, b, Gm, Cm, Hin, Hout, alpha, damp, exact_cd, stats, n_groups, s,
                                screen);
}
