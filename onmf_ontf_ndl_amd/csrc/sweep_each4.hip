// Per-column stopping instantiations of the tile sweep kernel with 4 waves (64 columns) per workgroup.
#include "sweep_tile.h"

int launch_sweep_tile_each4(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* H, float alpha,
                           int exact_cd, int sub_iter, EachStop es, hipStream_t s) {
    const int nb = (r + 15) / 16;
    float* hb[3] = {H, H, H};
    ScreenDev none = {};
    int groups = 0;
#define EACH_CASE(NBV, HALFV)                                                                              \
    case NBV:                                                                                              \
        return run_tile<NBV, HALFV, false, 1, true, 4>(c, r, b, Gm, Cm, hb, alpha, exact_cd, 0, sub_iter,     \
                                                    nullptr, &groups, none, s, es)
    switch (nb) {
        EACH_CASE(1, false);
        EACH_CASE(2, false);
        EACH_CASE(3, false);
        EACH_CASE(4, false);
        EACH_CASE(5, false);
        EACH_CASE(6, false);
        EACH_CASE(7, false);
        EACH_CASE(8, false);
        EACH_CASE(9, false);
        EACH_CASE(10, false);
        EACH_CASE(11, false);
        EACH_CASE(12, true);
        EACH_CASE(13, true);
        EACH_CASE(14, true);
        EACH_CASE(15, true);
        EACH_CASE(16, true);
    }
#undef EACH_CASE
    return onmf_fail(c, ONMF_ERR_ARG, "sweep: unsupported r=%d", r);
}
