// PROBE-ONLY timing hooks for the stash helpers of csrc/ncw_common.h / ncw_mlp.h (DESIGN.md "what the stash loads cost").
// Included only when NCW_PROBE_BUILD is defined, which neuralrecon-w_amd/build.py does only for libraries built beside the
// product under NCW_BUILD_TAG=<tag> (libneuconw_hip_<tag>.so).  A library built with these returns GARBAGE gradients:
//   -DNCW_EXP_NOSTORE   stash stores are skipped (readers see uninitialised memory)
//   -DNCW_EXP_NOLOAD    stash loads return constants
#pragma once
#ifdef NCW_EXP_NOSTORE
#define NCW_EXP_STORE_HOOK() return
#else
#define NCW_EXP_STORE_HOOK()
#endif
#ifdef NCW_EXP_NOLOAD
template <int RB>
NCW_DEV void ncw_exp_fill(CVec<RB>& c) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int q = 0; q < 16; ++q) c.v[rb][q] = 0.37f + 0.01f * q;
}
NCW_DEV void ncw_exp_fill(f32x16& v) {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.37f + 0.01f * q;
}
#define NCW_EXP_LOAD_HOOK(x) \
    do {                     \
        ncw_exp_fill(x);     \
        return;              \
    } while (0)
#else
#define NCW_EXP_LOAD_HOOK(x)
#endif
