// Helpers shared by the fused MLP kernels (SDF / colour / background NeRF): point sources,
// activation epilogues, tile bookkeeping.
#pragma once
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

template <class P>
struct Fast { static constexpr bool v = (P::id == NCW_PREC_BF16); };

// ---- point of lane ---------------------------------------------------------------------------
// mode 0: x[p];  1: o + d z[p];  2: o + d (z_i + dist_i / 2) with dist_i = z_{i+1} - z_i, last =
// sample_dist  (renderer.py:586-597 / :172-179)
NCW_DEV void load_point(const NcwPoints& s, int64_t p, float (&xs)[3], int64_t& ray) {
    if (s.mode == 0) {
        xs[0] = s.x[p * 3 + 0]; xs[1] = s.x[p * 3 + 1]; xs[2] = s.x[p * 3 + 2];
        ray = p;
        return;
    }
    const int64_t r = p / s.per_ray;
    ray = r;
    float zz = s.z[p];
    if (s.mode == 2) {
        const int i = (int)(p - r * s.per_ray);
        const float dist = (i + 1 < s.per_ray) ? s.z[p + 1] - zz : s.sample_dist[r];
        zz = zz + dist * 0.5f;
    }
    xs[0] = s.rays_o[r * 3 + 0] + s.rays_d[r * 3 + 0] * zz;
    xs[1] = s.rays_o[r * 3 + 1] + s.rays_d[r * 3 + 1] * zz;
    xs[2] = s.rays_o[r * 3 + 2] + s.rays_d[r * 3 + 2] * zz;
}

// wave's tile index and the (clamped) point of this lane; returns false if the whole tile is empty
NCW_DEV bool tile_setup(int64_t n, int64_t& tile, int64_t& p, bool& valid, int lane) {
    tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= n) return false;
    p = tile * 32 + (lane & 31);
    valid = p < n;
    if (!valid) p = n - 1;
    return true;
}

// act = Softplus100(acc); optionally stash y (next layer's input) and s = Softplus'
template <class P, int RB>
NCW_DEV void softplus_epilogue(Act<P, RB>& act, CVec<RB>& acc, typename P::selem* st_h, typename P::selem* st_s,
                               size_t tile, int lane) {
    if (st_s) {
        CVec<RB> sv;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y, s;
                softplus100<Fast<P>::v>(acc.v[rb][r], y, s);
                acc.v[rb][r] = y;
                sv.v[rb][r] = s;
            }
        stash_store<RB>(st_s, tile, sv, lane);
    } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y, s;
                softplus100<Fast<P>::v>(acc.v[rb][r], y, s);
                acc.v[rb][r] = y;
            }
    }
    if (st_h) stash_store<RB>(st_h, tile, acc, lane);
    to_act(act, acc);
}

// act = relu(acc), optional stash of the post-activation
template <class P, int RB>
NCW_DEV void relu_epilogue(Act<P, RB>& act, CVec<RB>& acc, typename P::selem* st, size_t tile, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.v[rb][r] = fmaxf(acc.v[rb][r], 0.f);
    if (st) stash_store<RB>(st, tile, acc, lane);
    to_act(act, acc);
}

// zbar = ubar * [y > 0]  where y is the stashed post-relu activation
template <class P, int RB>
NCW_DEV void relu_backward(CVec<RB>& u, const typename P::selem* st_y, size_t tile, int lane) {
    CVec<RB> y;
    stash_load<RB>(y, st_y, tile, lane);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) u.v[rb][r] = y.v[rb][r] > 0.f ? u.v[rb][r] : 0.f;
}

template <int RB>
NCW_DEV void cvec_copy(CVec<RB>& d, const CVec<RB>& s) {
#pragma unroll
    for (int i = 0; i < RB; ++i) d.v[i] = s.v[i];
}

#define NCW_LAUNCH_TILES(kernel, n, st, ...)                                                         \
    do {                                                                                            \
        const int64_t tiles__ = ((n) + 31) / 32;                                                    \
        const int64_t blocks__ = (tiles__ + 3) / 4;                                                 \
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks__), dim3(256), 0, st, __VA_ARGS__);        \
        NCW_CHECK_LAUNCH();                                                                         \
    } while (0)

// ---- auxiliary input blocks shared by the colour net and the background NeRF ------------------------
// AUX1 (3 blocks, 96): [gamma_4(view dir) (27) | appearance embedding a (n_a <= 69) | 0...]
//   (models/neuconw.py:131-139, models/nerf.py:159-160,173)
template <bool FAST>
NCW_DEV void build_aux1(CVec<3>& aux, const float (&dir)[3], const float* __restrict__ a_row, int n_a, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * rb + ncw_feat_of(r, 0) + 4 * h;
            float v;
            if (32 * rb + ncw_feat_of(r, 0) + 4 < 27) {  // both halves inside the direction encoding
                v = freq_feature<3, 4, FAST>(dir, f);
            } else if (32 * rb + ncw_feat_of(r, 0) >= 27) {  // both halves in the embedding / padding
                v = (f - 27 < n_a) ? a_row[f - 27] : 0.f;
            } else {
                v = (f < 27) ? freq_feature<3, 4, FAST>(dir, f) : ((f - 27 < n_a) ? a_row[f - 27] : 0.f);
            }
            aux.v[rb][r] = v;
        }
}

// d_a[ray][j] += sum over the wave's points of the AUX1-part adjoint (features 27 .. 27+n_a)
NCW_DEV void accumulate_d_a(const CVec<3>& q, float* __restrict__ d_a, int64_t ray, int n_a, bool valid, int lane) {
    const int h = lane >> 5;
    const int64_t r0 = __shfl(ray, 0, 64);
    const bool uniform = __all((ray == r0) ? 1 : 0);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * rb + ncw_feat_of(r, 0) + 4 < 27) continue;
            const int f = 32 * rb + ncw_feat_of(r, 0) + 4 * h;
            const int j = f - 27;
            float v = valid ? q.v[rb][r] : 0.f;
            if (uniform) {
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
                if ((lane & 31) == 0 && j >= 0 && j < n_a) atomicAdd(d_a + ray * n_a + j, v);
            } else {
                if (valid && j >= 0 && j < n_a) atomicAdd(d_a + ray * n_a + j, v);
            }
        }
}
