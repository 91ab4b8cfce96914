// SDF network at W = 512 in the EXACT-fp32 parity mode: sdf_infer, sdf_fwd, sdf_bwd in the streamed-weights structure of
// ncw_sdf16.hip (weights from L2 through a register ring, the workgroup's activations in ONE LDS buffer rewritten in place) --
// the arithmetic and the stashes of the generic kernels of ncw_sdf.hip (models/neuconw.py:263-296 forward, the analytic adjoint
// pass for the normals, the second-order backward), which at RB = 16 keep 256 accumulator + 256 activation registers per wave
// and spill 1-2 KB per lane (the shipped yaml's step in fp32: 58.6 ms at 2048 rays).
//
// fp32 MFMA is `v_mfma_f32_32x32x2_f32` (64 cycles per SIMD, 157 TFLOP/s chip-wide): a "k-unit" here is ONE k-step = one
// register r of a C-layout block (feature 32 rb + (r&3) + 8 (r>>2) + 4 h on half h), 256 units per 512-wide layer; the A
// fragment of a unit is one float per lane (the packed fp32 matrix is [unit][out-block][64 lanes]), the B fragment one float
// per lane read from LDS ([tile][256 units][64 lanes] floats = 64 KiB per tile, T = 2 tiles per workgroup).  Per layer and
// workgroup: 2 blocks x 2 tiles x 256 MFMAs per wave = 65.5 k cycles per SIMD pair against 1 MiB of weights (16 k cycles of
// the CU's L2 port) and 128 KiB of LDS reads: MFMA-bound.  Wave w owns output blocks w and w + 8.
#include "ncw_mlp.h"

#ifndef NCW_HALF_F16  // fp32 kernels: compiled once, with the default (bf16) build of the library

namespace {

constexpr int F_WAVES = 8, F_T = 2;
constexpr int F_KU = 256;  // k-units (single k-steps) of a 512-wide layer
constexpr int F_GU = 20;   // k-units of gamma (39 features: block 0 = 16 units, block 1 = the 4 units holding 32..38)
constexpr int F_GS = 24;   // gbuf units per tile: gamma / qbar_0 (20) + the d_sdf unit (index F_GU)
constexpr int F_D = 16;    // weight prefetch distance (k-units)

typedef __attribute__((address_space(3))) float f_lf;
typedef const __attribute__((address_space(1))) float* f_gp;

struct FW { float f[F_D][2]; };  // register ring: units q .. q + D - 1 of the wave's two output blocks

NCW_DEV float f_ld(const void* w, int rb_stride, int ob, int u, int lane) {
    return ((f_gp)w)[((size_t)u * rb_stride + ob) * 64 + lane];
}
NCW_DEV f32x16 f_mfma(float a, float b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

NCW_DEV void f_prefetch(FW& r, const void* w, int rb_stride, int wave, int lane) {
#pragma unroll
    for (int d = 0; d < F_D; ++d) {
        r.f[d][0] = f_ld(w, rb_stride, wave, d, lane);
        r.f[d][1] = f_ld(w, rb_stride, wave + 8, d, lane);
    }
}

// acc[j][t] += W[block wave + 8 j][.] . in[tile t][.] over the 256 k-units of a hidden layer
NCW_DEV void f_mma(f32x16 (&acc)[2][F_T], FW& r, const void* w, int rb_stride, int wave, const f_lf* in, int lane) {
    for (int q0 = 0; q0 < F_KU; q0 += F_D) {  // ring slot k holds unit q0 + k: compile-time register indices
#pragma unroll
        for (int k = 0; k < F_D; ++k) {
            const int q = q0 + k;
            const float a0 = r.f[k][0], a1 = r.f[k][1];
            if (q + F_D < F_KU) {
                r.f[k][0] = f_ld(w, rb_stride, wave, q + F_D, lane);
                r.f[k][1] = f_ld(w, rb_stride, wave + 8, q + F_D, lane);
            }
#pragma unroll
            for (int t = 0; t < F_T; ++t) {
                const float b = in[(t * F_KU + q) * 64 + lane];
                acc[0][t] = f_mfma(a0, b, acc[0][t]);
                acc[1][t] = f_mfma(a1, b, acc[1][t]);
            }
        }
    }
}

// the F_GU gamma k-units (units first .. first + F_GU - 1 of the matrix) against gbuf
NCW_DEV void f_mma_gamma(f32x16 (&acc)[2][F_T], const void* w, int rb_stride, int first, int wave, const f_lf* gbuf, int lane) {
#pragma unroll 4
    for (int q = 0; q < F_GU; ++q) {
        const float g0 = f_ld(w, rb_stride, wave, first + q, lane), g1 = f_ld(w, rb_stride, wave + 8, first + q, lane);
#pragma unroll
        for (int t = 0; t < F_T; ++t) {
            acc[0][t] = f_mfma(g0, gbuf[(t * F_GS + q) * 64 + lane], acc[0][t]);
            acc[1][t] = f_mfma(g1, gbuf[(t * F_GS + q) * 64 + lane], acc[1][t]);
        }
    }
}

// one output block of one tile (gamma rows of a transposed matrix, the sdf row): acc += W[block ob][.] . in[tile t][.]
NCW_DEV void f_mma1(f32x16& acc, const void* w, int rb_stride, int ob, const f_lf* in, int t, int lane) {
#pragma unroll 8
    for (int q = 0; q < F_KU; ++q) acc = f_mfma(f_ld(w, rb_stride, ob, q, lane), in[(t * F_KU + q) * 64 + lane], acc);
}

// C-layout block ob of tile t -> its 16 k-units of the next layer's input
NCW_DEV void f_store_units(f_lf* buf, int t, int ob, const f32x16& v, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[(t * F_KU + ob * 16 + r) * 64 + lane] = v[r];
}

NCW_DEV f32x16 f_bias(const float* bp, int ob, int lane) {
    CVec<1> b1;
    load_bias(b1, bp + ob * 32, lane);
    return b1.v[0];
}
NCW_DEV f32x16 f_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
NCW_DEV void f_fill(f32x16 (&acc)[2][F_T], const f32x16& b0, const f32x16& b1) {
#pragma unroll
    for (int t = 0; t < F_T; ++t) { acc[0][t] = b0; acc[1][t] = b1; }
}
NCW_DEV f32x16 f_softplus(const f32x16& z) {
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) { float yy, s; softplus100<false>(z[r], yy, s); y[r] = yy; }
    return y;
}
// phi'(z) = 1 - exp(-100 h) from the stashed post-activation h (ncw_sdf.hip load_sprime_block, fp32 form)
NCW_DEV f32x16 f_sprime(const float* __restrict__ st_h, size_t tile, int ob, int lane) {
    f32x16 sv;
    stash_load_block(sv, st_h, tile, 16, ob, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = 1.f - expf(-100.f * sv[r]);
    return sv;
}
// the first F_GU units of a 2-block C-layout vector (gamma, qbar_0) -> gbuf units 0 .. F_GU - 1 of tile tw
NCW_DEV void f_store_gamma(f_lf* gbuf, int tw, const CVec<2>& g, int lane) {
#pragma unroll
    for (int q = 0; q < F_GU; ++q) gbuf[(tw * F_GS + q) * 64 + lane] = g.v[q >> 4][q & 15];
}

#define F_LDS_DECL()                                                                         \
    __shared__ __attribute__((aligned(16))) char lds[F_T * F_KU * 256 + F_T * F_GS * 256];   \
    f_lf* const abuf = (f_lf*)(ncw_lchar*)lds;                                               \
    f_lf* const gbuf = abuf + F_T * F_KU * 64;                                               \
    const int lane = ncw_lane();                                                             \
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));               \
    const int L = net.n_layers;                                                              \
    constexpr int T = F_T;                                                                   \
    const int64_t tile0 = (int64_t)blockIdx.x * T

// forward chain shared by sdf_infer and sdf_fwd: gamma -> layers 0 .. L-2; leaves h_{L-1} in abuf and the first units of
// `w_after` (the sdf row is read directly; w_feat for sdf_fwd) in the ring.  STASH: gamma, h_1 .. h_{L-1}.
// STASH 2: gamma + h_l (training); 1: h_l only (forward-only render: the adjoint sweep's scratch); 0: nothing (sdf_infer)
template <int STASH>
NCW_DEV void f_forward_chain(const NcwSdfNet& net, const NcwPoints& src, int64_t n, int64_t tile0, f_lf* abuf, f_lf* gbuf, FW& r,
                             f32x16 (&acc)[2][F_T], const void* w_after, int lane, int wave, const NcwSdfStash& st) {
    constexpr int T = F_T;
    const int L = net.n_layers;
    if (wave < T) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, false>(gam, xs, lane);
        if (STASH == 2) stash_store<2>((float*)st.gamma, (size_t)(tile0 + wave), gam, lane);
        f_store_gamma(gbuf, wave, gam, lane);
    }
    {   // layer 0: K = 39 (the F_GU gamma units)
        f_prefetch(r, L - 1 > 1 ? net.w[1] : w_after, 16, wave, lane);
        f_fill(acc, f_bias(net.b[0], wave, lane), f_bias(net.b[0], wave + 8, lane));
        ncw_lds_barrier();
        f_mma_gamma(acc, net.w[0], 16, 0, wave, gbuf, lane);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 y = f_softplus(acc[j][t]);
                if (STASH >= 1) stash_store_block_keep((float*)st.h[1], (size_t)(tile0 + t), 16, wave + 8 * j, y, lane);
                f_store_units(abuf, t, wave + 8 * j, y, lane);
            }
    }
    for (int l = 1; l < L - 1; ++l) {  // r = first units of w[l]
        const f32x16 b0 = f_bias(net.b[l], wave, lane), b1 = f_bias(net.b[l], wave + 8, lane);
        ncw_lds_barrier();  // layer l-1 outputs of all waves are in abuf
        f_fill(acc, b0, b1);
        f_mma(acc, r, net.w[l], 16, wave, abuf, lane);
        f_prefetch(r, l + 1 < L - 1 ? net.w[l + 1] : w_after, 16, wave, lane);
        if (l == net.skip_layer) f_mma_gamma(acc, net.w[l], 16, F_KU, wave, gbuf, lane);  // units 256.. = the gamma columns
        ncw_lds_barrier();  // every wave has read abuf: overwrite in place
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 y = f_softplus(acc[j][t]);
                if (STASH >= 1) stash_store_block_keep((float*)st.h[l + 1], (size_t)(tile0 + t), 16, wave + 8 * j, y, lane);
                f_store_units(abuf, t, wave + 8 * j, y, lane);
            }
    }
}

__global__ __launch_bounds__(64 * F_WAVES) void sdf_infer16f_kernel(NcwSdfNet net, NcwPoints src, int64_t n, float* __restrict__ sdf) {
    F_LDS_DECL();
    FW r;
    f32x16 acc[2][T];
    NcwSdfStash none = {};
    f_forward_chain<0>(net, src, n, tile0, abuf, gbuf, r, acc, net.w[1], lane, wave, none);  // (w_after unused: any valid matrix)
    ncw_lds_barrier();
    if (wave < T) {  // sdf row
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        f_mma1(o.v[0], net.w[L - 1], 1, 0, abuf, wave, lane);
        const int64_t p = (tile0 + wave) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / net.scale;
    }
}

// ------------------------------------------------------------------------------------------------
// sdf_fwd: forward chain with the activation stash, feature rows, sdf row, then the analytic adjoint pass
// t_{l-1} = (W_l^T t_l) * phi'(z_{l-1}) with the t_l stash and grad = J_gamma^T g_gamma (sdf_fwd_kernel, ncw_sdf.hip).
// The gamma output blocks (16, 17) of the transposed skip layer and the two blocks of W_0^T are 2 blocks x T tiles jobs:
// wave w < 2 T takes block (w & 1) of tile (w >> 1) and keeps that g_gamma block to the end.
// ------------------------------------------------------------------------------------------------
template <bool TRAIN>  // false: forward-only render -- the same outputs bit for bit; no gamma / t_l stash
__global__ __launch_bounds__(64 * F_WAVES) void sdf_fwd16f_kernel(NcwSdfNet net, NcwPoints src, int64_t n, float* __restrict__ sdf,
                                                                 float* __restrict__ grad, NcwSdfStash st) {
    typedef float SE;
    F_LDS_DECL();
    const int jb = wave & 1, jt = wave >> 1;
    const bool gjob = wave < 2 * T;
    FW r;
    f32x16 acc[2][T];
    f_forward_chain<(TRAIN ? 2 : 1)>(net, src, n, tile0, abuf, gbuf, r, acc, net.w_feat, lane, wave, st);
    // ---- feature rows (r = first units of w_feat) and sdf row; then the adjoint's first vector ----------------------
    {
        const float wt1_0 = f_ld(net.wt[L - 1], 16, wave, 0, lane), wt1_1 = f_ld(net.wt[L - 1], 16, wave + 8, 0, lane);
        const f32x16 b0 = f_bias(net.b_feat, wave, lane), b1 = f_bias(net.b_feat, wave + 8, lane);
        ncw_lds_barrier();  // h_{L-1} complete in abuf
        f_fill(acc, b0, b1);
        f_mma(acc, r, net.w_feat, 16, wave, abuf, lane);
        if (L - 2 >= 1) f_prefetch(r, net.wt[L - 2], (L - 2 == net.skip_layer) ? 18 : 16, wave, lane);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) stash_store_block((SE*)st.feat, (size_t)(tile0 + t), 16, wave + 8 * j, acc[j][t], lane);
        if (wave < T) {
            CVec<1> o;
            load_bias(o, net.b[L - 1], lane);
            f_mma1(o.v[0], net.w[L - 1], 1, 0, abuf, wave, lane);
            const int64_t p = (tile0 + wave) * 32 + (lane & 31);
            if (p < n && lane < 32) sdf[p] = o.v[0][0] / net.scale;
        }
        // a_{L-2} = W_{L-1}^T e_0 (the same for every point): k-unit 0 carries feature 0 on half 0 (feature 4 on half 1)
        const float e0 = lane < 32 ? 1.f : 0.f;
        const f32x16 a0 = f_mfma(wt1_0, e0, f_zero()), a1 = f_mfma(wt1_1, e0, f_zero());
        ncw_lds_barrier();  // the feature rows and the sdf row have read h_{L-1}: overwrite abuf with t_{L-2}
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 sv = f_sprime((const SE*)st.h[L - 1], (size_t)(tile0 + t), wave + 8 * j, lane);
                const f32x16& aa = j ? a1 : a0;
#pragma unroll
                for (int q = 0; q < 16; ++q) sv[q] *= aa[q];
                if (TRAIN) stash_store_block((SE*)st.t[L - 2], (size_t)(tile0 + t), 16, wave + 8 * j, sv, lane);
                f_store_units(abuf, t, wave + 8 * j, sv, lane);
            }
    }
    // ---- adjoint layers l = L-2 .. 1: t_{l-1} = (W_l^T t_l) * phi'(z_{l-1});  r = first units of wt[l] -------------
    f32x16 gg = f_zero();
    for (int l = L - 2; l >= 1; --l) {
        const bool skip = (l == net.skip_layer);
        ncw_lds_barrier();  // t_l complete in abuf
        f_fill(acc, f_zero(), f_zero());
        f_mma(acc, r, net.wt[l], skip ? 18 : 16, wave, abuf, lane);
        if (l - 1 >= 1) f_prefetch(r, net.wt[l - 1], (l - 1 == net.skip_layer) ? 18 : 16, wave, lane);
        if (skip && gjob) f_mma1(gg, net.wt[l], 18, 16 + jb, abuf, jt, lane);  // gamma rows of the transposed skip layer
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 sv = f_sprime((const SE*)st.h[l], (size_t)(tile0 + t), wave + 8 * j, lane);
#pragma unroll
                for (int q = 0; q < 16; ++q) sv[q] *= acc[j][t][q];
                if (TRAIN) stash_store_block((SE*)st.t[l - 1], (size_t)(tile0 + t), 16, wave + 8 * j, sv, lane);
                f_store_units(abuf, t, wave + 8 * j, sv, lane);
            }
    }
    // ---- adjoint layer 0: g_gamma += W_0^T t_0 (2 out-blocks), then grad = J_gamma^T g_gamma ------------------------
    ncw_lds_barrier();
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int64_t p = (tile0 + jt) * 32 + (lane & 31), ray;
    const bool valid = gjob && p < n;
    if (gjob) {
        f_mma1(gg, net.wt[0], 2, jb, abuf, jt, lane);
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        const int h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int f0 = 32 * jb + ncw_feat_of(q, 0);
            if (f0 >= 39) continue;  // (block 1 holds features 32..38 only)
            int comp;
            const float dv = freq_feature_deriv<3, 6, false>(xs, f0 + 4 * h, comp);
            const float c = gg[q] * dv;
            nx += comp == 0 ? c : 0.f;
            ny += comp == 1 ? c : 0.f;
            nz += comp == 2 ? c : 0.f;
        }
        nx = half_pair_sum(nx); ny = half_pair_sum(ny); nz = half_pair_sum(nz);
    }
    // combine the two blocks of a tile (waves 2 jt and 2 jt + 1) through LDS (the gamma region is free now)
    f_lf* part = gbuf;
    if (gjob && jb == 1 && lane < 32) {
        part[(jt * 32 + lane) * 3 + 0] = nx; part[(jt * 32 + lane) * 3 + 1] = ny; part[(jt * 32 + lane) * 3 + 2] = nz;
    }
    ncw_lds_barrier();
    if (gjob && jb == 0 && lane < 32 && valid) {
        grad[p * 3 + 0] = nx + part[(jt * 32 + lane) * 3 + 0];
        grad[p * 3 + 1] = ny + part[(jt * 32 + lane) * 3 + 1];
        grad[p * 3 + 2] = nz + part[(jt * 32 + lane) * 3 + 2];
    }
}

// ------------------------------------------------------------------------------------------------
// sdf_bwd (second order): (1) backward of the adjoint pass l = 0 .. L-2 (tbar = W qbar, abar = tbar phi',
// zbar2 = tbar 100 t (1 - phi')), (2) backward of the forward pass l = L-1 .. 0 (ubar = W^T zbar,
// zbar = ubar phi' + zbar2) -- the arithmetic and the stash of sdf_bwd_kernel (ncw_sdf.hip).
// gbuf: units 0 .. F_GU-1 of a tile = qbar_0 = J_gamma nbar, unit F_GU = the d_sdf unit.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * F_WAVES) void sdf_bwd16f_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                 const float* __restrict__ d_sdf,
                                                                 const float* __restrict__ d_grad, NcwSdfStash st) {
    typedef float SE;
    F_LDS_DECL();
    if (wave < T) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        const bool valid = p < n;
        if (!valid) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        const float vmask = valid ? 1.f : 0.f;  // padded lanes must contribute nothing to the weight gradients
        const float nb[3] = {d_grad[p * 3 + 0] * vmask, d_grad[p * 3 + 1] * vmask, d_grad[p * 3 + 2] * vmask};
        const float dsdf = d_sdf[p] * vmask / net.scale;
        const int h = lane >> 5;
        CVec<2> q0;  // qbar_0 = J_gamma nbar
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (32 * rb + ncw_feat_of(q, 0) >= 39) {
                    q0.v[rb][q] = 0.f;
                    continue;
                }
                int comp;
                const float dv = freq_feature_deriv<3, 6, false>(xs, 32 * rb + ncw_feat_of(q, 0) + 4 * h, comp);
                q0.v[rb][q] = dv * (comp == 0 ? nb[0] : (comp == 1 ? nb[1] : nb[2]));
            }
        stash_store<2>((SE*)st.qbar[0], (size_t)(tile0 + wave), q0, lane);
        f_store_gamma(gbuf, wave, q0, lane);
        CVec<1> zs, one;
        cvec_zero(zs);
        cvec_zero(one);
        zs.v[0][0] = (lane < 32) ? dsdf : 0.f;
        one.v[0][0] = (lane < 32) ? vmask : 0.f;
        stash_store<1>((SE*)st.zsdf, (size_t)(tile0 + wave), zs, lane);
        stash_store<1>((SE*)st.one, (size_t)(tile0 + wave), one, lane);
        gbuf[(wave * F_GS + F_GU) * 64 + lane] = zs.v[0][0];  // k-unit 0 of the d_sdf block (feature 0 on half 0)
    }
    FW r;
    f32x16 acc[2][T];
    // one output block of one tile: tbar -> zbar2_l (temporarily in zbar[l]), abar_l = qbar_{l+1} (stash + LDS)
    auto adj_epilogue = [&](const f32x16& tbar, int l, int t, int ob) {
        const f32x16 sv = f_sprime((const SE*)st.h[l + 1], (size_t)(tile0 + t), ob, lane);
        f32x16 tv, z2, ab;
        stash_load_block(tv, (const SE*)st.t[l], (size_t)(tile0 + t), 16, ob, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            z2[q] = tbar[q] * 100.f * tv[q] * (1.f - sv[q]);  // a_l phi''(z_l) = 100 t_l (1 - s_l)
            ab[q] = tbar[q] * sv[q];
        }
        stash_store_block_keep((SE*)st.zbar[l], (size_t)(tile0 + t), 16, ob, z2, lane);  // zbar2_l: re-read by pass 2
        stash_store_block((SE*)st.qbar[l + 1], (size_t)(tile0 + t), 16, ob, ab, lane);
        f_store_units(abuf, t, ob, ab, lane);
    };
    {   // (1) layer 0: tbar = W_0 qbar_0 (the F_GU gamma units)
        f_prefetch(r, 1 <= L - 2 ? net.w[1] : net.wt_feat, 16, wave, lane);
        f_fill(acc, f_zero(), f_zero());
        ncw_lds_barrier();
        f_mma_gamma(acc, net.w[0], 16, 0, wave, gbuf, lane);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) adj_epilogue(acc[j][t], 0, t, wave + 8 * j);
    }
    for (int l = 1; l <= L - 2; ++l) {  // r = first units of w[l]
        ncw_lds_barrier();
        f_fill(acc, f_zero(), f_zero());
        f_mma(acc, r, net.w[l], 16, wave, abuf, lane);
        f_prefetch(r, l + 1 <= L - 2 ? net.w[l + 1] : net.wt_feat, 16, wave, lane);
        if (l == net.skip_layer) f_mma_gamma(acc, net.w[l], 16, F_KU, wave, gbuf, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) adj_epilogue(acc[j][t], l, t, wave + 8 * j);
    }
    // ---- (2) u = wt_feat dfeat + wt[L-1] d_sdf  (r = first units of wt_feat) -----------------------------------------
    // zbar_l = u phi'(z_l) + zbar2_l  -> stash zbar[l] (+ LDS when a further layer consumes it)
    auto fwd_epilogue = [&](const f32x16& u, int l, int t, int ob) {
        const f32x16 sv = f_sprime((const SE*)st.h[l + 1], (size_t)(tile0 + t), ob, lane);
        f32x16 z2;
        stash_load_block(z2, (const SE*)st.zbar[l], (size_t)(tile0 + t), 16, ob, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) z2[q] = u[q] * sv[q] + z2[q];
        stash_store_block((SE*)st.zbar[l], (size_t)(tile0 + t), 16, ob, z2, lane);
        if (l > 0) f_store_units(abuf, t, ob, z2, lane);
    };
    {
        // the dfeat blocks of this wave: stash -> B units, over qbar_{L-1} in abuf, which nobody reads (it only goes to
        // the stash) and of which this wave owns exactly the units it overwrites
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 df;
                stash_load_block(df, (const SE*)st.dfeat, (size_t)(tile0 + t), 16, wave + 8 * j, lane);
                f_store_units(abuf, t, wave + 8 * j, df, lane);
            }
        const float wl0 = f_ld(net.wt[L - 1], 16, wave, 0, lane), wl1 = f_ld(net.wt[L - 1], 16, wave + 8, 0, lane);  // K = 1
        ncw_lds_barrier();  // dfeat complete in abuf
        f_fill(acc, f_zero(), f_zero());
        f_mma(acc, r, net.wt_feat, 16, wave, abuf, lane);
        if (L - 2 > 0) f_prefetch(r, net.wt[L - 2], (L - 2 == net.skip_layer) ? 18 : 16, wave, lane);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[0][t] = f_mfma(wl0, gbuf[(t * F_GS + F_GU) * 64 + lane], acc[0][t]);
            acc[1][t] = f_mfma(wl1, gbuf[(t * F_GS + F_GU) * 64 + lane], acc[1][t]);
        }
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) fwd_epilogue(acc[j][t], L - 2, t, wave + 8 * j);
    }
    for (int l = L - 2; l >= 1; --l) {  // u = wt[l] zbar_l, zbar_{l-1} = u phi'(z_{l-1}) + zbar2_{l-1}
        ncw_lds_barrier();
        f_fill(acc, f_zero(), f_zero());
        f_mma(acc, r, net.wt[l], (l == net.skip_layer) ? 18 : 16, wave, abuf, lane);
        if (l - 1 > 0) f_prefetch(r, net.wt[l - 1], (l - 1 == net.skip_layer) ? 18 : 16, wave, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) fwd_epilogue(acc[j][t], l - 1, t, wave + 8 * j);
    }
}

}  // namespace

#define F_GRID(n) dim3((unsigned)((((n) + 31) / 32 + F_T - 1) / F_T))

int ncw_sdf_infer16f_launch(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st) {
    hipLaunchKernelGGL(sdf_infer16f_kernel, F_GRID(n), dim3(64 * F_WAVES), 0, st, *net, src, n, sdf);
    NCW_CHECK_LAUNCH();
    return 0;
}
int ncw_sdf_fwd16f_launch(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad, const NcwSdfStash& stash,
                          hipStream_t st) {
    if (stash.t[0] == nullptr)  // forward-only render
        hipLaunchKernelGGL(sdf_fwd16f_kernel<false>, F_GRID(n), dim3(64 * F_WAVES), 0, st, *net, src, n, sdf, grad, stash);
    else
        hipLaunchKernelGGL(sdf_fwd16f_kernel<true>, F_GRID(n), dim3(64 * F_WAVES), 0, st, *net, src, n, sdf, grad, stash);
    NCW_CHECK_LAUNCH();
    return 0;
}
int ncw_sdf_bwd16f_launch(const NcwSdfNet* net, const NcwPoints& src, int64_t n, const float* d_sdf, const float* d_grad,
                          const NcwSdfStash& stash, hipStream_t st) {
    hipLaunchKernelGGL(sdf_bwd16f_kernel, F_GRID(n), dim3(64 * F_WAVES), 0, st, *net, src, n, d_sdf, d_grad, stash);
    NCW_CHECK_LAUNCH();
    return 0;
}

#endif  // !NCW_HALF_F16
