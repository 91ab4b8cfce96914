// Weights-stationary fused MLP kernels, W = 256, 16-bit: sdf_fwdB, nerf_fwdB, nerf_bwdB.
// (The 8-wave / half-layer pilot of round 1 and the sdf_bwd / color_fwd ports that did not beat the
// two-workgroups-per-CU kernels were removed in round 2, the burst inference kernel sdf_inferB -- superseded by
// ncw_pp.hip's fine-interleaved sdf_inferC -- in round 3; numbers in DESIGN.md.)
#include "ncw_mlp.h"

namespace {

// ------------------------------------------------------------------------------------------------
// The structure: WEIGHTS STATIONARY IN REGISTERS, ACTIVATIONS THROUGH LDS.
// A workgroup of 8 waves owns 128 points (4 tiles of 32).  Wave w owns OUTPUT BLOCK w of every hidden layer:
// its slice of the layer's packed matrix (16 k-units x 1 KiB = 64 registers per lane) is loaded global -> registers,
// the NEXT layer's slice while the current one is being used (the loads have a whole layer to land), and is reused
// for the 4 tiles.  The activations of the 128 points live in LDS in B-FRAGMENT form ([tile][k-unit][64 lanes][16 B],
// 64 KiB, double-buffered): every wave reads all of them (16 ds_read_b128 per tile) and writes the two units of its
// own output block.  No LDS-DMA, no per-chunk barriers (one barrier per layer), the same LDS read bytes per MFMA
// as the weights-through-LDS kernels.  The unit order of the packed weights is unchanged: unit 2 rb + t of the
// next layer IS the bf16 image of registers 8t..8t+7 of C-layout block rb (ncw_common.h), which is what a wave
// holds in its accumulator.
// ------------------------------------------------------------------------------------------------
constexpr int SB_WAVES = 8, SB_TILES = 4;

// phi'(z) = 1 - exp(-100 h) from the stashed post-activation h (ncw_sdf.hip load_sprime_block)
NCW_DEV void load_sprime_block_bf16(f32x16& sv, const ncw_h16* __restrict__ st_h, size_t tile, int RB, int rb, int lane) {
    stash_load_block(sv, st_h, tile, RB, rb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = 1.f - __builtin_amdgcn_exp2f(sv[r] * -144.26950408889634f);
}
constexpr int SB_ACT = SB_TILES * 16 * 1024;   // one activation buffer: 4 tiles x 16 units x 1 KiB
constexpr int SB_GAM = SB_TILES * 3 * 1024;    // gamma: 3 units per tile

typedef __attribute__((address_space(3))) bf16x8 sb_lfrag;

template <int NU>
NCW_DEV void sb_load_slice(bf16x8* a, const void* w, int rb_stride, int ob, int u0, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) a[q] = ncw_ld_frag<bf16x8>(w, (size_t)(u0 + q) * rb_stride + ob, lane);
}

// ------------------------------------------------------------------------------------------------
// sdf_fwd in the weights-stationary structure: forward chain with the activation stash, feature
// layer, sdf row, then the analytic adjoint pass a_{l-1} = W_l^T (a_l * phi'(z_l)) with the t_l stash and
// grad = J_gamma^T g_gamma -- the same arithmetic as sdf_fwd_kernel (ncw_sdf.hip), W = 256 bf16.
// The two gamma output blocks of the transposed skip layer and of W_0^T are 2 blocks x 4 tiles = 8 jobs:
// wave w takes block (w & 1) of tile (w >> 1) and keeps that g_gamma block (16 registers) to the end.
// ------------------------------------------------------------------------------------------------
NCW_DEV void sb_store_units(sb_lfrag* buf, int t, int ob, const f32x16& v, int lane) {
    Act<PrecBF16, 1> o;
    to_act_block<1>(o, 0, v);
    buf[(t * 16 + 2 * ob) * 64 + lane] = o.f[0];
    buf[(t * 16 + 2 * ob + 1) * 64 + lane] = o.f[1];
}

// TRAIN = false: forward-only render -- bit for bit the same outputs; of the stash only h_l (the adjoint sweep's scratch) and
// feat (the colour network's input) are written, not gamma, not t_l (include/neuconw_hip.h, NcwSdfStash).
template <bool TRAIN>
__global__ __launch_bounds__(64 * SB_WAVES) void sdf_fwdB_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                float* __restrict__ sdf, float* __restrict__ grad,
                                                                NcwSdfStash st) {
    typedef ncw_h16 SE;
    __shared__ __attribute__((aligned(16))) char lds[2 * SB_ACT + SB_GAM];
    sb_lfrag* const abuf0 = (sb_lfrag*)(ncw_lchar*)lds;
    sb_lfrag* const abuf1 = abuf0 + SB_ACT / 16;
    sb_lfrag* const gbuf = abuf0 + 2 * SB_ACT / 16;
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int L = net.n_layers;
    const int64_t tile0 = (int64_t)blockIdx.x * SB_TILES;
    const int jb = wave & 1, jt = wave >> 1;  // this wave's gamma job: block jb of tile jt
    if (wave < SB_TILES) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, true>(gam, xs, lane);
        if (TRAIN) stash_store<2>((SE*)st.gamma, (size_t)(tile0 + wave), gam, lane);
        Act<PrecBF16, 2> ga;
        to_act(ga, gam);
#pragma unroll
        for (int q = 0; q < 3; ++q) gbuf[(wave * 3 + q) * 64 + lane] = ga.f[q];
    }
    bf16x8 wa[16], wb[16], wg[3];
    auto bias_block = [&](const float* bp) {
        CVec<1> b1;
        load_bias(b1, bp + wave * 32, lane);
        return b1.v[0];
    };
    // ---- layer 0 ------------------------------------------------------------------------------------
    {
        bf16x8 w0[3];
        sb_load_slice<3>(w0, net.w[0], 8, wave, 0, lane);
        sb_load_slice<16>(wa, L - 1 > 1 ? net.w[1] : net.w_feat, 8, wave, 0, lane);
        const f32x16 bias = bias_block(net.b[0]);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < SB_TILES; ++t) {
            f32x16 acc = bias;
#pragma unroll
            for (int q = 0; q < 3; ++q) acc = NCW_MFMA_H(w0[q], gbuf[(t * 3 + q) * 64 + lane], acc, 0, 0, 0);
            f32x16 yv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { float y, s; softplus100<true>(acc[r], y, s); yv[r] = y; }
            stash_store_block_keep((SE*)st.h[1], (size_t)(tile0 + t), 8, wave, yv, lane);
            sb_store_units(abuf0, t, wave, yv, lane);
        }
    }
    int cur = 0;
    // ---- hidden layers 1 .. L-2 (wa = slice of w[l]) ---------------------------------------------------
    for (int l = 1; l < L - 1; ++l) {
        const bool skip = (l == net.skip_layer);
        if (skip) sb_load_slice<3>(wg, net.w[l], 8, wave, 16, lane);
        sb_load_slice<16>(wb, l + 1 < L - 1 ? net.w[l + 1] : net.w_feat, 8, wave, 0, lane);  // next: hidden or feature layer
        const f32x16 bias = bias_block(net.b[l]);
        ncw_lds_barrier();
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 acc0 = bias, acc1 = bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
            if (skip) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    acc0 = NCW_MFMA_H(wg[q], gbuf[(tp * 3 + q) * 64 + lane], acc0, 0, 0, 0);
                    acc1 = NCW_MFMA_H(wg[q], gbuf[((tp + 1) * 3 + q) * 64 + lane], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& acc = j ? acc1 : acc0;
                f32x16 yv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { float y, s; softplus100<true>(acc[r], y, s); yv[r] = y; }
                stash_store_block_keep((SE*)st.h[l + 1], (size_t)(tile0 + tp + j), 8, wave, yv, lane);
                sb_store_units(out, tp + j, wave, yv, lane);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- feature layer (wa = slice of w_feat) and sdf row; prefetch the adjoint's first slice ---------------
    {
        bf16x8 wt1 = ((const __attribute__((address_space(1))) bf16x8*)net.wt[L - 1])[(size_t)wave * 64 + lane];  // unit 0, block ob
        if (L - 2 >= 1) sb_load_slice<16>(wb, net.wt[L - 2], (L - 2 == net.skip_layer) ? 10 : 8, wave, 0, lane);
        const f32x16 bias = bias_block(net.b_feat);
        ncw_lds_barrier();  // h_{L-1} complete in abuf[cur]
        const sb_lfrag* in = cur ? abuf1 : abuf0;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 acc0 = bias, acc1 = bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
            stash_store_block((SE*)st.feat, (size_t)(tile0 + tp), 8, wave, acc0, lane);
            stash_store_block((SE*)st.feat, (size_t)(tile0 + tp + 1), 8, wave, acc1, lane);
        }
        if (wave < SB_TILES) {
            bf16x8 w1[16];
            sb_load_slice<16>(w1, net.w[L - 1], 1, 0, 0, lane);
            CVec<1> o;
            load_bias(o, net.b[L - 1], lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) o.v[0] = NCW_MFMA_H(w1[q], in[(wave * 16 + q) * 64 + lane], o.v[0], 0, 0, 0);
            const int64_t p = (tile0 + wave) * 32 + (lane & 31);
            if (p < n && lane < 32) sdf[p] = o.v[0][0] / net.scale;
        }
        // ---- adjoint start: a_{L-2} = W_{L-1}^T e_0 (the same for every point); t_{L-2} = a * phi'(z_{L-2}) ----
        bf16x8 e0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) e0f[e] = (ncw_h16)0.f;
        e0f[0] = (ncw_h16)(lane < 32 ? 1.f : 0.f);
        f32x16 a0;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.f;
        a0 = NCW_MFMA_H(wt1, e0f, a0, 0, 0, 0);
        sb_lfrag* out = cur ? abuf0 : abuf1;  // free: its readers finished before the barrier above
#pragma unroll
        for (int t = 0; t < SB_TILES; ++t) {
            f32x16 sv;
            load_sprime_block_bf16(sv, (const SE*)st.h[L - 1], (size_t)(tile0 + t), 8, wave, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] *= a0[r];
            if (TRAIN) stash_store_block((SE*)st.t[L - 2], (size_t)(tile0 + t), 8, wave, sv, lane);
            sb_store_units(out, t, wave, sv, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- adjoint layers l = L-2 .. 1: t_{l-1} = (W_l^T t_l) * phi'(z_{l-1});  wa = slice of wt[l] ---------------
    f32x16 gg;
#pragma unroll
    for (int r = 0; r < 16; ++r) gg[r] = 0.f;
    for (int l = L - 2; l >= 1; --l) {
        const bool skip = (l == net.skip_layer);
        const int stride = skip ? 10 : 8;
        (void)stride;
        if (!skip && l - 1 >= 1) sb_load_slice<16>(wb, net.wt[l - 1], (l - 1 == net.skip_layer) ? 10 : 8, wave, 0, lane);
        ncw_lds_barrier();  // t_l complete in abuf[cur]
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& acc = j ? acc1 : acc0;
                f32x16 sv;
                load_sprime_block_bf16(sv, (const SE*)st.h[l], (size_t)(tile0 + tp + j), 8, wave, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] *= acc[r];
                if (TRAIN) stash_store_block((SE*)st.t[l - 1], (size_t)(tile0 + tp + j), 8, wave, sv, lane);
                sb_store_units(out, tp + j, wave, sv, lane);
            }
        }
        if (skip) {  // the gamma columns of the transposed skip layer: out-blocks 8, 9 (one (block, tile) job per wave)
            sb_load_slice<16>(wb, net.wt[l], 10, 8 + jb, 0, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) gg = NCW_MFMA_H(wb[q], in[(jt * 16 + q) * 64 + lane], gg, 0, 0, 0);
            if (l - 1 >= 1) sb_load_slice<16>(wb, net.wt[l - 1], (l - 1 == net.skip_layer) ? 10 : 8, wave, 0, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- adjoint layer 0: g_gamma += W_0^T t_0 (2 out-blocks), then grad = J_gamma^T g_gamma -----------------------
    sb_load_slice<16>(wb, net.wt[0], 2, jb, 0, lane);
    ncw_lds_barrier();
    {
        const sb_lfrag* in = cur ? abuf1 : abuf0;
#pragma unroll
        for (int q = 0; q < 16; ++q) gg = NCW_MFMA_H(wb[q], in[(jt * 16 + q) * 64 + lane], gg, 0, 0, 0);
    }
    int64_t p = (tile0 + jt) * 32 + (lane & 31), ray;
    const bool valid = p < n;
    if (!valid) p = n - 1;
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f0 = 32 * jb + ncw_feat_of(r, 0);
        if (f0 >= 39) continue;  // (block 1 holds features 32..38 only)
        int comp;
        const float dv = freq_feature_deriv<3, 6, true>(xs, f0 + 4 * h, comp);
        const float c = gg[r] * dv;
        nx += comp == 0 ? c : 0.f;
        ny += comp == 1 ? c : 0.f;
        nz += comp == 2 ? c : 0.f;
    }
    nx = half_pair_sum(nx); ny = half_pair_sum(ny); nz = half_pair_sum(nz);
    // combine the two blocks of a tile (waves 2 jt and 2 jt + 1) through LDS (the gamma region is free now)
    typedef __attribute__((address_space(3))) float lfloat;
    lfloat* part = (lfloat*)gbuf;
    if (jb == 1 && lane < 32) {
        part[(jt * 32 + lane) * 3 + 0] = nx; part[(jt * 32 + lane) * 3 + 1] = ny; part[(jt * 32 + lane) * 3 + 2] = nz;
    }
    ncw_lds_barrier();
    if (jb == 0 && lane < 32 && valid) {
        grad[p * 3 + 0] = nx + part[(jt * 32 + lane) * 3 + 0];
        grad[p * 3 + 1] = ny + part[(jt * 32 + lane) * 3 + 1];
        grad[p * 3 + 2] = nz + part[(jt * 32 + lane) * 3 + 2];
    }
}


// ------------------------------------------------------------------------------------------------
// nerf_fwd (models/nerf.py:86-183 on the inverted-sphere points of renderer.py:176-186) in the weights-stationary
// structure, W = 256, 16-bit: trunk with the gamma(p) skip, density, feature layer, appearance head,
// raw rgb -- the same arithmetic and stash as nerf_fwd_kernel (ncw_nerf.hip).  The 128-wide head layers are
// 4 output blocks x 4 tiles: wave w takes block (w & 3) for the tile pair (w >> 2).
// xbuf ([4 tiles][6 units]) holds gamma(p) during the trunk and AUX1 = [gamma(dir) | appearance code] for the head.
// ------------------------------------------------------------------------------------------------
constexpr int SB_X = SB_TILES * 6 * 1024;

NCW_DEV void inverted_sphere8(const float (&x)[3], float (&p4)[4]) {  // renderer.py:181-186
    float r = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    r = fminf(fmaxf(r, 1.0f), 1e10f);
    p4[0] = x[0] / r; p4[1] = x[1] / r; p4[2] = x[2] / r; p4[3] = 1.0f / r;
}

// TRAIN = false: forward-only render -- nothing is stashed (st.gp == NULL selects it; st.aux_bias stays an input).
template <bool TRAIN>
__global__ __launch_bounds__(64 * SB_WAVES) void nerf_fwdB_kernel(NcwNerfNet net, NcwPoints src, const float* __restrict__ x4,
                                                                 int64_t n, const float* __restrict__ a,
                                                                 float* __restrict__ density, float* __restrict__ rgb,
                                                                 NcwNerfStash st) {
    typedef ncw_h16 SE;
    __shared__ __attribute__((aligned(16))) char lds[2 * SB_ACT + SB_X + SB_TILES * 32 * 4];
    sb_lfrag* const abuf0 = (sb_lfrag*)(ncw_lchar*)lds;
    sb_lfrag* const abuf1 = abuf0 + SB_ACT / 16;
    sb_lfrag* const xbuf = abuf0 + 2 * SB_ACT / 16;
    typedef __attribute__((address_space(3))) int sb_lint;
    sb_lint* const rbuf = (sb_lint*)(xbuf + SB_X / 16);  // ray index of every point of the workgroup (per-ray head bias)
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile0 = (int64_t)blockIdx.x * SB_TILES;
    const int D = net.D;
    n = points_count(src, n);            // mode 4: the selection's size, read on the device
    if (tile0 * 32 >= n) return;         // (uniform: before any barrier)
    int64_t pp = 0;
    bool pvalid = false;
    if (wave < SB_TILES) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        pvalid = p < n;
        if (!pvalid) p = n - 1;
        pp = point_slot(src, p);         // density / rgb are addressed by the ray sample
        float p4[4];
        if (x4) {
            p4[0] = x4[p * 4 + 0]; p4[1] = x4[p * 4 + 1]; p4[2] = x4[p * 4 + 2]; p4[3] = x4[p * 4 + 3];
            ray = p;
        } else {
            float xs[3];
            load_point(src, p, xs, ray);
            inverted_sphere8(xs, p4);
        }
        const float dir[3] = {src.rays_d[ray * 3 + 0], src.rays_d[ray * 3 + 1], src.rays_d[ray * 3 + 2]};
        if (lane < 32) rbuf[wave * 32 + lane] = (int)ray;
        CVec<3> gp;
        freq_encode<3, 4, 10, true>(gp, p4, lane);
        if (TRAIN) stash_store<3>((SE*)st.gp, (size_t)(tile0 + wave), gp, lane);
        Act<PrecBF16, 3> gpa;
        to_act(gpa, gp);
#pragma unroll
        for (int q = 0; q < 6; ++q) xbuf[(wave * 6 + q) * 64 + lane] = gpa.f[q];
        if (TRAIN) {
            CVec<3> aux1;
            build_aux1<true>(aux1, dir, a + ray * net.n_a, net.n_a, lane);
            stash_store<3>((SE*)st.aux1, (size_t)(tile0 + wave), aux1, lane);  // re-read for the head (same bf16 values)
        }
    }
    bf16x8 wa[16], wb[16], wx[6];
    auto bias_of = [&](const float* bp, int ob) {
        CVec<1> b1;
        load_bias(b1, bp + ob * 32, lane);
        return b1.v[0];
    };
    auto relu16 = [](const f32x16& v) {
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = ncw_relu(v[r]);
        return y;
    };
    // ---- trunk layer 0 (K = 84: the 6 units of gamma(p)) ---------------------------------------------------
    {
        bf16x8 w0[6];
        sb_load_slice<6>(w0, net.w_p[0], 8, wave, 0, lane);
        sb_load_slice<16>(wa, D > 1 ? net.w_p[1] : net.w_feat, 8, wave, 0, lane);
        const f32x16 bias = bias_of(net.b_p[0], wave);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < SB_TILES; ++t) {
            f32x16 acc = bias;
#pragma unroll
            for (int q = 0; q < 6; ++q) acc = NCW_MFMA_H(w0[q], xbuf[(t * 6 + q) * 64 + lane], acc, 0, 0, 0);
            const f32x16 y = relu16(acc);
            if (TRAIN) stash_store_block((SE*)st.h[1], (size_t)(tile0 + t), 8, wave, y, lane);
            sb_store_units(abuf0, t, wave, y, lane);
        }
    }
    int cur = 0;
    // ---- trunk layers 1 .. D-1 -----------------------------------------------------------------------------
    for (int i = 1; i < D; ++i) {
        const bool skip = (i == net.skip + 1);
        if (skip) sb_load_slice<6>(wx, net.w_p[i], 8, wave, 16, lane);  // units 16..21 = the gamma(p) columns
        sb_load_slice<16>(wb, i + 1 < D ? net.w_p[i + 1] : net.w_feat, 8, wave, 0, lane);
        const f32x16 bias = bias_of(net.b_p[i], wave);
        ncw_lds_barrier();
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 acc0 = bias, acc1 = bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
            if (skip) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc0 = NCW_MFMA_H(wx[q], xbuf[(tp * 6 + q) * 64 + lane], acc0, 0, 0, 0);
                    acc1 = NCW_MFMA_H(wx[q], xbuf[((tp + 1) * 6 + q) * 64 + lane], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 y = relu16(j ? acc1 : acc0);
                if (TRAIN) stash_store_block((SE*)st.h[i + 1], (size_t)(tile0 + tp + j), 8, wave, y, lane);
                sb_store_units(out, tp + j, wave, y, lane);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- density (waves 0..3), feature layer (wa = slice of w_feat), AUX1 into xbuf -------------------------------
    const int hb = wave & 3, hp = wave >> 2;  // head job: block hb, tiles 2 hp and 2 hp + 1
    {
        sb_load_slice<16>(wb, net.w_a[0], 4, hb, 0, lane);   // head layer 0: feature columns
        sb_load_slice<6>(wx, net.w_a[0], 4, hb, 16, lane);   //               AUX1 columns (units 16..21)
        const f32x16 bias = bias_of(net.b_feat, wave);
        ncw_lds_barrier();  // h_D complete in abuf[cur]; nobody reads gamma(p) in xbuf any more
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
        if (wave < SB_TILES) {
            if (TRAIN || st.aux_bias == nullptr) {  // (with the per-ray fp32 head columns the AUX1 operand is not multiplied)
                CVec<3> aux1;
                if (TRAIN) {
                    stash_load<3>(aux1, (const SE*)st.aux1, (size_t)(tile0 + wave), lane);
                } else {  // nothing was stashed: rebuild it (the 16-bit image below is the one the stash would have held)
                    const int64_t ray = rbuf[wave * 32 + (lane & 31)];
                    const float dir[3] = {src.rays_d[ray * 3 + 0], src.rays_d[ray * 3 + 1], src.rays_d[ray * 3 + 2]};
                    build_aux1<true>(aux1, dir, a + ray * net.n_a, net.n_a, lane);
                }
                Act<PrecBF16, 3> aux1a;
                to_act(aux1a, aux1);
#pragma unroll
                for (int q = 0; q < 6; ++q) xbuf[(wave * 6 + q) * 64 + lane] = aux1a.f[q];
            }
            bf16x8 w1[16];
            sb_load_slice<16>(w1, net.w_alpha, 1, 0, 0, lane);
            CVec<1> o;
            load_bias(o, net.b_alpha, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) o.v[0] = NCW_MFMA_H(w1[q], in[(wave * 16 + q) * 64 + lane], o.v[0], 0, 0, 0);
            if (pvalid && lane < 32) density[pp] = o.v[0][0];
        }
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 acc0 = bias, acc1 = bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
            if (TRAIN) {
                stash_store_block((SE*)st.featn, (size_t)(tile0 + tp), 8, wave, acc0, lane);
                stash_store_block((SE*)st.featn, (size_t)(tile0 + tp + 1), 8, wave, acc1, lane);
            }
            sb_store_units(out, tp, wave, acc0, lane);
            sb_store_units(out, tp + 1, wave, acc1, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- appearance head (nerf.py:131-139,173-174): layer 0 takes [feature | AUX1], the others 128 -> 128 ---------
    for (int i = 0; i < net.n_head; ++i) {
        bf16x8 wn[8];
        if (i + 1 < net.n_head) sb_load_slice<8>(wn, net.w_a[i + 1], 4, hb, 0, lane);
        const f32x16 bias = bias_of(net.b_a[i], hb);
        ncw_lds_barrier();
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
        const int ta = 2 * hp, tb = 2 * hp + 1;
        f32x16 acc0 = bias, acc1 = bias;
        if (i == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(ta * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[(tb * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
            if (st.aux_bias != nullptr) {  // the AUX1 columns: per-ray fp32 rows of ncw_aux_ray_bias instead of 16-bit operands
                ncw_add_ray_bias_block(acc0, st.aux_bias + (size_t)rbuf[ta * 32 + (lane & 31)] * 128, hb, lane);
                ncw_add_ray_bias_block(acc1, st.aux_bias + (size_t)rbuf[tb * 32 + (lane & 31)] * 128, hb, lane);
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc0 = NCW_MFMA_H(wx[q], xbuf[(ta * 6 + q) * 64 + lane], acc0, 0, 0, 0);
                    acc1 = NCW_MFMA_H(wx[q], xbuf[(tb * 6 + q) * 64 + lane], acc1, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(ta * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[(tb * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
        }
        const f32x16 y0 = relu16(acc0), y1 = relu16(acc1);
        if (TRAIN) {
            stash_store_block((SE*)st.e[i], (size_t)(tile0 + ta), 4, hb, y0, lane);
            stash_store_block((SE*)st.e[i], (size_t)(tile0 + tb), 4, hb, y1, lane);
        }
        sb_store_units(out, ta, hb, y0, lane);
        sb_store_units(out, tb, hb, y1, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) wa[q] = wn[q];
        cur ^= 1;
    }
    // ---- raw rgb (nerf.py:181), waves 0..3 ---------------------------------------------------------------------
    ncw_lds_barrier();
    if (wave < SB_TILES) {
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        bf16x8 w1[8];
        sb_load_slice<8>(w1, net.w_rgb, 1, 0, 0, lane);
        CVec<1> o;
        load_bias(o, net.b_rgb, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[0] = NCW_MFMA_H(w1[q], in[(wave * 16 + q) * 64 + lane], o.v[0], 0, 0, 0);
        if (pvalid && lane < 32) {
            rgb[pp * 3 + 0] = o.v[0][0];
            rgb[pp * 3 + 1] = o.v[0][1];
            rgb[pp * 3 + 2] = o.v[0][2];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// nerf_bwd in the weights-stationary structure, W = 256, 16-bit: the data-gradient chain of
// nerf_bwd_kernel (ncw_nerf.hip) -- rgb head reversed, appearance head, feature / density, trunk -- emitting the
// z-bar stashes the weight-gradient GEMMs read and the per-ray appearance-code gradient d_a.
// ------------------------------------------------------------------------------------------------
NCW_DEV f32x16 sb_relu_bwd(const f32x16& u, const ncw_h16* __restrict__ st_y, size_t tile, int RB, int rb, int lane) {
    f32x16 y, z;
    stash_load_block(y, st_y, tile, RB, rb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = y[r] > 0.f ? u[r] : 0.f;
    return z;
}

// d_a[ray][j] += sum over the tile's points of block rb of the AUX1 adjoint (features 27 .. 27 + n_a), as accumulate_d_a
NCW_DEV void sb_accumulate_d_a(int rb, const f32x16& q, float* __restrict__ d_a, int64_t ray, int n_a, bool valid, int lane) {
    const int h = lane >> 5;
    const int64_t r0 = __shfl(ray, 0, 64);
    const bool uniform = __all((ray == r0) ? 1 : 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 32 * rb + ncw_feat_of(r, 0) + 4 * h;
        const int j = f - 27;
        float v = valid ? q[r] : 0.f;
        if (uniform) {
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
            if ((lane & 31) == 0 && j >= 0 && j < n_a) atomicAdd(d_a + ray * n_a + j, v);
        } else {
            if (valid && j >= 0 && j < n_a) atomicAdd(d_a + ray * n_a + j, v);
        }
    }
}

__global__ __launch_bounds__(64 * SB_WAVES) void nerf_bwdB_kernel(NcwNerfNet net, NcwPoints src, int64_t n,
                                                                 const float* __restrict__ d_density,
                                                                 const float* __restrict__ d_rgb, float* __restrict__ d_a,
                                                                 NcwNerfStash st) {
    typedef ncw_h16 SE;
    __shared__ __attribute__((aligned(16))) char lds[2 * SB_ACT + SB_X];
    sb_lfrag* const abuf0 = (sb_lfrag*)(ncw_lchar*)lds;
    sb_lfrag* const abuf1 = abuf0 + SB_ACT / 16;
    sb_lfrag* const xbuf = abuf0 + 2 * SB_ACT / 16;   // unit 0 of a tile: d_rgb block, unit 1: d_density block
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile0 = (int64_t)blockIdx.x * SB_TILES;
    const int D = net.D, NH = net.n_head;
    const int hb = wave & 3, hp = wave >> 2, ta = 2 * hp, tb = 2 * hp + 1;
    typedef const __attribute__((address_space(1))) bf16x8* gfrag;
    n = points_count(src, n);            // mode 4: the selection's size, read on the device
    if (tile0 * 32 >= n) return;         // (uniform: before any barrier)
    if (wave < SB_TILES) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31);
        const bool valid = p < n;
        if (!valid) p = n - 1;
        p = point_slot(src, p);          // d_density / d_rgb are addressed by the ray sample
        const float vm = valid ? 1.f : 0.f;
        CVec<1> zr, zal;
        cvec_zero(zr);
        cvec_zero(zal);
        if (lane < 32) {
            zr.v[0][0] = d_rgb[p * 3 + 0] * vm;
            zr.v[0][1] = d_rgb[p * 3 + 1] * vm;
            zr.v[0][2] = d_rgb[p * 3 + 2] * vm;
            zal.v[0][0] = d_density[p] * vm;
        }
        stash_store<1>((SE*)st.zrgb, (size_t)(tile0 + wave), zr, lane);
        stash_store<1>((SE*)st.zalpha, (size_t)(tile0 + wave), zal, lane);
        Act<PrecBF16, 1> za1;
        to_act(za1, zr);
        xbuf[(wave * 6 + 0) * 64 + lane] = za1.f[0];
        to_act(za1, zal);
        xbuf[(wave * 6 + 1) * 64 + lane] = za1.f[0];
    }
    bf16x8 wa[16], wb[16];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    int cur = 0;
    // ---- appearance head reversed: wave = (block hb, tiles ta, tb) -------------------------------------------------
    f32x16 ue0, ue1;
    {
        const bf16x8 wr = ((gfrag)net.wt_rgb)[(size_t)hb * 64 + lane];  // wt_rgb: 4 out-blocks, unit 0 (K = 3)
        if (NH > 1) sb_load_slice<8>(wa, net.wt_a[NH - 1], 4, hb, 0, lane);
        else sb_load_slice<8>(wa, net.wt_a[0], 11, wave, 0, lane);
        ncw_lds_barrier();  // d_rgb / d_density units visible
        ue0 = NCW_MFMA_H(wr, xbuf[(ta * 6 + 0) * 64 + lane], zero16, 0, 0, 0);
        ue1 = NCW_MFMA_H(wr, xbuf[(tb * 6 + 0) * 64 + lane], zero16, 0, 0, 0);
    }
    for (int i = NH - 1; i >= 1; --i) {
        sb_lfrag* out = cur ? abuf1 : abuf0;
        const f32x16 z0 = sb_relu_bwd(ue0, (const SE*)st.e[i], (size_t)(tile0 + ta), 4, hb, lane);
        const f32x16 z1 = sb_relu_bwd(ue1, (const SE*)st.e[i], (size_t)(tile0 + tb), 4, hb, lane);
        stash_store_block((SE*)st.ze[i], (size_t)(tile0 + ta), 4, hb, z0, lane);
        stash_store_block((SE*)st.ze[i], (size_t)(tile0 + tb), 4, hb, z1, lane);
        sb_store_units(out, ta, hb, z0, lane);
        sb_store_units(out, tb, hb, z1, lane);
        if (i - 1 >= 1) sb_load_slice<8>(wb, net.wt_a[i - 1], 4, hb, 0, lane);
        else sb_load_slice<8>(wb, net.wt_a[0], 11, wave, 0, lane);  // next: q = wt_a[0] ze_0, block = wave
        ncw_lds_barrier();
        ue0 = zero16; ue1 = zero16;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ue0 = NCW_MFMA_H(wa[q], out[(ta * 16 + q) * 64 + lane], ue0, 0, 0, 0);
            ue1 = NCW_MFMA_H(wa[q], out[(tb * 16 + q) * 64 + lane], ue1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- ze_0, then q = wt_a[0] ze_0: blocks 0..7 = d(feature) (block = wave), blocks 8..10 = AUX1 adjoint -> d_a -----
    {
        sb_lfrag* out = cur ? abuf1 : abuf0;
        const f32x16 z0 = sb_relu_bwd(ue0, (const SE*)st.e[0], (size_t)(tile0 + ta), 4, hb, lane);
        const f32x16 z1 = sb_relu_bwd(ue1, (const SE*)st.e[0], (size_t)(tile0 + tb), 4, hb, lane);
        stash_store_block((SE*)st.ze[0], (size_t)(tile0 + ta), 4, hb, z0, lane);
        stash_store_block((SE*)st.ze[0], (size_t)(tile0 + tb), 4, hb, z1, lane);
        sb_store_units(out, ta, hb, z0, lane);
        sb_store_units(out, tb, hb, z1, lane);
        sb_load_slice<16>(wb, net.wt_feat, 8, wave, 0, lane);  // next: u = wt_feat zf
        ncw_lds_barrier();
        const sb_lfrag* in = out;
        sb_lfrag* out2 = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 a0 = zero16, a1 = zero16;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], a0, 0, 0, 0);
                a1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], a1, 0, 0, 0);
            }
            stash_store_block((SE*)st.zfeat, (size_t)(tile0 + tp), 8, wave, a0, lane);
            stash_store_block((SE*)st.zfeat, (size_t)(tile0 + tp + 1), 8, wave, a1, lane);
            sb_store_units(out2, tp, wave, a0, lane);
            sb_store_units(out2, tp + 1, wave, a1, lane);
        }
        // the 3 AUX1 blocks x 4 tiles = 12 jobs: wave w takes jobs w and w + 8
        for (int j = wave; j < 12; j += SB_WAVES) {
            const int b = j % 3, t = j / 3;
            bf16x8 wq[8];
            sb_load_slice<8>(wq, net.wt_a[0], 11, 8 + b, 0, lane);
            f32x16 qa = zero16;
#pragma unroll
            for (int q = 0; q < 8; ++q) qa = NCW_MFMA_H(wq[q], in[(t * 16 + q) * 64 + lane], qa, 0, 0, 0);
            int64_t p = (tile0 + t) * 32 + (lane & 31);
            const bool valid = p < n;
            if (!valid) p = n - 1;
            p = point_slot(src, p);
            const int64_t ray = (src.mode == 0) ? p : p / src.per_ray;
            sb_accumulate_d_a(b, qa, d_a, ray, net.n_a, valid, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;  // zf lives in out2
    }
    // ---- u = wt_feat zf + wt_alpha d_density;  za_{D-1} = relu'(h_D) u --------------------------------------------
    {
        const bf16x8 wal = ((gfrag)net.wt_alpha)[(size_t)wave * 64 + lane];  // wt_alpha: 8 out-blocks, unit 0 (K = 1)
        if (D - 1 > 0) sb_load_slice<16>(wb, net.wt_p[D - 1], (D - 1 == net.skip + 1) ? 11 : 8, wave, 0, lane);
        ncw_lds_barrier();  // zf complete
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 u0 = zero16, u1 = zero16;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                u0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], u0, 0, 0, 0);
                u1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], u1, 0, 0, 0);
            }
            u0 = NCW_MFMA_H(wal, xbuf[(tp * 6 + 1) * 64 + lane], u0, 0, 0, 0);
            u1 = NCW_MFMA_H(wal, xbuf[((tp + 1) * 6 + 1) * 64 + lane], u1, 0, 0, 0);
            const f32x16 z0 = sb_relu_bwd(u0, (const SE*)st.h[D], (size_t)(tile0 + tp), 8, wave, lane);
            const f32x16 z1 = sb_relu_bwd(u1, (const SE*)st.h[D], (size_t)(tile0 + tp + 1), 8, wave, lane);
            stash_store_block((SE*)st.zp[D - 1], (size_t)(tile0 + tp), 8, wave, z0, lane);
            stash_store_block((SE*)st.zp[D - 1], (size_t)(tile0 + tp + 1), 8, wave, z1, lane);
            sb_store_units(out, tp, wave, z0, lane);
            sb_store_units(out, tp + 1, wave, z1, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- trunk reversed: za_{i-1} = relu'(h_i) (wt_p[i] za_i), i = D-1 .. 1 (wa = slice of wt_p[i]) ----------------------
    for (int i = D - 1; i >= 1; --i) {
        if (i - 1 >= 1) sb_load_slice<16>(wb, net.wt_p[i - 1], (i - 1 == net.skip + 1) ? 11 : 8, wave, 0, lane);
        ncw_lds_barrier();
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SB_TILES; tp += 2) {
            f32x16 u0 = zero16, u1 = zero16;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                u0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], u0, 0, 0, 0);
                u1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], u1, 0, 0, 0);
            }
            const f32x16 z0 = sb_relu_bwd(u0, (const SE*)st.h[i], (size_t)(tile0 + tp), 8, wave, lane);
            const f32x16 z1 = sb_relu_bwd(u1, (const SE*)st.h[i], (size_t)(tile0 + tp + 1), 8, wave, lane);
            stash_store_block((SE*)st.zp[i - 1], (size_t)(tile0 + tp), 8, wave, z0, lane);
            stash_store_block((SE*)st.zp[i - 1], (size_t)(tile0 + tp + 1), 8, wave, z1, lane);
            if (i - 1 >= 1) {
                sb_store_units(out, tp, wave, z0, lane);
                sb_store_units(out, tp + 1, wave, z1, lane);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
}


}  // namespace

int NCW_FN(ncw_sdf_fwd8_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad, const NcwSdfStash& stash,
                        hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    if (stash.t[0] == nullptr)  // forward-only render
        hipLaunchKernelGGL(sdf_fwdB_kernel<false>, dim3((unsigned)((tiles + SB_TILES - 1) / SB_TILES)), dim3(64 * SB_WAVES), 0, st,
                           *net, src, n, sdf, grad, stash);
    else
        hipLaunchKernelGGL(sdf_fwdB_kernel<true>, dim3((unsigned)((tiles + SB_TILES - 1) / SB_TILES)), dim3(64 * SB_WAVES), 0, st,
                           *net, src, n, sdf, grad, stash);
    NCW_CHECK_LAUNCH();
    return 0;
}

int NCW_FN(ncw_nerf_fwd8_launch)(const NcwNerfNet* net, const NcwPoints& src, const float* x4, int64_t n, const float* a, float* density,
                         float* rgb, const NcwNerfStash& stash, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    if (stash.gp == nullptr)  // forward-only render
        hipLaunchKernelGGL(nerf_fwdB_kernel<false>, dim3((unsigned)((tiles + SB_TILES - 1) / SB_TILES)), dim3(64 * SB_WAVES), 0, st,
                           *net, src, x4, n, a, density, rgb, stash);
    else
        hipLaunchKernelGGL(nerf_fwdB_kernel<true>, dim3((unsigned)((tiles + SB_TILES - 1) / SB_TILES)), dim3(64 * SB_WAVES), 0, st,
                           *net, src, x4, n, a, density, rgb, stash);
    NCW_CHECK_LAUNCH();
    return 0;
}

int NCW_FN(ncw_nerf_bwd8_launch)(const NcwNerfNet* net, const NcwPoints& src, int64_t n, const float* d_density, const float* d_rgb,
                         float* d_a, const NcwNerfStash& stash, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    hipLaunchKernelGGL(nerf_bwdB_kernel, dim3((unsigned)((tiles + SB_TILES - 1) / SB_TILES)), dim3(64 * SB_WAVES), 0, st, *net, src, n,
                       d_density, d_rgb, d_a, stash);
    NCW_CHECK_LAUNCH();
    return 0;
}
