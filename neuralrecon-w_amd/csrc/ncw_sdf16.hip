// SDF network at W = 512 (the width the reference ships, config/train_*.yaml), 16-bit operands: sdf_infer, sdf_fwd and
// sdf_bwd with the WEIGHTS STREAMED FROM L2 and the activations in LDS -- the arithmetic and the stashes of the generic
// kernels of ncw_sdf.hip (models/neuconw.py:263-296 forward, the analytic adjoint pass for the normals, the second-order
// backward), which at RB = 16 need 256 accumulator registers per wave and spill 1.1-1.8 KB per lane.
//
// A layer is 512 KB of 16-bit weights = the whole register file of a CU, so the weights-stationary structure of
// ncw_sdf8.hip does not carry over.  Here a workgroup of 8 waves owns T tiles of 32 points (T = 4: 128 points; T = 2 for
// launches that would not fill the chip otherwise); wave w owns OUTPUT BLOCKS w and w + 8 of every layer.  Per k-unit
// (16 input features) it loads its two A fragments (2 x 16 B per lane) from the packed matrix in L2, S16_D units ahead
// into a small register ring, reads the T B fragments of the unit from LDS and issues 2 T MFMAs.  The activations of
// the workgroup live in ONE LDS buffer ([tile][32 k-units][64 lanes][16 B] = T x 32 KiB) that every layer rewrites IN
// PLACE: all waves read all of it during the MFMA phase (the layer's whole output sits in 32 T accumulator registers per
// wave), a barrier, then every wave overwrites the two blocks it owns.  Per layer and workgroup at T = 4: 2,048 MFMAs
// (16,384 cycles per SIMD quad) against 512 KB from L2 (8,192 cycles at 64 B/clk/CU; the 9.2 MB of forward + transposed
// matrices do not fit the 4 MB L2 of an XCD, so part of it comes from the memory-side cache) and 1 MB of LDS reads
// (8,192 cycles).  Nothing spills.
#include "ncw_mlp.h"

#ifdef S16_EXP_NOSTASH  // timing experiment only: no stash stores (the backward reads garbage) -- probe libraries only
#ifndef NCW_PROBE_BUILD
#error "S16_EXP_NOSTASH is a probe-only timing hook: build with NCW_BUILD_TAG=<tag> (neuralrecon-w_amd/build.py)"
#endif
#define stash_store_block(...) ((void)0)
#define stash_store_block_keep(...) ((void)0)
#endif

namespace {

constexpr int S16_WAVES = 8;
constexpr int S16_KU = 32;  // k-units of a 512-wide layer
#ifndef S16_DEPTH
#define S16_DEPTH 4
#endif
constexpr int S16_D = S16_DEPTH;  // weight prefetch distance (k-units)

typedef __attribute__((address_space(3))) bf16x8 s16_lfrag;
typedef const __attribute__((address_space(1))) bf16x8* s16_gfrag;

struct S16W { bf16x8 f[S16_D][2]; };  // register ring: units q .. q + D - 1 of the wave's two output blocks

NCW_DEV bf16x8 s16_ld(const void* w, int rb_stride, int ob, int u, int lane) {
    return ncw_ld_frag<bf16x8>(w, (size_t)u * rb_stride + ob, lane);
}

// the first S16_D units of a matrix (issued a layer ahead: they land during the previous epilogue and the barriers)
NCW_DEV void s16_prefetch(S16W& r, const void* w, int rb_stride, int wave, int lane) {
#pragma unroll
    for (int d = 0; d < S16_D; ++d) {
        r.f[d][0] = s16_ld(w, rb_stride, wave, d, lane);
        r.f[d][1] = s16_ld(w, rb_stride, wave + 8, d, lane);
    }
}

// acc[j][t] += W[block wave + 8 j][.] . in[tile t][.] over NU k-units; r holds units 0 .. D-1 on entry and is free on exit
// (BS = 2: `in` is a split-precision buffer [tile][unit][hi | lo]; the hi fragments are read)
template <int T, int NU, int BS = 1>
NCW_DEV void s16_mma(f32x16 (&acc)[2][T], S16W& r, const void* w, int rb_stride, int wave, const s16_lfrag* in, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        const bf16x8 a0 = r.f[q % S16_D][0], a1 = r.f[q % S16_D][1];
        if (q + S16_D < NU) {
            r.f[q % S16_D][0] = s16_ld(w, rb_stride, wave, q + S16_D, lane);
            r.f[q % S16_D][1] = s16_ld(w, rb_stride, wave + 8, q + S16_D, lane);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bf16x8 b = in[((t * S16_KU + q) * BS) * 64 + lane];
            acc[0][t] = NCW_MFMA_H(a0, b, acc[0][t], 0, 0, 0);
            acc[1][t] = NCW_MFMA_H(a1, b, acc[1][t], 0, 0, 0);
        }
    }
}

// the same with the weights as hi + lo pairs (`w` / `wlo`: a packed matrix and its residual matrix, two register rings): every
// B fragment read from LDS feeds FOUR MFMAs (NcwSdfNet.wt_lo: the adjoint sweep of the fp16 mode, ncw_split.hip sdf_fwdSA_kernel)
template <int T, int NU>
NCW_DEV void s16_mma_hl(f32x16 (&acc)[2][T], S16W& r, S16W& rl, const void* w, const void* wlo, int rb_stride, int wave,
                        const s16_lfrag* in, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        const bf16x8 a0 = r.f[q % S16_D][0], a1 = r.f[q % S16_D][1], l0 = rl.f[q % S16_D][0], l1 = rl.f[q % S16_D][1];
        if (q + S16_D < NU) {
            r.f[q % S16_D][0] = s16_ld(w, rb_stride, wave, q + S16_D, lane);
            r.f[q % S16_D][1] = s16_ld(w, rb_stride, wave + 8, q + S16_D, lane);
            rl.f[q % S16_D][0] = s16_ld(wlo, rb_stride, wave, q + S16_D, lane);
            rl.f[q % S16_D][1] = s16_ld(wlo, rb_stride, wave + 8, q + S16_D, lane);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bf16x8 b = in[(t * S16_KU + q) * 64 + lane];
            acc[0][t] = NCW_MFMA_H(a0, b, acc[0][t], 0, 0, 0);
            acc[1][t] = NCW_MFMA_H(a1, b, acc[1][t], 0, 0, 0);
            acc[0][t] = NCW_MFMA_H(l0, b, acc[0][t], 0, 0, 0);
            acc[1][t] = NCW_MFMA_H(l1, b, acc[1][t], 0, 0, 0);
        }
    }
}

// ... and with BOTH operands as hi + lo pairs (round 6): `in` is a split buffer [tile][unit][hi | lo]; W_hi b_hi + W_lo b_hi + W_hi b_lo,
// six MFMAs per (unit, tile), consecutive MFMAs on different accumulators
template <int T, int NU>
NCW_DEV void s16_mma_hl2(f32x16 (&acc)[2][T], S16W& r, S16W& rl, const void* w, const void* wlo, int rb_stride, int wave,
                         const s16_lfrag* in, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        const bf16x8 a0 = r.f[q % S16_D][0], a1 = r.f[q % S16_D][1], l0 = rl.f[q % S16_D][0], l1 = rl.f[q % S16_D][1];
        if (q + S16_D < NU) {
            r.f[q % S16_D][0] = s16_ld(w, rb_stride, wave, q + S16_D, lane);
            r.f[q % S16_D][1] = s16_ld(w, rb_stride, wave + 8, q + S16_D, lane);
            rl.f[q % S16_D][0] = s16_ld(wlo, rb_stride, wave, q + S16_D, lane);
            rl.f[q % S16_D][1] = s16_ld(wlo, rb_stride, wave + 8, q + S16_D, lane);
        }
        bf16x8 bh[T], bl[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            bh[t] = in[((t * S16_KU + q) * 2) * 64 + lane];
            bl[t] = in[((t * S16_KU + q) * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) { acc[0][t] = NCW_MFMA_H(a0, bh[t], acc[0][t], 0, 0, 0); acc[1][t] = NCW_MFMA_H(a1, bh[t], acc[1][t], 0, 0, 0); }
#pragma unroll
        for (int t = 0; t < T; ++t) { acc[0][t] = NCW_MFMA_H(l0, bh[t], acc[0][t], 0, 0, 0); acc[1][t] = NCW_MFMA_H(l1, bh[t], acc[1][t], 0, 0, 0); }
#pragma unroll
        for (int t = 0; t < T; ++t) { acc[0][t] = NCW_MFMA_H(a0, bl[t], acc[0][t], 0, 0, 0); acc[1][t] = NCW_MFMA_H(a1, bl[t], acc[1][t], 0, 0, 0); }
    }
}

// the 3 gamma k-units (32..34) of the forward-orientation skip layer against gbuf (units 0..2 of a tile)
template <int T>
NCW_DEV void s16_mma_gamma(f32x16 (&acc)[2][T], const void* w, int wave, const s16_lfrag* gbuf, int lane) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const bf16x8 g0 = s16_ld(w, 16, wave, S16_KU + q, lane), g1 = s16_ld(w, 16, wave + 8, S16_KU + q, lane);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[0][t] = NCW_MFMA_H(g0, gbuf[(t * 4 + q) * 64 + lane], acc[0][t], 0, 0, 0);
            acc[1][t] = NCW_MFMA_H(g1, gbuf[(t * 4 + q) * 64 + lane], acc[1][t], 0, 0, 0);
        }
    }
}

// one output block of one tile (gamma columns of a transposed matrix, the sdf row): acc += W[block ob][.] . in[tile t][.]
// (BS = 2: a split buffer; PLANE 0 = its hi fragments, 1 = its lo fragments)
template <int NU, int BS = 1, int PLANE = 0>
NCW_DEV void s16_mma1(f32x16& acc, const void* w, int rb_stride, int ob, const s16_lfrag* in, int t, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q)
        acc = NCW_MFMA_H(s16_ld(w, rb_stride, ob, q, lane), in[((t * S16_KU + q) * BS + PLANE) * 64 + lane], acc, 0, 0, 0);
}

// hi and lo fragments of the two k-units of one output block into a split buffer [tile][unit][hi | lo]
NCW_DEV void s16_store_units_hl(s16_lfrag* buf, int t, int ob, const f32x16& v, int lane) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        bf16x8 hi, lo;
        ncw_split8(v, tt, hi, lo);
        buf[((t * S16_KU + 2 * ob + tt) * 2 + 0) * 64 + lane] = hi;
        buf[((t * S16_KU + 2 * ob + tt) * 2 + 1) * 64 + lane] = lo;
    }
}

NCW_DEV void s16_store_units(s16_lfrag* buf, int t, int ob, const f32x16& v, int lane) {
    Act<PrecBF16, 1> o;
    to_act_block<1>(o, 0, v);
    buf[(t * S16_KU + 2 * ob) * 64 + lane] = o.f[0];
    buf[(t * S16_KU + 2 * ob + 1) * 64 + lane] = o.f[1];
}

NCW_DEV f32x16 s16_bias(const float* bp, int ob, int lane) {
    CVec<1> b1;
    load_bias(b1, bp + ob * 32, lane);
    return b1.v[0];
}

NCW_DEV f32x16 s16_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

template <int T>
NCW_DEV void s16_fill(f32x16 (&acc)[2][T], const f32x16& b0, const f32x16& b1) {
#pragma unroll
    for (int t = 0; t < T; ++t) { acc[0][t] = b0; acc[1][t] = b1; }
}

// TU: the value-only launches run the hidden layers in t-units (ncw_common.h softplus_tu: gamma and the biases enter x 100 log2 e,
// the sdf row's result is divided by it, the hidden matrices are unchanged)
template <bool TU = false>
NCW_DEV f32x16 s16_softplus(const f32x16& z) {
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = softplus_sel<TU>(z[r]);
    return y;
}

// phi'(z) = 1 - exp(-100 h) from the stashed post-activation h
NCW_DEV f32x16 s16_sprime(const ncw_h16* __restrict__ st_h, size_t tile, int ob, int lane) {
    f32x16 sv;
    stash_load_block(sv, st_h, tile, 16, ob, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = 1.f - __builtin_amdgcn_exp2f(sv[r] * -144.26950408889634f);
    return sv;
}

// ... from h as an fp16 hi + lo pair (st_lo = NcwSdfStash.s: the residual stash of the split value chain; adj_mode 2)
NCW_DEV f32x16 s16_sprime_hl(const ncw_h16* __restrict__ st_h, const ncw_h16* __restrict__ st_lo, size_t tile, int ob, int lane) {
    f32x16 sv, lv;
    stash_load_block(sv, st_h, tile, 16, ob, lane);
    stash_load_block(lv, st_lo, tile, 16, ob, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = 1.f - __builtin_amdgcn_exp2f((sv[r] + lv[r]) * -144.26950408889634f);
    return sv;
}

// abuf: the activations, rewritten in place by every layer; gbuf: per tile 4 units (gamma / qbar_0: 3, the d_sdf unit: 1)
#define S16_LDS_DECL()                                                                \
    __shared__ __attribute__((aligned(16))) char lds[T * S16_KU * 1024 + T * 4096];   \
    s16_lfrag* const abuf = (s16_lfrag*)(ncw_lchar*)lds;                              \
    s16_lfrag* const gbuf = abuf + T * S16_KU * 64;                                   \
    const int lane = ncw_lane();                                                      \
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));        \
    const int L = net.n_layers;                                                       \
    const int64_t tile0 = (int64_t)blockIdx.x * T

// ------------------------------------------------------------------------------------------------
// sdf_infer: the sampler's SDF queries (renderer.py:482-566), the octree refresh and the 512^3 grid sweep
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(64 * S16_WAVES) void sdf_infer16_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                    float* __restrict__ sdf) {
    S16_LDS_DECL();
    for (int tw = wave; tw < T; tw += S16_WAVES) {  // gamma of tile tw, straight into LDS as k-units 0..2
        int64_t p = (tile0 + tw) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, true>(gam, xs, lane);
        cvec_scale<2>(gam, NCW_TU);  // t-units
        Act<PrecBF16, 2> ga;
        to_act(ga, gam);
#pragma unroll
        for (int q = 0; q < 3; ++q) gbuf[(tw * 4 + q) * 64 + lane] = ga.f[q];
    }
    S16W r;
    f32x16 acc[2][T];
    {   // layer 0: K = 39 (3 units)
        bf16x8 w0[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) w0[j][q] = s16_ld(net.w[0], 16, wave + 8 * j, q, lane);
        if (L - 1 > 1) s16_prefetch(r, net.w[1], 16, wave, lane);
        const f32x16 b0 = s16_bias(net.b[0], wave, lane) * NCW_TU, b1 = s16_bias(net.b[0], wave + 8, lane) * NCW_TU;
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 a = j ? b1 : b0;
#pragma unroll
                for (int q = 0; q < 3; ++q) a = NCW_MFMA_H(w0[j][q], gbuf[(t * 4 + q) * 64 + lane], a, 0, 0, 0);
                s16_store_units(abuf, t, wave + 8 * j, s16_softplus<true>(a), lane);
            }
    }
    for (int l = 1; l < L - 1; ++l) {
        const f32x16 b0 = s16_bias(net.b[l], wave, lane) * NCW_TU, b1 = s16_bias(net.b[l], wave + 8, lane) * NCW_TU;
        ncw_lds_barrier();  // layer l-1 outputs of all waves are in abuf
        s16_fill<T>(acc, b0, b1);
        s16_mma<T, S16_KU>(acc, r, net.w[l], 16, wave, abuf, lane);
        if (l + 1 < L - 1) s16_prefetch(r, net.w[l + 1], 16, wave, lane);
        if (l == net.skip_layer) s16_mma_gamma<T>(acc, net.w[l], wave, gbuf, lane);
        ncw_lds_barrier();  // every wave has read abuf: overwrite in place
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) s16_store_units(abuf, t, wave + 8 * j, s16_softplus<true>(acc[j][t]), lane);
    }
    ncw_lds_barrier();
    for (int tw = wave; tw < T; tw += S16_WAVES) {  // sdf row
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        cvec_scale<1>(o, NCW_TU);
        s16_mma1<S16_KU>(o.v[0], net.w[L - 1], 1, 0, abuf, tw, lane);
        const int64_t p = (tile0 + tw) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / (net.scale * NCW_TU);
    }
}

// The part of sdf_fwd after the forward chain: feature rows, (sdf row,) the analytic adjoint sweep with the t_l stash and
// grad = J_gamma^T g_gamma.  On entry h_{L-1} of the T tiles is in abuf (BS = 1: plain fragments; BS = 2: the split chain's
// [tile][unit][hi | lo] buffer, whose hi fragments are read) and r holds the first units of w_feat; abuf is then reused in the
// plain layout.  SDF_ROW = false: the caller's (split-precision) chain has written sdf already.
// TRAIN = false: forward-only render -- t_l is not stashed (feat is: the colour network reads it).
// ADJ 1: the sweep's transposed weights as hi + lo pairs (net.wt_lo; round 5 -- the normals are multiplied by dist * inv_s inside the
// compositor's sigmoid: ncw_split.hip sdf_fwdSA_kernel has the measurements).
// ADJ 2 (round 6, BS = 2 only: the split kernel's 128 KiB buffer): t_l as a hi + lo pair too -- W_hi^T t_hi + W_lo^T t_hi + W_hi^T t_lo,
// i.e. the normals at the accuracy of the value chain.  At W = 512 with the shipped 8 + 16 samples ONE sample carries most of a
// ray's weight and the colour network reads the normal of that sample: on trained weights the plain sweep's 4e-4 put 2 % of the
// rays above 1e-4, W^T alone as a pair made the worst ray worse (profiles/r05/emul_timed_batch_shipped.log), both operands --
// together with the colour network's activations as pairs -- bring the worst of 256 rays to 2e-5 (profiles/r06/
// emul_timed_batch_shipped_tangent*.log).  The stash t_l stays the single-rounded hi part: the backward is unchanged.
template <int T, int BS, bool SDF_ROW, bool TRAIN, int ADJ = 0>
NCW_DEV void s16_fwd_tail(const NcwSdfNet& net, const NcwPoints& src, int64_t n, float* __restrict__ sdf, float* __restrict__ grad,
                          const NcwSdfStash& st, s16_lfrag* abuf, s16_lfrag* gbuf, S16W& r, f32x16 (&acc)[2][T], int lane, int wave,
                          int64_t tile0) {
    typedef ncw_h16 SE;
    const int L = net.n_layers;
    const int jb = wave & 1, jt = wave >> 1;
    const bool gjob = wave < 2 * T;
    S16W rl;  // the residual matrices' register ring (ADJ)
    // ---- feature layer (r = first units of w_feat) and sdf row; then the adjoint's first vector ------------------
    {
        const bf16x8 wt1_0 = s16_ld(net.wt[L - 1], 16, wave, 0, lane), wt1_1 = s16_ld(net.wt[L - 1], 16, wave + 8, 0, lane);
        const f32x16 b0 = s16_bias(net.b_feat, wave, lane), b1 = s16_bias(net.b_feat, wave + 8, lane);
        ncw_lds_barrier();  // h_{L-1} complete in abuf
        s16_fill<T>(acc, b0, b1);
        s16_mma<T, S16_KU, BS>(acc, r, net.w_feat, 16, wave, abuf, lane);
        if (L - 2 >= 1) s16_prefetch(r, net.wt[L - 2], (L - 2 == net.skip_layer) ? 18 : 16, wave, lane);
        if (ADJ && L - 2 >= 1) s16_prefetch(rl, net.wt_lo[L - 2], (L - 2 == net.skip_layer) ? 18 : 16, wave, lane);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) stash_store_block((SE*)st.feat, (size_t)(tile0 + t), 16, wave + 8 * j, acc[j][t], lane);
        if (SDF_ROW && wave < T) {
            CVec<1> o;
            load_bias(o, net.b[L - 1], lane);
            s16_mma1<S16_KU, BS>(o.v[0], net.w[L - 1], 1, 0, abuf, wave, lane);
            const int64_t p = (tile0 + wave) * 32 + (lane & 31);
            if (p < n && lane < 32) sdf[p] = o.v[0][0] / net.scale;
        }
        // a_{L-2} = W_{L-1}^T e_0 (the same for every point); t_{L-2} = a * phi'(z_{L-2})
        bf16x8 e0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) e0f[e] = (ncw_h16)0.f;
        e0f[0] = (ncw_h16)(lane < 32 ? 1.f : 0.f);
        f32x16 a0 = NCW_MFMA_H(wt1_0, e0f, s16_zero(), 0, 0, 0), a1 = NCW_MFMA_H(wt1_1, e0f, s16_zero(), 0, 0, 0);
        if (ADJ) {
            a0 = NCW_MFMA_H(s16_ld(net.wt_lo[L - 1], 16, wave, 0, lane), e0f, a0, 0, 0, 0);
            a1 = NCW_MFMA_H(s16_ld(net.wt_lo[L - 1], 16, wave + 8, 0, lane), e0f, a1, 0, 0, 0);
        }
        ncw_lds_barrier();  // the feature layer and the sdf row have read h_{L-1}: overwrite abuf with t_{L-2}
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 sv = (ADJ == 2 && st.s[L - 1] != nullptr)
                                ? s16_sprime_hl((const SE*)st.h[L - 1], (const SE*)st.s[L - 1], (size_t)(tile0 + t), wave + 8 * j, lane)
                                : s16_sprime((const SE*)st.h[L - 1], (size_t)(tile0 + t), wave + 8 * j, lane);
                const f32x16& aa = j ? a1 : a0;
#pragma unroll
                for (int q = 0; q < 16; ++q) sv[q] *= aa[q];
                if (TRAIN) stash_store_block((SE*)st.t[L - 2], (size_t)(tile0 + t), 16, wave + 8 * j, sv, lane);
                if (ADJ == 2) s16_store_units_hl(abuf, t, wave + 8 * j, sv, lane);
                else s16_store_units(abuf, t, wave + 8 * j, sv, lane);
            }
    }
    // ---- adjoint layers l = L-2 .. 1: t_{l-1} = (W_l^T t_l) * phi'(z_{l-1});  r = first units of wt[l] -------------
    f32x16 gg = s16_zero();
    for (int l = L - 2; l >= 1; --l) {
        const bool skip = (l == net.skip_layer);
        ncw_lds_barrier();  // t_l complete in abuf
        s16_fill<T>(acc, s16_zero(), s16_zero());
        if (ADJ == 2) s16_mma_hl2<T, S16_KU>(acc, r, rl, net.wt[l], net.wt_lo[l], skip ? 18 : 16, wave, abuf, lane);
        else if (ADJ) s16_mma_hl<T, S16_KU>(acc, r, rl, net.wt[l], net.wt_lo[l], skip ? 18 : 16, wave, abuf, lane);
        else s16_mma<T, S16_KU>(acc, r, net.wt[l], skip ? 18 : 16, wave, abuf, lane);
        if (l - 1 >= 1) s16_prefetch(r, net.wt[l - 1], (l - 1 == net.skip_layer) ? 18 : 16, wave, lane);
        if (ADJ && l - 1 >= 1) s16_prefetch(rl, net.wt_lo[l - 1], (l - 1 == net.skip_layer) ? 18 : 16, wave, lane);
        if (skip && gjob) {  // gamma columns of the skip layer
            s16_mma1<S16_KU, (ADJ == 2 ? 2 : 1)>(gg, net.wt[l], 18, 16 + jb, abuf, jt, lane);
            if (ADJ) s16_mma1<S16_KU, (ADJ == 2 ? 2 : 1)>(gg, net.wt_lo[l], 18, 16 + jb, abuf, jt, lane);
            if (ADJ == 2) s16_mma1<S16_KU, 2, 1>(gg, net.wt[l], 18, 16 + jb, abuf, jt, lane);
        }
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 sv = (ADJ == 2 && st.s[l] != nullptr)
                                ? s16_sprime_hl((const SE*)st.h[l], (const SE*)st.s[l], (size_t)(tile0 + t), wave + 8 * j, lane)
                                : s16_sprime((const SE*)st.h[l], (size_t)(tile0 + t), wave + 8 * j, lane);
#pragma unroll
                for (int q = 0; q < 16; ++q) sv[q] *= acc[j][t][q];
                if (TRAIN) stash_store_block((SE*)st.t[l - 1], (size_t)(tile0 + t), 16, wave + 8 * j, sv, lane);
                if (ADJ == 2) s16_store_units_hl(abuf, t, wave + 8 * j, sv, lane);
                else s16_store_units(abuf, t, wave + 8 * j, sv, lane);
            }
    }
    // ---- adjoint layer 0: g_gamma += W_0^T t_0 (2 out-blocks), then grad = J_gamma^T g_gamma ------------------------
    ncw_lds_barrier();
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int64_t p = (tile0 + jt) * 32 + (lane & 31), ray;
    const bool valid = gjob && p < n;
    if (gjob) {
        s16_mma1<S16_KU, (ADJ == 2 ? 2 : 1)>(gg, net.wt[0], 2, jb, abuf, jt, lane);
        if (ADJ) s16_mma1<S16_KU, (ADJ == 2 ? 2 : 1)>(gg, net.wt_lo[0], 2, jb, abuf, jt, lane);
        if (ADJ == 2) s16_mma1<S16_KU, 2, 1>(gg, net.wt[0], 2, jb, abuf, jt, lane);
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
            const float dv = freq_feature_deriv<3, 6, ADJ == 0>(xs, f0 + 4 * h, comp);  // (ADJ: sinf / cosf like the split value chain's gamma)
            const float c = gg[q] * dv;
            nx += comp == 0 ? c : 0.f;
            ny += comp == 1 ? c : 0.f;
            nz += comp == 2 ? c : 0.f;
        }
        nx = half_pair_sum(nx); ny = half_pair_sum(ny); nz = half_pair_sum(nz);
    }
    // combine the two blocks of a tile (waves 2 jt and 2 jt + 1) through LDS (the gamma region is free now)
    typedef __attribute__((address_space(3))) float lfloat;
    lfloat* part = (lfloat*)gbuf;
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
// sdf_fwd: forward chain with the activation stash, feature layer, sdf row, then the analytic adjoint pass
// t_{l-1} = (W_l^T t_l) * phi'(z_{l-1}) with the t_l stash and grad = J_gamma^T g_gamma (sdf_fwd_kernel, ncw_sdf.hip).
// The gamma output blocks (16, 17) of the transposed skip layer and the two blocks of W_0^T are 2 blocks x T tiles
// jobs: wave w < 2 T takes block (w & 1) of tile (w >> 1) and keeps that g_gamma block to the end (2 T <= 8).
// ------------------------------------------------------------------------------------------------
// TRAIN = false (sdf_render16_kernel): forward-only render -- bit for bit the same outputs; of the stash only h_l (the adjoint
// sweep's scratch) and feat are written.
template <int T, bool TRAIN>
NCW_DEV void sdf_fwd16_body(const NcwSdfNet& net, const NcwPoints& src, int64_t n, float* __restrict__ sdf, float* __restrict__ grad,
                            const NcwSdfStash& st) {
    typedef ncw_h16 SE;
    static_assert(2 * T <= S16_WAVES, "one gamma job per wave");
    S16_LDS_DECL();
    if (wave < T) {
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
        for (int q = 0; q < 3; ++q) gbuf[(wave * 4 + q) * 64 + lane] = ga.f[q];
    }
    S16W r;
    f32x16 acc[2][T];
    {   // layer 0
        bf16x8 w0[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) w0[j][q] = s16_ld(net.w[0], 16, wave + 8 * j, q, lane);
        s16_prefetch(r, L - 1 > 1 ? net.w[1] : net.w_feat, 16, wave, lane);
        const f32x16 b0 = s16_bias(net.b[0], wave, lane), b1 = s16_bias(net.b[0], wave + 8, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 a = j ? b1 : b0;
#pragma unroll
                for (int q = 0; q < 3; ++q) a = NCW_MFMA_H(w0[j][q], gbuf[(t * 4 + q) * 64 + lane], a, 0, 0, 0);
                const f32x16 y = s16_softplus(a);
                stash_store_block_keep((SE*)st.h[1], (size_t)(tile0 + t), 16, wave + 8 * j, y, lane);
                s16_store_units(abuf, t, wave + 8 * j, y, lane);
            }
    }
    for (int l = 1; l < L - 1; ++l) {  // r = first units of w[l]
        const f32x16 b0 = s16_bias(net.b[l], wave, lane), b1 = s16_bias(net.b[l], wave + 8, lane);
        ncw_lds_barrier();
        s16_fill<T>(acc, b0, b1);
        s16_mma<T, S16_KU>(acc, r, net.w[l], 16, wave, abuf, lane);
        s16_prefetch(r, l + 1 < L - 1 ? net.w[l + 1] : net.w_feat, 16, wave, lane);  // next: hidden or feature layer
        if (l == net.skip_layer) s16_mma_gamma<T>(acc, net.w[l], wave, gbuf, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 y = s16_softplus(acc[j][t]);
                stash_store_block_keep((SE*)st.h[l + 1], (size_t)(tile0 + t), 16, wave + 8 * j, y, lane);
                s16_store_units(abuf, t, wave + 8 * j, y, lane);
            }
    }
    s16_fwd_tail<T, 1, true, TRAIN>(net, src, n, sdf, grad, st, abuf, gbuf, r, acc, lane, wave, tile0);
}

template <int T>
__global__ __launch_bounds__(64 * S16_WAVES) void sdf_fwd16_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                  float* __restrict__ sdf, float* __restrict__ grad,
                                                                  NcwSdfStash st) {
    sdf_fwd16_body<T, true>(net, src, n, sdf, grad, st);
}
template <int T>
__global__ __launch_bounds__(64 * S16_WAVES) void sdf_render16_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                     float* __restrict__ sdf, float* __restrict__ grad,
                                                                     NcwSdfStash st) {
    sdf_fwd16_body<T, false>(net, src, n, sdf, grad, st);
}

// ------------------------------------------------------------------------------------------------
// sdf_bwd (second order): (1) backward of the adjoint pass l = 0 .. L-2 (tbar = W qbar, abar = tbar phi',
// zbar2 = tbar 100 t (1 - phi')), (2) backward of the forward pass l = L-1 .. 0 (ubar = W^T zbar,
// zbar = ubar phi' + zbar2) -- the arithmetic and the stash of sdf_bwd_kernel (ncw_sdf.hip).
// gbuf: units 0..2 of a tile = qbar_0 = J_gamma nbar, unit 3 = the d_sdf block.
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(64 * S16_WAVES) void sdf_bwd16_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                  const float* __restrict__ d_sdf,
                                                                  const float* __restrict__ d_grad, NcwSdfStash st) {
    typedef ncw_h16 SE;
    S16_LDS_DECL();
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
                const float dv = freq_feature_deriv<3, 6, true>(xs, 32 * rb + ncw_feat_of(q, 0) + 4 * h, comp);
                q0.v[rb][q] = dv * (comp == 0 ? nb[0] : (comp == 1 ? nb[1] : nb[2]));
            }
        stash_store<2>((SE*)st.qbar[0], (size_t)(tile0 + wave), q0, lane);
        Act<PrecBF16, 2> q0a;
        to_act(q0a, q0);
#pragma unroll
        for (int q = 0; q < 3; ++q) gbuf[(wave * 4 + q) * 64 + lane] = q0a.f[q];
        CVec<1> zs, one;
        cvec_zero(zs);
        cvec_zero(one);
        zs.v[0][0] = (lane < 32) ? dsdf : 0.f;
        one.v[0][0] = (lane < 32) ? vmask : 0.f;
        stash_store<1>((SE*)st.zsdf, (size_t)(tile0 + wave), zs, lane);
        stash_store<1>((SE*)st.one, (size_t)(tile0 + wave), one, lane);
        Act<PrecBF16, 1> zsa;
        to_act(zsa, zs);
        gbuf[(wave * 4 + 3) * 64 + lane] = zsa.f[0];
    }
    S16W r;
    f32x16 acc[2][T];
    // one output block of one tile: tbar -> zbar2_l (temporarily in zbar[l]), abar_l = qbar_{l+1} (stash + LDS)
    auto adj_epilogue = [&](const f32x16& tbar, int l, int t, int ob) {
        const f32x16 sv = s16_sprime((const SE*)st.h[l + 1], (size_t)(tile0 + t), ob, lane);
        f32x16 tv, z2, ab;
        stash_load_block(tv, (const SE*)st.t[l], (size_t)(tile0 + t), 16, ob, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            z2[q] = tbar[q] * 100.f * tv[q] * (1.f - sv[q]);  // a_l phi''(z_l) = 100 t_l (1 - s_l)
            ab[q] = tbar[q] * sv[q];
        }
        stash_store_block_keep((SE*)st.zbar[l], (size_t)(tile0 + t), 16, ob, z2, lane);  // zbar2_l: re-read by pass 2
        stash_store_block((SE*)st.qbar[l + 1], (size_t)(tile0 + t), 16, ob, ab, lane);
        s16_store_units(abuf, t, ob, ab, lane);
    };
    {   // (1) layer 0: tbar = W_0 qbar_0 (3 units)
        bf16x8 w0[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) w0[j][q] = s16_ld(net.w[0], 16, wave + 8 * j, q, lane);
        s16_prefetch(r, 1 <= L - 2 ? net.w[1] : net.wt_feat, 16, wave, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 a = s16_zero();
#pragma unroll
                for (int q = 0; q < 3; ++q) a = NCW_MFMA_H(w0[j][q], gbuf[(t * 4 + q) * 64 + lane], a, 0, 0, 0);
                adj_epilogue(a, 0, t, wave + 8 * j);
            }
    }
    for (int l = 1; l <= L - 2; ++l) {  // r = first units of w[l]
        ncw_lds_barrier();
        s16_fill<T>(acc, s16_zero(), s16_zero());
        s16_mma<T, S16_KU>(acc, r, net.w[l], 16, wave, abuf, lane);
        s16_prefetch(r, l + 1 <= L - 2 ? net.w[l + 1] : net.wt_feat, 16, wave, lane);
        if (l == net.skip_layer) s16_mma_gamma<T>(acc, net.w[l], wave, gbuf, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) adj_epilogue(acc[j][t], l, t, wave + 8 * j);
    }
    // ---- (2) u = wt_feat dfeat + wt[L-1] d_sdf  (r = first units of wt_feat) -----------------------------------------
    // zbar_l = u phi'(z_l) + zbar2_l  -> stash zbar[l] (+ LDS when a further layer consumes it)
    auto fwd_epilogue = [&](const f32x16& u, int l, int t, int ob) {
        const f32x16 sv = s16_sprime((const SE*)st.h[l + 1], (size_t)(tile0 + t), ob, lane);
        f32x16 z2;
        stash_load_block(z2, (const SE*)st.zbar[l], (size_t)(tile0 + t), 16, ob, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) z2[q] = u[q] * sv[q] + z2[q];
        stash_store_block((SE*)st.zbar[l], (size_t)(tile0 + t), 16, ob, z2, lane);
        if (l > 0) s16_store_units(abuf, t, ob, z2, lane);
    };
    {
        // the dfeat blocks of this wave: stash -> B fragments, over qbar_{L-1} in abuf, which nobody reads (it only goes
        // to the stash) and of which this wave owns exactly the units it overwrites
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 df;
                stash_load_block(df, (const SE*)st.dfeat, (size_t)(tile0 + t), 16, wave + 8 * j, lane);
                s16_store_units(abuf, t, wave + 8 * j, df, lane);
            }
        const bf16x8 wl0 = s16_ld(net.wt[L - 1], 16, wave, 0, lane), wl1 = s16_ld(net.wt[L - 1], 16, wave + 8, 0, lane);  // K = 1 unit
        ncw_lds_barrier();  // dfeat complete in abuf
        s16_fill<T>(acc, s16_zero(), s16_zero());
        s16_mma<T, S16_KU>(acc, r, net.wt_feat, 16, wave, abuf, lane);
        if (L - 2 > 0) s16_prefetch(r, net.wt[L - 2], (L - 2 == net.skip_layer) ? 18 : 16, wave, lane);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[0][t] = NCW_MFMA_H(wl0, gbuf[(t * 4 + 3) * 64 + lane], acc[0][t], 0, 0, 0);
            acc[1][t] = NCW_MFMA_H(wl1, gbuf[(t * 4 + 3) * 64 + lane], acc[1][t], 0, 0, 0);
        }
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) fwd_epilogue(acc[j][t], L - 2, t, wave + 8 * j);
    }
    for (int l = L - 2; l >= 1; --l) {  // u = wt[l] zbar_l, zbar_{l-1} = u phi'(z_{l-1}) + zbar2_{l-1}
        ncw_lds_barrier();
        s16_fill<T>(acc, s16_zero(), s16_zero());
        s16_mma<T, S16_KU>(acc, r, net.wt[l], (l == net.skip_layer) ? 18 : 16, wave, abuf, lane);
        if (l - 1 > 0) s16_prefetch(r, net.wt[l - 1], (l - 1 == net.skip_layer) ? 18 : 16, wave, lane);
        ncw_lds_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) fwd_epilogue(acc[j][t], l - 1, t, wave + 8 * j);
    }
}

#ifdef NCW_HALF_F16
// ------------------------------------------------------------------------------------------------
// Split-precision VALUE path at W = 512 (fp16 build; DESIGN.md 3.1b, the W = 256 counterpart is ncw_split.hip): the value
// chain gamma -> Softplus layers -> sdf row with BOTH operands as fp16 hi + lo pairs, W h ~= W_hi h_hi + W_lo h_hi + W_hi h_lo
// (three MFMAs, f32 accumulate), in the streamed-weights structure of this file: T = 2 tiles per workgroup (the activation
// buffer holds hi and lo: [tile][32 units][hi | lo] = 128 KiB), wave w owns output blocks w and w + 8, per k-unit four A
// fragments (hi, lo of both blocks) come through a register ring S16_D units ahead.  Twice the weight stream of the plain
// kernels (hi + lo) for three times the MFMAs: 1 MiB per layer per workgroup = 16 k cycles of the CU's L2 port against 24.6 k
// MFMA cycles.  Feature rows, adjoint sweep and backward stay plain fp16 (s16_fwd_tail reads the hi fragments).
// ------------------------------------------------------------------------------------------------
constexpr int S16S_T = 2;
struct S16WS { bf16x8 f[S16_D][4]; };  // ring: {block w hi, block w lo, block w+8 hi, block w+8 lo} of units q .. q + D - 1

NCW_DEV void s16s_ld_unit(bf16x8 (&f)[4], const void* w, const void* wlo, int rb_stride, int wave, int u, int lane) {
    f[0] = s16_ld(w, rb_stride, wave, u, lane);
    f[1] = s16_ld(wlo, rb_stride, wave, u, lane);
    f[2] = s16_ld(w, rb_stride, wave + 8, u, lane);
    f[3] = s16_ld(wlo, rb_stride, wave + 8, u, lane);
}

NCW_DEV void s16s_prefetch(S16WS& r, const void* w, const void* wlo, int rb_stride, int wave, int lane) {
#pragma unroll
    for (int d = 0; d < S16_D; ++d) s16s_ld_unit(r.f[d], w, wlo, rb_stride, wave, d, lane);
}

NCW_DEV void s16s_split8(const f32x16& v, int t, bf16x8& hi, bf16x8& lo) { ncw_split8(v, t, hi, lo); }

// six MFMAs per (unit, tile): blocks w and w + 8, {hi.hi, lo.hi, hi.lo}; consecutive MFMAs go to different accumulators
NCW_DEV void s16s_unit(f32x16 (&acc)[2][S16S_T], const bf16x8 (&a)[4], const s16_lfrag* in, int upt, int u, int lane) {
    bf16x8 bh[S16S_T], bl[S16S_T];
#pragma unroll
    for (int t = 0; t < S16S_T; ++t) {
        bh[t] = in[((t * upt + u) * 2) * 64 + lane];
        bl[t] = in[((t * upt + u) * 2 + 1) * 64 + lane];
    }
#pragma unroll
    for (int t = 0; t < S16S_T; ++t) { acc[0][t] = NCW_MFMA_H(a[0], bh[t], acc[0][t], 0, 0, 0); acc[1][t] = NCW_MFMA_H(a[2], bh[t], acc[1][t], 0, 0, 0); }
#pragma unroll
    for (int t = 0; t < S16S_T; ++t) { acc[0][t] = NCW_MFMA_H(a[1], bh[t], acc[0][t], 0, 0, 0); acc[1][t] = NCW_MFMA_H(a[3], bh[t], acc[1][t], 0, 0, 0); }
#pragma unroll
    for (int t = 0; t < S16S_T; ++t) { acc[0][t] = NCW_MFMA_H(a[0], bl[t], acc[0][t], 0, 0, 0); acc[1][t] = NCW_MFMA_H(a[2], bl[t], acc[1][t], 0, 0, 0); }
}

NCW_DEV void s16s_mma(f32x16 (&acc)[2][S16S_T], S16WS& r, const void* w, const void* wlo, int wave, const s16_lfrag* in, int lane) {
#pragma unroll
    for (int q = 0; q < S16_KU; ++q) {
        bf16x8 a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = r.f[q % S16_D][k];
        if (q + S16_D < S16_KU) s16s_ld_unit(r.f[q % S16_D], w, wlo, 16, wave, q + S16_D, lane);
        // keep the ring's loads where they are issued: hipcc otherwise sinks them next to their use (s_waitcnt vmcnt(0 / 1) in front of every
        // k-unit instead of 12-15 loads in flight).  Measured, round 6: value chain 0.607 -> 0.569 ms per 49,152 points, 10.9 -> 10.45 ms per
        // 1,048,576 (profiles/r06/w512_probe.log); a ring of 8 units instead of 4 is slower (registers).
        __builtin_amdgcn_sched_barrier(0);
        s16s_unit(acc, a, in, S16_KU, q, lane);
    }
}

// value chain: leaves h_{L-1} (hi | lo) of the 2 tiles in sbuf, the first units of w_feat are NOT prefetched (the caller's
// plain ring does that).  STASH 2: gamma, h_1 .. h_{L-1} (fp16 roundings = the hi parts); 1 (forward-only render): h_l only, the
// adjoint sweep's scratch; 0 (sdf_infer): nothing.
template <int STASH>
NCW_DEV void s16s_value_chain(const NcwSdfNet& net, const NcwPoints& src, int64_t n, int64_t tile0, s16_lfrag* sbuf, s16_lfrag* gsbuf,
                              int lane, int wave, float* __restrict__ sdf, const NcwSdfStash& st) {
    typedef ncw_h16 SE;
    constexpr int T = S16S_T;
    constexpr bool TU = STASH == 0;  // value-only launches: t-units
    const float bscale = TU ? NCW_TU : 1.f;
    const int L = net.n_layers;
    if (wave < T) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, false>(gam, xs, lane);  // sinf / cosf: the hardware v_sin / v_cos are not fp32-accurate
        if (STASH == 2) stash_store<2>((SE*)st.gamma, (size_t)(tile0 + wave), gam, lane);
        if (TU) cvec_scale<2>(gam, NCW_TU);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bf16x8 hi, lo;
            s16s_split8(gam.v[q >> 1], q & 1, hi, lo);
            gsbuf[((wave * 4 + q) * 2 + 0) * 64 + lane] = hi;
            gsbuf[((wave * 4 + q) * 2 + 1) * 64 + lane] = lo;
        }
    }
    S16WS r;
    f32x16 acc[2][T];
    auto epilogue = [&](int l_out) {  // Softplus, stash, hi / lo fragments of this wave's two blocks into sbuf (in place)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 y = s16_softplus<TU>(acc[j][t]);
                const int ob = wave + 8 * j;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    bf16x8 hi, lo;
                    s16s_split8(y, tt, hi, lo);
                    sbuf[((t * S16_KU + 2 * ob + tt) * 2 + 0) * 64 + lane] = hi;
                    sbuf[((t * S16_KU + 2 * ob + tt) * 2 + 1) * 64 + lane] = lo;
                    // the stash h (the backward's operand, the adjoint sweep's phi') takes the SAME hi bits the LDS holds -- registers 8 tt .. 8 tt + 7
                    // are groups 2 tt, 2 tt + 1 of the stash block: no second conversion -- and, adj_mode 2, NcwSdfStash.s their residuals
                    // (phi' = 1 - exp(-100 (h + h_lo)) then carries no fp16 rounding of h)
                    if (STASH >= 1) {
                        typedef ncw_h16 h4 __attribute__((ext_vector_type(4)));
                        const size_t at = (((size_t)(tile0 + t) * 16 + ob) * 4 + 2 * tt) * 64 + lane;
                        h4* qh = reinterpret_cast<h4*>(st.h[l_out]) + at;
                        const h4 a = {hi[0], hi[1], hi[2], hi[3]}, b = {hi[4], hi[5], hi[6], hi[7]};
                        qh[0] = a;
                        qh[64] = b;
                        if (st.s[l_out] != nullptr) {
                            h4* ql = reinterpret_cast<h4*>(st.s[l_out]) + at;
                            const h4 c = {lo[0], lo[1], lo[2], lo[3]}, d = {lo[4], lo[5], lo[6], lo[7]};
                            ql[0] = c;
                            ql[64] = d;
                        }
                    }
                }
            }
    };
    {   // layer 0: K = 39 (3 units of gamma)
        bf16x8 w0[3][4];
#pragma unroll
        for (int q = 0; q < 3; ++q) s16s_ld_unit(w0[q], net.w[0], net.w_lo[0], 16, wave, q, lane);
        if (L - 1 > 1) s16s_prefetch(r, net.w[1], net.w_lo[1], 16, wave, lane);
        s16_fill<T>(acc, s16_bias(net.b[0], wave, lane) * bscale, s16_bias(net.b[0], wave + 8, lane) * bscale);
        ncw_lds_barrier();  // gamma visible
#pragma unroll
        for (int q = 0; q < 3; ++q) s16s_unit(acc, w0[q], gsbuf, 4, q, lane);
        epilogue(1);
    }
    for (int l = 1; l < L - 1; ++l) {
        const f32x16 b0 = s16_bias(net.b[l], wave, lane) * bscale, b1 = s16_bias(net.b[l], wave + 8, lane) * bscale;
        ncw_lds_barrier();  // layer l-1 outputs of all waves are in sbuf
        s16_fill<T>(acc, b0, b1);
        s16s_mma(acc, r, net.w[l], net.w_lo[l], wave, sbuf, lane);
        if (l + 1 < L - 1) s16s_prefetch(r, net.w[l + 1], net.w_lo[l + 1], 16, wave, lane);
        if (l == net.skip_layer) {  // the gamma columns: units 32..34, loaded just in time (once per launch)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                bf16x8 g[4];
                s16s_ld_unit(g, net.w[l], net.w_lo[l], 16, wave, S16_KU + q, lane);
                s16s_unit(acc, g, gsbuf, 4, q, lane);
            }
        }
        ncw_lds_barrier();  // every wave has read sbuf: overwrite in place
        epilogue(l + 1);
    }
    ncw_lds_barrier();
    if (wave < T) {  // sdf row (1 output block), tile = wave
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        if (TU) cvec_scale<1>(o, NCW_TU);
#pragma unroll
        for (int c = 0; c < S16_KU; c += 8) {
            bf16x8 vh[8], vl[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { vh[q] = s16_ld(net.w[L - 1], 1, 0, c + q, lane); vl[q] = s16_ld(net.w_lo[L - 1], 1, 0, c + q, lane); }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bf16x8 bh = sbuf[((wave * S16_KU + c + q) * 2) * 64 + lane], bl = sbuf[((wave * S16_KU + c + q) * 2 + 1) * 64 + lane];
                o.v[0] = NCW_MFMA_H(vh[q], bh, o.v[0], 0, 0, 0);
                o.v[0] = NCW_MFMA_H(vl[q], bh, o.v[0], 0, 0, 0);
                o.v[0] = NCW_MFMA_H(vh[q], bl, o.v[0], 0, 0, 0);
            }
        }
        const int64_t p = (tile0 + wave) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / (net.scale * bscale);
    }
}

#define S16S_LDS_DECL()                                                                                  \
    __shared__ __attribute__((aligned(16))) char lds[S16S_T * S16_KU * 2 * 1024 + S16S_T * 4 * 2 * 1024]; \
    s16_lfrag* const sbuf = (s16_lfrag*)(ncw_lchar*)lds;                                                 \
    s16_lfrag* const gsbuf = sbuf + S16S_T * S16_KU * 2 * 64;                                            \
    const int lane = ncw_lane();                                                                         \
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                           \
    const int64_t tile0 = (int64_t)blockIdx.x * S16S_T

__global__ __launch_bounds__(64 * S16_WAVES) void sdf_inferS16_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                     float* __restrict__ sdf) {
    S16S_LDS_DECL();
    NcwSdfStash none = {};
    s16s_value_chain<0>(net, src, n, tile0, sbuf, gsbuf, lane, wave, sdf, none);
}

template <bool TRAIN, int ADJ>  // TRAIN false: forward-only render (no gamma / t_l stash); ADJ 1: adjoint sweep with hi + lo weights, 2: and hi + lo t_l
__global__ __launch_bounds__(64 * S16_WAVES) void sdf_fwdS16_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                   float* __restrict__ sdf, float* __restrict__ grad, NcwSdfStash st) {
    S16S_LDS_DECL();
    s16s_value_chain<(TRAIN ? 2 : 1)>(net, src, n, tile0, sbuf, gsbuf, lane, wave, sdf, st);
    // the plain tail: feature rows from the hi fragments (BS = 2), adjoint sweep in the plain layout over the same LDS
    S16W r;
    f32x16 acc[2][S16S_T];
    s16_prefetch(r, net.w_feat, 16, wave, lane);
    s16_fwd_tail<S16S_T, 2, false, TRAIN, ADJ>(net, src, n, sdf, grad, st, sbuf, gsbuf, r, acc, lane, wave, tile0);
}
#endif  // NCW_HALF_F16

// Tiles per workgroup.  A workgroup costs about (12 + 10.7 T) k cycles per layer (the weight stream from L2 is paid once per
// workgroup whatever T is, the MFMAs per tile: fitted from T = 2 vs 4) and the launch runs
// ceil(workgroups / 256 CUs) rounds: T is chosen to minimise rounds x cost.  At the reference's batch (2048 rays x 24
// samples = 1536 tiles, scripts/train.sh:16-19) T = 3 gives 512 workgroups = exactly two rounds (T = 4: 384 = 1.5 rounds,
// i.e. two), at 1024 rays 256 workgroups = one round; long launches (the grid sweep) take T = 4.
// NCW_SDF16_T = 2 | 3 | 4 forces a value (tests: the small cases never reach T = 3 / 4 on their own).
int s16_tiles_per_wg(int64_t tiles) {
    static const int forced = getenv("NCW_SDF16_T") ? atoi(getenv("NCW_SDF16_T")) : 0;
    if (forced >= 2 && forced <= 4) return forced;
    int best = 2;
    double best_cost = 1e30;
    for (int T = 2; T <= 4; ++T) {
        const int64_t wgs = (tiles + T - 1) / T;
        const double cost = (double)((wgs + 255) / 256) * (12.0 + 10.7 * T);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = T; }
    }
    return best;
}

}  // namespace

#define S16_LAUNCH(KERNEL, ...)                                                                                              \
    do {                                                                                                                     \
        const int64_t tiles = (n + 31) / 32;                                                                                 \
        const int T = s16_tiles_per_wg(tiles);                                                                               \
        if (T == 4)                                                                                                          \
            hipLaunchKernelGGL(KERNEL<4>, dim3((unsigned)((tiles + 3) / 4)), dim3(64 * S16_WAVES), 0, st, __VA_ARGS__);      \
        else if (T == 3)                                                                                                     \
            hipLaunchKernelGGL(KERNEL<3>, dim3((unsigned)((tiles + 2) / 3)), dim3(64 * S16_WAVES), 0, st, __VA_ARGS__);      \
        else                                                                                                                 \
            hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)((tiles + 1) / 2)), dim3(64 * S16_WAVES), 0, st, __VA_ARGS__);      \
        NCW_CHECK_LAUNCH();                                                                                                  \
        return 0;                                                                                                            \
    } while (0)

int NCW_FN(ncw_sdf_infer16_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st) {
    S16_LAUNCH(sdf_infer16_kernel, *net, src, n, sdf);
}

int NCW_FN(ncw_sdf_fwd16_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad,
                                 const NcwSdfStash& stash, hipStream_t st) {
    if (stash.t[0] == nullptr) S16_LAUNCH(sdf_render16_kernel, *net, src, n, sdf, grad, stash);  // forward-only render
    S16_LAUNCH(sdf_fwd16_kernel, *net, src, n, sdf, grad, stash);
}

int NCW_FN(ncw_sdf_bwd16_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, const float* d_sdf, const float* d_grad,
                                 const NcwSdfStash& stash, hipStream_t st) {
    S16_LAUNCH(sdf_bwd16_kernel, *net, src, n, d_sdf, d_grad, stash);
}

#ifdef NCW_HALF_F16
// fp16 build, W = 512, nets that carry residual matrices (NcwSdfNet.w_lo): the split-precision value path
int ncw_sdf_inferS16_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    hipLaunchKernelGGL(sdf_inferS16_kernel, dim3((unsigned)((tiles + S16S_T - 1) / S16S_T)), dim3(64 * S16_WAVES), 0, st, *net, src,
                       n, sdf);
    NCW_CHECK_LAUNCH();
    return 0;
}

int ncw_sdf_fwdS16_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad,
                              const NcwSdfStash& stash, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    bool adj = net->n_layers >= 3;  // every transposed residual present: the adjoint sweep takes hi + lo weights
    for (int l = 0; l < net->n_layers; ++l) adj = adj && net->wt_lo[l] != nullptr;
    const bool render = stash.t[0] == nullptr;  // forward-only render
    const dim3 grid((unsigned)((tiles + S16S_T - 1) / S16S_T)), block(64 * S16_WAVES);
    if (adj && net->adj_mode == 2) {  // ... and t_l as a hi + lo pair
        if (render) hipLaunchKernelGGL((sdf_fwdS16_kernel<false, 2>), grid, block, 0, st, *net, src, n, sdf, grad, stash);
        else hipLaunchKernelGGL((sdf_fwdS16_kernel<true, 2>), grid, block, 0, st, *net, src, n, sdf, grad, stash);
    } else if (adj) {
        if (render) hipLaunchKernelGGL((sdf_fwdS16_kernel<false, 1>), grid, block, 0, st, *net, src, n, sdf, grad, stash);
        else hipLaunchKernelGGL((sdf_fwdS16_kernel<true, 1>), grid, block, 0, st, *net, src, n, sdf, grad, stash);
    } else {
        if (render) hipLaunchKernelGGL((sdf_fwdS16_kernel<false, 0>), grid, block, 0, st, *net, src, n, sdf, grad, stash);
        else hipLaunchKernelGGL((sdf_fwdS16_kernel<true, 0>), grid, block, 0, st, *net, src, n, sdf, grad, stash);
    }
    NCW_CHECK_LAUNCH();
    return 0;
}
#endif
