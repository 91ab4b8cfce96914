// Fused colour-network kernels: RenderingNetwork (models/neuconw.py:59-170) with the appearance
// head, forward and backward, one launch each, activations register-resident (ncw_common.h).
#include <stdlib.h>

#include "ncw_mlp.h"


namespace NCW_NS {

// AUX2 (1 block): [points (3) | normals (3) | 0...]   (neuconw.py:147-148)
NCW_DEV void build_aux2(CVec<1>& aux, const float (&x)[3], const float (&nrm)[3], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = 0.f;
        if (ncw_feat_of(r, 0) < 6) {
            const int f = ncw_feat_of(r, 0) + 4 * h;
            v = f == 0 ? x[0] : f == 1 ? x[1] : f == 2 ? x[2] : f == 3 ? nrm[0] : f == 4 ? nrm[1] : f == 5 ? nrm[2] : 0.f;
        }
        aux.v[0][r] = v;
    }
}

// acc += W in with W as a 16-bit hi + lo pair (w_lo: the residual matrix h16(W - h16(W)), same packed shape; nullptr = hi only):
// two passes of the weight ring over the same activations.  self_bytes = first-chunk bytes of this matrix shape (what the
// call before must have prefetched: the lo pass re-uses it), w_next / next_bytes as mma_stream.
template <bool SPLIT, int RB_IN, int RB_OUT, int K_REAL, int SLOT, class P>
NCW_DEV void mma_stream_split(CVec<RB_OUT>& acc, const Act<P, RB_IN>& in, WRing& ring, const typename P::welem* __restrict__ w_hi,
                              const void* w_lo, int self_bytes, const void* w_next, int next_bytes, int lane) {
    typedef typename P::welem WE;
    if (SPLIT) {  // compile-time: a run-time branch here triples the inlined ring loops and the kernel spills
        mma_stream<RB_IN, RB_OUT, K_REAL, SLOT>(acc, in, ring, w_hi, w_lo, self_bytes, lane);
        mma_stream<RB_IN, RB_OUT, K_REAL, SLOT>(acc, in, ring, (const WE*)w_lo, w_next, next_bytes, lane);
    } else {
        mma_stream<RB_IN, RB_OUT, K_REAL, SLOT>(acc, in, ring, w_hi, w_next, next_bytes, lane);
    }
}

// hi = h16(x) and lo = h16(x - h16(x)) of one block of a C-layout vector as B fragments: an fp16 hi + lo ACTIVATION pair.  Block by
// block, so that the f32 block dies as soon as its two fragments exist (RBF = 16: 256 accumulators next to 2 x 128 fragment registers)
template <class P, int RB>
NCW_DEV void to_act_block_hl(Act<P, RB>& hi, Act<P, RB>& lo, int rb, const f32x16& v) {
#pragma unroll
    for (int t = 0; t < 2; ++t) ncw_split8(v, t, hi.f[2 * rb + t], lo.f[2 * rb + t]);  // (ONE conversion per element: ncw_common.h)
}

// relu_epilogue that also leaves the lo halves of the post-activations (ASPLIT; the stash keeps the single-rounded values)
template <class P, int RB>
NCW_DEV void relu_epilogue_hl(Act<P, RB>& act, Act<P, RB>& act_lo, CVec<RB>& acc, typename P::selem* st, size_t tile, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x16 yv;
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = ncw_relu(acc.v[rb][r]);
        if (st) stash_store_block(st, tile, RB, rb, yv, lane);
        to_act_block_hl<P, RB>(act, act_lo, rb, yv);
    }
}

// acc += W_hi (in + in_lo) + W_lo in: weights AND activations as fp16 hi + lo pairs (round 6, ASPLIT) in TWO passes of the weight ring --
// the W_hi pass feeds every fragment to two MFMAs (the hi and the lo half of the activations), the W_lo pass to one.
// Chaining like mma_stream_split: self_bytes = first-chunk bytes of this matrix shape.
template <int RB_IN, int RB_OUT, int K_REAL, int SLOT, class P>
NCW_DEV void mma_stream_split3(CVec<RB_OUT>& acc, const Act<P, RB_IN>& in, const Act<P, RB_IN>& in_lo, WRing& ring,
                               const typename P::welem* __restrict__ w_hi, const void* w_lo, int self_bytes, const void* w_next,
                               int next_bytes, int lane) {
    typedef typename P::welem WE;
    mma_stream2<RB_IN, RB_OUT, K_REAL, SLOT>(acc, in, in_lo, ring, w_hi, w_lo, self_bytes, lane);
    mma_stream<RB_IN, RB_OUT, K_REAL, SLOT>(acc, in, ring, (const WE*)w_lo, w_next, next_bytes, lane);
}

template <class P, int RBF, int RBH, int RBC, bool WIDE = false>
struct ColShapes {
    // both colour kernels fit 256 registers at d_feature = 256: 2 workgroups / CU; with a 512-wide feature
    // vector (RBF = 16: 256 accumulators for xyz_encoding_final alone) they own the CU like the SDF forward -- and so does the
    // forward with hi + lo ACTIVATIONS at any width (WIDE: 64 more live registers per trunk layer)
    static constexpr int OCC = (RBF >= 16 || WIDE) ? 1 : 2;
    static constexpr int SLOT = RingSlot<RBC, OCC>::bytes;
    static constexpr int FCB_F = ncw_first_chunk_bytes<P, RBF, 32 * RBF, RBF, SLOT>();
    static constexpr int FCB_E0 = ncw_first_chunk_bytes<P, RBF + 3, 32 * RBF + 96, RBH, SLOT>();
    static constexpr int FCB_E = ncw_first_chunk_bytes<P, RBH, 32 * RBH, RBH, SLOT>();
    static constexpr int FCB_L0 = ncw_first_chunk_bytes<P, RBH + 1, 32 * RBH + 6, RBC, SLOT>();
    static constexpr int FCB_L = ncw_first_chunk_bytes<P, RBC, 32 * RBC, RBC, SLOT>();
    static constexpr int FCB_LAST = ncw_first_chunk_bytes<P, RBC, 32 * RBC, 1, SLOT>();
    static constexpr int FCB_TLAST = ncw_first_chunk_bytes<P, 1, 3, RBC, SLOT>();
    static constexpr int FCB_TL0 = ncw_first_chunk_bytes<P, RBC, 32 * RBC, RBH + 1, SLOT>();
    static constexpr int FCB_TE0 = ncw_first_chunk_bytes<P, RBH, 32 * RBH, RBF + 3, SLOT>();
};

// TRAIN = false (color_render_kernel): the forward-only render -- the same arithmetic bit for bit, NOTHING is stashed
// (validation / novel views / vertex colours: rendering/renderer.py:785-916 under no_grad, :951-961 `rgb`).
// SPLIT 1: weights as fp16 hi + lo pairs (+ lin0's [points | normals] inputs); 2 (round 6): the ACTIVATIONS of every layer as pairs
// too (f, the head's and the trunk's post-activations: a third pass W_hi x_lo per layer) -- at the shipped shape (W = 512, 8 + 16
// samples: one sample carries a ray) their rounding was, with the normals, what kept 2 % of the rays above 1e-4 on trained weights
// (profiles/r06/emul_timed_batch_shipped_tangent*.log: 1.2e-4 -> 2e-5 once both are pairs).  Forward only, like SPLIT 1.
template <class P, int RBF, int RBH, int RBC, int SPLIT, bool TRAIN>
NCW_DEV void color_fwd_body(const NcwColorNet& net, const NcwPoints& src, int64_t n, const float* __restrict__ normals,
                            const float* __restrict__ a, const void* __restrict__ feat_stash, float* __restrict__ rgb,
                            const NcwColorStash& st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    auto stp = [](void* p) -> SE* { return TRAIN ? (SE*)p : nullptr; };  // compile-time null: the helpers' `if (st)` folds away
    typedef ColShapes<P, RBF, RBH, RBC, SPLIT == 2> SH;
    constexpr bool AS = SPLIT == 2;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    ring_prologue(ring, net.w_f, SH::FCB_F);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    float xs[3];
    load_point(src, p, xs, ray);
    const float dir[3] = {src.rays_d[ray * 3 + 0], src.rays_d[ray * 3 + 1], src.rays_d[ray * 3 + 2]};
    const float nrm[3] = {normals[p * 3 + 0], normals[p * 3 + 1], normals[p * 3 + 2]};

    Act<P, 3> aux1a;
    {
        CVec<3> aux1;
        build_aux1<Fast<P>::v>(aux1, dir, a + ray * net.n_a, net.n_a, lane);
        if (TRAIN) stash_store<3>((SE*)st.aux1, tile, aux1, lane);
        to_act(aux1a, aux1);
        // per-ray part of the head's first layer evaluated in fp32 by ncw_aux_ray_bias: the AUX1 operand of THIS pass is
        // zero (the stash above keeps the real one for the backward and the weight gradients)
        if (st.aux_bias != nullptr) ncw_act_zero3(aux1a);
    }
    Act<P, 1> aux2a, aux2lo;  // (SPLIT: [points | normals] as an fp16 hi + lo pair, ONE conversion per element: ncw_common.h ncw_split8)
    {
        CVec<1> aux2;
        build_aux2(aux2, xs, nrm, lane);
        if (TRAIN) stash_store<1>((SE*)st.aux2, tile, aux2, lane);
        if constexpr (SPLIT != 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) ncw_split8(aux2.v[0], t, aux2a.f[t], aux2lo.f[t]);
        } else {
            to_act(aux2a, aux2);
        }
    }
    // f = xyz_encoding_final(feat)   (no activation, neuconw.py:128,136)
    Act<P, RBF + 3> cat1;
    Act<P, (AS ? RBF + 3 : 1)> cat1_lo;  // (AS) [lo(f) | 0]
    {
        Act<P, RBF> fin;
#pragma unroll
        for (int rb = 0; rb < RBF; ++rb) {
            f32x16 v;
            stash_load_block(v, (const SE*)feat_stash, tile, RBF, rb, lane);
            to_act_block<RBF>(fin, rb, v);
        }
        CVec<RBF> f;
        load_bias(f, net.b_f, lane);
        mma_stream_split<(SPLIT != 0), RBF, RBF, 32 * RBF, SH::SLOT>(f, fin, ring, (const WE*)net.w_f, net.w_f_lo, SH::FCB_F, net.w_e[0],
                                                                SH::FCB_E0, lane);
        if (TRAIN) stash_store<RBF>((SE*)st.f, tile, f, lane);
        if constexpr (AS) {
            Act<P, RBF> fa, fl;
#pragma unroll
            for (int rb = 0; rb < RBF; ++rb) to_act_block_hl<P, RBF>(fa, fl, rb, f.v[rb]);
            act_concat<RBF, 3>(cat1, fa, aux1a);
            Act<P, 3> z3;
            ncw_act_zero3(z3);
            act_concat<RBF, 3>(cat1_lo, fl, z3);
        } else {
            Act<P, RBF> fa;
            to_act(fa, f);
            act_concat<RBF, 3>(cat1, fa, aux1a);
        }
    }
    // appearance head (neuconw.py:111-127,137-140)
    Act<P, RBH> ea;
    Act<P, (AS ? RBH : 1)> ea_lo;
    {
        CVec<RBH> e;
        load_bias(e, net.b_e[0], lane);
        // + W_e0[:, dir | a] . [gamma_4(d) | a] of this point's ray (fp32, [R][32 RBH])
        if (st.aux_bias != nullptr) ncw_add_ray_bias<RBH>(e, st.aux_bias + ray * (32 * RBH), lane);
        const void* wn = net.n_head > 1 ? net.w_e[1] : net.w_l[0];
        const int nb = net.n_head > 1 ? SH::FCB_E : SH::FCB_L0;
        if constexpr (AS) {
            mma_stream_split3<RBF + 3, RBH, 32 * RBF + 96, SH::SLOT>(e, cat1, cat1_lo, ring, (const WE*)net.w_e[0], net.w_e_lo[0],
                                                                     SH::FCB_E0, wn, nb, lane);
            relu_epilogue_hl<P, RBH>(ea, ea_lo, e, stp(st.e[0]), tile, lane);
        } else {
            mma_stream_split<(SPLIT != 0), RBF + 3, RBH, 32 * RBF + 96, SH::SLOT>(e, cat1, ring, (const WE*)net.w_e[0], net.w_e_lo[0],
                                                                             SH::FCB_E0, wn, nb, lane);
            relu_epilogue<P, RBH>(ea, e, stp(st.e[0]), tile, lane);
        }
        for (int i = 1; i < net.n_head; ++i) {
            load_bias(e, net.b_e[i], lane);
            const void* wn2 = i + 1 < net.n_head ? net.w_e[i + 1] : net.w_l[0];
            const int nb2 = i + 1 < net.n_head ? SH::FCB_E : SH::FCB_L0;
            if constexpr (AS) {
                mma_stream_split3<RBH, RBH, 32 * RBH, SH::SLOT>(e, ea, ea_lo, ring, (const WE*)net.w_e[i], net.w_e_lo[i], SH::FCB_E, wn2,
                                                                nb2, lane);
                relu_epilogue_hl<P, RBH>(ea, ea_lo, e, stp(st.e[i]), tile, lane);
            } else {
                mma_stream_split<(SPLIT != 0), RBH, RBH, 32 * RBH, SH::SLOT>(e, ea, ring, (const WE*)net.w_e[i], net.w_e_lo[i], SH::FCB_E,
                                                                        wn2, nb2, lane);
                relu_epilogue<P, RBH>(ea, e, stp(st.e[i]), tile, lane);
            }
        }
    }
    // trunk (neuconw.py:158-166)
    Act<P, RBC> xa;
    Act<P, (AS ? RBC : 1)> xa_lo;
    CVec<RBC> x;
    const int last = net.n_lin - 1;
    {
        Act<P, RBH + 1> cat2;
        act_concat<RBH, 1>(cat2, ea, aux2a);
        load_bias(x, net.b_l[0], lane);
        if constexpr (SPLIT != 0) {
            // lin0 takes [points | normals | e] (models/neuconw.py:147-148,158): with the weights as hi + lo pairs the remaining coherent
            // term on trained weights was the fp16 rounding of POINTS and NORMALS themselves (scripts/diag/emul_timed_batch.py
            // --candidates: a surface ray at 8.8e-5 -> 1.3e-5).  Third pass of the ring: W_hi . lo([p | n]) -- the e blocks of the
            // operand are zero, the AUX2 block holds h16(v - h16(v)).  Forward only (the stash keeps the single-rounded AUX2).
            const int nxt = 1 == last ? SH::FCB_LAST : SH::FCB_L;
            if constexpr (AS) {  // W_hi against both halves in one pass of the ring, then W_lo
                Act<P, RBH + 1> cat2lo;  // [lo(e) | lo([p | n])]
#pragma unroll
                for (int i = 0; i < 2 * RBH; ++i) cat2lo.f[i] = ea_lo.f[i];
                cat2lo.f[2 * RBH] = aux2lo.f[0];
                cat2lo.f[2 * RBH + 1] = aux2lo.f[1];
                mma_stream_split3<RBH + 1, RBC, 32 * RBH + 6, SH::SLOT>(x, cat2, cat2lo, ring, (const WE*)net.w_l[0], net.w_l_lo[0], SH::FCB_L0,
                                                                        net.w_l[1], nxt, lane);
            } else {  // round 5's three passes (two workgroups per CU: a merged pass's second operand does not fit 256 registers)
                mma_stream<RBH + 1, RBC, 32 * RBH + 6, SH::SLOT>(x, cat2, ring, (const WE*)net.w_l[0], net.w_l_lo[0], SH::FCB_L0, lane);
                mma_stream<RBH + 1, RBC, 32 * RBH + 6, SH::SLOT>(x, cat2, ring, (const WE*)net.w_l_lo[0], net.w_l[0], SH::FCB_L0, lane);
#pragma unroll
                for (int i = 0; i < 2 * RBH; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cat2.f[i][e] = (ncw_h16)0.f;  // in place: the e blocks of the lo operand are zero
                cat2.f[2 * RBH] = aux2lo.f[0];
                cat2.f[2 * RBH + 1] = aux2lo.f[1];
                mma_stream<RBH + 1, RBC, 32 * RBH + 6, SH::SLOT>(x, cat2, ring, (const WE*)net.w_l[0], net.w_l[1], nxt, lane);
            }
        } else {
            mma_stream_split<false, RBH + 1, RBC, 32 * RBH + 6, SH::SLOT>(x, cat2, ring, (const WE*)net.w_l[0], net.w_l_lo[0], SH::FCB_L0,
                                                                          net.w_l[1], 1 == last ? SH::FCB_LAST : SH::FCB_L, lane);
        }
        if constexpr (AS) relu_epilogue_hl<P, RBC>(xa, xa_lo, x, stp(st.x[0]), tile, lane);
        else relu_epilogue<P, RBC>(xa, x, stp(st.x[0]), tile, lane);
    }
    for (int l = 1; l < last; ++l) {
        load_bias(x, net.b_l[l], lane);
        if constexpr (AS) {
            mma_stream_split3<RBC, RBC, 32 * RBC, SH::SLOT>(x, xa, xa_lo, ring, (const WE*)net.w_l[l], net.w_l_lo[l], SH::FCB_L,
                                                            net.w_l[l + 1], l + 1 == last ? SH::FCB_LAST : SH::FCB_L, lane);
            relu_epilogue_hl<P, RBC>(xa, xa_lo, x, stp(st.x[l]), tile, lane);
        } else {
            mma_stream_split<(SPLIT != 0), RBC, RBC, 32 * RBC, SH::SLOT>(x, xa, ring, (const WE*)net.w_l[l], net.w_l_lo[l], SH::FCB_L,
                                                                    net.w_l[l + 1], l + 1 == last ? SH::FCB_LAST : SH::FCB_L, lane);
            relu_epilogue<P, RBC>(xa, x, stp(st.x[l]), tile, lane);
        }
    }
    CVec<1> o;
    load_bias(o, net.b_l[last], lane);
    if constexpr (AS)
        mma_stream_split3<RBC, 1, 32 * RBC, SH::SLOT>(o, xa, xa_lo, ring, (const WE*)net.w_l[last], net.w_l_lo[last], SH::FCB_LAST,
                                                      nullptr, 0, lane);
    else
        mma_stream_split<(SPLIT != 0), RBC, 1, 32 * RBC, SH::SLOT>(o, xa, ring, (const WE*)net.w_l[last], net.w_l_lo[last], SH::FCB_LAST,
                                                              nullptr, 0, lane);
    if (valid && lane < 32) {  // features 0,1,2 <-> registers 0,1,2 of half 0; sigmoid (neuconw.py:168-169)
        rgb[p * 3 + 0] = sigmoidf_<Fast<P>::v>(o.v[0][0]);
        rgb[p * 3 + 1] = sigmoidf_<Fast<P>::v>(o.v[0][1]);
        rgb[p * 3 + 2] = sigmoidf_<Fast<P>::v>(o.v[0][2]);
    }
}

template <class P, int RBF, int RBH, int RBC, int SPLIT = 0>
__global__ __launch_bounds__(64 * NCW_WG_WAVES, ((RBF >= 16 || SPLIT == 2) ? 1 : 2)) void color_fwd_kernel(NcwColorNet net, NcwPoints src, int64_t n,
                                                                      const float* __restrict__ normals,
                                                                      const float* __restrict__ a,
                                                                      const void* __restrict__ feat_stash,
                                                                      float* __restrict__ rgb, NcwColorStash st) {
    color_fwd_body<P, RBF, RBH, RBC, SPLIT, true>(net, src, n, normals, a, feat_stash, rgb, st);
}
template <class P, int RBF, int RBH, int RBC, int SPLIT = 0>
__global__ __launch_bounds__(64 * NCW_WG_WAVES, ((RBF >= 16 || SPLIT == 2) ? 1 : 2)) void color_render_kernel(NcwColorNet net, NcwPoints src, int64_t n,
                                                                         const float* __restrict__ normals,
                                                                         const float* __restrict__ a,
                                                                         const void* __restrict__ feat_stash,
                                                                         float* __restrict__ rgb, NcwColorStash st) {
    color_fwd_body<P, RBF, RBH, RBC, SPLIT, false>(net, src, n, normals, a, feat_stash, rgb, st);
}

template <class P, int RBF, int RBH, int RBC>
__global__ __launch_bounds__(64 * NCW_WG_WAVES, (RBF >= 16 ? 1 : 2)) void color_bwd_kernel(NcwColorNet net, NcwPoints src, int64_t n,
                                                                      const float* __restrict__ rgb,
                                                                      const float* __restrict__ d_rgb,
                                                                      float* __restrict__ d_grad, float* __restrict__ d_a,
                                                                      float* __restrict__ d_a_rows,
                                                                      void* __restrict__ dfeat_stash, NcwColorStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    typedef ColShapes<P, RBF, RBH, RBC> SH;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    const int last = net.n_lin - 1;
    ring_prologue(ring, net.wt_l[last], SH::FCB_TLAST);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    ray = (src.mode == 0) ? p : p / src.per_ray;
    const float vm = valid ? 1.f : 0.f;

    CVec<1> zo;
    cvec_zero(zo);
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float y = rgb[p * 3 + c];
            zo.v[0][c] = d_rgb[p * 3 + c] * y * (1.f - y) * vm;
        }
    }
    stash_store<1>((SE*)st.zo, tile, zo, lane);
    Act<P, 1> zoa;
    to_act(zoa, zo);
    CVec<RBC> u;
    cvec_zero(u);
    mma_stream<1, RBC, 3, SH::SLOT>(u, zoa, ring, (const WE*)net.wt_l[last], net.wt_l[last - 1],
                                    last - 1 == 0 ? SH::FCB_TL0 : SH::FCB_L, lane);
    Act<P, RBC> za;
    for (int l = last - 1; l >= 1; --l) {
        relu_backward<P, RBC>(za, u, (const SE*)st.x[l], (SE*)st.zx[l], tile, lane);
        cvec_zero(u);
        mma_stream<RBC, RBC, 32 * RBC, SH::SLOT>(u, za, ring, (const WE*)net.wt_l[l], net.wt_l[l - 1],
                                                  l - 1 == 0 ? SH::FCB_TL0 : SH::FCB_L, lane);
    }
    CVec<RBH> ue;
    {
        relu_backward<P, RBC>(za, u, (const SE*)st.x[0], (SE*)st.zx[0], tile, lane);
        CVec<RBH + 1> q;
        cvec_zero(q);
        const void* wn = net.n_head > 1 ? net.wt_e[net.n_head - 1] : net.wt_e[0];
        const int nb = net.n_head > 1 ? SH::FCB_E : SH::FCB_TE0;
        mma_stream<RBC, RBH + 1, 32 * RBC, SH::SLOT>(q, za, ring, (const WE*)net.wt_l[0], wn, nb, lane);
#pragma unroll
        for (int rb = 0; rb < RBH; ++rb) ue.v[rb] = q.v[rb];
        // d normals = AUX2 features 3,4,5: f=3 <-> (r=3,h=0); f=4 <-> (r=0,h=1); f=5 <-> (r=1,h=1)
        if (valid) {
            if (lane < 32) d_grad[p * 3 + 0] += q.v[RBH][3];
            else {
                d_grad[p * 3 + 1] += q.v[RBH][0];
                d_grad[p * 3 + 2] += q.v[RBH][1];
            }
        }
    }
    Act<P, RBH> zea;
    for (int i = net.n_head - 1; i >= 1; --i) {
        relu_backward<P, RBH>(zea, ue, (const SE*)st.e[i], (SE*)st.ze[i], tile, lane);
        cvec_zero(ue);
        mma_stream<RBH, RBH, 32 * RBH, SH::SLOT>(ue, zea, ring, (const WE*)net.wt_e[i], net.wt_e[i - 1],
                                                  i - 1 == 0 ? SH::FCB_TE0 : SH::FCB_E, lane);
    }
    {
        relu_backward<P, RBH>(zea, ue, (const SE*)st.e[0], (SE*)st.ze[0], tile, lane);
        CVec<RBF + 3> q;
        cvec_zero(q);
        mma_stream<RBH, RBF + 3, 32 * RBH, SH::SLOT>(q, zea, ring, (const WE*)net.wt_e[0], net.wt_f, SH::FCB_F, lane);
        CVec<3> qa;
        qa.v[0] = q.v[RBF]; qa.v[1] = q.v[RBF + 1]; qa.v[2] = q.v[RBF + 2];
        accumulate_d_a(qa, d_a, ray, net.n_a, valid, lane, d_a_rows, p);
        CVec<RBF> zf;
#pragma unroll
        for (int rb = 0; rb < RBF; ++rb) zf.v[rb] = q.v[rb];
        stash_store<RBF>((SE*)st.zf, tile, zf, lane);
        Act<P, RBF> zfa;
        to_act(zfa, zf);
        CVec<RBF> df;
        cvec_zero(df);
        mma_stream<RBF, RBF, 32 * RBF, SH::SLOT>(df, zfa, ring, (const WE*)net.wt_f, nullptr, 0, lane);
        stash_store<RBF>((SE*)dfeat_stash, tile, df, lane);
    }
}

}  // namespace NCW_NS
using namespace NCW_NS;

static bool color_ok(const NcwColorNet* net) {
    return net && net->n_head >= 1 && net->n_head <= 4 && net->n_lin >= 2 && net->n_lin <= 8 && net->n_a >= 0 &&
           net->n_a <= 69;
}

#define NCW_COLOR_DISPATCH(KERNEL, ...)                                                                          \
    do {                                                                                                         \
        const int key = net->rbf * 10000 + net->rbh * 100 + net->rbc;                                            \
        if (prec == NCW_PREC_F32) {                                                                              \
            if (key == 20102) NCW_LAUNCH_TILES((KERNEL<PrecF32, 2, 1, 2>), n, st, __VA_ARGS__);                  \
            else if (key == 20408) NCW_LAUNCH_TILES((KERNEL<PrecF32, 2, 4, 8>), n, st, __VA_ARGS__);             \
            else if (key == 80408) NCW_LAUNCH_TILES((KERNEL<PrecF32, 8, 4, 8>), n, st, __VA_ARGS__);             \
            else if (key == 160408) NCW_LAUNCH_TILES((KERNEL<PrecF32, 16, 4, 8>), n, st, __VA_ARGS__);           \
            else return NCW_E_UNSUPPORTED;                                                                       \
        } else {                                                                                                 \
            if (key == 20102) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 2, 1, 2>), n, st, __VA_ARGS__);                 \
            else if (key == 20408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 2, 4, 8>), n, st, __VA_ARGS__);            \
            else if (key == 80408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 8, 4, 8>), n, st, __VA_ARGS__);            \
            else if (key == 160408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 16, 4, 8>), n, st, __VA_ARGS__);          \
            else return NCW_E_UNSUPPORTED;                                                                       \
        }                                                                                                        \
    } while (0)

#ifndef NCW_HALF_F16
extern "C" int ncw_color_fwd_f16(const NcwColorNet*, int, const NcwPoints*, int64_t, const float*, const float*, const void*, float*,
                                 const NcwColorStash*, void*);
extern "C" int ncw_color_bwd_f16(const NcwColorNet*, int, const NcwPoints*, int64_t, const float*, const float*, float*, float*, float*,
                                 void*, const NcwColorStash*, void*);
#endif

extern "C" int NCW_FN(ncw_color_fwd)(const NcwColorNet* net, int prec, const NcwPoints* pts, int64_t n, const float* normals,
                                     const float* a, const void* feat_stash, float* rgb, const NcwColorStash* stash,
                                     void* stream) {
    NCW_FORWARD_F16(prec, ncw_color_fwd_f16(net, NCW_PREC_BF16, pts, n, normals, a, feat_stash, rgb, stash, stream));
    if (!color_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (pts->mode == 4) return NCW_E_UNSUPPORTED;  // point selections: background NeRF kernels only
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool render = stash->aux1 == nullptr;  // forward-only render: nothing is stashed (include/neuconw_hip.h, NcwColorStash)
    if (net->w_f_lo != nullptr) {  // forward matrices as hi + lo pairs (NcwColorNet.w_*_lo): the 16-bit widths that ship
        if (prec == NCW_PREC_F32) return NCW_E_BADARG;
        const int key = net->rbf * 10000 + net->rbh * 100 + net->rbc;
        for (int i = 0; i < net->n_head; ++i) if (net->w_e_lo[i] == nullptr) return NCW_E_BADARG;
        for (int l = 0; l < net->n_lin; ++l) if (net->w_l_lo[l] == nullptr) return NCW_E_BADARG;
#define NCW_COLOR_SPLIT_DISPATCH(KERNEL, S)                                                                                               \
        do {                                                                                                                                    \
            if (key == 20102) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 2, 1, 2, S>), n, st, *net, *pts, n, normals, a, feat_stash, rgb, *stash);        \
            else if (key == 20408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 2, 4, 8, S>), n, st, *net, *pts, n, normals, a, feat_stash, rgb, *stash);   \
            else if (key == 80408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 8, 4, 8, S>), n, st, *net, *pts, n, normals, a, feat_stash, rgb, *stash);   \
            else if (key == 160408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 16, 4, 8, S>), n, st, *net, *pts, n, normals, a, feat_stash, rgb, *stash); \
            else return NCW_E_UNSUPPORTED;                                                                                                      \
        } while (0)
        if (net->act_split) {  // the activations as hi + lo pairs too (the two widths that ship: d_feature 256 / 512)
#define NCW_COLOR_ASPLIT_DISPATCH(KERNEL)                                                                                                  \
        do {                                                                                                                                    \
            if (key == 80408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 8, 4, 8, 2>), n, st, *net, *pts, n, normals, a, feat_stash, rgb, *stash);        \
            else if (key == 160408) NCW_LAUNCH_TILES((KERNEL<PrecBF16, 16, 4, 8, 2>), n, st, *net, *pts, n, normals, a, feat_stash, rgb, *stash); \
            else return NCW_E_UNSUPPORTED;                                                                                                      \
        } while (0)
            if (render) NCW_COLOR_ASPLIT_DISPATCH(color_render_kernel);
            else NCW_COLOR_ASPLIT_DISPATCH(color_fwd_kernel);
            return 0;
        }
        if (render) NCW_COLOR_SPLIT_DISPATCH(color_render_kernel, 1);
        else NCW_COLOR_SPLIT_DISPATCH(color_fwd_kernel, 1);
        return 0;
    }
    if (render) NCW_COLOR_DISPATCH(color_render_kernel, *net, *pts, n, normals, a, feat_stash, rgb, *stash);
    else NCW_COLOR_DISPATCH(color_fwd_kernel, *net, *pts, n, normals, a, feat_stash, rgb, *stash);
    return 0;
}

extern "C" int NCW_FN(ncw_color_bwd)(const NcwColorNet* net, int prec, const NcwPoints* pts, int64_t n, const float* rgb,
                                     const float* d_rgb, float* d_grad, float* d_a, float* d_a_rows, void* dfeat_stash,
                                     const NcwColorStash* stash, void* stream) {
    NCW_FORWARD_F16(prec, ncw_color_bwd_f16(net, NCW_PREC_BF16, pts, n, rgb, d_rgb, d_grad, d_a, d_a_rows, dfeat_stash, stash, stream));
    if (!color_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (pts->mode == 4) return NCW_E_UNSUPPORTED;  // point selections: background NeRF kernels only
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_COLOR_DISPATCH(color_bwd_kernel, *net, *pts, n, rgb, d_rgb, d_grad, d_a, d_a_rows, dfeat_stash, *stash);
    return 0;
}
