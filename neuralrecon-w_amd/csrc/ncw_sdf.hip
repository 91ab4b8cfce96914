// Fused SDF-network kernels (models/neuconw.py:183-296) for gfx950.
//   sdf_infer : x -> sdf                       (SDFNetwork.sdf :281-282; sampler / octree / mesh grid)
//   sdf_fwd   : x -> sdf, feat, d sdf/dx       (forward :263-279 + gradient :284-296 as the analytic
//                                               adjoint pass of SURVEY 8a-2) + activation stash
//   sdf_bwd   : first- and second-order backward of the above (what create_graph=True makes autograd
//               compute in the reference), emitting the operands of the weight-gradient GEMMs.
// One wave = 32 points, activations register-resident in MFMA C layout (ncw_common.h); every layer
// of the network runs inside ONE launch; inter-layer activations never round-trip HBM (only the
// stash that the backward needs is written, once, coalesced).
#include <stdlib.h>

#include "ncw_mlp.h"

NCW_DEV NcwPoints points_from_x(const float* x) {
    NcwPoints s;
    s.x = x; s.rays_o = nullptr; s.rays_d = nullptr; s.z = nullptr; s.sample_dist = nullptr;
    s.per_ray = 1; s.mode = 0;
    return s;
}

// Shapes of the packed SDF matrices as seen by the weight ring (first-chunk sizes for prefetch).
namespace NCW_NS {

template <class P, int RB, int OCC = 1>
struct SdfShapes {
    static constexpr int SLOT = RingSlot<RB, OCC>::bytes;
    static constexpr int FCB_W0 = ncw_first_chunk_bytes<P, 2, 39, RB, SLOT>();              // lin0: K = gamma
    static constexpr int FCB_WH = ncw_first_chunk_bytes<P, RB, 32 * RB, RB, SLOT>();        // hidden
    static constexpr int FCB_WS = ncw_first_chunk_bytes<P, RB + 2, 32 * RB + 39, RB, SLOT>();  // skip layer
    static constexpr int FCB_W1 = ncw_first_chunk_bytes<P, RB, 32 * RB, 1, SLOT>();         // sdf row
    static constexpr int FCB_TL = ncw_first_chunk_bytes<P, 1, 1, RB, SLOT>();               // wt[L-1]
    static constexpr int FCB_TS = ncw_first_chunk_bytes<P, RB, 32 * RB, RB + 2, SLOT>();    // wt[skip]
    static constexpr int FCB_T0 = ncw_first_chunk_bytes<P, RB, 32 * RB, 2, SLOT>();         // wt[0]
};

// Softplus'(z_l) recovered from the stashed activation h_{l+1} = Softplus(z_l):  s = 1 - exp(-100 h)
// (exactly sigmoid(100 z); 1 above torch's threshold to f32 rounding).  Saves a whole stash vector per
// layer (write in the forward, reads in the adjoint pass and in the backward).
template <class P, class SE>
NCW_DEV void load_sprime_block(f32x16& sv, const SE* __restrict__ st_h, size_t tile, int RB, int rb, int lane) {
    stash_load_block(sv, st_h, tile, RB, rb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (Fast<P>::v) sv[r] = 1.f - __builtin_amdgcn_exp2f(sv[r] * -144.26950408889634f);
        else sv[r] = -expm1f(-100.f * sv[r]);
    }
}

// z_l = b_l + W_l u_l for a hidden layer l (1 <= l <= L-2), honouring the skip concatenation
template <class P, int RB, int SLOT, class GammaFn>
NCW_DEV void sdf_hidden_layer(CVec<RB>& acc, const Act<P, RB>& act, GammaFn&& gamma_fn, const NcwSdfNet& net, int l,
                              WRing& ring, const void* w_next, int next_bytes, int lane) {
    typedef typename P::welem WE;
    load_bias(acc, net.b[l], lane);
    if (l == net.skip_layer) {
        // gamma is re-materialised here (recomputed / reloaded from the stash) instead of being kept
        // live across all layers: 16-32 registers less on the critical allocation
        Act<P, 2> gact;
        gamma_fn(gact);
        CatB<P, RB, 2> cat(act, gact);  // [h | gamma] without a copy
        mma_stream_b<RB, 32 * RB + 39, SLOT, RB, P>(acc, cat, ring, (const WE*)net.w[l], w_next, next_bytes, lane);
    } else {
        mma_stream<RB, RB, 32 * RB, SLOT>(acc, act, ring, (const WE*)net.w[l], w_next, next_bytes, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// inference
// ---------------------------------------------------------------------------------------------
template <class P, int RB>
__global__ __launch_bounds__(64 * NCW_WG_WAVES) void sdf_infer_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                      float* __restrict__ sdf) {
    typedef typename P::welem WE;
    typedef SdfShapes<P, RB> SH;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    const int L = net.n_layers;
    ring_prologue(ring, net.w[0], SH::FCB_W0);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;

    auto make_gamma = [&](Act<P, 2>& g) {
        CVec<2> gam;
        freq_encode<2, 3, 6, Fast<P>::v>(gam, xs, lane);
        to_act(g, gam);
    };
    CVec<RB> acc;
    Act<P, RB> act;
    auto next_of = [&](int l, const void*& w, int& bytes) {  // matrix consumed after forward layer l
        const int m = l + 1;
        w = net.w[m];
        bytes = (m == L - 1) ? SH::FCB_W1 : (m == net.skip_layer ? SH::FCB_WS : SH::FCB_WH);
    };
    const void* wn;
    int nbts;
    load_bias(acc, net.b[0], lane);
    next_of(0, wn, nbts);
    {
        Act<P, 2> gact;
        make_gamma(gact);
        mma_stream<2, RB, 39, SH::SLOT>(acc, gact, ring, (const WE*)net.w[0], wn, nbts, lane);
    }
    softplus_epilogue<P, RB>(act, acc, nullptr, nullptr, 0, lane);
    for (int l = 1; l < L - 1; ++l) {
        next_of(l, wn, nbts);
        sdf_hidden_layer<P, RB, SH::SLOT>(acc, act, make_gamma, net, l, ring, wn, nbts, lane);
        softplus_epilogue<P, RB>(act, acc, nullptr, nullptr, 0, lane);
    }
    CVec<1> o;
    load_bias(o, net.b[L - 1], lane);
    mma_stream<RB, 1, 32 * RB, SH::SLOT>(o, act, ring, (const WE*)net.w[L - 1], nullptr, 0, lane);
    if (valid && lane < 32) sdf[p] = o.v[0][0] / net.scale;
}

// ---------------------------------------------------------------------------------------------
// forward + analytic input gradient + stash
// ---------------------------------------------------------------------------------------------
// TRAIN = false (sdf_render_kernel): the forward-only render -- the same arithmetic bit for bit; of the stash only h_l (the
// adjoint sweep's scratch) and feat (the colour network's input) are written: no gamma (recomputed for the skip layer: the
// same 16-bit / f32 image the stash would have returned), no t_l.
template <class P, int RB, bool TRAIN>
NCW_DEV void sdf_fwd_body(const NcwSdfNet& net, const NcwPoints& src, int64_t n, float* __restrict__ sdf, float* __restrict__ grad,
                          const NcwSdfStash& st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    typedef SdfShapes<P, RB> SH;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    const int L = net.n_layers;
    ring_prologue(ring, net.w[0], SH::FCB_W0);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;

    auto reload_gamma = [&](Act<P, 2>& g) {  // gamma lives in the stash between layer 0 and the skip layer
        CVec<2> gm;
        if (TRAIN) stash_load<2>(gm, (const SE*)st.gamma, tile, lane);
        else freq_encode<2, 3, 6, Fast<P>::v>(gm, xs, lane);
        to_act(g, gm);
    };
    CVec<RB> acc;
    Act<P, RB> act;
    auto next_of = [&](int l, const void*& w, int& bytes) {
        const int m = l + 1;
        w = net.w[m];
        bytes = (m == L - 1) ? SH::FCB_W1 : (m == net.skip_layer ? SH::FCB_WS : SH::FCB_WH);
    };
    const void* wn;
    int nbts;
    load_bias(acc, net.b[0], lane);
    next_of(0, wn, nbts);
    {
        CVec<2> gam;
        freq_encode<2, 3, 6, Fast<P>::v>(gam, xs, lane);
        if (TRAIN) stash_store<2>((SE*)st.gamma, tile, gam, lane);
        Act<P, 2> gact;
        to_act(gact, gam);
        mma_stream<2, RB, 39, SH::SLOT>(acc, gact, ring, (const WE*)net.w[0], wn, nbts, lane);
    }
    softplus_epilogue<P, RB>(act, acc, (SE*)st.h[1], nullptr, tile, lane);
    for (int l = 1; l < L - 1; ++l) {
        next_of(l, wn, nbts);
        sdf_hidden_layer<P, RB, SH::SLOT>(acc, act, reload_gamma, net, l, ring, wn, nbts, lane);
        softplus_epilogue<P, RB>(act, acc, (SE*)st.h[l + 1], nullptr, tile, lane);
    }
    {
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        mma_stream<RB, 1, 32 * RB, SH::SLOT>(o, act, ring, (const WE*)net.w[L - 1], net.w_feat, SH::FCB_WH, lane);
        if (valid && lane < 32) sdf[p] = o.v[0][0] / net.scale;
        load_bias(acc, net.b_feat, lane);
        mma_stream<RB, RB, 32 * RB, SH::SLOT>(acc, act, ring, (const WE*)net.w_feat, net.wt[L - 1], SH::FCB_TL, lane);
        stash_store<RB>((SE*)st.feat, tile, acc, lane);
    }
    // ---- adjoint pass: a_{L-2} = W_{L-1}[0,:]; t_l = a_l * s_l; a_{l-1} = W_l^T t_l --------------
    auto adj_next = [&](int l, const void*& w, int& bytes) {  // matrix consumed after adjoint layer l
        const int m = l - 1;
        if (m < 0) { w = nullptr; bytes = 0; return; }
        w = net.wt[m];
        bytes = (m == 0) ? SH::FCB_T0 : (m == net.skip_layer ? SH::FCB_TS : SH::FCB_WH);
    };
    CVec<RB> a;
    {
        CVec<1> e0;
        cvec_zero(e0);
        e0.v[0][0] = (lane < 32) ? 1.f : 0.f;  // feature 0 <-> (r=0, h=0)
        Act<P, 1> e0a;
        to_act(e0a, e0);
        cvec_zero(a);
        adj_next(L - 1, wn, nbts);
        mma_stream<1, RB, 1, SH::SLOT>(a, e0a, ring, (const WE*)net.wt[L - 1], wn, nbts, lane);
    }
    CVec<2> gg;
    cvec_zero(gg);
    for (int l = L - 2; l >= 0; --l) {
        Act<P, RB> ta;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x16 sv;
            load_sprime_block<P>(sv, (const SE*)st.h[l + 1], tile, RB, rb, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] *= a.v[rb][r];
            if (TRAIN) stash_store_block((SE*)st.t[l], tile, RB, rb, sv, lane);
            to_act_block<RB>(ta, rb, sv);
        }
        const WE* wt = (const WE*)net.wt[l];
        adj_next(l, wn, nbts);
        if (l == 0) {
            mma_stream<RB, 2, 32 * RB, SH::SLOT>(gg, ta, ring, wt, wn, nbts, lane);
        } else if (l == net.skip_layer) {
            CVec<RB + 2> q;
            cvec_zero(q);
            mma_stream<RB, RB + 2, 32 * RB, SH::SLOT>(q, ta, ring, wt, wn, nbts, lane);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a.v[rb] = q.v[rb];
            gg.v[0] = q.v[RB];
            gg.v[1] = q.v[RB + 1];
        } else {
            cvec_zero(a);
            mma_stream<RB, RB, 32 * RB, SH::SLOT>(a, ta, ring, wt, wn, nbts, lane);
        }
    }
    // ---- grad = J_gamma(x)^T g_gamma -----------------------------------------------------------
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int h = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * rb + ncw_feat_of(r, 0) >= 39) continue;
            int comp;
            const float dv = freq_feature_deriv<3, 6, Fast<P>::v>(xs, 32 * rb + ncw_feat_of(r, 0) + 4 * h, comp);
            const float c = gg.v[rb][r] * dv;
            nx += comp == 0 ? c : 0.f;
            ny += comp == 1 ? c : 0.f;
            nz += comp == 2 ? c : 0.f;
        }
    nx = half_pair_sum(nx); ny = half_pair_sum(ny); nz = half_pair_sum(nz);
    if (valid && lane < 32) {
        grad[p * 3 + 0] = nx; grad[p * 3 + 1] = ny; grad[p * 3 + 2] = nz;
    }
}

template <class P, int RB>
__global__ __launch_bounds__(64 * NCW_WG_WAVES) void sdf_fwd_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                    float* __restrict__ sdf, float* __restrict__ grad,
                                                                    NcwSdfStash st) {
    sdf_fwd_body<P, RB, true>(net, src, n, sdf, grad, st);
}
template <class P, int RB>
__global__ __launch_bounds__(64 * NCW_WG_WAVES) void sdf_render_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                       float* __restrict__ sdf, float* __restrict__ grad,
                                                                       NcwSdfStash st) {
    sdf_fwd_body<P, RB, false>(net, src, n, sdf, grad, st);
}

// ---------------------------------------------------------------------------------------------
// backward (second order)
// ---------------------------------------------------------------------------------------------
template <class P, int RB>
// W = 512 (RB = 16): 256 accumulator registers alone -- one workgroup per CU like the forward
__global__ __launch_bounds__(64 * NCW_WG_WAVES, (RB >= 16 ? 1 : 2)) void sdf_bwd_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                    const float* __restrict__ d_sdf,
                                                                    const float* __restrict__ d_grad, NcwSdfStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    typedef SdfShapes<P, RB, (RB >= 16 ? 1 : 2)> SH;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    const int h = lane >> 5;
    const int L = net.n_layers;
    ring_prologue(ring, net.w[0], SH::FCB_W0);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
    const float vmask = valid ? 1.f : 0.f;  // padded lanes must contribute nothing to weight grads
    float nb[3] = {d_grad[p * 3 + 0] * vmask, d_grad[p * 3 + 1] * vmask, d_grad[p * 3 + 2] * vmask};
    const float dsdf = d_sdf[p] * vmask / net.scale;

    // ---- (1) backward of the adjoint pass, l = 0 .. L-2 ------------------------------------------
    CVec<2> q0;  // qbar_0 = J_gamma nbar
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * rb + ncw_feat_of(r, 0) >= 39) {
                q0.v[rb][r] = 0.f;
                continue;
            }
            int comp;
            const float dv = freq_feature_deriv<3, 6, Fast<P>::v>(xs, 32 * rb + ncw_feat_of(r, 0) + 4 * h, comp);
            q0.v[rb][r] = dv * (comp == 0 ? nb[0] : (comp == 1 ? nb[1] : nb[2]));
        }
    stash_store<2>((SE*)st.qbar[0], tile, q0, lane);
    Act<P, 2> q0a;
    to_act(q0a, q0);
    Act<P, RB> qa;
    CVec<RB> tb;
    const void* wn;
    int nbts;
    for (int l = 0; l <= L - 2; ++l) {
        cvec_zero(tb);
        const WE* w = (const WE*)net.w[l];
        if (l + 1 <= L - 2) {
            wn = net.w[l + 1];
            nbts = (l + 1 == net.skip_layer) ? SH::FCB_WS : SH::FCB_WH;
        } else {
            wn = net.wt_feat;
            nbts = SH::FCB_WH;
        }
        if (l == 0) {
            mma_stream<2, RB, 39, SH::SLOT>(tb, q0a, ring, w, wn, nbts, lane);
        } else if (l == net.skip_layer) {
            Act<P, RB + 2> cat;
            act_concat<RB, 2>(cat, qa, q0a);
            mma_stream<RB + 2, RB, 32 * RB + 39, SH::SLOT>(tb, cat, ring, w, wn, nbts, lane);
        } else {
            mma_stream<RB, RB, 32 * RB, SH::SLOT>(tb, qa, ring, w, wn, nbts, lane);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x16 sv, tv, z2, ab;
            load_sprime_block<P>(sv, (const SE*)st.h[l + 1], tile, RB, rb, lane);
            stash_load_block(tv, (const SE*)st.t[l], tile, RB, rb, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tbar = tb.v[rb][r];
                // a_l phi''(z_l) = 100 t_l (1 - s_l)   (0 above the Softplus threshold where s == 1)
                z2[r] = tbar * 100.f * tv[r] * (1.f - sv[r]);
                ab[r] = tbar * sv[r];  // abar_l
            }
            stash_store_block_keep((SE*)st.zbar[l], tile, RB, rb, z2, lane);  // temporarily zbar2_l (re-read by pass 2)
            stash_store_block((SE*)st.qbar[l + 1], tile, RB, rb, ab, lane);  // qbar_{l+1} = abar_l (read by wgrad only)
            to_act_block<RB>(qa, rb, ab);
        }
    }
    // ---- (2) backward of the forward pass, l = L-1 .. 0 -------------------------------------------
    CVec<RB> u;
    {
        CVec<1> zs;
        cvec_zero(zs);
        zs.v[0][0] = (lane < 32) ? dsdf : 0.f;
        stash_store<1>((SE*)st.zsdf, tile, zs, lane);
        CVec<1> one;
        cvec_zero(one);
        one.v[0][0] = (lane < 32) ? vmask : 0.f;
        stash_store<1>((SE*)st.one, tile, one, lane);
        Act<P, 1> zsa;
        to_act(zsa, zs);
        Act<P, RB> dfa;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x16 df;
            stash_load_block(df, (const SE*)st.dfeat, tile, RB, rb, lane);
            to_act_block<RB>(dfa, rb, df);
        }
        cvec_zero(u);
        mma_stream<RB, RB, 32 * RB, SH::SLOT>(u, dfa, ring, (const WE*)net.wt_feat, net.wt[L - 1], SH::FCB_TL, lane);
        const int m = L - 2;  // next: wt[L-2] unless the loop below needs no matrix at all (L-2 == 0)
        wn = (m > 0) ? net.wt[m] : nullptr;
        nbts = (m == net.skip_layer) ? SH::FCB_TS : SH::FCB_WH;
        mma_stream<1, RB, 1, SH::SLOT>(u, zsa, ring, (const WE*)net.wt[L - 1], wn, nbts, lane);
    }
    for (int l = L - 2; l >= 0; --l) {
        Act<P, RB> za;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x16 sv, z2;
            load_sprime_block<P>(sv, (const SE*)st.h[l + 1], tile, RB, rb, lane);
            stash_load_block(z2, (const SE*)st.zbar[l], tile, RB, rb, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) z2[r] = u.v[rb][r] * sv[r] + z2[r];
            stash_store_block((SE*)st.zbar[l], tile, RB, rb, z2, lane);  // final zbar_l (read by wgrad only)
            to_act_block<RB>(za, rb, z2);
        }
        if (l > 0) {
            cvec_zero(u);
            wn = (l - 1 > 0) ? net.wt[l - 1] : nullptr;
            nbts = (l - 1 == net.skip_layer) ? SH::FCB_TS : SH::FCB_WH;
            if (l == net.skip_layer) mma_stream<RB, RB, 32 * RB, SH::SLOT, RB + 2>(u, za, ring, (const WE*)net.wt[l], wn, nbts, lane);
            else mma_stream<RB, RB, 32 * RB, SH::SLOT>(u, za, ring, (const WE*)net.wt[l], wn, nbts, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
}  // namespace NCW_NS
using namespace NCW_NS;

static bool sdf_net_ok(const NcwSdfNet* net) {
    return net && net->multires == 6 && net->n_layers >= 2 && net->n_layers <= NCW_MAX_LAYERS;
}

#define NCW_SDF_DISPATCH(KERNEL, ...)                                                                   \
    do {                                                                                                \
        if (net->rb == 2) {                                                                             \
            if (prec == NCW_PREC_F32) NCW_LAUNCH_TILES((KERNEL<PrecF32, 2>), n, st, __VA_ARGS__);       \
            else NCW_LAUNCH_TILES((KERNEL<PrecBF16, 2>), n, st, __VA_ARGS__);                           \
        } else if (net->rb == 8) {                                                                      \
            if (prec == NCW_PREC_F32) NCW_LAUNCH_TILES((KERNEL<PrecF32, 8>), n, st, __VA_ARGS__);       \
            else NCW_LAUNCH_TILES((KERNEL<PrecBF16, 8>), n, st, __VA_ARGS__);                           \
        } else if (net->rb == 16) {                                                                     \
            if (prec == NCW_PREC_F32) NCW_LAUNCH_TILES((KERNEL<PrecF32, 16>), n, st, __VA_ARGS__);      \
            else NCW_LAUNCH_TILES((KERNEL<PrecBF16, 16>), n, st, __VA_ARGS__);                          \
        } else return NCW_E_UNSUPPORTED;                                                                \
    } while (0)

int NCW_FN(ncw_sdf_fwd8_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad, const NcwSdfStash& stash,
                        hipStream_t st);  // ncw_sdf8.hip

int NCW_FN(ncw_sdf_inferC_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st);  // ncw_pp.hip

// W = 512, 16-bit: weights streamed from L2, activations in LDS (ncw_sdf16.hip)
int NCW_FN(ncw_sdf_infer16_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st);
int NCW_FN(ncw_sdf_fwd16_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad,
                                 const NcwSdfStash& stash, hipStream_t st);
int NCW_FN(ncw_sdf_bwd16_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, const float* d_sdf, const float* d_grad,
                                 const NcwSdfStash& stash, hipStream_t st);
// W = 512, exact fp32 (the parity mode at the shipped width): the same structure with f32 MFMAs (ncw_sdf16f.hip)
#ifndef NCW_HALF_F16
int ncw_sdf_infer16f_launch(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st);
int ncw_sdf_fwd16f_launch(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad, const NcwSdfStash& stash,
                          hipStream_t st);
int ncw_sdf_bwd16f_launch(const NcwSdfNet* net, const NcwPoints& src, int64_t n, const float* d_sdf, const float* d_grad,
                          const NcwSdfStash& stash, hipStream_t st);
static bool sdf16f_on(const NcwSdfNet* net, int prec) { return net->rb == 16 && prec == NCW_PREC_F32 && net->n_layers >= 3; }
#else
static bool sdf16f_on(const NcwSdfNet*, int) { return false; }
#endif

static bool sdf16_on(const NcwSdfNet* net, int prec) {
    return net->rb == 16 && prec == NCW_PREC_BF16 && net->n_layers >= 3;
}

// fp16 build, W = 256: the split-precision value path of ncw_split.hip whenever the net carries residual matrices (w_lo)
#ifdef NCW_HALF_F16
int ncw_sdf_inferS_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st);
int ncw_sdf_fwdS_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad,
                            const NcwSdfStash& stash, hipStream_t st);
int ncw_sdf_inferS16_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st);  // ncw_sdf16.hip
int ncw_sdf_fwdS16_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad,
                              const NcwSdfStash& stash, hipStream_t st);
static bool sdf_split_on(const NcwSdfNet* net, int prec) {
    if (!((net->rb == 8 || net->rb == 16) && prec == NCW_PREC_BF16 && net->n_layers >= 3 && net->n_layers <= NCW_MAX_LAYERS)) return false;
    for (int l = 0; l < net->n_layers; ++l)
        if (net->w_lo[l] == nullptr) return false;
    return true;
}
#endif

static int sdf_infer_any(const NcwSdfNet* net, int prec, const NcwPoints& src, int64_t n, float* sdf, void* stream) {
    if (!sdf_net_ok(net) || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (src.mode == 4) return NCW_E_UNSUPPORTED;  // point selections: background NeRF kernels only
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#ifdef NCW_HALF_F16
    if (sdf_split_on(net, prec))
        return net->rb == 16 ? ncw_sdf_inferS16_launch_f16(net, src, n, sdf, st) : ncw_sdf_inferS_launch_f16(net, src, n, sdf, st);
#endif
    if (sdf16_on(net, prec)) return NCW_FN(ncw_sdf_infer16_launch)(net, src, n, sdf, st);
#ifndef NCW_HALF_F16
    if (sdf16f_on(net, prec)) return ncw_sdf_infer16f_launch(net, src, n, sdf, st);
#endif
    // W = 256, 16-bit: the fine-interleaved kernel of ncw_pp.hip (0.165 ms per 131,072 points; round 2's burst kernel 0.198,
    // the weights-through-LDS kernel below -- fp32 and the other widths -- 0.26)
    if (net->rb == 8 && prec == NCW_PREC_BF16 && net->n_layers >= 3 && net->n_layers <= 12)
        return NCW_FN(ncw_sdf_inferC_launch)(net, src, n, sdf, st);
    NCW_SDF_DISPATCH(sdf_infer_kernel, *net, src, n, sdf);
    return 0;
}

#ifndef NCW_HALF_F16
extern "C" int ncw_sdf_infer_f16(const NcwSdfNet*, int, const float*, int64_t, float*, void*);
extern "C" int ncw_sdf_infer_points_f16(const NcwSdfNet*, int, const NcwPoints*, int64_t, float*, void*);
extern "C" int ncw_sdf_infer_rays_f16(const NcwSdfNet*, int, const float*, const float*, const float*, int, int, float*, void*);
extern "C" int ncw_sdf_fwd_f16(const NcwSdfNet*, int, const NcwPoints*, int64_t, float*, float*, const NcwSdfStash*, void*);
extern "C" int ncw_sdf_bwd_f16(const NcwSdfNet*, int, const NcwPoints*, int64_t, const float*, const float*, const NcwSdfStash*, void*);
#endif

extern "C" int NCW_FN(ncw_sdf_infer)(const NcwSdfNet* net, int prec, const float* x, int64_t n, float* sdf, void* stream) {
    NCW_FORWARD_F16(prec, ncw_sdf_infer_f16(net, NCW_PREC_BF16, x, n, sdf, stream));
    NcwPoints src = {};
    src.x = x; src.rays_o = nullptr; src.rays_d = nullptr; src.z = nullptr; src.sample_dist = nullptr;
    src.per_ray = 1; src.mode = 0;
    return sdf_infer_any(net, prec, src, n, sdf, stream);
}

extern "C" int NCW_FN(ncw_sdf_infer_points)(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t count, float* sdf,
                                            void* stream) {
    NCW_FORWARD_F16(prec, ncw_sdf_infer_points_f16(net, NCW_PREC_BF16, pts, count, sdf, stream));
    if (!pts) return NCW_E_BADARG;
    return sdf_infer_any(net, prec, *pts, count, sdf, stream);
}

extern "C" int NCW_FN(ncw_sdf_infer_rays)(const NcwSdfNet* net, int prec, const float* rays_o, const float* rays_d,
                                          const float* z, int R, int n, float* sdf, void* stream) {
    NCW_FORWARD_F16(prec, ncw_sdf_infer_rays_f16(net, NCW_PREC_BF16, rays_o, rays_d, z, R, n, sdf, stream));
    if (R < 0 || n <= 0) return NCW_E_BADARG;
    NcwPoints src = {};
    src.x = nullptr; src.rays_o = rays_o; src.rays_d = rays_d; src.z = z; src.sample_dist = nullptr;
    src.per_ray = n; src.mode = 1;
    return sdf_infer_any(net, prec, src, (int64_t)R * n, sdf, stream);
}

extern "C" int NCW_FN(ncw_sdf_fwd)(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t n, float* sdf, float* grad,
                                   const NcwSdfStash* stash, void* stream) {
    NCW_FORWARD_F16(prec, ncw_sdf_fwd_f16(net, NCW_PREC_BF16, pts, n, sdf, grad, stash, stream));
    if (!sdf_net_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (pts->mode == 4) return NCW_E_UNSUPPORTED;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#ifdef NCW_HALF_F16
    if (sdf_split_on(net, prec))
        return net->rb == 16 ? ncw_sdf_fwdS16_launch_f16(net, *pts, n, sdf, grad, *stash, st)
                             : ncw_sdf_fwdS_launch_f16(net, *pts, n, sdf, grad, *stash, st);
#endif
    // W = 256, 16-bit: the weights-stationary kernel of ncw_sdf8.hip (0.57 vs 0.67 ms per 131,072 points for the
    // weights-through-LDS kernel below, which serves fp32 and the other widths)
    if (net->rb == 8 && prec == NCW_PREC_BF16 && net->n_layers >= 3)
        return NCW_FN(ncw_sdf_fwd8_launch)(net, *pts, n, sdf, grad, *stash, st);
    if (sdf16_on(net, prec)) return NCW_FN(ncw_sdf_fwd16_launch)(net, *pts, n, sdf, grad, *stash, st);
#ifndef NCW_HALF_F16
    if (sdf16f_on(net, prec)) return ncw_sdf_fwd16f_launch(net, *pts, n, sdf, grad, *stash, st);
#endif
    if (stash->t[0] == nullptr) NCW_SDF_DISPATCH(sdf_render_kernel, *net, *pts, n, sdf, grad, *stash);  // forward-only render
    else NCW_SDF_DISPATCH(sdf_fwd_kernel, *net, *pts, n, sdf, grad, *stash);
    return 0;
}

extern "C" int NCW_FN(ncw_sdf_bwd)(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* d_sdf,
                                   const float* d_grad, const NcwSdfStash* stash, void* stream) {
    NCW_FORWARD_F16(prec, ncw_sdf_bwd_f16(net, NCW_PREC_BF16, pts, n, d_sdf, d_grad, stash, stream));
    if (!sdf_net_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (pts->mode == 4) return NCW_E_UNSUPPORTED;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (sdf16_on(net, prec)) return NCW_FN(ncw_sdf_bwd16_launch)(net, *pts, n, d_sdf, d_grad, *stash, st);
#ifndef NCW_HALF_F16
    if (sdf16f_on(net, prec)) return ncw_sdf_bwd16f_launch(net, *pts, n, d_sdf, d_grad, *stash, st);
#endif
    NCW_SDF_DISPATCH(sdf_bwd_kernel, *net, *pts, n, d_sdf, d_grad, *stash);
    return 0;
}
