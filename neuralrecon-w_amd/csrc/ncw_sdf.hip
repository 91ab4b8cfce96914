// Fused SDF-network kernels (models/neuconw.py:183-296) for gfx950.
//   sdf_infer : x -> sdf                       (SDFNetwork.sdf :281-282; sampler / octree / mesh grid)
//   sdf_fwd   : x -> sdf, feat, d sdf/dx       (forward :263-279 + gradient :284-296 as the analytic
//                                               adjoint pass of SURVEY 8a-2) + activation stash
//   sdf_bwd   : first- and second-order backward of the above (what create_graph=True makes autograd
//               compute in the reference), emitting the operands of the weight-gradient GEMMs.
// One wave = 32 points, activations register-resident in MFMA C layout (ncw_common.h); every layer
// of the network runs inside ONE launch; inter-layer activations never round-trip HBM (only the
// stash that the backward needs is written, once, coalesced).
#include "ncw_mlp.h"

NCW_DEV NcwPoints points_from_x(const float* x) {
    NcwPoints s;
    s.x = x; s.rays_o = nullptr; s.rays_d = nullptr; s.z = nullptr; s.sample_dist = nullptr;
    s.per_ray = 1; s.mode = 0;
    return s;
}

// ---------------------------------------------------------------------------------------------
// inference
// ---------------------------------------------------------------------------------------------
template <class P, int RB>
__global__ __launch_bounds__(256) void sdf_infer_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                        float* __restrict__ sdf) {
    typedef typename P::welem WE;
    const int lane = ncw_lane();
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;

    CVec<2> gam;
    freq_encode<2, 3, 6, Fast<P>::v>(gam, xs, lane);
    Act<P, 2> gact;
    to_act(gact, gam);

    CVec<RB> acc;
    Act<P, RB> act;
    load_bias(acc, net.b[0], lane);
    mma<2, RB, 39>(acc, gact, (const WE*)net.w[0], lane);
    softplus_epilogue<P, RB>(act, acc, nullptr, nullptr, 0, lane);
    const int L = net.n_layers;
    for (int l = 1; l < L - 1; ++l) {
        load_bias(acc, net.b[l], lane);
        const WE* w = (const WE*)net.w[l];
        mma<RB, RB, 32 * RB>(acc, act, w, lane);
        if (l == net.skip_layer) mma<2, RB, 39>(acc, gact, w + ncw_packed_elems(RB, RB), lane);
        softplus_epilogue<P, RB>(act, acc, nullptr, nullptr, 0, lane);
    }
    CVec<1> o;
    load_bias(o, net.b[L - 1], lane);
    mma<RB, 1, 32 * RB>(o, act, (const WE*)net.w[L - 1], lane);
    if (valid && lane < 32) sdf[p] = o.v[0][0] / net.scale;
}

// ---------------------------------------------------------------------------------------------
// forward + analytic input gradient + stash
// ---------------------------------------------------------------------------------------------
template <class P, int RB>
__global__ __launch_bounds__(256) void sdf_fwd_kernel(NcwSdfNet net, NcwPoints src, int64_t n, float* __restrict__ sdf,
                                                      float* __restrict__ grad, NcwSdfStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    const int lane = ncw_lane();
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
    const int L = net.n_layers;

    CVec<2> gam;
    freq_encode<2, 3, 6, Fast<P>::v>(gam, xs, lane);
    stash_store<2>((SE*)st.gamma, tile, gam, lane);
    Act<P, 2> gact;
    to_act(gact, gam);

    CVec<RB> acc;
    Act<P, RB> act;
    load_bias(acc, net.b[0], lane);
    mma<2, RB, 39>(acc, gact, (const WE*)net.w[0], lane);
    softplus_epilogue<P, RB>(act, acc, (SE*)st.h[1], (SE*)st.s[0], tile, lane);
    for (int l = 1; l < L - 1; ++l) {
        load_bias(acc, net.b[l], lane);
        const WE* w = (const WE*)net.w[l];
        mma<RB, RB, 32 * RB>(acc, act, w, lane);
        if (l == net.skip_layer) mma<2, RB, 39>(acc, gact, w + ncw_packed_elems(RB, RB), lane);
        softplus_epilogue<P, RB>(act, acc, (SE*)st.h[l + 1], (SE*)st.s[l], tile, lane);
    }
    {
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        mma<RB, 1, 32 * RB>(o, act, (const WE*)net.w[L - 1], lane);
        if (valid && lane < 32) sdf[p] = o.v[0][0] / net.scale;
        load_bias(acc, net.b_feat, lane);
        mma<RB, RB, 32 * RB>(acc, act, (const WE*)net.w_feat, lane);
        stash_store<RB>((SE*)st.feat, tile, acc, lane);
    }
    // ---- adjoint pass: a_{L-2} = W_{L-1}[0,:]; t_l = a_l * s_l; a_{l-1} = W_l^T t_l --------------
    CVec<RB> a;
    {
        CVec<1> e0;
        cvec_zero(e0);
        e0.v[0][0] = (lane < 32) ? 1.f : 0.f;  // feature 0 <-> (r=0, h=0)
        Act<P, 1> e0a;
        to_act(e0a, e0);
        cvec_zero(a);
        mma<1, RB, 1>(a, e0a, (const WE*)net.wt[L - 1], lane);
    }
    CVec<2> gg;
    cvec_zero(gg);
    for (int l = L - 2; l >= 0; --l) {
        CVec<RB> sv;
        stash_load<RB>(sv, (const SE*)st.s[l], tile, lane);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) a.v[rb][r] *= sv.v[rb][r];
        stash_store<RB>((SE*)st.t[l], tile, a, lane);
        Act<P, RB> ta;
        to_act(ta, a);
        const WE* wt = (const WE*)net.wt[l];
        if (l == 0) {
            mma<RB, 2, 32 * RB>(gg, ta, wt, lane);
        } else if (l == net.skip_layer) {
            CVec<RB + 2> q;
            cvec_zero(q);
            mma<RB, RB + 2, 32 * RB>(q, ta, wt, lane);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a.v[rb] = q.v[rb];
            gg.v[0] = q.v[RB];
            gg.v[1] = q.v[RB + 1];
        } else {
            cvec_zero(a);
            mma<RB, RB, 32 * RB>(a, ta, wt, lane);
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

// ---------------------------------------------------------------------------------------------
// backward (second order)
// ---------------------------------------------------------------------------------------------
template <class P, int RB>
__global__ __launch_bounds__(256) void sdf_bwd_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                      const float* __restrict__ d_sdf, const float* __restrict__ d_grad,
                                                      NcwSdfStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    const int lane = ncw_lane();
    const int h = lane >> 5;
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
    const int L = net.n_layers;
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
    for (int l = 0; l <= L - 2; ++l) {
        cvec_zero(tb);
        const WE* w = (const WE*)net.w[l];
        if (l == 0) {
            mma<2, RB, 39>(tb, q0a, w, lane);
        } else {
            mma<RB, RB, 32 * RB>(tb, qa, w, lane);
            if (l == net.skip_layer) mma<2, RB, 39>(tb, q0a, w + ncw_packed_elems(RB, RB), lane);
        }
        CVec<RB> sv, tv;
        stash_load<RB>(sv, (const SE*)st.s[l], tile, lane);
        stash_load<RB>(tv, (const SE*)st.t[l], tile, lane);
        CVec<RB> z2;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tbar = tb.v[rb][r];
                const float s = sv.v[rb][r];
                // a_l phi''(z_l) = 100 t_l (1 - s_l)   (0 above the Softplus threshold where s == 1)
                z2.v[rb][r] = tbar * 100.f * tv.v[rb][r] * (1.f - s);
                tb.v[rb][r] = tbar * s;  // abar_l
            }
        stash_store<RB>((SE*)st.zbar[l], tile, z2, lane);    // temporarily zbar2_l
        stash_store<RB>((SE*)st.qbar[l + 1], tile, tb, lane);  // qbar_{l+1} = abar_l
        to_act(qa, tb);
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
        CVec<RB> df;
        stash_load<RB>(df, (const SE*)st.dfeat, tile, lane);
        Act<P, RB> dfa;
        to_act(dfa, df);
        cvec_zero(u);
        mma<RB, RB, 32 * RB>(u, dfa, (const WE*)net.wt_feat, lane);
        mma<1, RB, 1>(u, zsa, (const WE*)net.wt[L - 1], lane);
    }
    for (int l = L - 2; l >= 0; --l) {
        CVec<RB> sv, z2;
        stash_load<RB>(sv, (const SE*)st.s[l], tile, lane);
        stash_load<RB>(z2, (const SE*)st.zbar[l], tile, lane);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) u.v[rb][r] = u.v[rb][r] * sv.v[rb][r] + z2.v[rb][r];
        stash_store<RB>((SE*)st.zbar[l], tile, u, lane);
        if (l > 0) {
            Act<P, RB> za;
            to_act(za, u);
            cvec_zero(u);
            if (l == net.skip_layer) mma<RB, RB, 32 * RB, RB + 2>(u, za, (const WE*)net.wt[l], lane);
            else mma<RB, RB, 32 * RB>(u, za, (const WE*)net.wt[l], lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
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

static int sdf_infer_any(const NcwSdfNet* net, int prec, const NcwPoints& src, int64_t n, float* sdf, void* stream) {
    if (!sdf_net_ok(net) || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_SDF_DISPATCH(sdf_infer_kernel, *net, src, n, sdf);
    return 0;
}

extern "C" int ncw_sdf_infer(const NcwSdfNet* net, int prec, const float* x, int64_t n, float* sdf, void* stream) {
    NcwPoints src;
    src.x = x; src.rays_o = nullptr; src.rays_d = nullptr; src.z = nullptr; src.sample_dist = nullptr;
    src.per_ray = 1; src.mode = 0;
    return sdf_infer_any(net, prec, src, n, sdf, stream);
}

extern "C" int ncw_sdf_infer_rays(const NcwSdfNet* net, int prec, const float* rays_o, const float* rays_d,
                                  const float* z, int R, int n, float* sdf, void* stream) {
    if (R < 0 || n <= 0) return NCW_E_BADARG;
    NcwPoints src;
    src.x = nullptr; src.rays_o = rays_o; src.rays_d = rays_d; src.z = z; src.sample_dist = nullptr;
    src.per_ray = n; src.mode = 1;
    return sdf_infer_any(net, prec, src, (int64_t)R * n, sdf, stream);
}

extern "C" int ncw_sdf_fwd(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t n, float* sdf, float* grad,
                           const NcwSdfStash* stash, void* stream) {
    if (!sdf_net_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_SDF_DISPATCH(sdf_fwd_kernel, *net, *pts, n, sdf, grad, *stash);
    return 0;
}

extern "C" int ncw_sdf_bwd(const NcwSdfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* d_sdf,
                           const float* d_grad, const NcwSdfStash* stash, void* stream) {
    if (!sdf_net_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_SDF_DISPATCH(sdf_bwd_kernel, *net, *pts, n, d_sdf, d_grad, *stash);
    return 0;
}
