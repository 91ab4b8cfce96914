// Fused background-NeRF kernels: models/nerf.py:86-183 evaluated on the inverted-sphere
// reparametrisation of renderer.py:176-186, forward and backward, one launch each.
#include "ncw_mlp.h"

// 4-D inverted-sphere point (renderer.py:181-186): r = clip(|p|, 1, 1e10); p4 = [p / r, 1 / r]
NCW_DEV void inverted_sphere(const float (&x)[3], float (&p4)[4]) {
    float r = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    r = fminf(fmaxf(r, 1.0f), 1e10f);
    p4[0] = x[0] / r; p4[1] = x[1] / r; p4[2] = x[2] / r; p4[3] = 1.0f / r;
}

template <class P, int RBN, int RBH>
__global__ __launch_bounds__(256) void nerf_fwd_kernel(NcwNerfNet net, NcwPoints src, const float* __restrict__ x4,
                                                       int64_t n, const float* __restrict__ a,
                                                       float* __restrict__ density, float* __restrict__ rgb,
                                                       NcwNerfStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    const int lane = ncw_lane();
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    float p4[4];
    if (x4) {
        p4[0] = x4[p * 4 + 0]; p4[1] = x4[p * 4 + 1]; p4[2] = x4[p * 4 + 2]; p4[3] = x4[p * 4 + 3];
        ray = p;
    } else {
        float xs[3];
        load_point(src, p, xs, ray);
        inverted_sphere(xs, p4);
    }
    const float dir[3] = {src.rays_d[ray * 3 + 0], src.rays_d[ray * 3 + 1], src.rays_d[ray * 3 + 2]};

    CVec<3> gp;
    freq_encode<3, 4, 10, Fast<P>::v>(gp, p4, lane);
    stash_store<3>((SE*)st.gp, tile, gp, lane);
    Act<P, 3> gpa;
    to_act(gpa, gp);
    CVec<3> aux1;
    build_aux1<Fast<P>::v>(aux1, dir, a + ray * net.n_a, net.n_a, lane);
    stash_store<3>((SE*)st.aux1, tile, aux1, lane);
    Act<P, 3> aux1a;
    to_act(aux1a, aux1);

    // trunk (nerf.py:162-167)
    CVec<RBN> acc;
    Act<P, RBN> ha;
    load_bias(acc, net.b_p[0], lane);
    mma<3, RBN, 84>(acc, gpa, (const WE*)net.w_p[0], lane);
    relu_epilogue<P, RBN>(ha, acc, (SE*)st.h[1], tile, lane);
    for (int i = 1; i < net.D; ++i) {
        load_bias(acc, net.b_p[i], lane);
        const WE* w = (const WE*)net.w_p[i];
        mma<RBN, RBN, 32 * RBN>(acc, ha, w, lane);
        if (i == net.skip + 1) mma<3, RBN, 84>(acc, gpa, w + ncw_packed_elems(RBN, RBN), lane);
        relu_epilogue<P, RBN>(ha, acc, (SE*)st.h[i + 1], tile, lane);
    }
    // density + feature (nerf.py:170-171)
    {
        CVec<1> o;
        load_bias(o, net.b_alpha, lane);
        mma<RBN, 1, 32 * RBN>(o, ha, (const WE*)net.w_alpha, lane);
        if (valid && lane < 32) density[p] = o.v[0][0];
    }
    Act<P, RBN> fa;
    load_bias(acc, net.b_feat, lane);
    mma<RBN, RBN, 32 * RBN>(acc, ha, (const WE*)net.w_feat, lane);
    stash_store<RBN>((SE*)st.featn, tile, acc, lane);
    to_act(fa, acc);
    // appearance head (nerf.py:131-139,173-174)
    CVec<RBH> e;
    Act<P, RBH> ea;
    {
        load_bias(e, net.b_a[0], lane);
        const WE* w = (const WE*)net.w_a[0];
        mma<RBN, RBH, 32 * RBN>(e, fa, w, lane);
        mma<3, RBH, 96>(e, aux1a, w + ncw_packed_elems(RBH, RBN), lane);
        relu_epilogue<P, RBH>(ea, e, (SE*)st.e[0], tile, lane);
    }
    for (int i = 1; i < net.n_head; ++i) {
        load_bias(e, net.b_a[i], lane);
        mma<RBH, RBH, 32 * RBH>(e, ea, (const WE*)net.w_a[i], lane);
        relu_epilogue<P, RBH>(ea, e, (SE*)st.e[i], tile, lane);
    }
    CVec<1> o;
    load_bias(o, net.b_rgb, lane);
    mma<RBH, 1, 32 * RBH>(o, ea, (const WE*)net.w_rgb, lane);
    if (valid && lane < 32) {  // raw rgb, no sigmoid (nerf.py:181)
        rgb[p * 3 + 0] = o.v[0][0];
        rgb[p * 3 + 1] = o.v[0][1];
        rgb[p * 3 + 2] = o.v[0][2];
    }
}

template <class P, int RBN, int RBH>
__global__ __launch_bounds__(256) void nerf_bwd_kernel(NcwNerfNet net, NcwPoints src, int64_t n,
                                                       const float* __restrict__ d_density,
                                                       const float* __restrict__ d_rgb, float* __restrict__ d_a,
                                                       NcwNerfStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    const int lane = ncw_lane();
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    ray = (src.mode == 0) ? p : p / src.per_ray;
    const float vm = valid ? 1.f : 0.f;

    CVec<1> zr;
    cvec_zero(zr);
    if (lane < 32) {
        zr.v[0][0] = d_rgb[p * 3 + 0] * vm;
        zr.v[0][1] = d_rgb[p * 3 + 1] * vm;
        zr.v[0][2] = d_rgb[p * 3 + 2] * vm;
    }
    stash_store<1>((SE*)st.zrgb, tile, zr, lane);
    Act<P, 1> zra;
    to_act(zra, zr);
    CVec<RBH> ue;
    cvec_zero(ue);
    mma<1, RBH, 3>(ue, zra, (const WE*)net.wt_rgb, lane);
    for (int i = net.n_head - 1; i >= 1; --i) {
        relu_backward<P, RBH>(ue, (const SE*)st.e[i], tile, lane);
        stash_store<RBH>((SE*)st.ze[i], tile, ue, lane);
        Act<P, RBH> za;
        to_act(za, ue);
        cvec_zero(ue);
        mma<RBH, RBH, 32 * RBH>(ue, za, (const WE*)net.wt_a[i], lane);
    }
    CVec<RBN> u;
    {
        relu_backward<P, RBH>(ue, (const SE*)st.e[0], tile, lane);
        stash_store<RBH>((SE*)st.ze[0], tile, ue, lane);
        Act<P, RBH> za;
        to_act(za, ue);
        CVec<RBN + 3> q;
        cvec_zero(q);
        mma<RBH, RBN + 3, 32 * RBH>(q, za, (const WE*)net.wt_a[0], lane);
        CVec<3> qa;
        qa.v[0] = q.v[RBN]; qa.v[1] = q.v[RBN + 1]; qa.v[2] = q.v[RBN + 2];
        accumulate_d_a(qa, d_a, ray, net.n_a, valid, lane);
        CVec<RBN> zf;
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb) zf.v[rb] = q.v[rb];
        stash_store<RBN>((SE*)st.zfeat, tile, zf, lane);
        Act<P, RBN> zfa;
        to_act(zfa, zf);
        CVec<1> zal;
        cvec_zero(zal);
        zal.v[0][0] = (lane < 32) ? d_density[p] * vm : 0.f;
        stash_store<1>((SE*)st.zalpha, tile, zal, lane);
        Act<P, 1> zala;
        to_act(zala, zal);
        cvec_zero(u);
        mma<RBN, RBN, 32 * RBN>(u, zfa, (const WE*)net.wt_feat, lane);
        mma<1, RBN, 1>(u, zala, (const WE*)net.wt_alpha, lane);
    }
    for (int i = net.D - 1; i >= 0; --i) {
        relu_backward<P, RBN>(u, (const SE*)st.h[i + 1], tile, lane);
        stash_store<RBN>((SE*)st.zp[i], tile, u, lane);
        if (i > 0) {
            Act<P, RBN> za;
            to_act(za, u);
            cvec_zero(u);
            if (i == net.skip + 1) mma<RBN, RBN, 32 * RBN, RBN + 3>(u, za, (const WE*)net.wt_p[i], lane);
            else mma<RBN, RBN, 32 * RBN>(u, za, (const WE*)net.wt_p[i], lane);
        }
    }
}

static bool nerf_ok(const NcwNerfNet* net) {
    return net && net->D >= 2 && net->D <= 8 && net->n_head >= 1 && net->n_head <= 4 && net->n_a >= 0 &&
           net->n_a <= 69 && net->skip >= 0 && net->skip < net->D - 1;
}

#define NCW_NERF_DISPATCH(KERNEL, ...)                                                               \
    do {                                                                                             \
        if (net->rbn == 2 && net->rbh == 1) {                                                        \
            if (prec == NCW_PREC_F32) NCW_LAUNCH_TILES((KERNEL<PrecF32, 2, 1>), n, st, __VA_ARGS__); \
            else NCW_LAUNCH_TILES((KERNEL<PrecBF16, 2, 1>), n, st, __VA_ARGS__);                     \
        } else if (net->rbn == 8 && net->rbh == 4) {                                                 \
            if (prec == NCW_PREC_F32) NCW_LAUNCH_TILES((KERNEL<PrecF32, 8, 4>), n, st, __VA_ARGS__); \
            else NCW_LAUNCH_TILES((KERNEL<PrecBF16, 8, 4>), n, st, __VA_ARGS__);                     \
        } else return NCW_E_UNSUPPORTED;                                                             \
    } while (0)

extern "C" int ncw_nerf_fwd(const NcwNerfNet* net, int prec, const NcwPoints* pts, const float* x4, int64_t n,
                            const float* a, float* density, float* rgb, const NcwNerfStash* stash, void* stream) {
    if (!nerf_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_NERF_DISPATCH(nerf_fwd_kernel, *net, *pts, x4, n, a, density, rgb, *stash);
    return 0;
}

extern "C" int ncw_nerf_bwd(const NcwNerfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* d_density,
                            const float* d_rgb, float* d_a, const NcwNerfStash* stash, void* stream) {
    if (!nerf_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_NERF_DISPATCH(nerf_bwd_kernel, *net, *pts, n, d_density, d_rgb, d_a, *stash);
    return 0;
}
