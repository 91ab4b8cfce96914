// Fused colour-network kernels: RenderingNetwork (models/neuconw.py:59-170) with the appearance
// head, forward and backward, one launch each, activations register-resident (ncw_common.h).
#include "ncw_mlp.h"

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

template <class P, int RBF, int RBH, int RBC>
__global__ __launch_bounds__(256) void color_fwd_kernel(NcwColorNet net, NcwPoints src, int64_t n,
                                                        const float* __restrict__ normals, const float* __restrict__ a,
                                                        const void* __restrict__ feat_stash, float* __restrict__ rgb,
                                                        NcwColorStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    const int lane = ncw_lane();
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    float xs[3];
    load_point(src, p, xs, ray);
    const float dir[3] = {src.rays_d[ray * 3 + 0], src.rays_d[ray * 3 + 1], src.rays_d[ray * 3 + 2]};
    const float nrm[3] = {normals[p * 3 + 0], normals[p * 3 + 1], normals[p * 3 + 2]};

    CVec<3> aux1;
    build_aux1<Fast<P>::v>(aux1, dir, a + ray * net.n_a, net.n_a, lane);
    stash_store<3>((SE*)st.aux1, tile, aux1, lane);
    Act<P, 3> aux1a;
    to_act(aux1a, aux1);
    CVec<1> aux2;
    build_aux2(aux2, xs, nrm, lane);
    stash_store<1>((SE*)st.aux2, tile, aux2, lane);
    Act<P, 1> aux2a;
    to_act(aux2a, aux2);

    // f = xyz_encoding_final(feat)   (no activation, neuconw.py:128,136)
    Act<P, RBF> fa;
    {
        CVec<RBF> ft;
        stash_load<RBF>(ft, (const SE*)feat_stash, tile, lane);
        Act<P, RBF> fin;
        to_act(fin, ft);
        CVec<RBF> f;
        load_bias(f, net.b_f, lane);
        mma<RBF, RBF, 32 * RBF>(f, fin, (const WE*)net.w_f, lane);
        stash_store<RBF>((SE*)st.f, tile, f, lane);
        to_act(fa, f);
    }
    // appearance head (neuconw.py:111-127,137-140)
    Act<P, RBH> ea;
    {
        CVec<RBH> e;
        load_bias(e, net.b_e[0], lane);
        const WE* w = (const WE*)net.w_e[0];
        mma<RBF, RBH, 32 * RBF>(e, fa, w, lane);
        mma<3, RBH, 96>(e, aux1a, w + ncw_packed_elems(RBH, RBF), lane);
        relu_epilogue<P, RBH>(ea, e, (SE*)st.e[0], tile, lane);
        for (int i = 1; i < net.n_head; ++i) {
            load_bias(e, net.b_e[i], lane);
            mma<RBH, RBH, 32 * RBH>(e, ea, (const WE*)net.w_e[i], lane);
            relu_epilogue<P, RBH>(ea, e, (SE*)st.e[i], tile, lane);
        }
    }
    // trunk (neuconw.py:158-166)
    Act<P, RBC> xa;
    CVec<RBC> x;
    {
        load_bias(x, net.b_l[0], lane);
        const WE* w = (const WE*)net.w_l[0];
        mma<RBH, RBC, 32 * RBH>(x, ea, w, lane);
        mma<1, RBC, 6>(x, aux2a, w + ncw_packed_elems(RBC, RBH), lane);
        relu_epilogue<P, RBC>(xa, x, (SE*)st.x[0], tile, lane);
    }
    for (int l = 1; l < net.n_lin - 1; ++l) {
        load_bias(x, net.b_l[l], lane);
        mma<RBC, RBC, 32 * RBC>(x, xa, (const WE*)net.w_l[l], lane);
        relu_epilogue<P, RBC>(xa, x, (SE*)st.x[l], tile, lane);
    }
    CVec<1> o;
    load_bias(o, net.b_l[net.n_lin - 1], lane);
    mma<RBC, 1, 32 * RBC>(o, xa, (const WE*)net.w_l[net.n_lin - 1], lane);
    if (valid && lane < 32) {  // features 0,1,2 <-> registers 0,1,2 of half 0; sigmoid (neuconw.py:168-169)
        rgb[p * 3 + 0] = sigmoidf_<Fast<P>::v>(o.v[0][0]);
        rgb[p * 3 + 1] = sigmoidf_<Fast<P>::v>(o.v[0][1]);
        rgb[p * 3 + 2] = sigmoidf_<Fast<P>::v>(o.v[0][2]);
    }
}

template <class P, int RBF, int RBH, int RBC>
__global__ __launch_bounds__(256) void color_bwd_kernel(NcwColorNet net, NcwPoints src, int64_t n,
                                                        const float* __restrict__ rgb, const float* __restrict__ d_rgb,
                                                        float* __restrict__ d_grad, float* __restrict__ d_a,
                                                        void* __restrict__ dfeat_stash, NcwColorStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    const int lane = ncw_lane();
    int64_t tile, p, ray;
    bool valid;
    if (!tile_setup(n, tile, p, valid, lane)) return;
    float xs[3];
    load_point(src, p, xs, ray);
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
    mma<1, RBC, 3>(u, zoa, (const WE*)net.wt_l[net.n_lin - 1], lane);
    for (int l = net.n_lin - 2; l >= 1; --l) {
        relu_backward<P, RBC>(u, (const SE*)st.x[l], tile, lane);
        stash_store<RBC>((SE*)st.zx[l], tile, u, lane);
        Act<P, RBC> za;
        to_act(za, u);
        cvec_zero(u);
        mma<RBC, RBC, 32 * RBC>(u, za, (const WE*)net.wt_l[l], lane);
    }
    CVec<RBH> ue;
    {
        relu_backward<P, RBC>(u, (const SE*)st.x[0], tile, lane);
        stash_store<RBC>((SE*)st.zx[0], tile, u, lane);
        Act<P, RBC> za;
        to_act(za, u);
        CVec<RBH + 1> q;
        cvec_zero(q);
        mma<RBC, RBH + 1, 32 * RBC>(q, za, (const WE*)net.wt_l[0], lane);
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
    for (int i = net.n_head - 1; i >= 1; --i) {
        relu_backward<P, RBH>(ue, (const SE*)st.e[i], tile, lane);
        stash_store<RBH>((SE*)st.ze[i], tile, ue, lane);
        Act<P, RBH> za;
        to_act(za, ue);
        cvec_zero(ue);
        mma<RBH, RBH, 32 * RBH>(ue, za, (const WE*)net.wt_e[i], lane);
    }
    {
        relu_backward<P, RBH>(ue, (const SE*)st.e[0], tile, lane);
        stash_store<RBH>((SE*)st.ze[0], tile, ue, lane);
        Act<P, RBH> za;
        to_act(za, ue);
        CVec<RBF + 3> q;
        cvec_zero(q);
        mma<RBH, RBF + 3, 32 * RBH>(q, za, (const WE*)net.wt_e[0], lane);
        CVec<3> qa;
        qa.v[0] = q.v[RBF]; qa.v[1] = q.v[RBF + 1]; qa.v[2] = q.v[RBF + 2];
        accumulate_d_a(qa, d_a, ray, net.n_a, valid, lane);
        CVec<RBF> zf;
#pragma unroll
        for (int rb = 0; rb < RBF; ++rb) zf.v[rb] = q.v[rb];
        stash_store<RBF>((SE*)st.zf, tile, zf, lane);
        Act<P, RBF> zfa;
        to_act(zfa, zf);
        CVec<RBF> df;
        cvec_zero(df);
        mma<RBF, RBF, 32 * RBF>(df, zfa, (const WE*)net.wt_f, lane);
        stash_store<RBF>((SE*)dfeat_stash, tile, df, lane);
    }
}

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

extern "C" int ncw_color_fwd(const NcwColorNet* net, int prec, const NcwPoints* pts, int64_t n, const float* normals,
                             const float* a, const void* feat_stash, float* rgb, const NcwColorStash* stash,
                             void* stream) {
    if (!color_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_COLOR_DISPATCH(color_fwd_kernel, *net, *pts, n, normals, a, feat_stash, rgb, *stash);
    return 0;
}

extern "C" int ncw_color_bwd(const NcwColorNet* net, int prec, const NcwPoints* pts, int64_t n, const float* rgb,
                             const float* d_rgb, float* d_grad, float* d_a, void* dfeat_stash,
                             const NcwColorStash* stash, void* stream) {
    if (!color_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    NCW_COLOR_DISPATCH(color_bwd_kernel, *net, *pts, n, rgb, d_rgb, d_grad, d_a, dfeat_stash, *stash);
    return 0;
}
