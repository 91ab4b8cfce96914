// Fused background-NeRF kernels: models/nerf.py:86-183 evaluated on the inverted-sphere
// reparametrisation of renderer.py:176-186, forward and backward, one launch each.
#include "ncw_mlp.h"

int NCW_FN(ncw_nerf_fwd8_launch)(const NcwNerfNet* net, const NcwPoints& src, const float* x4, int64_t n, const float* a, float* density,
                         float* rgb, const NcwNerfStash& stash, hipStream_t st);  // ncw_sdf8.hip
int NCW_FN(ncw_nerf_bwd8_launch)(const NcwNerfNet* net, const NcwPoints& src, int64_t n, const float* d_density, const float* d_rgb,
                         float* d_a, const NcwNerfStash& stash, hipStream_t st);

// 4-D inverted-sphere point (renderer.py:181-186): r = clip(|p|, 1, 1e10); p4 = [p / r, 1 / r]
namespace NCW_NS {

NCW_DEV void inverted_sphere(const float (&x)[3], float (&p4)[4]) {
    float r = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    r = fminf(fmaxf(r, 1.0f), 1e10f);
    p4[0] = x[0] / r; p4[1] = x[1] / r; p4[2] = x[2] / r; p4[3] = 1.0f / r;
}

template <class P, int RBN, int RBH, int OCC = 1>
struct NerfShapes {
    static constexpr int SLOT = RingSlot<RBN, OCC>::bytes;
    static constexpr int FCB_P0 = ncw_first_chunk_bytes<P, 3, 84, RBN, SLOT>();
    static constexpr int FCB_P = ncw_first_chunk_bytes<P, RBN, 32 * RBN, RBN, SLOT>();
    static constexpr int FCB_PS = ncw_first_chunk_bytes<P, RBN + 3, 32 * RBN + 84, RBN, SLOT>();
    static constexpr int FCB_ALPHA = ncw_first_chunk_bytes<P, RBN, 32 * RBN, 1, SLOT>();
    static constexpr int FCB_A0 = ncw_first_chunk_bytes<P, RBN + 3, 32 * RBN + 96, RBH, SLOT>();
    static constexpr int FCB_A = ncw_first_chunk_bytes<P, RBH, 32 * RBH, RBH, SLOT>();
    static constexpr int FCB_RGB = ncw_first_chunk_bytes<P, RBH, 32 * RBH, 1, SLOT>();
    static constexpr int FCB_TRGB = ncw_first_chunk_bytes<P, 1, 3, RBH, SLOT>();
    static constexpr int FCB_TA0 = ncw_first_chunk_bytes<P, RBH, 32 * RBH, RBN + 3, SLOT>();
    static constexpr int FCB_TALPHA = ncw_first_chunk_bytes<P, 1, 1, RBN, SLOT>();
    static constexpr int FCB_TPS = ncw_first_chunk_bytes<P, RBN, 32 * RBN, RBN + 3, SLOT>();
};

// TRAIN = false (nerf_render_kernel): the forward-only render -- the same arithmetic bit for bit, nothing is stashed.
template <class P, int RBN, int RBH, bool TRAIN>
NCW_DEV void nerf_fwd_body(const NcwNerfNet& net, const NcwPoints& src, const float* __restrict__ x4, int64_t n,
                           const float* __restrict__ a, float* __restrict__ density, float* __restrict__ rgb,
                           const NcwNerfStash& st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    auto stp = [](void* p) -> SE* { return TRAIN ? (SE*)p : nullptr; };  // compile-time null: the helpers' `if (st)` folds away
    typedef NerfShapes<P, RBN, RBH> SH;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    n = points_count(src, n);  // mode 4 (dead-background elimination): the selection's size, read on the device
    if ((int64_t)blockIdx.x * (blockDim.x >> 6) * 32 >= n) return;  // surplus workgroup (uniform: before any barrier / DMA)
    ring_prologue(ring, net.w_p[0], SH::FCB_P0);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    const int64_t ps = point_slot(src, p);  // density / rgb are addressed by the ray sample
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

    Act<P, 3> gpa;
    {
        CVec<3> gp;
        freq_encode<3, 4, 10, Fast<P>::v>(gp, p4, lane);
        if (TRAIN) stash_store<3>((SE*)st.gp, tile, gp, lane);
        to_act(gpa, gp);
    }
    Act<P, 3> aux1a;
    {
        CVec<3> aux1;
        build_aux1<Fast<P>::v>(aux1, dir, a + ray * net.n_a, net.n_a, lane);
        if (TRAIN) stash_store<3>((SE*)st.aux1, tile, aux1, lane);
        to_act(aux1a, aux1);
        if (st.aux_bias != nullptr) ncw_act_zero3(aux1a);  // the per-ray part of the head comes from ncw_aux_ray_bias (fp32)
    }
    auto next_trunk = [&](int i, const void*& w, int& bytes) {  // matrix consumed after trunk layer i
        const int m = i + 1;
        if (m < net.D) {
            w = net.w_p[m];
            bytes = (m == net.skip + 1) ? SH::FCB_PS : SH::FCB_P;
        } else {
            w = net.w_alpha;
            bytes = SH::FCB_ALPHA;
        }
    };
    const void* wn;
    int nb;
    // trunk (nerf.py:162-167)
    CVec<RBN> acc;
    Act<P, RBN> ha;
    load_bias(acc, net.b_p[0], lane);
    next_trunk(0, wn, nb);
    mma_stream<3, RBN, 84, SH::SLOT>(acc, gpa, ring, (const WE*)net.w_p[0], wn, nb, lane);
    relu_epilogue<P, RBN>(ha, acc, stp(st.h[1]), tile, lane);
    for (int i = 1; i < net.D; ++i) {
        load_bias(acc, net.b_p[i], lane);
        next_trunk(i, wn, nb);
        if (i == net.skip + 1) {
            Act<P, RBN + 3> cat;
            act_concat<RBN, 3>(cat, ha, gpa);
            mma_stream<RBN + 3, RBN, 32 * RBN + 84, SH::SLOT>(acc, cat, ring, (const WE*)net.w_p[i], wn, nb, lane);
        } else {
            mma_stream<RBN, RBN, 32 * RBN, SH::SLOT>(acc, ha, ring, (const WE*)net.w_p[i], wn, nb, lane);
        }
        relu_epilogue<P, RBN>(ha, acc, stp(st.h[i + 1]), tile, lane);
    }
    // density + feature (nerf.py:170-171)
    {
        CVec<1> o;
        load_bias(o, net.b_alpha, lane);
        mma_stream<RBN, 1, 32 * RBN, SH::SLOT>(o, ha, ring, (const WE*)net.w_alpha, net.w_feat, SH::FCB_P, lane);
        if (valid && lane < 32) density[ps] = o.v[0][0];
    }
    Act<P, RBN + 3> cat1;
    {
        load_bias(acc, net.b_feat, lane);
        mma_stream<RBN, RBN, 32 * RBN, SH::SLOT>(acc, ha, ring, (const WE*)net.w_feat, net.w_a[0], SH::FCB_A0, lane);
        if (TRAIN) stash_store<RBN>((SE*)st.featn, tile, acc, lane);
        Act<P, RBN> fa;
        to_act(fa, acc);
        act_concat<RBN, 3>(cat1, fa, aux1a);
    }
    // appearance head (nerf.py:131-139,173-174)
    CVec<RBH> e;
    Act<P, RBH> ea;
    {
        load_bias(e, net.b_a[0], lane);
        if (st.aux_bias != nullptr) ncw_add_ray_bias<RBH>(e, st.aux_bias + ray * (32 * RBH), lane);
        wn = net.n_head > 1 ? net.w_a[1] : net.w_rgb;
        nb = net.n_head > 1 ? SH::FCB_A : SH::FCB_RGB;
        mma_stream<RBN + 3, RBH, 32 * RBN + 96, SH::SLOT>(e, cat1, ring, (const WE*)net.w_a[0], wn, nb, lane);
        relu_epilogue<P, RBH>(ea, e, stp(st.e[0]), tile, lane);
    }
    for (int i = 1; i < net.n_head; ++i) {
        load_bias(e, net.b_a[i], lane);
        wn = i + 1 < net.n_head ? net.w_a[i + 1] : net.w_rgb;
        nb = i + 1 < net.n_head ? SH::FCB_A : SH::FCB_RGB;
        mma_stream<RBH, RBH, 32 * RBH, SH::SLOT>(e, ea, ring, (const WE*)net.w_a[i], wn, nb, lane);
        relu_epilogue<P, RBH>(ea, e, stp(st.e[i]), tile, lane);
    }
    CVec<1> o;
    load_bias(o, net.b_rgb, lane);
    mma_stream<RBH, 1, 32 * RBH, SH::SLOT>(o, ea, ring, (const WE*)net.w_rgb, nullptr, 0, lane);
    if (valid && lane < 32) {  // raw rgb, no sigmoid (nerf.py:181)
        rgb[ps * 3 + 0] = o.v[0][0];
        rgb[ps * 3 + 1] = o.v[0][1];
        rgb[ps * 3 + 2] = o.v[0][2];
    }
}

template <class P, int RBN, int RBH>
__global__ __launch_bounds__(64 * NCW_WG_WAVES) void nerf_fwd_kernel(NcwNerfNet net, NcwPoints src,
                                                                     const float* __restrict__ x4, int64_t n,
                                                                     const float* __restrict__ a,
                                                                     float* __restrict__ density, float* __restrict__ rgb,
                                                                     NcwNerfStash st) {
    nerf_fwd_body<P, RBN, RBH, true>(net, src, x4, n, a, density, rgb, st);
}
template <class P, int RBN, int RBH>
__global__ __launch_bounds__(64 * NCW_WG_WAVES) void nerf_render_kernel(NcwNerfNet net, NcwPoints src,
                                                                        const float* __restrict__ x4, int64_t n,
                                                                        const float* __restrict__ a,
                                                                        float* __restrict__ density, float* __restrict__ rgb,
                                                                        NcwNerfStash st) {
    nerf_fwd_body<P, RBN, RBH, false>(net, src, x4, n, a, density, rgb, st);
}

template <class P, int RBN, int RBH>
__global__ __launch_bounds__(64 * NCW_WG_WAVES, 2) void nerf_bwd_kernel(NcwNerfNet net, NcwPoints src, int64_t n,
                                                                     const float* __restrict__ d_density,
                                                                     const float* __restrict__ d_rgb,
                                                                     float* __restrict__ d_a, float* __restrict__ d_a_rows,
                                                                     NcwNerfStash st) {
    typedef typename P::welem WE;
    typedef typename P::selem SE;
    typedef NerfShapes<P, RBN, RBH, 2> SH;
    NCW_RING_DECL(SH::SLOT);
    const int lane = ncw_lane();
    n = points_count(src, n);  // mode 4: the selection's size, read on the device
    if ((int64_t)blockIdx.x * (blockDim.x >> 6) * 32 >= n) return;  // surplus workgroup (uniform: before any barrier / DMA)
    ring_prologue(ring, net.wt_rgb, SH::FCB_TRGB);
    int64_t tile, p, ray;
    bool valid;
    tile_setup(n, tile, p, valid, lane);
    const int64_t ps = point_slot(src, p);  // cotangents (and the d_a_rows row) are addressed by the ray sample
    ray = (src.mode == 0) ? ps : ps / src.per_ray;
    const float vm = valid ? 1.f : 0.f;

    CVec<1> zr;
    cvec_zero(zr);
    if (lane < 32) {
        zr.v[0][0] = d_rgb[ps * 3 + 0] * vm;
        zr.v[0][1] = d_rgb[ps * 3 + 1] * vm;
        zr.v[0][2] = d_rgb[ps * 3 + 2] * vm;
    }
    stash_store<1>((SE*)st.zrgb, tile, zr, lane);
    Act<P, 1> zra;
    to_act(zra, zr);
    CVec<RBH> ue;
    cvec_zero(ue);
    {
        const void* wn = net.n_head > 1 ? net.wt_a[net.n_head - 1] : net.wt_a[0];
        const int nb = net.n_head > 1 ? SH::FCB_A : SH::FCB_TA0;
        mma_stream<1, RBH, 3, SH::SLOT>(ue, zra, ring, (const WE*)net.wt_rgb, wn, nb, lane);
    }
    Act<P, RBH> zea;
    for (int i = net.n_head - 1; i >= 1; --i) {
        relu_backward<P, RBH>(zea, ue, (const SE*)st.e[i], (SE*)st.ze[i], tile, lane);
        cvec_zero(ue);
        mma_stream<RBH, RBH, 32 * RBH, SH::SLOT>(ue, zea, ring, (const WE*)net.wt_a[i], net.wt_a[i - 1],
                                                  i - 1 == 0 ? SH::FCB_TA0 : SH::FCB_A, lane);
    }
    CVec<RBN> u;
    {
        relu_backward<P, RBH>(zea, ue, (const SE*)st.e[0], (SE*)st.ze[0], tile, lane);
        CVec<RBN + 3> q;
        cvec_zero(q);
        mma_stream<RBH, RBN + 3, 32 * RBH, SH::SLOT>(q, zea, ring, (const WE*)net.wt_a[0], net.wt_feat, SH::FCB_P, lane);
        CVec<3> qa;
        qa.v[0] = q.v[RBN]; qa.v[1] = q.v[RBN + 1]; qa.v[2] = q.v[RBN + 2];
        accumulate_d_a(qa, d_a, ray, net.n_a, valid, lane, d_a_rows, ps);
        CVec<RBN> zf;
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb) zf.v[rb] = q.v[rb];
        stash_store<RBN>((SE*)st.zfeat, tile, zf, lane);
        Act<P, RBN> zfa;
        to_act(zfa, zf);
        CVec<1> zal;
        cvec_zero(zal);
        zal.v[0][0] = (lane < 32) ? d_density[ps] * vm : 0.f;
        stash_store<1>((SE*)st.zalpha, tile, zal, lane);
        Act<P, 1> zala;
        to_act(zala, zal);
        cvec_zero(u);
        mma_stream<RBN, RBN, 32 * RBN, SH::SLOT>(u, zfa, ring, (const WE*)net.wt_feat, net.wt_alpha, SH::FCB_TALPHA, lane);
        const int m = net.D - 1;  // first trunk matrix of the reverse sweep (only needed when m > 0)
        const void* wn = m > 0 ? net.wt_p[m] : nullptr;
        const int nb = (m == net.skip + 1) ? SH::FCB_TPS : SH::FCB_P;
        mma_stream<1, RBN, 1, SH::SLOT>(u, zala, ring, (const WE*)net.wt_alpha, wn, nb, lane);
    }
    Act<P, RBN> za;
    for (int i = net.D - 1; i >= 0; --i) {
        relu_backward<P, RBN>(za, u, (const SE*)st.h[i + 1], (SE*)st.zp[i], tile, lane);
        if (i > 0) {
            cvec_zero(u);
            const void* wn = (i - 1 > 0) ? net.wt_p[i - 1] : nullptr;
            const int nb = (i - 1 == net.skip + 1) ? SH::FCB_TPS : SH::FCB_P;
            if (i == net.skip + 1) mma_stream<RBN, RBN, 32 * RBN, SH::SLOT, RBN + 3>(u, za, ring, (const WE*)net.wt_p[i], wn, nb, lane);
            else mma_stream<RBN, RBN, 32 * RBN, SH::SLOT>(u, za, ring, (const WE*)net.wt_p[i], wn, nb, lane);
        }
    }
}

}  // namespace NCW_NS
using namespace NCW_NS;

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

#ifndef NCW_HALF_F16
extern "C" int ncw_nerf_fwd_f16(const NcwNerfNet*, int, const NcwPoints*, const float*, int64_t, const float*, float*, float*,
                                const NcwNerfStash*, void*);
extern "C" int ncw_nerf_bwd_f16(const NcwNerfNet*, int, const NcwPoints*, int64_t, const float*, const float*, float*, float*,
                                const NcwNerfStash*, void*);
#endif

extern "C" int NCW_FN(ncw_nerf_fwd)(const NcwNerfNet* net, int prec, const NcwPoints* pts, const float* x4, int64_t n,
                                    const float* a, float* density, float* rgb, const NcwNerfStash* stash, void* stream) {
    NCW_FORWARD_F16(prec, ncw_nerf_fwd_f16(net, NCW_PREC_BF16, pts, x4, n, a, density, rgb, stash, stream));
    if (!nerf_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (pts->mode == 4 && (!pts->idx || !pts->count || x4)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // W = 256, 16-bit: the weights-stationary kernel of ncw_sdf8.hip (0.33 vs 0.40 ms per 135,168 points for the
    // weights-through-LDS kernel below, which serves fp32 and the other widths)
    if (net->rbn == 8 && net->rbh == 4 && prec == NCW_PREC_BF16 && net->n_head >= 1 && net->n_head <= 4)
        return NCW_FN(ncw_nerf_fwd8_launch)(net, *pts, x4, n, a, density, rgb, *stash, st);
    if (stash->gp == nullptr) NCW_NERF_DISPATCH(nerf_render_kernel, *net, *pts, x4, n, a, density, rgb, *stash);  // forward-only render
    else NCW_NERF_DISPATCH(nerf_fwd_kernel, *net, *pts, x4, n, a, density, rgb, *stash);
    return 0;
}

// fp16 build: the split-precision refinement of ncw_split.hip (W = 256 networks that carry residual matrices)
#ifdef NCW_HALF_F16
int ncw_nerf_refineS_launch_f16(const NcwNerfNet* net, const NcwPoints& src, int64_t n, const float* aux_bias, float* density,
                                float* rgb, hipStream_t st);
#else
extern "C" int ncw_nerf_refine_f16(const NcwNerfNet*, int, const NcwPoints*, int64_t, const float*, float*, float*, void*);
#endif

extern "C" int NCW_FN(ncw_nerf_refine)(const NcwNerfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* aux_bias,
                                       float* density, float* rgb, void* stream) {
    NCW_FORWARD_F16(prec, ncw_nerf_refine_f16(net, NCW_PREC_BF16, pts, n, aux_bias, density, rgb, stream));
#ifdef NCW_HALF_F16
    if (!nerf_ok(net) || !pts || !aux_bias || !density || !rgb || n < 0 || prec != NCW_PREC_BF16) return NCW_E_BADARG;
    if (pts->mode != 4 || !pts->idx || !pts->count) return NCW_E_BADARG;   // a device-made selection (ncw_bg_select)
    if (net->rbn != 8 || net->rbh != 4) return NCW_E_UNSUPPORTED;
    bool lo = net->w_alpha_lo != nullptr && net->w_feat_lo != nullptr && net->w_rgb_lo != nullptr;
    for (int i = 0; i < net->D; ++i) lo = lo && net->w_p_lo[i] != nullptr;
    for (int i = 0; i < net->n_head; ++i) lo = lo && net->w_a_lo[i] != nullptr;
    if (!lo) return NCW_E_BADARG;
    if (n == 0) return 0;
    return ncw_nerf_refineS_launch_f16(net, *pts, n, aux_bias, density, rgb, (hipStream_t)stream);
#else
    return NCW_E_UNSUPPORTED;  // fp16 mode only (prec 2)
#endif
}

extern "C" int NCW_FN(ncw_nerf_bwd)(const NcwNerfNet* net, int prec, const NcwPoints* pts, int64_t n, const float* d_density,
                                    const float* d_rgb, float* d_a, float* d_a_rows, const NcwNerfStash* stash, void* stream) {
    NCW_FORWARD_F16(prec, ncw_nerf_bwd_f16(net, NCW_PREC_BF16, pts, n, d_density, d_rgb, d_a, d_a_rows, stash, stream));
    if (!nerf_ok(net) || !pts || !stash || n < 0 || (prec != NCW_PREC_F32 && prec != NCW_PREC_BF16)) return NCW_E_BADARG;
    if (pts->mode == 4 && (!pts->idx || !pts->count)) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // W = 256, 16-bit: the weights-stationary kernel of ncw_sdf8.hip (0.44 vs 0.49 ms per 135,168 points); the
    // weights-through-LDS kernel below serves fp32, the other widths and the order-fixed d_a_rows path
    if (d_a_rows == nullptr && net->rbn == 8 && net->rbh == 4 && prec == NCW_PREC_BF16 && net->n_head >= 1 &&
        net->n_head <= 4)
        return NCW_FN(ncw_nerf_bwd8_launch)(net, *pts, n, d_density, d_rgb, d_a, *stash, st);
    NCW_NERF_DISPATCH(nerf_bwd_kernel, *net, *pts, n, d_density, d_rgb, d_a, d_a_rows, *stash);
    return 0;
}
