// Fused SDF-network kernels (models/neuconw.py:183-296) for gfx950.
//   sdf_infer : x -> sdf            (SDFNetwork.sdf, :281-282; sampler / octree refresh / mesh grid)
// One wave = 32 points, activations register-resident in MFMA C layout (ncw_common.h); every layer
// of the network runs inside ONE launch, no activation ever touches HBM.
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

template <class P>
struct Fast { static constexpr bool v = (P::id == NCW_PREC_BF16); };

// act = Softplus100(acc)
template <class P, int RB>
NCW_DEV void softplus_act(Act<P, RB>& act, CVec<RB>& acc) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y, s;
            softplus100<Fast<P>::v>(acc.v[rb][r], y, s);
            acc.v[rb][r] = y;
        }
    to_act(act, acc);
}

// Where a wave's 32 points come from: an explicit [n,3] array, or ray samples o + d z.
struct PointSrc {
    const float* x;       // [n,3] or null
    const float* rays_o;  // [R,3]
    const float* rays_d;  // [R,3]
    const float* z;       // [R,per_ray]
    int per_ray;
};
NCW_DEV void load_point(const PointSrc& s, int64_t p, float (&xs)[3]) {
    if (s.x) {
        xs[0] = s.x[p * 3 + 0]; xs[1] = s.x[p * 3 + 1]; xs[2] = s.x[p * 3 + 2];
    } else {
        const int64_t r = p / s.per_ray;
        const float zz = s.z[p];
        xs[0] = s.rays_o[r * 3 + 0] + s.rays_d[r * 3 + 0] * zz;
        xs[1] = s.rays_o[r * 3 + 1] + s.rays_d[r * 3 + 1] * zz;
        xs[2] = s.rays_o[r * 3 + 2] + s.rays_d[r * 3 + 2] * zz;
    }
}

template <class P, int RB>
__global__ __launch_bounds__(256) void sdf_infer_kernel(NcwSdfNet net, PointSrc src, int64_t n,
                                                        float* __restrict__ sdf) {
    typedef typename P::welem WE;
    const int lane = ncw_lane();
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int64_t p = tile * 32 + (lane & 31);
    if (tile * 32 >= n) return;
    const bool valid = p < n;
    if (!valid) p = n - 1;
    float xs[3];
    load_point(src, p, xs);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;

    CVec<2> gam;
    freq_encode<2, 3, 6, Fast<P>::v>(gam, xs, lane);
    Act<P, 2> gact;
    to_act(gact, gam);

    CVec<RB> acc;
    Act<P, RB> act;
    load_bias(acc, net.b[0], lane);
    mma<2, RB, 39>(acc, gact, (const WE*)net.w[0], lane);
    softplus_act<P, RB>(act, acc);
    const int L = net.n_layers;
    for (int l = 1; l < L - 1; ++l) {
        load_bias(acc, net.b[l], lane);
        const WE* w = (const WE*)net.w[l];
        mma<RB, RB, 32 * RB>(acc, act, w, lane);
        if (l == net.skip_layer) mma<2, RB, 39>(acc, gact, w + ncw_packed_elems(RB, RB), lane);
        softplus_act<P, RB>(act, acc);
    }
    CVec<1> o;
    load_bias(o, net.b[L - 1], lane);
    mma<RB, 1, 32 * RB>(o, act, (const WE*)net.w[L - 1], lane);
    if (valid && lane < 32) sdf[p] = o.v[0][0] / net.scale;
}

template <class P, int RB>
static int launch_sdf_infer(const NcwSdfNet* net, const PointSrc& src, int64_t n, float* sdf, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    const int64_t blocks = (tiles + 3) / 4;
    hipLaunchKernelGGL((sdf_infer_kernel<P, RB>), dim3((unsigned)blocks), dim3(256), 0, st, *net, src, n, sdf);
    NCW_CHECK_LAUNCH();
    return 0;
}

static int sdf_infer_dispatch(const NcwSdfNet* net, int prec, const PointSrc& src, int64_t n, float* sdf, void* stream) {
    if (!net || n < 0 || net->multires != 6 || net->n_layers < 2 || net->n_layers > NCW_MAX_LAYERS) return NCW_E_BADARG;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define NCW_DISPATCH(RBV)                                                                        \
    if (net->rb == RBV) {                                                                        \
        if (prec == NCW_PREC_F32) return launch_sdf_infer<PrecF32, RBV>(net, src, n, sdf, st);   \
        if (prec == NCW_PREC_BF16) return launch_sdf_infer<PrecBF16, RBV>(net, src, n, sdf, st); \
        return NCW_E_BADARG;                                                                     \
    }
    NCW_DISPATCH(2)
    NCW_DISPATCH(8)
    NCW_DISPATCH(16)
#undef NCW_DISPATCH
    return NCW_E_UNSUPPORTED;
}

extern "C" int ncw_sdf_infer(const NcwSdfNet* net, int prec, const float* x, int64_t n, float* sdf, void* stream) {
    PointSrc src{x, nullptr, nullptr, nullptr, 1};
    return sdf_infer_dispatch(net, prec, src, n, sdf, stream);
}

extern "C" int ncw_sdf_infer_rays(const NcwSdfNet* net, int prec, const float* rays_o, const float* rays_d,
                                  const float* z, int R, int n, float* sdf, void* stream) {
    if (R < 0 || n <= 0) return NCW_E_BADARG;
    PointSrc src{nullptr, rays_o, rays_d, z, n};
    return sdf_infer_dispatch(net, prec, src, (int64_t)R * n, sdf, stream);
}
