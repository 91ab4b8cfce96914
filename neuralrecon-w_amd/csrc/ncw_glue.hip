// Per-step glue of render() and the loss as single launches.  The step is GPU-bound (rocprof: kernel time sums to the
// step time) and a dependent kernel boundary costs 1.5-2 us, so the ~60 tiny torch kernels of ray normalisation, the
// variance network, the loss and their backward were 0.25 ms of a 3.9 ms step; each group below is one launch.
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

namespace {

// renderer.py:793-806
__global__ __launch_bounds__(256) void ray_prologue_kernel(const float* __restrict__ rays, int ncols, int64_t R, float ox, float oy,
                                                           float oz, float radius, float* __restrict__ rays_o,
                                                           float* __restrict__ rays_d, float* __restrict__ near,
                                                           float* __restrict__ far, float* __restrict__ depth_gt,
                                                           float* __restrict__ depth_w) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* s = rays + r * ncols;
    // torch divides a tensor by a Python scalar as a * (1.0f / b) (ATen div_true_kernel_cuda): same arithmetic, bit for bit
    const float inv = 1.0f / radius;
    rays_o[r * 3 + 0] = (s[0] - ox) * inv;
    rays_o[r * 3 + 1] = (s[1] - oy) * inv;
    rays_o[r * 3 + 2] = (s[2] - oz) * inv;
    rays_d[r * 3 + 0] = s[3]; rays_d[r * 3 + 1] = s[4]; rays_d[r * 3 + 2] = s[5];
    near[r] = s[6] * inv;
    far[r] = s[7] * inv;
    depth_gt[r] = ncols >= 10 ? s[8] * inv : 0.f;
    depth_w[r] = ncols >= 10 ? s[9] : 0.f;
}

__global__ void inv_s_fwd_kernel(const float* __restrict__ variance, float* __restrict__ inv_s, float* __restrict__ s_val) {
    const float v = fminf(fmaxf(expf(variance[0] * 10.0f), 1e-6f), 1e6f);
    inv_s[0] = v;
    if (s_val) s_val[0] = 1.0f / v;
}

// fixed-order block reduction: thread t sums elements t, t+256, ... ; then a fixed tree
__device__ __forceinline__ float block_sum_fixed(float v, float* sm) {
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    const float t = sm[0];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void inv_s_bwd_kernel(const float* __restrict__ d, int64_t R, const float* __restrict__ inv_s,
                                                        float* __restrict__ d_var) {
    __shared__ float sm[256];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < R; i += 256) a += d[i];
    const float tot = block_sum_fixed(a, sm);
    if (threadIdx.x == 0) {
        const float v = inv_s[0];
        d_var[0] = (v > 1e-6f && v < 1e6f) ? tot * 10.0f * v : 0.f;
    }
}

// losses.py:21-43
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ color, const float* __restrict__ rgbs, int64_t R,
                                                       const float* __restrict__ ge, const float* __restrict__ me, int64_t n_mask,
                                                       const float* __restrict__ sfm, int64_t n_sfm, float coef, float igr_w,
                                                       float mask_w, float depth_w, float* __restrict__ loss) {
    __shared__ float sm[256];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int64_t i = threadIdx.x; i < R * 3; i += 256) a += fabsf(color[i] - rgbs[i]);
    for (int64_t i = threadIdx.x; i < n_mask; i += 256) b += me[i];
    for (int64_t i = threadIdx.x; i < n_sfm; i += 256) c += sfm[i];
    const float sa = block_sum_fixed(a, sm), sb = block_sum_fixed(b, sm), sc = block_sum_fixed(c, sm);
    if (threadIdx.x == 0) {
        float l = coef * (sa / ((float)R + 1e-5f));
        if (ge) l += coef * (igr_w * ge[0]);
        if (n_mask > 0) l += coef * (mask_w * (sb / (float)n_mask));
        if (n_sfm > 0) l += coef * (depth_w * (sc / (float)n_sfm));
        loss[0] = l;
    }
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ d_loss, const float* __restrict__ color,
                                                       const float* __restrict__ rgbs, int64_t R, int64_t n_mask, int64_t n_sfm,
                                                       float coef, float igr_w, float mask_w, float depth_w,
                                                       float* __restrict__ d_color, float* __restrict__ d_ge,
                                                       float* __restrict__ d_me, float* __restrict__ d_sfm) {
    const float g = d_loss[0] * coef;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < R * 3) {
        const float e = color[i] - rgbs[i];
        d_color[i] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * g / ((float)R + 1e-5f);  // torch's sgn at 0 is 0
    }
    if (d_me && i < n_mask) d_me[i] = g * mask_w / (float)n_mask;
    if (d_sfm && i < n_sfm) d_sfm[i] = g * depth_w / (float)n_sfm;
    if (d_ge && i == 0) d_ge[0] = g * igr_w;
}

// Per-ray part of the appearance head's first layer (models/neuconw.py:137-140: e = [f | gamma_4(view dir) | a] -> static_linear_0):
// W[:, col0 : col0 + 27 + n_a] . [gamma_4(d_ray) | a_ray] is the SAME for every sample of a ray, so it is evaluated once per
// ray in fp32 (libm sin / cos, fmaf) and enters the fused colour kernel as a per-ray bias.  One workgroup per ray.
__global__ __launch_bounds__(128) void aux_ray_bias_kernel(const float* __restrict__ w, int ldw, int col0, int n_out,
                                                          const float* __restrict__ rays_d, const float* __restrict__ a, int n_a,
                                                          float* __restrict__ out, int ldo) {
    __shared__ float x[96];
    const int64_t ray = blockIdx.x;
    const int t = threadIdx.x;
    const int K = 27 + n_a;
    if (t < K) {
        float v;
        if (t < 27) {  // gamma_4: [d (3) | sin(2^k d) (3) | cos(2^k d) (3)] k = 0..3  (models/neuconw.py:7-55)
            const float d[3] = {rays_d[ray * 3 + 0], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
            if (t < 3) v = d[t];
            else {
                const int j = t - 3, k = j / 6, rem = j - 6 * k;
                const float arg = d[rem % 3] * (float)(1 << k);
                v = rem >= 3 ? cosf(arg) : sinf(arg);
            }
        } else {
            v = a[ray * n_a + (t - 27)];
        }
        x[t] = v;
    }
    __syncthreads();
    for (int j = t; j < n_out; j += 128) {
        const float* wr = w + (size_t)j * ldw + col0;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(wr[k], x[k], acc);
        out[ray * ldo + j] = acc;
    }
}

}  // namespace

extern "C" int ncw_aux_ray_bias(const float* w, int ldw, int col0, int n_out, const float* rays_d, const float* a, int n_a,
                                int64_t R, float* out, int ldo, void* stream) {
    if (R <= 0) return 0;
    if (!w || !rays_d || !a || !out || n_a < 0 || n_a > 69 || n_out <= 0 || ldo < n_out || col0 < 0 || col0 + 27 + n_a > ldw)
        return NCW_E_BADARG;
    hipLaunchKernelGGL(aux_ray_bias_kernel, dim3((unsigned)R), dim3(128), 0, (hipStream_t)stream, w, ldw, col0, n_out, rays_d, a,
                       n_a, out, ldo);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_ray_prologue(const float* rays, int ncols, int64_t R, const float* origin_host, float radius, float* rays_o,
                                float* rays_d, float* near, float* far, float* depth_gt, float* depth_weight, void* stream) {
    if (R <= 0) return 0;
    if (!rays || ncols < 8 || !origin_host || !rays_o || !rays_d || !near || !far || !depth_gt || !depth_weight || radius == 0.f)
        return NCW_E_BADARG;
    hipLaunchKernelGGL(ray_prologue_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, ncols, R,
                       origin_host[0], origin_host[1], origin_host[2], radius, rays_o, rays_d, near, far, depth_gt, depth_weight);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_inv_s_fwd(const float* variance, float* inv_s, float* s_val, void* stream) {
    if (!variance || !inv_s) return NCW_E_BADARG;
    hipLaunchKernelGGL(inv_s_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, variance, inv_s, s_val);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_inv_s_bwd(const float* d_inv_s_rays, int64_t R, const float* inv_s, float* d_variance, void* stream) {
    if (!d_inv_s_rays || !inv_s || !d_variance || R < 0) return NCW_E_BADARG;
    hipLaunchKernelGGL(inv_s_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_inv_s_rays, R, inv_s, d_variance);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_loss_fwd(const float* color, const float* rgbs, int64_t R, const float* gradient_error, const float* mask_error,
                            int64_t n_mask, const float* sfm, int64_t n_sfm, float coef, float igr_w, float mask_w, float depth_w,
                            float* loss, void* stream) {
    if (!color || !rgbs || !loss || R <= 0 || n_mask < 0 || n_sfm < 0 || (n_mask > 0 && !mask_error) || (n_sfm > 0 && !sfm))
        return NCW_E_BADARG;
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, color, rgbs, R, gradient_error, mask_error,
                       n_mask, sfm, n_sfm, coef, igr_w, mask_w, depth_w, loss);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_loss_bwd(const float* d_loss, const float* color, const float* rgbs, int64_t R, int64_t n_mask, int64_t n_sfm,
                            float coef, float igr_w, float mask_w, float depth_w, float* d_color, float* d_gradient_error,
                            float* d_mask_error, float* d_sfm, void* stream) {
    if (!d_loss || !color || !rgbs || !d_color || R <= 0 || n_mask < 0 || n_sfm < 0) return NCW_E_BADARG;
    int64_t n = R * 3;
    if (n_mask > n) n = n_mask;
    if (n_sfm > n) n = n_sfm;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_loss, color, rgbs, R,
                       n_mask, n_sfm, coef, igr_w, mask_w, depth_w, d_color, d_gradient_error, d_mask_error, d_sfm);
    NCW_CHECK_LAUNCH();
    return 0;
}
