// Optimiser step over the flat parameter / gradient buffers (trainer.FlatParams): global-norm clip + Adam in ONE
// elementwise launch.  Replaces, for the flat layout, torch.nn.utils.clip_grad_norm_ (train.py:61) followed by
// torch.optim.Adam(eps=1e-7).step() (utils/__init__.py:23-31) -- same arithmetic per element (torch's
// _single_tensor_adam: lerp, mul+addcmul, sqrt/bias_correction2_sqrt + eps, addcdiv); torch's multi-tensor kernels
// split one 2 M-element tensor into 32 workgroups and take ~0.3 ms per step, this takes ~10 us.
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float step_size, float beta1,
                                                        float beta2, float eps, float bc2_sqrt,
                                                        const float* __restrict__ total_norm, float max_norm) {
    float coef = 1.f;
    if (total_norm != nullptr) {
        const float tn = total_norm[0];
        // a non-finite gradient norm (an fp16 overflow, a NaN batch) would turn every parameter and both moments into
        // NaN through coef: the step is skipped instead -- nothing is written (the reference would not recover either)
        if (!(tn < __builtin_inff())) return;
        coef = fminf(max_norm / (tn + 1e-6f), 1.f);  // clip_grad_norm_
    }
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    const float w1 = 1.f - beta1, w2 = 1.f - beta2;
    if (i4 + 4 <= n) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + i4), gg = *reinterpret_cast<f32x4*>(g + i4);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + i4), vv = *reinterpret_cast<f32x4*>(v + i4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gc = gg[c] * coef;
            gg[c] = gc;
            mm[c] = mm[c] + w1 * (gc - mm[c]);
            vv[c] = vv[c] * beta2 + w2 * gc * gc;
            const float denom = sqrtf(vv[c]) / bc2_sqrt + eps;
            pp[c] = pp[c] - step_size * (mm[c] / denom);
        }
        *reinterpret_cast<f32x4*>(p + i4) = pp;
        *reinterpret_cast<f32x4*>(g + i4) = gg;
        *reinterpret_cast<f32x4*>(m + i4) = mm;
        *reinterpret_cast<f32x4*>(v + i4) = vv;
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float gc = g[i] * coef;
            g[i] = gc;
            const float mi = m[i] + w1 * (gc - m[i]);
            const float vi = v[i] * beta2 + w2 * gc * gc;
            m[i] = mi;
            v[i] = vi;
            p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        }
    }
}

extern "C" int ncw_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float step_size,
                             float beta1, float beta2, float eps, float bias_correction2_sqrt, const float* total_norm,
                             float max_norm, void* stream) {
    if (n <= 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !(bias_correction2_sqrt > 0.f)) return NCW_E_BADARG;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return NCW_E_BADARG;
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param,
                       grad, exp_avg, exp_avg_sq, n, step_size, beta1, beta2, eps, bias_correction2_sqrt, total_norm,
                       max_norm);
    NCW_CHECK_LAUNCH();
    return 0;
}
