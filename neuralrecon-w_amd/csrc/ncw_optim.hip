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

// ---- device-resident optimiser state (NcwAdamState): applied-step counter, bias corrections, clip coefficient and the
// fp16 loss scale all live on the device, so (a) a skipped (non-finite) step does not advance the bias-correction
// step, (b) the loss scale backs off / grows without a device->host round trip, (c) the whole update is HIP-graph
// capturable (nothing step-dependent is a kernel argument).
__global__ void adam_prep_kernel(NcwAdamState* __restrict__ st, const float* __restrict__ total_norm, float lr,
                                 const float* __restrict__ lr_dev, float beta1, float beta2, float max_norm,
                                 float* __restrict__ loss_scale, int growth_interval, float scale_min, float scale_max) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float tn = total_norm ? total_norm[0] : 0.f;
    const bool finite = tn < __builtin_inff() && tn == tn;
    if (!finite) {
        st->skip_now = 1;
        st->skipped += 1;
        st->good = 0;
        if (loss_scale) {  // back off: halve (a power of two stays exact), never below scale_min
            const float s = fmaxf(loss_scale[0] * 0.5f, scale_min);
            loss_scale[0] = s;
            loss_scale[1] = 1.f / s;
        }
        return;
    }
    st->skip_now = 0;
    const int t = st->step + 1;
    st->step = t;
    st->good += 1;
    if (loss_scale && growth_interval > 0 && st->good >= growth_interval) {
        const float s = fminf(loss_scale[0] * 2.f, scale_max);
        loss_scale[0] = s;
        loss_scale[1] = 1.f / s;
        st->good = 0;
    }
    const double l = lr_dev ? (double)lr_dev[0] : (double)lr;
    // torch's non-capturable path computes both corrections in double on the host (_single_tensor_adam)
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    st->step_size = (float)(l / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
    st->coef = (total_norm && max_norm > 0.f) ? fminf(max_norm / (tn + 1e-6f), 1.f) : 1.f;  // clip_grad_norm_
    st->last_norm = tn;
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, float beta1, float beta2, float eps,
                                                       const NcwAdamState* __restrict__ st) {
    if (st->skip_now) return;  // nothing is written: one overflowed fp16 step must not poison parameters and moments
    const float coef = st->coef, step_size = st->step_size, bc2_sqrt = st->bc2_sqrt;
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

// ---- 2-norm of the flat gradient buffer (the `total_norm` of clip_grad_norm_, train.py:61) as ONE launch with a FIXED
// summation order: NORM_BLOCKS blocks reduce contiguous, 16-byte-aligned chunks (lane-strided f32x4 loads, a wavefront shuffle
// tree, the four wave partials through LDS) into scratch[block]; the last block to arrive (a device ticket) adds the partials in
// block order, takes the square root and re-arms the ticket.  Run-to-run reproducible (no float atomics), non-finite inputs
// propagate (the fp16 loss-scale guard of adam_prep_kernel reads the result).
constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void grad_norm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ scratch,
                                                        float* __restrict__ norm) {
    __shared__ float wsum[4];
    __shared__ int last;
    const int64_t quads = (n + 3) / 4;
    const int64_t per = (quads + NORM_BLOCKS - 1) / NORM_BLOCKS;
    const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < quads ? q0 + per : quads;
    float acc = 0.f;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += 256) {
        const int64_t i = q * 4;
        if (i + 4 <= n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
            acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
            for (int64_t j = i; j < n; ++j) acc += g[j] * g[j];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        __threadfence();
        unsigned* ticket = reinterpret_cast<unsigned*>(scratch + NORM_BLOCKS);
        last = (atomicAdd(ticket, 1u) == NORM_BLOCKS - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the 256 partials in block order: lane-strided pairs, then the same shuffle tree (fixed order)
    float t = __builtin_nontemporal_load(scratch + threadIdx.x);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        norm[0] = sqrtf((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    }
}

extern "C" int64_t ncw_grad_norm_scratch_floats(void) { return NORM_BLOCKS + 4; }

extern "C" int ncw_grad_norm(const float* grad, int64_t n, float* scratch, float* norm, void* stream) {
    if (!grad || !scratch || !norm || n < 0 || ((uintptr_t)grad & 15) != 0) return NCW_E_BADARG;
    // The arrival ticket is armed by a stream-ordered memset in front of EVERY launch (capture-safe: a memset node): a launch that
    // was aborted before its last block arrived (device fault, killed graph replay) cannot leave a stale count behind.  One
    // scratch buffer belongs to one stream at a time.
    if (hipMemsetAsync(scratch + NORM_BLOCKS, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return NCW_E_BADARG;
    hipLaunchKernelGGL(grad_norm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, (hipStream_t)stream, grad, n, scratch, norm);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, NcwAdamState* state,
                                 const float* total_norm, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                                 float max_norm, float* loss_scale, int growth_interval, float scale_min, float scale_max,
                                 void* stream) {
    if (n <= 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state) return NCW_E_BADARG;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return NCW_E_BADARG;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, total_norm, lr, lr_dev, beta1, beta2,
                       max_norm, loss_scale, growth_interval, scale_min, scale_max);
    NCW_CHECK_LAUNCH();
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, beta1, beta2, eps, state);
    NCW_CHECK_LAUNCH();
    return 0;
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
