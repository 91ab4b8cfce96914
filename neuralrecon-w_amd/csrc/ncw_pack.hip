// Weight packing / gradient unpacking kernels (descriptor-table driven: ONE launch per step packs
// every Linear of every network).  See ncw_common.h for the fragment order.
//
// Replaces: nn.utils.weight_norm reparametrisation W = g v/|v| (models/neuconw.py:104-105,
// 256-257) and its autograd backward; the skip-layer 1/sqrt(2) (neuconw.py:273) is folded in here.
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

// element index inside a packed [32*rb_out x 32*rb_in] matrix of logical entry (o, k)
__device__ __forceinline__ size_t packed_index(int o, int k, int rb_out, int prec) {
    const int ro = o >> 5, i = o & 31;
    const int rb = k >> 5, kk = k & 31;
    const int h = (kk >> 2) & 1;
    const int r = (kk & 3) + 4 * (kk >> 3);
    const int lane = i + 32 * h;
    if (prec == NCW_PREC_F32) return ((size_t)(rb * 16 + r) * rb_out + ro) * 64 + lane;
    const int t = r >> 3, e = r & 7;
    return (((size_t)(2 * rb + t) * rb_out + ro) * 64 + lane) * 8 + e;
}

__device__ __forceinline__ int find_desc(const int32_t* __restrict__ prefix, int n, int row) {
    int lo = 0, hi = n;  // prefix[lo] <= row < prefix[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= row) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One WAVE per matrix row (4 rows per workgroup): the row norm is a wave reduction, no LDS, no barriers.
__global__ __launch_bounds__(256) void pack_kernel(const NcwPackDesc* __restrict__ descs,
                                                   const int32_t* __restrict__ prefix, int n, int total_rows) {
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grow >= total_rows) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int d = find_desc(prefix, n, grow);
    const NcwPackDesc& D = descs[d];
    const int i = grow - prefix[d];  // row within the descriptor
    const int row = D.row0 + i;
    const int ld = D.ld;
    const float* srow = D.src + (size_t)row * ld;
    float coef = D.scale;
    if (D.g != nullptr) {
        float ss = 0.f;
        for (int c = lane; c < ld; c += 64) ss += srow[c] * srow[c];
        ss = wave_sum(ss);
        coef *= D.g[row] / sqrtf(ss);
    }
    const int o_log = D.drow0 + i;
    const int rb_out = D.rb_out, prec = D.prec, transpose = D.transpose;
    void* dst_w = D.dst_w;
    for (int s = 0; s < D.nseg; ++s) {
        const NcwSeg sg = D.seg[s];
        for (int c = lane; c < sg.ncols; c += 64) {
            const float v = srow[sg.col0 + c] * coef;
            const int k_log = sg.dcol0 + c;
            const int o = transpose ? k_log : o_log;
            const int k = transpose ? o_log : k_log;
            const size_t idx = packed_index(o, k, rb_out, prec);
            if (prec == NCW_PREC_F32) reinterpret_cast<float*>(dst_w)[idx] = v;
            else if (prec == NCW_PREC_F16) {
                const _Float16 hi = (_Float16)v;  // residual: the low half of a split (hi + lo) matrix
                reinterpret_cast<_Float16*>(dst_w)[idx] = D.residual ? (_Float16)(v - (float)hi) : hi;
            } else {
                const __bf16 hi = (__bf16)v;
                reinterpret_cast<__bf16*>(dst_w)[idx] = D.residual ? (__bf16)(v - (float)hi) : hi;
            }
        }
    }
    if (D.dst_b != nullptr && D.bias != nullptr && lane == 0) {
        // packed bias [rb][h][16]: feature f = 32 rb + (r&3) + 8 (r>>2) + 4 h
        const int f = o_log, rb = f >> 5, kk = f & 31;
        const int h = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
        D.dst_b[(rb * 2 + h) * 16 + r] = D.bias[row];
    }
}

__global__ __launch_bounds__(256) void unpack_kernel(const NcwUnpackDesc* __restrict__ descs,
                                                     const int32_t* __restrict__ prefix, int n, int total_rows) {
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grow >= total_rows) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int d = find_desc(prefix, n, grow);
    const NcwUnpackDesc& D = descs[d];
    const int i = grow - prefix[d];
    const int row = D.row0 + i;
    const float* drow = D.dw + (size_t)(D.drow0 + i) * D.ldw;
    const float* vrow = D.src + (size_t)row * D.ld;
    float* out = D.d_src + (size_t)row * D.ld;
    const float gm = (D.grad_mul != 0.f ? D.grad_mul : 1.0f) * (D.grad_mul_dev ? D.grad_mul_dev[0] : 1.0f);
    const float scale = D.scale * gm;
    const int accumulate = D.accumulate;
    if (D.g == nullptr) {
        for (int s = 0; s < D.nseg; ++s) {
            const NcwSeg sg = D.seg[s];
            for (int c = lane; c < sg.ncols; c += 64) {
                const float gval = drow[sg.dcol0 + c] * scale;
                float* o = out + sg.col0 + c;
                *o = accumulate ? *o + gval : gval;
            }
        }
    } else {
        // weight-norm backward on the full source row (segments cover the whole row)
        float ss = 0.f, dot = 0.f;
        for (int s = 0; s < D.nseg; ++s) {
            const NcwSeg sg = D.seg[s];
            for (int c = lane; c < sg.ncols; c += 64) {
                const float v = vrow[sg.col0 + c];
                ss += v * v;
                dot += drow[sg.dcol0 + c] * scale * v;
            }
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float inv = 1.f / sqrtf(ss);
        const float gbar = dot * inv;  // sum(Wbar * v_hat)
        const float gg = D.g[row];
        for (int s = 0; s < D.nseg; ++s) {
            const NcwSeg sg = D.seg[s];
            for (int c = lane; c < sg.ncols; c += 64) {
                const float v = vrow[sg.col0 + c];
                const float gval = gg * inv * (drow[sg.dcol0 + c] * scale - gbar * v * inv);
                float* o = out + sg.col0 + c;
                *o = accumulate ? *o + gval : gval;
            }
        }
        if (lane == 0 && D.d_g != nullptr) D.d_g[row] = accumulate ? D.d_g[row] + gbar : gbar;
    }
    if (lane == 0 && D.d_bias != nullptr && D.db != nullptr) {
        const float b = D.db[D.drow0 + i] * gm;
        D.d_bias[row] = accumulate ? D.d_bias[row] + b : b;
    }
}

extern "C" int ncw_pack_weights(const NcwPackDesc* descs, const int32_t* row_prefix, int n, int total_rows,
                                void* stream) {
    if (n <= 0 || total_rows <= 0) return 0;
    hipLaunchKernelGGL(pack_kernel, dim3((total_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, descs, row_prefix, n,
                       total_rows);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_unpack_grads(const NcwUnpackDesc* descs, const int32_t* row_prefix, int n, int total_rows,
                                void* stream) {
    if (n <= 0 || total_rows <= 0) return 0;
    hipLaunchKernelGGL(unpack_kernel, dim3((total_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, descs, row_prefix, n,
                       total_rows);
    NCW_CHECK_LAUNCH();
    return 0;
}

// ---- stash <-> row-major converters ---------------------------------------------------------------
template <class SE, bool TO_ROWS>
__global__ void stash_rows_kernel(SE* stash, float* rows, int64_t n, int F, int rb) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= n) return;
    const int64_t p = tile * 32 + (lane & 31);
    const int h = lane >> 5;
    for (int b = 0; b < rb; ++b)
        for (int g = 0; g < 4; ++g) {
            SE* s = stash + ((((size_t)tile * rb + b) * 4 + g) * 64 + lane) * 4;
            for (int c = 0; c < 4; ++c) {
                const int f = b * 32 + 8 * g + 4 * h + c;
                if (TO_ROWS) {
                    if (p < n && f < F) rows[p * F + f] = (float)s[c];
                } else {
                    s[c] = (p < n && f < F) ? (SE)rows[p * F + f] : (SE)0.f;
                }
            }
        }
}

extern "C" int ncw_stash_from_rows(int prec, const float* rows, int64_t n, int F, int rb, void* stash, void* stream) {
    if (n <= 0) return 0;
    if (F > 32 * rb) return NCW_E_BADARG;
    const int64_t tiles = (n + 31) / 32;
    dim3 grid((unsigned)((tiles + 3) / 4));
    if (prec == NCW_PREC_F32)
        hipLaunchKernelGGL((stash_rows_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, (float*)stash,
                           const_cast<float*>(rows), n, F, rb);
    else if (prec == NCW_PREC_F16)
        hipLaunchKernelGGL((stash_rows_kernel<_Float16, false>), grid, dim3(256), 0, (hipStream_t)stream, (_Float16*)stash,
                           const_cast<float*>(rows), n, F, rb);
    else
        hipLaunchKernelGGL((stash_rows_kernel<__bf16, false>), grid, dim3(256), 0, (hipStream_t)stream, (__bf16*)stash,
                           const_cast<float*>(rows), n, F, rb);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_stash_to_rows(int prec, const void* stash, int64_t n, int F, int rb, float* rows, void* stream) {
    if (n <= 0) return 0;
    if (F > 32 * rb) return NCW_E_BADARG;
    const int64_t tiles = (n + 31) / 32;
    dim3 grid((unsigned)((tiles + 3) / 4));
    if (prec == NCW_PREC_F32)
        hipLaunchKernelGGL((stash_rows_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           (float*)const_cast<void*>(stash), rows, n, F, rb);
    else if (prec == NCW_PREC_F16)
        hipLaunchKernelGGL((stash_rows_kernel<_Float16, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           (_Float16*)const_cast<void*>(stash), rows, n, F, rb);
    else
        hipLaunchKernelGGL((stash_rows_kernel<__bf16, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           (__bf16*)const_cast<void*>(stash), rows, n, F, rb);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_abi_version(void) { return 18; }

extern "C" int ncw_device_info(char* buf, int buflen) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) {
        if (buf && buflen > 0) buf[0] = 0;
        return 0;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
    if (buf && buflen > 0) {
        int i = 0;
        for (; i < buflen - 1 && p.gcnArchName[i]; ++i) buf[i] = p.gcnArchName[i];
        buf[i] = 0;
    }
    return p.multiProcessorCount;
}
