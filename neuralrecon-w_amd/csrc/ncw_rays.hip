// Per-ray kernels of the NeuS-style sampler and compositor (rendering/renderer.py), one WAVE per
// ray: elementwise math across the 64 lanes, transmittance / CDF as wavefront scans, the
// inverse-CDF lookup and the sorted merges as rank computations in LDS.  HBM-bound scan/gather
// work (~60 B per ray-sample): the point is few launches and coalesced rows, not MFMA.
//
//   ncw_sample_coarse   renderer.py:488-514   coarse z, outside z, sample_dist (+ perturb)
//   ncw_upsample        renderer.py:257-341 + sample_pdf :15-48
//   ncw_sort_merge      renderer.py:343-363 (cat_z_vals), :566, :835-836
//   ncw_boundary        renderer.py:549-565
//   ncw_composite_fwd / _bwd   renderer.py:205-216 (bg alpha), :586-783 (render_core tail)
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

// A ray's samples live in LDS (one wave per ray).  The file is compiled TWICE (neuralrecon-w_amd/build.py): the standard object
// handles up to 512 samples per ray (two workgroups per CU in the compositor backward) and forwards larger rays -- up to 1088: the
// reference's own defaults, config/defaults.py:8-9,32: 512 + 512 samples + 32 outside -- to the `_big` entry points of the second
// object (-DNCW_RAYS_BIG: the same kernels with 17 elements per lane, 122 KB of LDS in the compositor backward).
#ifdef NCW_RAYS_BIG
#define RAY_MAXN 1088
#define NCW_RAYNS ncw_rays_big
#define NCW_RAYFN(name) name##_big
#else
#define RAY_MAXN 512            // max samples per ray handled in LDS by this object
#define NCW_RAYNS ncw_rays_std
#define NCW_RAYFN(name) name
#endif
#define RAY_MAXN_BIG 1088
#define RAY_CH (RAY_MAXN / 64)  // elements per lane in a chunked scan

namespace NCW_RAYNS {

NCW_DEV float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// torch.linspace(start, end, steps)[i]  (ATen RangeFactories: symmetric evaluation)
NCW_DEV float torch_linspace(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// Exclusive scan over a[0..m) in LDS (one wave), op = multiply (MUL) or add; out may alias a.
// Returns the total (product / sum of all m elements) in every lane.
template <bool MUL>
NCW_DEV float wave_excl_scan(const float* a, float* out, int m, int lane) {
    const int per = (m + 63) >> 6;
    const int b = lane * per;
    float loc[RAY_CH];
    float tot = MUL ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < RAY_CH; ++k) {
        if (k < per) {
            const int i = b + k;
            loc[k] = (i < m) ? a[i] : (MUL ? 1.f : 0.f);
            tot = MUL ? tot * loc[k] : tot + loc[k];
        }
    }
    // inclusive scan of lane totals
    float inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc = MUL ? inc * t : inc + t;
    }
    float run = __shfl_up(inc, 1, 64);
    if (lane == 0) run = MUL ? 1.f : 0.f;
    const float total = __shfl(inc, 63, 64);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < RAY_CH; ++k) {
        if (k < per) {
            const int i = b + k;
            if (i < m) out[i] = run;
            run = MUL ? run * loc[k] : run + loc[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    return total;
}

NCW_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// coarse samples -- renderer.py:488-514.  near/far: ray near/far (after an optional coarse-octree
// override); s_near/s_far: sampling window (== near/far without a fine octree).
// ------------------------------------------------------------------------------------------------
__global__ void sample_coarse_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                     const float* __restrict__ s_near, const float* __restrict__ s_far, int R,
                                     int n_samples, int n_out, const float* __restrict__ rand_shift,
                                     const float* __restrict__ rand_out, float* __restrict__ z,
                                     float* __restrict__ z_out, float* __restrict__ sample_dist) {
    const int r = blockIdx.x * blockDim.y + threadIdx.y;
    if (r >= R) return;
    const float sn = s_near[r], sf = s_far[r], fr = far[r];
    for (int i = threadIdx.x; i < n_samples; i += blockDim.x) {
        float v = sn + (sf - sn) * torch_linspace(0.f, 1.f, n_samples, i);
        if (rand_shift) v = v + (sf - sn) * (rand_shift[r] - 0.5f) * 2.0f / (float)n_samples;
        z[(size_t)r * n_samples + i] = v;
    }
    if (threadIdx.x == 0) sample_dist[r] = (sf - sn) / (float)n_samples;
    const float oe = 1.0f - 1.0f / ((float)n_out + 1.0f);
    for (int j = threadIdx.x; j < n_out; j += blockDim.x) {
        // element j of the flipped array = element q = n_out-1-j of z_vals_outside
        const int q = n_out - 1 - j;
        float zo = torch_linspace(1e-3f, oe, n_out, q);
        if (rand_out) {
            const float cur = zo;
            const float nxt = (q + 1 < n_out) ? torch_linspace(1e-3f, oe, n_out, q + 1) : cur;
            const float prv = (q > 0) ? torch_linspace(1e-3f, oe, n_out, q - 1) : cur;
            const float upper = (q + 1 < n_out) ? 0.5f * (nxt + cur) : cur;
            const float lower = (q > 0) ? 0.5f * (cur + prv) : cur;
            zo = lower + (upper - lower) * rand_out[(size_t)r * n_out + q];
        }
        z_out[(size_t)r * n_out + j] = fr / zo + 1.0f / (float)n_samples;
    }
}

// ------------------------------------------------------------------------------------------------
// up_sample + sample_pdf(det=True)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ z, const float* __restrict__ sdf,
                                                       int R, int n, float inv_s, int n_new,
                                                       float* __restrict__ z_new) {
    __shared__ float sm[4][4][RAY_MAXN];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    float* zs = sm[wv][0];
    float* sd = sm[wv][1];
    float* t0 = sm[wv][2];  // radius -> cos -> alpha -> cdf
    float* t1 = sm[wv][3];
    const float ox = rays_o[r * 3], oy = rays_o[r * 3 + 1], oz = rays_o[r * 3 + 2];
    const float dx = rays_d[r * 3], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
    const int m = n - 1;
    for (int i = lane; i < n; i += 64) {
        const float zz = z[(size_t)r * n + i];
        zs[i] = zz;
        sd[i] = sdf[(size_t)r * n + i];
        const float px = ox + dx * zz, py = oy + dy * zz, pz = oz + dz * zz;
        t0[i] = sqrtf(px * px + py * py + pz * pz);
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < m; i += 64) t1[i] = (sd[i + 1] - sd[i]) / (zs[i + 1] - zs[i] + 1e-5f);
    __builtin_amdgcn_wave_barrier();
    float alpha_l[RAY_CH];
#pragma unroll
    for (int k = 0; k < RAY_CH; ++k) {
        const int i = lane + 64 * k;
        alpha_l[k] = 0.f;
        if (i < m) {
            const float pc0 = (i > 0) ? t1[i - 1] : 0.f;
            float cv = fminf(pc0, t1[i]);
            cv = fminf(fmaxf(cv, -1e3f), 0.f);
            const bool inside = (t0[i] < 1.0f) || (t0[i + 1] < 1.0f);
            cv = inside ? cv : 0.f;
            const float mid = (sd[i] + sd[i + 1]) * 0.5f;
            const float dist = zs[i + 1] - zs[i];
            const float pe = mid - cv * dist * 0.5f, ne = mid + cv * dist * 0.5f;
            const float pc = sigmoid_acc(pe * inv_s), nc = sigmoid_acc(ne * inv_s);
            alpha_l[k] = (pc - nc + 1e-5f) / (pc + 1e-5f);
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < RAY_CH; ++k) {
        const int i = lane + 64 * k;
        if (i < m) t0[i] = 1.0f - alpha_l[k] + 1e-7f;
    }
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<true>(t0, t0, m, lane);  // t0 = T_i
    float wsum = 0.f;
#pragma unroll
    for (int k = 0; k < RAY_CH; ++k) {
        const int i = lane + 64 * k;
        if (i < m) {
            const float w = alpha_l[k] * t0[i] + 1e-5f;  // sample_pdf: weights + 1e-5
            t1[i] = w;
            wsum += w;
        }
    }
    wsum = wave_sum(wsum);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < m; i += 64) t1[i] = t1[i] / wsum;  // pdf
    __builtin_amdgcn_wave_barrier();
    // cdf[0] = 0, cdf[i+1] = inclusive cumsum -> exclusive scan over m+1 slots of [pdf..., x]
    if (lane == 0) t1[m] = 0.f;
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<false>(t1, t0, m + 1, lane);  // t0[0..n) = cdf
    for (int j = lane; j < n_new; j += 64) {
        const float u = torch_linspace(0.5f / (float)n_new, 1.0f - 0.5f / (float)n_new, n_new, j);
        int lo = 0, hi = n;  // count of cdf entries <= u  (searchsorted right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (t0[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, n - 1);
        const float cb = t0[below], ca = t0[above];
        float denom = ca - cb;
        denom = (denom < 1e-5f) ? 1.0f : denom;
        const float t = (u - cb) / denom;
        z_new[(size_t)r * n_new + j] = zs[below] + t * (zs[above] - zs[below]);
    }
}

// ------------------------------------------------------------------------------------------------
// stable sort of cat([a, b]) per ray with an optional payload (sdf) -- rank by counting in LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_merge_kernel(const float* __restrict__ a, int na, const float* __restrict__ b,
                                                         int nb, const float* __restrict__ pa,
                                                         const float* __restrict__ pb, int R, float* __restrict__ out,
                                                         float* __restrict__ pout) {
    __shared__ float sm[4][RAY_MAXN];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;
    const int n = na + nb;
    float* c = sm[wv];
    for (int i = lane; i < n; i += 64) c[i] = (i < na) ? a[(size_t)r * na + i] : b[(size_t)r * nb + (i - na)];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {
        const float v = c[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float w = c[j];
            rank += (w < v || (w == v && j < i)) ? 1 : 0;
        }
        out[(size_t)r * n + rank] = v;
        if (pout) pout[(size_t)r * n + rank] = (i < na) ? pa[(size_t)r * na + i] : pb[(size_t)r * nb + (i - na)];
    }
}

// boundary samples -- renderer.py:549-565: nb//2 uniform in [near, z_first), rest in (z_last, far]
__global__ void boundary_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                const float* __restrict__ z, int n, int R, int nb, float* __restrict__ zb) {
    const int r = blockIdx.x * blockDim.y + threadIdx.y;
    if (r >= R) return;
    const int n_near = nb / 2, n_far = nb - n_near;
    const float nr = near[r], fr = far[r], z0 = z[(size_t)r * n], z1 = z[(size_t)r * n + n - 1];
    for (int j = threadIdx.x; j < nb; j += blockDim.x) {
        float v;
        if (j < n_near) v = nr + (z0 - nr) * torch_linspace(0.f, 1.f, n_near + 1, j);
        else v = z1 + (fr - z1) * torch_linspace(0.f, 1.f, n_far + 1, j - n_near + 1);
        zb[(size_t)r * nb + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Compositor forward.  Per ray: S inside samples (+ O outside, bg arrays have S+O columns).
// ------------------------------------------------------------------------------------------------
struct CompArgs {
    const float *rays_o, *rays_d, *z, *z_feed, *sample_dist, *sdf, *grad, *rgb, *density, *bg_rgb, *inv_s;
    const float* background_rgb;  // [3] or null
    const float* cos_anneal_dev;  // [1] or null
    float cos_anneal;
    int R, S, O, has_bg, trim_sphere;
};
struct CompOut {
    float *color, *color_sphere, *color_bg, *weights, *weights_sum, *cdf, *inside, *depth, *normals, *eik;
    float *mid_z, *dists, *bg_alpha, *weights_max;
};

NCW_DEV float iter_cos_of(float tc, float c) {
    return -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - c) + fmaxf(-tc, 0.f) * c);
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs A, CompOut Q) {
    __shared__ float sm[4][4][RAY_MAXN];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wv;
    if (r >= A.R) return;
    const int S = A.S, M = A.has_bg ? A.S + A.O : A.S;
    float* al0 = sm[wv][0];  // unmasked alpha (inside samples)
    float* alm = sm[wv][1];  // merged alpha
    float* tA = sm[wv][2];
    float* tB = sm[wv][3];
    const float ox = A.rays_o[r * 3], oy = A.rays_o[r * 3 + 1], oz = A.rays_o[r * 3 + 2];
    const float dx = A.rays_d[r * 3], dy = A.rays_d[r * 3 + 1], dz = A.rays_d[r * 3 + 2];
    const float inv_s = A.inv_s[0], sdist = A.sample_dist[r];
    const float* zr = A.z + (size_t)r * S;
    float eik_num = 0.f, eik_den = 0.f;
    // ---- per-sample quantities -------------------------------------------------------------
    for (int i = lane; i < S; i += 64) {
        const float zi = zr[i];
        const float dist = (i + 1 < S) ? zr[i + 1] - zi : sdist;
        const float mid = zi + dist * 0.5f;
        const float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
        const float pn = sqrtf(px * px + py * py + pz * pz);
        const size_t q = (size_t)r * S + i;
        const float gx = A.grad[q * 3], gy = A.grad[q * 3 + 1], gz = A.grad[q * 3 + 2];
        const float tc = dx * gx + dy * gy + dz * gz;
        const float ic = iter_cos_of(tc, A.cos_anneal_dev ? A.cos_anneal_dev[0] : A.cos_anneal);
        const float sd = A.sdf[q];
        const float en = sd + ic * dist * 0.5f, ep = sd - ic * dist * 0.5f;
        const float pc = sigmoid_acc(ep * inv_s), nc = sigmoid_acc(en * inv_s);
        float al = (pc - nc + 1e-5f) / (pc + 1e-5f);
        al = fminf(fmaxf(al, 0.f), 1.f);
        const float inside = pn < 1.0f ? 1.f : 0.f;
        const float relax = pn < 1.2f ? 1.f : 0.f;
        al0[i] = al;
        Q.cdf[q] = pc;
        Q.inside[q] = inside;
        Q.mid_z[q] = mid;
        Q.dists[q] = dist;
        const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
        eik_num += relax * (gn - 1.0f) * (gn - 1.0f);
        eik_den += relax;
    }
    // ---- background alpha (renderer.py:205-207) ---------------------------------------------
    if (A.has_bg) {
        const float* zf = A.z_feed + (size_t)r * M;
        for (int j = lane; j < M; j += 64) {
            const float dist = (j + 1 < M) ? zf[j + 1] - zf[j] : sdist;
            const float den = A.density[(size_t)r * M + j];
            const float sp = den > 20.f ? den : log1pf(expf(den));  // F.softplus, threshold 20
            const float ba = 1.0f - expf(-sp * dist);
            tB[j] = ba;
            Q.bg_alpha[(size_t)r * M + j] = ba;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- depth from the un-masked alpha (renderer.py:641, 365-378) ----------------------------
    for (int i = lane; i < S; i += 64) tA[i] = 1.0f - al0[i] + 1e-7f;
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<true>(tA, tA, S, lane);
    float depth = 0.f;
    for (int i = lane; i < S; i += 64) depth += al0[i] * tA[i] * Q.mid_z[(size_t)r * S + i];
    depth = wave_sum(depth);
    // ---- merged alpha ---------------------------------------------------------------------
    for (int j = lane; j < M; j += 64) {
        float a;
        if (j < S) {
            const float ins = Q.inside[(size_t)r * S + j];
            a = ins > 0.f ? al0[j] : (A.has_bg ? tB[j] : 0.f);
        } else a = tB[j];
        alm[j] = a;
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < M; j += 64) tA[j] = 1.0f - alm[j] + 1e-7f;
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<true>(tA, tA, M, lane);  // tA = T merged
    float cr = 0.f, cg = 0.f, cb = 0.f, wsum = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, wmax = -3.4e38f;
    for (int j = lane; j < M; j += 64) {
        const float w = alm[j] * tA[j];
        Q.weights[(size_t)r * M + j] = w;
        wmax = fmaxf(wmax, w);
        float rr, gg, bb;
        bool ins = false;
        if (j < S) ins = Q.inside[(size_t)r * S + j] > 0.f;
        if (ins) {
            const size_t q = ((size_t)r * S + j) * 3;
            rr = A.rgb[q]; gg = A.rgb[q + 1]; bb = A.rgb[q + 2];
        } else if (A.has_bg) {
            const size_t q = ((size_t)r * M + j) * 3;
            rr = A.bg_rgb[q]; gg = A.bg_rgb[q + 1]; bb = A.bg_rgb[q + 2];
        } else rr = gg = bb = 0.f;
        cr += rr * w; cg += gg * w; cb += bb * w;
        if (j < S) {
            if (ins) wsum += w;
            const size_t q = ((size_t)r * S + j) * 3;
            nx += A.grad[q] * w; ny += A.grad[q + 1] * w; nz += A.grad[q + 2] * w;
        }
    }
    cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); wsum = wave_sum(wsum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
    nx = wave_sum(nx); ny = wave_sum(ny); nz = wave_sum(nz);
    if (A.background_rgb) {
        cr += A.background_rgb[0] * (1.0f - wsum);
        cg += A.background_rgb[1] * (1.0f - wsum);
        cb += A.background_rgb[2] * (1.0f - wsum);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- sphere-only colour (renderer.py:735-746) ---------------------------------------------
    for (int i = lane; i < S; i += 64) tA[i] = 1.0f - al0[i] * Q.inside[(size_t)r * S + i] + 1e-7f;
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<true>(tA, tA, S, lane);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    for (int i = lane; i < S; i += 64) {
        const float ins = Q.inside[(size_t)r * S + i];
        const float w = al0[i] * ins * tA[i];
        const size_t q = ((size_t)r * S + i) * 3;
        sr += A.rgb[q] * ins * w; sg += A.rgb[q + 1] * ins * w; sb += A.rgb[q + 2] * ins * w;
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb);
    // ---- background-only colour with the sphere trimmed (renderer.py:705-720) -----------------
    float br = 0.f, bgc = 0.f, bb2 = 0.f;
    if (A.has_bg) {
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < M; j += 64) {
            float a = tB[j];
            if (A.trim_sphere && j < S && Q.inside[(size_t)r * S + j] > 0.f) a = 0.f;
            alm[j] = a;
            tA[j] = 1.0f - a + 1e-7f;
        }
        __builtin_amdgcn_wave_barrier();
        wave_excl_scan<true>(tA, tA, M, lane);
        for (int j = lane; j < M; j += 64) {
            const float w = alm[j] * tA[j];
            const size_t q = ((size_t)r * M + j) * 3;
            br += A.bg_rgb[q] * w; bgc += A.bg_rgb[q + 1] * w; bb2 += A.bg_rgb[q + 2] * w;
        }
        br = wave_sum(br); bgc = wave_sum(bgc); bb2 = wave_sum(bb2);
    }
    eik_num = wave_sum(eik_num);
    eik_den = wave_sum(eik_den);
    if (lane == 0) {
        Q.color[r * 3] = cr; Q.color[r * 3 + 1] = cg; Q.color[r * 3 + 2] = cb;
        Q.color_sphere[r * 3] = sr; Q.color_sphere[r * 3 + 1] = sg; Q.color_sphere[r * 3 + 2] = sb;
        Q.color_bg[r * 3] = br; Q.color_bg[r * 3 + 1] = bgc; Q.color_bg[r * 3 + 2] = bb2;
        Q.weights_sum[r] = wsum;
        if (Q.weights_max) Q.weights_max[r] = wmax;
        Q.depth[r] = depth;
        Q.normals[r * 3] = nx; Q.normals[r * 3 + 1] = ny; Q.normals[r * 3 + 2] = nz;
        Q.eik[r] = eik_num; Q.eik[A.R + r] = eik_den;  // [2,R]: two contiguous per-ray arrays
    }
}

// ------------------------------------------------------------------------------------------------
// Compositor backward (SURVEY 8a-12 backward contract).  Upstream: d_color[R,3], d_wsum[R],
// d_depth[R], d_eik_num[R].  Reverse scans:  dL/dalpha_i = T_i wbar_i - (sum_{k>i} w_k wbar_k)/(1-alpha_i+eps)
// ------------------------------------------------------------------------------------------------
struct CompBwd {
    const float *d_color, *d_wsum, *d_depth, *d_eik;
    float *d_sdf, *d_grad, *d_rgb, *d_density, *d_bg_rgb, *d_inv_s;
    float grad_scale;  // every upstream cotangent is multiplied by it on load (fp16 loss scaling; a power of two is exact)
    const float* grad_scale_dev;  // dynamic part of the scale (device scalar) or nullptr
};

// suffix-exclusive sum: out[i] = sum_{k>i} a[k]
NCW_DEV void wave_suffix_excl_sum(const float* a, float* out, int m, int lane) {
    const int per = (m + 63) >> 6;
    const int b = lane * per;
    float loc[RAY_CH];
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < RAY_CH; ++k)
        if (k < per) {
            const int i = b + k;
            loc[k] = (i < m) ? a[i] : 0.f;
            tot += loc[k];
        }
    float inc = tot;  // inclusive suffix scan over lanes (towards higher lanes)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(inc, o, 64);
        if (lane + o < 64) inc += t;
    }
    float run = __shfl_down(inc, 1, 64);
    if (lane == 63) run = 0.f;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = RAY_CH - 1; k >= 0; --k)
        if (k < per) {
            const int i = b + k;
            if (i < m) out[i] = run;
            run += loc[k];
        }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs A, CompBwd G) {
    __shared__ float sm[4][5][RAY_MAXN];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wv;
    if (r >= A.R) return;
    const int S = A.S, M = A.has_bg ? A.S + A.O : A.S;
    float* al0 = sm[wv][0];
    float* alm = sm[wv][1];
    float* T = sm[wv][2];
    float* wk = sm[wv][3];
    float* dal0 = sm[wv][4];  // accumulates dL/d alpha0
    const float ox = A.rays_o[r * 3], oy = A.rays_o[r * 3 + 1], oz = A.rays_o[r * 3 + 2];
    const float dx = A.rays_d[r * 3], dy = A.rays_d[r * 3 + 1], dz = A.rays_d[r * 3 + 2];
    const float inv_s = A.inv_s[0], sdist = A.sample_dist[r], c = A.cos_anneal_dev ? A.cos_anneal_dev[0] : A.cos_anneal;
    const float* zr = A.z + (size_t)r * S;
    const float gs = G.grad_scale_dev ? G.grad_scale * G.grad_scale_dev[0] : G.grad_scale;
    const float dcr = G.d_color[r * 3] * gs, dcg = G.d_color[r * 3 + 1] * gs, dcb = G.d_color[r * 3 + 2] * gs;
    float dws = G.d_wsum[r] * gs;
    if (A.background_rgb)
        dws -= dcr * A.background_rgb[0] + dcg * A.background_rgb[1] + dcb * A.background_rgb[2];
    const float ddep = G.d_depth[r] * gs, deik = G.d_eik[r] * gs;
    // recompute alpha0, bg alpha
    for (int i = lane; i < S; i += 64) {
        const float zi = zr[i];
        const float dist = (i + 1 < S) ? zr[i + 1] - zi : sdist;
        const size_t q = (size_t)r * S + i;
        const float gx = A.grad[q * 3], gy = A.grad[q * 3 + 1], gz = A.grad[q * 3 + 2];
        const float tc = dx * gx + dy * gy + dz * gz;
        const float ic = iter_cos_of(tc, c);
        const float sd = A.sdf[q];
        const float pc = sigmoid_acc((sd - ic * dist * 0.5f) * inv_s), nc = sigmoid_acc((sd + ic * dist * 0.5f) * inv_s);
        const float raw = (pc - nc + 1e-5f) / (pc + 1e-5f);
        al0[i] = fminf(fmaxf(raw, 0.f), 1.f);
        dal0[i] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- chain B: depth ------------------------------------------------------------------
    for (int i = lane; i < S; i += 64) T[i] = 1.0f - al0[i] + 1e-7f;
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<true>(T, T, S, lane);
    for (int i = lane; i < S; i += 64) {
        const float zi = zr[i];
        const float dist = (i + 1 < S) ? zr[i + 1] - zi : sdist;
        const float wb = ddep * (zi + dist * 0.5f);
        wk[i] = al0[i] * T[i] * wb;  // w_k * wbar_k
    }
    __builtin_amdgcn_wave_barrier();
    wave_suffix_excl_sum(wk, wk, S, lane);
    for (int i = lane; i < S; i += 64) {
        const float zi = zr[i];
        const float dist = (i + 1 < S) ? zr[i + 1] - zi : sdist;
        dal0[i] += T[i] * ddep * (zi + dist * 0.5f) - wk[i] / (1.0f - al0[i] + 1e-7f);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- chain A: merged alpha -> colour, weights_sum ------------------------------------------
    float* bga = wk;  // reuse after chain B finished: bg alpha into a private array
    __shared__ float smb[4][2][RAY_MAXN];
    float* ba = smb[wv][0];
    float* wbar = smb[wv][1];
    (void)bga;
    if (A.has_bg) {
        const float* zf = A.z_feed + (size_t)r * M;
        for (int j = lane; j < M; j += 64) {
            const float dist = (j + 1 < M) ? zf[j + 1] - zf[j] : sdist;
            const float den = A.density[(size_t)r * M + j];
            const float sp = den > 20.f ? den : log1pf(expf(den));
            ba[j] = 1.0f - expf(-sp * dist);
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < M; j += 64) {
        bool ins = false;
        if (j < S) {
            const float zi = zr[j];
            const float dist = (j + 1 < S) ? zr[j + 1] - zi : sdist;
            const float mid = zi + dist * 0.5f;
            const float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
            ins = sqrtf(px * px + py * py + pz * pz) < 1.0f;
        }
        const float a = ins ? al0[j] : (A.has_bg ? ba[j] : 0.f);
        alm[j] = a;
        T[j] = 1.0f - a + 1e-7f;
        float rr, gg, bb;
        if (ins) {
            const size_t q = ((size_t)r * S + j) * 3;
            rr = A.rgb[q]; gg = A.rgb[q + 1]; bb = A.rgb[q + 2];
        } else if (A.has_bg) {
            const size_t q = ((size_t)r * M + j) * 3;
            rr = A.bg_rgb[q]; gg = A.bg_rgb[q + 1]; bb = A.bg_rgb[q + 2];
        } else rr = gg = bb = 0.f;
        wbar[j] = dcr * rr + dcg * gg + dcb * bb + (ins ? dws : 0.f);
    }
    __builtin_amdgcn_wave_barrier();
    wave_excl_scan<true>(T, T, M, lane);
    for (int j = lane; j < M; j += 64) wk[j] = alm[j] * T[j] * wbar[j];
    __builtin_amdgcn_wave_barrier();
    float* suf = wk;
    // keep w (for d_rgb) before overwriting: recompute from alm*T
    wave_suffix_excl_sum(wk, suf, M, lane);
    for (int j = lane; j < M; j += 64) {
        const float w = alm[j] * T[j];
        const float dalpha = T[j] * wbar[j] - suf[j] / (1.0f - alm[j] + 1e-7f);
        bool ins = false;
        if (j < S) {
            const float zi = zr[j];
            const float dist = (j + 1 < S) ? zr[j + 1] - zi : sdist;
            const float mid = zi + dist * 0.5f;
            const float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
            ins = sqrtf(px * px + py * py + pz * pz) < 1.0f;
        }
        if (j < S) {
            const size_t q = ((size_t)r * S + j) * 3;
            G.d_rgb[q] = ins ? dcr * w : 0.f;
            G.d_rgb[q + 1] = ins ? dcg * w : 0.f;
            G.d_rgb[q + 2] = ins ? dcb * w : 0.f;
            if (ins) dal0[j] += dalpha;
        }
        if (A.has_bg) {
            const size_t q = ((size_t)r * M + j) * 3;
            G.d_bg_rgb[q] = ins ? 0.f : dcr * w;
            G.d_bg_rgb[q + 1] = ins ? 0.f : dcg * w;
            G.d_bg_rgb[q + 2] = ins ? 0.f : dcb * w;
            // bg_alpha = 1 - exp(-softplus(den) dist): d/d den = exp(-sp dist) dist sigmoid(den)
            const float* zf = A.z_feed + (size_t)r * M;
            const float dist = (j + 1 < M) ? zf[j + 1] - zf[j] : sdist;
            const float den = A.density[(size_t)r * M + j];
            const float sg = den > 20.f ? 1.f : sigmoid_acc(den);
            const float dba = ins ? 0.f : dalpha;
            G.d_density[(size_t)r * M + j] = dba * (1.0f - ba[j]) * dist * sg;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- through alpha0 -> sdf, grad, inv_s; plus the eikonal term ------------------------------
    float ds_acc = 0.f;
    for (int i = lane; i < S; i += 64) {
        const float zi = zr[i];
        const float dist = (i + 1 < S) ? zr[i + 1] - zi : sdist;
        const float mid = zi + dist * 0.5f;
        const float px = ox + dx * mid, py = oy + dy * mid, pz = oz + dz * mid;
        const float pn = sqrtf(px * px + py * py + pz * pz);
        const float relax = pn < 1.2f ? 1.f : 0.f;
        const size_t q = (size_t)r * S + i;
        const float gx = A.grad[q * 3], gy = A.grad[q * 3 + 1], gz = A.grad[q * 3 + 2];
        const float tc = dx * gx + dy * gy + dz * gz;
        const float ic = iter_cos_of(tc, c);
        const float sd = A.sdf[q];
        const float ep = sd - ic * dist * 0.5f, en = sd + ic * dist * 0.5f;
        const float pc = sigmoid_acc(ep * inv_s), nc = sigmoid_acc(en * inv_s);
        const float raw = (pc - nc + 1e-5f) / (pc + 1e-5f);
        const float draw = (raw >= 0.f && raw <= 1.f) ? dal0[i] : 0.f;
        const float dpc = draw * (nc / ((pc + 1e-5f) * (pc + 1e-5f)));
        const float dnc = -draw / (pc + 1e-5f);
        const float dep_ = dpc * pc * (1.0f - pc), den_ = dnc * nc * (1.0f - nc);  // d/d(arg of sigmoid)
        ds_acc += dep_ * ep + den_ * en;
        const float dep = dep_ * inv_s, dne = den_ * inv_s;
        const float dic = (dne - dep) * dist * 0.5f;
        const float dtc = dic * (0.5f * (1.0f - c) * (tc < 1.0f ? 1.f : 0.f) + c * (tc < 0.f ? 1.f : 0.f));
        const float gn = sqrtf(gx * gx + gy * gy + gz * gz);
        const float ek = (gn > 0.f) ? deik * relax * 2.0f * (gn - 1.0f) / gn : 0.f;
        G.d_sdf[q] = dep + dne;
        G.d_grad[q * 3] = dtc * dx + ek * gx;
        G.d_grad[q * 3 + 1] = dtc * dy + ek * gy;
        G.d_grad[q * 3 + 2] = dtc * dz + ek * gz;
    }
    ds_acc = wave_sum(ds_acc);
    if (lane == 0) G.d_inv_s[r] = ds_acc;  // per-ray term; the caller sums them (no atomics: reproducible)
}

}  // namespace NCW_RAYNS
using namespace NCW_RAYNS;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
#ifndef NCW_RAYS_BIG
extern "C" int ncw_sample_coarse_big(const float*, const float*, const float*, const float*, int, int, int, const float*, const float*,
                                     float*, float*, float*, void*);
extern "C" int ncw_upsample_big(const float*, const float*, const float*, const float*, int, int, float, int, float*, void*);
extern "C" int ncw_sort_merge_big(const float*, int, const float*, int, const float*, const float*, int, float*, float*, void*);
extern "C" int ncw_composite_fwd_big(const NcwCompositeIn*, const NcwCompositeOut*, void*);
extern "C" int ncw_composite_bwd_big(const NcwCompositeIn*, const NcwCompositeGrad*, void*);
#define NCW_RAYS_FORWARD_BIG(cond, call) do { if (cond) return call; } while (0)
#else
#define NCW_RAYS_FORWARD_BIG(cond, call) do { } while (0)
#endif

extern "C" int NCW_RAYFN(ncw_sample_coarse)(const float* near, const float* far, const float* s_near, const float* s_far, int R,
                                 int n_samples, int n_outside, const float* rand_shift, const float* rand_out,
                                 float* z, float* z_out, float* sample_dist, void* stream) {
    if (R <= 0) return 0;
    NCW_RAYS_FORWARD_BIG(n_samples > RAY_MAXN && n_samples <= RAY_MAXN_BIG,
                         ncw_sample_coarse_big(near, far, s_near, s_far, R, n_samples, n_outside, rand_shift, rand_out, z, z_out, sample_dist, stream));
    if (n_samples < 1 || n_samples > RAY_MAXN) return NCW_E_BADARG;
    dim3 blk(64, 4);
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((R + 3) / 4), blk, 0, (hipStream_t)stream, near, far, s_near, s_far, R,
                       n_samples, n_outside, rand_shift, rand_out, z, z_out, sample_dist);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int NCW_RAYFN(ncw_upsample)(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int R, int n,
                            float inv_s, int n_new, float* z_new, void* stream) {
    if (R <= 0 || n_new <= 0) return 0;
    NCW_RAYS_FORWARD_BIG(n > RAY_MAXN - 1 && n <= RAY_MAXN_BIG - 1, ncw_upsample_big(rays_o, rays_d, z, sdf, R, n, inv_s, n_new, z_new, stream));
    if (n < 2 || n > RAY_MAXN - 1) return NCW_E_BADARG;
    hipLaunchKernelGGL(upsample_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z, sdf, R,
                       n, inv_s, n_new, z_new);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int NCW_RAYFN(ncw_sort_merge)(const float* a, int na, const float* b, int nb, const float* pa, const float* pb, int R,
                              float* out, float* pout, void* stream) {
    if (R <= 0 || na + nb <= 0) return 0;
    NCW_RAYS_FORWARD_BIG(na + nb > RAY_MAXN && na + nb <= RAY_MAXN_BIG, ncw_sort_merge_big(a, na, b, nb, pa, pb, R, out, pout, stream));
    if (na + nb > RAY_MAXN) return NCW_E_BADARG;
    hipLaunchKernelGGL(sort_merge_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, pa, pb, R,
                       out, pout);
    NCW_CHECK_LAUNCH();
    return 0;
}

#ifndef NCW_RAYS_BIG
extern "C" int ncw_boundary(const float* near, const float* far, const float* z, int n, int R, int nb, float* zb,
                            void* stream) {
    if (R <= 0 || nb <= 0) return 0;
    dim3 blk(64, 4);
    hipLaunchKernelGGL(boundary_kernel, dim3((R + 3) / 4), blk, 0, (hipStream_t)stream, near, far, z, n, R, nb, zb);
    NCW_CHECK_LAUNCH();
    return 0;
}
#endif

extern "C" int NCW_RAYFN(ncw_composite_fwd)(const NcwCompositeIn* in, const NcwCompositeOut* out, void* stream) {
    if (!in || !out) return NCW_E_BADARG;
    if (in->R <= 0) return 0;
    const int M = in->has_bg ? in->S + in->O : in->S;
    NCW_RAYS_FORWARD_BIG(M > RAY_MAXN && M <= RAY_MAXN_BIG, ncw_composite_fwd_big(in, out, stream));
    if (in->S < 1 || M > RAY_MAXN) return NCW_E_BADARG;
    CompArgs A;
    A.rays_o = in->rays_o; A.rays_d = in->rays_d; A.z = in->z; A.z_feed = in->z_feed; A.sample_dist = in->sample_dist;
    A.sdf = in->sdf; A.grad = in->grad; A.rgb = in->rgb; A.density = in->density; A.bg_rgb = in->bg_rgb;
    A.inv_s = in->inv_s; A.background_rgb = in->background_rgb; A.cos_anneal = in->cos_anneal; A.cos_anneal_dev = in->cos_anneal_dev;
    A.R = in->R; A.S = in->S; A.O = in->O; A.has_bg = in->has_bg; A.trim_sphere = in->trim_sphere;
    CompOut Q;
    Q.color = out->color; Q.color_sphere = out->color_sphere; Q.color_bg = out->color_bg; Q.weights = out->weights;
    Q.weights_sum = out->weights_sum; Q.cdf = out->cdf; Q.inside = out->inside; Q.depth = out->depth;
    Q.normals = out->normals; Q.eik = out->eik; Q.mid_z = out->mid_z; Q.dists = out->dists; Q.bg_alpha = out->bg_alpha; Q.weights_max = out->weights_max;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3((in->R + 3) / 4), dim3(256), 0, (hipStream_t)stream, A, Q);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int NCW_RAYFN(ncw_composite_bwd)(const NcwCompositeIn* in, const NcwCompositeGrad* g, void* stream) {
    if (!in || !g) return NCW_E_BADARG;
    if (in->R <= 0) return 0;
    const int M = in->has_bg ? in->S + in->O : in->S;
    NCW_RAYS_FORWARD_BIG(M > RAY_MAXN && M <= RAY_MAXN_BIG, ncw_composite_bwd_big(in, g, stream));
    if (in->S < 1 || M > RAY_MAXN) return NCW_E_BADARG;
    CompArgs A;
    A.rays_o = in->rays_o; A.rays_d = in->rays_d; A.z = in->z; A.z_feed = in->z_feed; A.sample_dist = in->sample_dist;
    A.sdf = in->sdf; A.grad = in->grad; A.rgb = in->rgb; A.density = in->density; A.bg_rgb = in->bg_rgb;
    A.inv_s = in->inv_s; A.background_rgb = in->background_rgb; A.cos_anneal = in->cos_anneal; A.cos_anneal_dev = in->cos_anneal_dev;
    A.R = in->R; A.S = in->S; A.O = in->O; A.has_bg = in->has_bg; A.trim_sphere = in->trim_sphere;
    CompBwd G;
    G.d_color = g->d_color; G.d_wsum = g->d_weights_sum; G.d_depth = g->d_depth; G.d_eik = g->d_eik_num;
    G.d_sdf = g->d_sdf; G.d_grad = g->d_grad; G.d_rgb = g->d_rgb; G.d_density = g->d_density; G.d_bg_rgb = g->d_bg_rgb;
    G.d_inv_s = g->d_inv_s;
    G.grad_scale = g->grad_scale != 0.f ? g->grad_scale : 1.0f;
    G.grad_scale_dev = g->grad_scale_dev;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((in->R + 3) / 4), dim3(256), 0, (hipStream_t)stream, A, G);
    NCW_CHECK_LAUNCH();
    return 0;
}

#ifndef NCW_RAYS_BIG  // ---- everything below does not keep a ray in LDS: the standard object only --------------------------------------
// out[r][j] (+)= sum_i rows[r * per_ray + i][j], i ascending: the order-fixed reduction of the per-point appearance-code
// adjoints (ncw_color_bwd / ncw_nerf_bwd with d_a_rows) to the per-ray gradient of the embedding lookup
// (models/neuconw.py:131-139, nerf.py:159-160: `a` is repeated over a ray's samples, so autograd sums over them).
__global__ __launch_bounds__(256) void ray_sum_rows_kernel(const float* __restrict__ rows, int64_t R, int per_ray, int n_cols,
                                                           float* __restrict__ out, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= R * n_cols) return;
    const int64_t r = e / n_cols;
    const int j = (int)(e - r * n_cols);
    const float* src = rows + (size_t)r * per_ray * n_cols + j;
    float s = 0.f;
    for (int i = 0; i < per_ray; ++i) s += src[(size_t)i * n_cols];
    out[e] = accumulate ? out[e] + s : s;
}

// out[idx[r]][j] += rows[r][j]: the backward of the appearance-embedding lookup `embedding_a(ts)`
// (lightning_modules/neuconw_system.py:70-75, renderer.py:808) -- torch's embedding_dense_backward takes 77 us for
// 1024 rows (it serialises duplicate indices); f32 atomics into the zero-filled gradient take ~3 us.  The fp32 parity
// mode keeps torch's deterministic kernel (atomics would make its result order-dependent for repeated images).
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ rows, const int64_t* __restrict__ idx,
                                                               int64_t R, int n_cols, int64_t n_out, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= R * n_cols) return;
    const int64_t r = e / n_cols;
    const int j = (int)(e - r * n_cols);
    const int64_t t = idx[r];
    if (t >= 0 && t < n_out) atomicAdd(out + t * n_cols + j, rows[e]);
}

extern "C" int ncw_scatter_add_rows(const float* rows, const int64_t* idx, int64_t R, int n_cols, int64_t n_out, float* out,
                                    void* stream) {
    if (R <= 0 || n_cols <= 0) return 0;
    if (!rows || !idx || !out || n_out <= 0) return NCW_E_BADARG;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((R * n_cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rows, idx, R, n_cols, n_out, out);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_ray_sum_rows(const float* rows, int64_t R, int per_ray, int n_cols, float* out, int accumulate,
                                void* stream) {
    if (R <= 0 || n_cols <= 0) return 0;
    if (!rows || !out || per_ray < 1) return NCW_E_BADARG;
    hipLaunchKernelGGL(ray_sum_rows_kernel, dim3((unsigned)((R * n_cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rows, R, per_ray, n_cols, out, accumulate);
    NCW_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Batch assembly from the HBM-resident ray cache: PhototourismDataset.__getitem__ for split "train"
// (datasets/phototourism.py:709-726) + the black-list test of NeuconWSystem.training_step
// (lightning_modules/neuconw_system.py:345-349) for a whole batch in one launch -- a row gather by `idx`:
//   with_semantics: cache row = [o(3) d(3) near far | ts | label | c10 c11 c12]  -> rays = [0:8] ++ [10:13]
//   without:        cache row = [o(3) d(3) near far | ts | c9 c10 c11]            -> rays = [0:8] ++ [9:12]
//   ts = long(row[8]); label = long(row[9]) (torch's float -> int64 truncation); keep = label not in mask_ids
// One thread per output row; rows are 48 / 52 B so a wave reads 3 KiB of contiguous-ish HBM per instruction group.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void batch_assemble_kernel(const float* __restrict__ all_rays, int ncols,
                                                             const float* __restrict__ all_rgbs,
                                                             const int64_t* __restrict__ idx, int64_t n_rows, int64_t B,
                                                             int with_semantics, float* __restrict__ rays,
                                                             int64_t* __restrict__ ts, int64_t* __restrict__ label,
                                                             float* __restrict__ rgbs, int4 ids, int n_ids,
                                                             uint8_t* __restrict__ keep) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    int64_t r = idx ? idx[b] : b;
    r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
    const float* src = all_rays + r * ncols;
    float* dst = rays + b * 11;
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] = src[c];
    const int tail = with_semantics ? 10 : 9;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[8 + c] = src[tail + c];
    ts[b] = (int64_t)src[8];
    const int64_t lab = with_semantics ? (int64_t)src[9] : 0;
    if (label) label[b] = lab;
    if (rgbs) {
        rgbs[b * 3] = all_rgbs[r * 3]; rgbs[b * 3 + 1] = all_rgbs[r * 3 + 1]; rgbs[b * 3 + 2] = all_rgbs[r * 3 + 2];
    }
    if (keep) {
        bool k = true;
        if (with_semantics) {
            if (n_ids > 0 && lab == ids.x) k = false;
            if (n_ids > 1 && lab == ids.y) k = false;
            if (n_ids > 2 && lab == ids.z) k = false;
            if (n_ids > 3 && lab == ids.w) k = false;
        }
        keep[b] = k ? 1 : 0;
    }
}

extern "C" int ncw_batch_assemble(const float* all_rays, int ncols, const float* all_rgbs, const int64_t* idx, int64_t n_rows,
                                  int64_t B, int with_semantics, float* rays, int64_t* ts, int64_t* label, float* rgbs,
                                  const int* mask_ids, int n_ids, uint8_t* keep, void* stream) {
    if (B <= 0) return 0;
    if (!all_rays || !rays || !ts || n_rows <= 0 || n_ids < 0 || n_ids > 4) return NCW_E_BADARG;
    if (ncols != (with_semantics ? 13 : 12) || (rgbs && !all_rgbs)) return NCW_E_BADARG;
    int4 ids = make_int4(n_ids > 0 ? mask_ids[0] : -1, n_ids > 1 ? mask_ids[1] : -1, n_ids > 2 ? mask_ids[2] : -1,
                         n_ids > 3 ? mask_ids[3] : -1);
    hipLaunchKernelGGL(batch_assemble_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, all_rays,
                       ncols, all_rgbs, idx, n_rows, B, with_semantics, rays, ts, label, rgbs, ids, n_ids, keep);
    NCW_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Per-ray loss terms computed inside render() (renderer.py:763-765 gradient_error, :869-877 mask_error,
// :892-897 sfm_depth_loss) and their backward: ~55 tiny torch launches per step become two.  One
// workgroup (the work is R elements + three sums).
//   gradient_error = sum(eik_num) / (sum(eik_den) + 1e-5)
//   mask_error[r]  = BCE(clip(weights_sum[r], 1e-3, 1 - 1e-3), mask[r]),  mask[r] = 0 iff label[r] in ids
//   sfm[r]         = (depth - depth_gt)^2 * w * [w > 0] * R / max(#{w > 0}, 1)   (the sync-free form: its mean
//                    over R equals the reference's mean over the selected rays; 0 if none is selected)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ float tail_mask(const int64_t* label, int r, int n_ids, const int* ids) {
    if (label == nullptr) return 1.f;
    const int64_t l = label[r];
    for (int i = 0; i < n_ids; ++i)
        if (l == ids[i]) return 0.f;
    return 1.f;
}

struct TailIds { int v[4]; };

__global__ __launch_bounds__(256) void ray_tail_fwd_kernel(const float* __restrict__ wsum, const int64_t* __restrict__ label,
                                                           const float* __restrict__ depth, const float* __restrict__ depth_gt,
                                                           const float* __restrict__ depth_w, const float* __restrict__ eik_num,
                                                           const float* __restrict__ eik_den, int R, int n_ids, TailIds ids,
                                                           float* __restrict__ mask_error, float* __restrict__ sfm,
                                                           float* __restrict__ scal) {
    __shared__ float sm[4];
    float a = 0.f, b = 0.f, c = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
        a += eik_num[r];
        b += eik_den[r];
        if (depth_w != nullptr) c += depth_w[r] > 0.f ? 1.f : 0.f;
    }
    a = block_sum256(a, sm);
    b = block_sum256(b, sm);
    c = block_sum256(c, sm);
    const float cnt = fmaxf(c, 1.f);
    if (threadIdx.x == 0) {
        scal[0] = a / (b + 1e-5f);
        scal[1] = b;
        scal[2] = cnt;
    }
    const float rs = (float)R / cnt;
    for (int r = threadIdx.x; r < R; r += 256) {
        if (mask_error != nullptr) {
            const float m = tail_mask(label, r, n_ids, ids.v);
            const float x = fminf(fmaxf(wsum[r], 1e-3f), 1.f - 1e-3f);
            mask_error[r] = -(m * fmaxf(logf(x), -100.f) + (1.f - m) * fmaxf(logf(1.f - x), -100.f));
        }
        if (sfm != nullptr) {
            const float w = depth_w[r], dd = depth[r] - depth_gt[r];
            sfm[r] = dd * dd * w * (w > 0.f ? 1.f : 0.f) * rs;
        }
    }
}

__global__ __launch_bounds__(256) void ray_tail_bwd_kernel(const float* __restrict__ wsum, const int64_t* __restrict__ label,
                                                           const float* __restrict__ depth, const float* __restrict__ depth_gt,
                                                           const float* __restrict__ depth_w, int R, int n_ids, TailIds ids,
                                                           const float* __restrict__ scal, const float* __restrict__ d_mask_error,
                                                           const float* __restrict__ d_sfm, const float* __restrict__ d_ge,
                                                           float* __restrict__ d_wsum, float* __restrict__ d_depth,
                                                           float* __restrict__ d_eik_num) {
    const float den = scal[1], cnt = scal[2];
    const float ge = d_ge != nullptr ? d_ge[0] / (den + 1e-5f) : 0.f;
    const float rs = (float)R / cnt;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        d_eik_num[r] = ge;
        float dw = 0.f;
        if (d_mask_error != nullptr) {
            const float w0 = wsum[r];
            if (w0 >= 1e-3f && w0 <= 1.f - 1e-3f) {  // clamp passes the gradient on the closed interval
                const float m = tail_mask(label, r, n_ids, ids.v);
                dw = d_mask_error[r] * (w0 - m) / fmaxf((1.f - w0) * w0, 1e-12f);
            }
        }
        d_wsum[r] = dw;
        float dd = 0.f;
        if (d_sfm != nullptr) {
            const float w = depth_w[r];
            dd = d_sfm[r] * 2.f * (depth[r] - depth_gt[r]) * w * (w > 0.f ? 1.f : 0.f) * rs;
        }
        d_depth[r] = dd;
    }
}

extern "C" int ncw_ray_tail_fwd(const float* weights_sum, const int64_t* label, const int* mask_ids, int n_ids,
                                const float* depth, const float* depth_gt, const float* depth_weight, const float* eik_num,
                                const float* eik_den, int R, float* mask_error, float* sfm_depth_loss, float* scalars,
                                void* stream) {
    if (R <= 0) return 0;
    if (!weights_sum || !eik_num || !eik_den || !scalars || n_ids < 0 || n_ids > 4 || (n_ids > 0 && !mask_ids))
        return NCW_E_BADARG;
    if (sfm_depth_loss && (!depth || !depth_gt || !depth_weight)) return NCW_E_BADARG;
    TailIds ids = {{0, 0, 0, 0}};
    for (int i = 0; i < n_ids; ++i) ids.v[i] = mask_ids[i];
    hipLaunchKernelGGL(ray_tail_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weights_sum, label, depth, depth_gt,
                       sfm_depth_loss ? depth_weight : nullptr, eik_num, eik_den, R, n_ids, ids, mask_error, sfm_depth_loss,
                       scalars);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_ray_tail_bwd(const float* weights_sum, const int64_t* label, const int* mask_ids, int n_ids,
                                const float* depth, const float* depth_gt, const float* depth_weight, int R,
                                const float* scalars, const float* d_mask_error, const float* d_sfm_depth_loss,
                                const float* d_gradient_error, float* d_weights_sum, float* d_depth, float* d_eik_num,
                                void* stream) {
    if (R <= 0) return 0;
    if (!weights_sum || !scalars || !d_weights_sum || !d_depth || !d_eik_num || n_ids < 0 || n_ids > 4 ||
        (n_ids > 0 && !mask_ids))
        return NCW_E_BADARG;
    if (d_sfm_depth_loss && (!depth || !depth_gt || !depth_weight)) return NCW_E_BADARG;
    TailIds ids = {{0, 0, 0, 0}};
    for (int i = 0; i < n_ids; ++i) ids.v[i] = mask_ids[i];
    hipLaunchKernelGGL(ray_tail_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, weights_sum, label,
                       depth, depth_gt, depth_weight, R, n_ids, ids, scalars, d_mask_error, d_sfm_depth_loss,
                       d_gradient_error, d_weights_sum, d_depth, d_eik_num);
    NCW_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Dead-background elimination: which ray samples need the background NeRF at all (include/neuconw_hip.h, ncw_bg_select).
// Three small launches: (1) one wave per ray counts its samples to keep (ballot), (2) one workgroup turns the counts into
// exclusive offsets, (3) one wave per ray writes its indices at its offset in sample order -- the list is ray-major and
// ascending whatever the launch shape (deterministic).  ~15 us for 1024 x 132 samples (a first single-workgroup version
// with one THREAD per ray took 250 us: uncoalesced reads of z).
// ------------------------------------------------------------------------------------------------
struct BgSel {
    const float *rays_o, *rays_d, *z, *sample_dist;
    int R, S, O;
};

// keep flag of sample i of ray r: the n_outside samples always; a primary sample iff the compositor's own inside_sphere
// (composite_fwd_kernel: section mid-point of the PRIMARY z, the last section ends sample_dist further) is 0
NCW_DEV bool bg_keep(const BgSel& A, int r, int i, const float (&o)[3], const float (&d)[3]) {
    const int M = A.S + A.O;
    if (i >= A.S) return i < M;
    // A.z = the PRIMARY z [R, S]: the compositor pairs column i < S of the background arrays with primary sample i by
    // INDEX (background_alpha[:, :n_samples] * (1 - inside_sphere), renderer.py:693), wherever z_feed's i-th point lies
    const float zi = A.z[(size_t)r * A.S + i];
    const float dist = (i + 1 < A.S) ? A.z[(size_t)r * A.S + i + 1] - zi : A.sample_dist[r];
    const float zz = zi + dist * 0.5f;
    const float x = o[0] + d[0] * zz, y = o[1] + d[1] * zz, w = o[2] + d[2] * zz;
    return !(sqrtf(x * x + y * y + w * w) < 1.0f);  // inside_sphere = (|p| < 1): renderer.py:637
}

template <bool WRITE>
__global__ __launch_bounds__(256) void bg_select_ray_kernel(BgSel A, int32_t* __restrict__ offs, int32_t* __restrict__ idx) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.R) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int M = A.S + A.O;
    const float o[3] = {A.rays_o[r * 3], A.rays_o[r * 3 + 1], A.rays_o[r * 3 + 2]};
    const float d[3] = {A.rays_d[r * 3], A.rays_d[r * 3 + 1], A.rays_d[r * 3 + 2]};
    int at = WRITE ? offs[r] : 0;
    for (int i0 = 0; i0 < M; i0 += 64) {
        const int i = i0 + lane;
        const bool keep = i < M && bg_keep(A, r, i, o, d);
        const unsigned long long m = __ballot(keep);
        if (WRITE && keep) idx[at + __popcll(m & ((1ull << lane) - 1ull))] = r * M + i;
        at += __popcll(m);
    }
    if (!WRITE && lane == 0) offs[r] = at;
}

// offs[r] (counts) -> exclusive prefix, offs[R] = count[0] = total; one workgroup, rays in chunks of its size
__global__ __launch_bounds__(1024) void bg_select_scan_kernel(int R, int32_t* __restrict__ offs, int32_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += blockDim.x) {
        const int r = r0 + tid;
        const int cnt = r < R ? offs[r] : 0;
        int v = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off, 64);
            if (lane >= off) v += u;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int wbase = 0, total = 0;
        for (int w = 0; w < nw; ++w) {
            if (w < wave) wbase += wsum[w];
            total += wsum[w];
        }
        const int base = base_s;
        if (r < R) offs[r] = base + wbase + v - cnt;
        __syncthreads();
        if (tid == 0) base_s = base + total;
        __syncthreads();
    }
    if (tid == 0) { offs[R] = base_s; count[0] = base_s; }
}

extern "C" int ncw_bg_select(const float* rays_o, const float* rays_d, const float* z, const float* sample_dist, int R,
                             int S, int O, int32_t* idx, int32_t* ray_offsets, int32_t* count, void* stream) {
    if (R < 0 || S < 0 || O < 0 || !idx || !count || !ray_offsets ||
        (R > 0 && (!rays_o || !rays_d || (S > 0 && !z) || !sample_dist)))
        return NCW_E_BADARG;
    if ((int64_t)R * (S + O) > 0x7fffffffLL) return NCW_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    BgSel A;
    A.rays_o = rays_o; A.rays_d = rays_d; A.z = z; A.sample_dist = sample_dist; A.R = R; A.S = S; A.O = O;
    if (R > 0) {
        hipLaunchKernelGGL(bg_select_ray_kernel<false>, dim3((R + 3) / 4), dim3(256), 0, st, A, ray_offsets, idx);
        NCW_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(bg_select_scan_kernel, dim3(1), dim3(1024), 0, st, R, ray_offsets, count);
    NCW_CHECK_LAUNCH();
    if (R > 0) {
        hipLaunchKernelGGL(bg_select_ray_kernel<true>, dim3((R + 3) / 4), dim3(256), 0, st, A, ray_offsets, idx);
        NCW_CHECK_LAUNCH();
    }
    return 0;
}
#endif  // !NCW_RAYS_BIG
