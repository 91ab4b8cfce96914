// Helpers shared by the fused MLP kernels (SDF / colour / background NeRF): point sources,
// activation epilogues, tile bookkeeping.
#pragma once
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

template <class P>
struct Fast { static constexpr bool v = (P::id == NCW_PREC_BF16); };

// Workgroup barrier that orders LDS traffic only: `__syncthreads()` makes hipcc drain the vector-memory queue as well
// (s_waitcnt vmcnt(0) in front of s_barrier), i.e. every barrier waits for the stash stores and the next layer's weight
// prefetch issued before it.  The kernels that use this one exchange data between waves through LDS only (a stash block is
// read back, if at all, by the thread that wrote it; register dependencies on outstanding loads are tracked by the
// compiler), so global traffic may stay in flight across the barrier.  NOT for the LDS-DMA weight ring (ncw_common.h):
// there the barrier is what publishes a `global_load_lds` to the other waves.
NCW_DEV void ncw_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- point of lane ---------------------------------------------------------------------------
// mode 0: x[p];  1: o + d z[p];  2: o + d (z_i + dist_i / 2) with dist_i = z_{i+1} - z_i, last =
// sample_dist  (renderer.py:586-597 / :172-179)
NCW_DEV float grid_linspace(float start, float end, int steps, int i) {  // torch.linspace element i
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// mode 4: points of the launch = min(n, *count); where outputs / cotangents of launch point p live (the ray sample)
NCW_DEV int64_t points_count(const NcwPoints& s, int64_t n) {
    if (s.mode != 4) return n;
    const int64_t c = (int64_t)s.count[0];
    return c < n ? c : n;
}
NCW_DEV int64_t point_slot(const NcwPoints& s, int64_t p) { return s.mode == 4 ? (int64_t)s.idx[p] : p; }

NCW_DEV void load_point(const NcwPoints& s, int64_t p, float (&xs)[3], int64_t& ray) {
    if (s.mode == 3) {
        const int64_t q = p + s.gstart;
        const int64_t d = s.gdim;
        const int iz = (int)(q % d), iy = (int)((q / d) % d), ix = (int)(q / (d * d));
        xs[0] = (grid_linspace(s.gmin[0], s.gmax[0], s.gdim, ix) - s.gorigin[0]) / s.gradius;
        xs[1] = (grid_linspace(s.gmin[1], s.gmax[1], s.gdim, iy) - s.gorigin[1]) / s.gradius;
        xs[2] = (grid_linspace(s.gmin[2], s.gmax[2], s.gdim, iz) - s.gorigin[2]) / s.gradius;
        ray = p;
        return;
    }
    if (s.mode == 0) {
        xs[0] = s.x[p * 3 + 0]; xs[1] = s.x[p * 3 + 1]; xs[2] = s.x[p * 3 + 2];
        ray = p;
        return;
    }
    if (s.mode == 4) p = s.idx[p];  // a selection of the mode-2 points (ncw_bg_select)
    const int64_t r = p / s.per_ray;
    ray = r;
    float zz = s.z[p];
    if (s.mode == 2 || s.mode == 4) {
        const int i = (int)(p - r * s.per_ray);
        const float dist = (i + 1 < s.per_ray) ? s.z[p + 1] - zz : s.sample_dist[r];
        zz = zz + dist * 0.5f;
    }
    xs[0] = s.rays_o[r * 3 + 0] + s.rays_d[r * 3 + 0] * zz;
    xs[1] = s.rays_o[r * 3 + 1] + s.rays_d[r * 3 + 1] * zz;
    xs[2] = s.rays_o[r * 3 + 2] + s.rays_d[r * 3 + 2] * zz;
}

// wave's tile index and the (clamped) point of this lane.  Waves whose tile lies beyond n stay in
// the kernel (they take part in the workgroup barriers of the weight ring) with valid == false; the
// stash arenas are padded to a whole number of workgroups so their stores land in padding.
NCW_DEV void tile_setup(int64_t n, int64_t& tile, int64_t& p, bool& valid, int lane) {
    tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    p = tile * 32 + (lane & 31);
    valid = p < n;
    if (!valid) p = n - 1;
}

// single-block stash store / activation conversion (keeps epilogue register pressure at one block)
template <class SE>
NCW_DEV void stash_store_block(SE* __restrict__ base, size_t tile, int RB, int rb, const f32x16& v, int lane) {
    NCW_EXP_STORE_HOOK();
    typedef SE vec4 __attribute__((ext_vector_type(4)));
    vec4* p = reinterpret_cast<vec4*>(base) + ((tile * RB + rb) * 4) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        vec4 t;
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = (SE)v[4 * g + c];
        NCW_STASH_ST(p[g * 64], t);
    }
}
// default cache policy (ncw_common.h "Cache policy of the stash traffic"): stashes the same or the next kernel re-reads
template <class SE>
NCW_DEV void stash_store_block_keep(SE* __restrict__ base, size_t tile, int RB, int rb, const f32x16& v, int lane) {
    NCW_EXP_STORE_HOOK();
    typedef SE vec4 __attribute__((ext_vector_type(4)));
    vec4* p = reinterpret_cast<vec4*>(base) + ((tile * RB + rb) * 4) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        vec4 t;
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = (SE)v[4 * g + c];
        p[g * 64] = t;
    }
}
template <class SE>
NCW_DEV void stash_load_block(f32x16& v, const SE* __restrict__ base, size_t tile, int RB, int rb, int lane) {
    NCW_EXP_LOAD_HOOK(v);
    typedef SE vec4 __attribute__((ext_vector_type(4)));
    const vec4* p = reinterpret_cast<const vec4*>(base) + ((tile * RB + rb) * 4) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        vec4 t = NCW_STASH_LD(p[g * 64]);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = (float)t[c];
    }
}
template <int RB>
NCW_DEV void to_act_block(Act<PrecF32, RB>& a, int rb, const f32x16& v) { a.v[rb] = v; }
template <int RB>
NCW_DEV void to_act_block(Act<PrecBF16, RB>& a, int rb, const f32x16& v) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) a.f[2 * rb + t][e] = (ncw_h16)v[8 * t + e];
}

// act = Softplus100(acc); optionally stash y (next layer's input) and s = Softplus'
template <class P, int RB>
NCW_DEV void softplus_epilogue(Act<P, RB>& act, CVec<RB>& acc, typename P::selem* st_h, typename P::selem* st_s,
                               size_t tile, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x16 yv, sv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y, s;
            softplus100<Fast<P>::v>(acc.v[rb][r], y, s);
            yv[r] = y;
            sv[r] = s;
        }
        if (st_s) stash_store_block(st_s, tile, RB, rb, sv, lane);
        if (st_h) stash_store_block_keep(st_h, tile, RB, rb, yv, lane);  // h_l: re-read by the adjoint sweep of the same kernel
        to_act_block<RB>(act, rb, yv);
    }
}

// act = relu(acc), optional stash of the post-activation
template <class P, int RB>
NCW_DEV void relu_epilogue(Act<P, RB>& act, CVec<RB>& acc, typename P::selem* st, size_t tile, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x16 yv;
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = ncw_relu(acc.v[rb][r]);
        if (st) stash_store_block(st, tile, RB, rb, yv, lane);
        to_act_block<RB>(act, rb, yv);
    }
}

// zbar = ubar * [y > 0]  where y is the stashed post-relu activation; stores zbar, returns it as Act
template <class P, int RB>
NCW_DEV void relu_backward(Act<P, RB>& za, const CVec<RB>& u, const typename P::selem* st_y, typename P::selem* st_z,
                           size_t tile, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x16 y, z;
        stash_load_block(y, st_y, tile, RB, rb, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = y[r] > 0.f ? u.v[rb][r] : 0.f;
        stash_store_block(st_z, tile, RB, rb, z, lane);
        to_act_block<RB>(za, rb, z);
    }
}

// ---- lazy B providers: the previous layer's epilogue evaluated block by block inside the MFMA stream ----
// u = Softplus100(z) (+ stash of u and of Softplus'), optionally followed by NX extra ready blocks
// (the gamma blocks of the SDF skip connection).
template <class P, int RB, int NX = 0>
struct SoftplusB {
    static constexpr int RB_IN = RB + NX;
    typedef typename P::selem SE;
    const CVec<RB>& z;
    SE* st_h;
    SE* st_s;
    size_t tile;
    int lane;
    const Act<P, (NX > 0 ? NX : 1)>* extra;
    Act<P, RB> act;
    NCW_DEV SoftplusB(const CVec<RB>& z_, SE* h, SE* s, size_t t, int l, const Act<P, (NX > 0 ? NX : 1)>* x = nullptr)
        : z(z_), st_h(h), st_s(s), tile(t), lane(l), extra(x) {}
    NCW_DEV void prepare(int rb) {
        if (rb >= RB) return;
        f32x16 yv, sv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y, sg;
            softplus100<Fast<P>::v>(z.v[rb][r], y, sg);
            yv[r] = y;
            sv[r] = sg;
        }
        if (st_s) stash_store_block(st_s, tile, RB, rb, sv, lane);
        if (st_h) stash_store_block_keep(st_h, tile, RB, rb, yv, lane);  // h_l: re-read by the adjoint sweep of the same kernel
        to_act_block<RB>(act, rb, yv);
    }
    NCW_DEV auto b(int rb, int sub) const {
        if constexpr (NX > 0) {
            if (rb >= RB) return unit_b<NX>(*extra, rb - RB, sub);
        }
        return unit_b<RB>(act, rb < RB ? rb : 0, sub);
    }
};

// u = relu(z) (+ stash), optionally followed by NX ready blocks (aux inputs / NeRF skip)
template <class P, int RB, int NX = 0>
struct ReluB {
    static constexpr int RB_IN = RB + NX;
    typedef typename P::selem SE;
    const CVec<RB>& z;
    SE* st;
    size_t tile;
    int lane;
    const Act<P, (NX > 0 ? NX : 1)>* extra;
    Act<P, RB> act;
    NCW_DEV ReluB(const CVec<RB>& z_, SE* s, size_t t, int l, const Act<P, (NX > 0 ? NX : 1)>* x = nullptr)
        : z(z_), st(s), tile(t), lane(l), extra(x) {}
    NCW_DEV void prepare(int rb) {
        if (rb >= RB) return;
        f32x16 yv;
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = ncw_relu(z.v[rb][r]);
        if (st) stash_store_block(st, tile, RB, rb, yv, lane);
        to_act_block<RB>(act, rb, yv);
    }
    NCW_DEV auto b(int rb, int sub) const {
        if constexpr (NX > 0) {
            if (rb >= RB) return unit_b<NX>(*extra, rb - RB, sub);
        }
        return unit_b<RB>(act, rb < RB ? rb : 0, sub);
    }
};

// u = z (identity; + stash), optionally followed by NX ready blocks
template <class P, int RB, int NX = 0>
struct LinearB {
    static constexpr int RB_IN = RB + NX;
    typedef typename P::selem SE;
    const CVec<RB>& z;
    SE* st;
    size_t tile;
    int lane;
    const Act<P, (NX > 0 ? NX : 1)>* extra;
    Act<P, RB> act;
    NCW_DEV LinearB(const CVec<RB>& z_, SE* s, size_t t, int l, const Act<P, (NX > 0 ? NX : 1)>* x = nullptr)
        : z(z_), st(s), tile(t), lane(l), extra(x) {}
    NCW_DEV void prepare(int rb) {
        if (rb >= RB) return;
        if (st) stash_store_block(st, tile, RB, rb, z.v[rb], lane);
        to_act_block<RB>(act, rb, z.v[rb]);
    }
    NCW_DEV auto b(int rb, int sub) const {
        if constexpr (NX > 0) {
            if (rb >= RB) return unit_b<NX>(*extra, rb - RB, sub);
        }
        return unit_b<RB>(act, rb < RB ? rb : 0, sub);
    }
};

template <int RB>
NCW_DEV void cvec_copy(CVec<RB>& d, const CVec<RB>& s) {
#pragma unroll
    for (int i = 0; i < RB; ++i) d.v[i] = s.v[i];
}

#define NCW_WG_WAVES 4
#define NCW_LAUNCH_TILES(kernel, n, st, ...)                                                         \
    do {                                                                                            \
        const int64_t tiles__ = ((n) + 31) / 32;                                                    \
        const int64_t blocks__ = (tiles__ + NCW_WG_WAVES - 1) / NCW_WG_WAVES;                       \
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks__), dim3(64 * NCW_WG_WAVES), 0, st, __VA_ARGS__); \
        NCW_CHECK_LAUNCH();                                                                         \
    } while (0)

// LDS slot size of the weight ring by network width and target occupancy (two slots per workgroup):
// OCC = 1: one 4-wave workgroup per CU (kernels that need > 256 registers per lane), 2 x 64 KiB;
// OCC = 2: two workgroups per CU (<= 256 registers: the backward chains), 2 x 32 KiB each, so one
//          workgroup's VALU epilogue / stash traffic overlaps the other's MFMAs.
template <int RB, int OCC = 1>
struct RingSlot { static constexpr int bytes = RB >= 8 ? (OCC >= 2 ? 32768 : 65536) : 16384; };

#define NCW_RING_DECL(SLOTB)                                              \
    __shared__ __attribute__((aligned(16))) char ring_mem__[2 * (SLOTB)]; \
    WRing ring;                                                           \
    ring.base = (ncw_lchar*)ring_mem__;                                   \
    ring.cur = 0;                                                         \
    ring.slot_bytes = (SLOTB)

// ---- auxiliary input blocks shared by the colour net and the background NeRF ------------------------
// AUX1 (3 blocks, 96): [gamma_4(view dir) (27) | appearance embedding a (n_a <= 69) | 0...]
//   (models/neuconw.py:131-139, models/nerf.py:159-160,173)
template <bool FAST>
NCW_DEV void build_aux1(CVec<3>& aux, const float (&dir)[3], const float* __restrict__ a_row, int n_a, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * rb + ncw_feat_of(r, 0) + 4 * h;
            float v;
            if (32 * rb + ncw_feat_of(r, 0) + 4 < 27) {  // both halves inside the direction encoding
                v = freq_feature<3, 4, FAST>(dir, f);
            } else if (32 * rb + ncw_feat_of(r, 0) >= 27) {  // both halves in the embedding / padding
                v = (f - 27 < n_a) ? a_row[f - 27] : 0.f;
            } else {
                v = (f < 27) ? freq_feature<3, 4, FAST>(dir, f) : ((f - 27 < n_a) ? a_row[f - 27] : 0.f);
            }
            aux.v[rb][r] = v;
        }
}

// 16-bit modes: the AUX1 columns of an appearance head's first layer are evaluated once per RAY in fp32 (ncw_aux_ray_bias) and
// enter the fused kernels as a per-ray bias row [32 RB] (f32, C-layout feature order = plain feature index); the AUX1
// operand of the forward pass is then zero.
NCW_DEV void ncw_act_zero3(Act<PrecF32, 3>& a) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.v[i][r] = 0.f;
}
NCW_DEV void ncw_act_zero3(Act<PrecBF16, 3>& a) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a.f[i][e] = (ncw_h16)0.f;
}
// block rb of the row: features 32 rb + 8 g + 4 h + c <-> register 4 g + c of half h
NCW_DEV void ncw_add_ray_bias_block(f32x16& v, const float* __restrict__ row, int rb, int lane) {
    const f32x4* ab = reinterpret_cast<const f32x4*>(row) + 8 * rb + (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = ab[2 * g];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] += t[c];
    }
}
template <int RB>
NCW_DEV void ncw_add_ray_bias(CVec<RB>& e, const float* __restrict__ row, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) ncw_add_ray_bias_block(e.v[rb], row, rb, lane);
}

// d_a[ray][j] += sum over the wave's points of the AUX1-part adjoint (features 27 .. 27+n_a)
// rows != nullptr: store this point's adjoint as row p of rows[n][n_a] instead (no atomics; ncw_ray_sum_rows adds a
// ray's rows in sample order -- the reproducible path of the fp32 parity mode)
NCW_DEV void accumulate_d_a(const CVec<3>& q, float* __restrict__ d_a, int64_t ray, int n_a, bool valid, int lane,
                            float* __restrict__ rows = nullptr, int64_t p = 0) {
    const int h = lane >> 5;
    if (rows != nullptr) {
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (32 * rb + ncw_feat_of(r, 0) + 4 < 27) continue;
                const int j = 32 * rb + ncw_feat_of(r, 0) + 4 * h - 27;
                if (valid && j >= 0 && j < n_a) rows[p * n_a + j] = q.v[rb][r];
            }
        return;
    }
    const int64_t r0 = __shfl(ray, 0, 64);
    const bool uniform = __all((ray == r0) ? 1 : 0);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * rb + ncw_feat_of(r, 0) + 4 < 27) continue;
            const int f = 32 * rb + ncw_feat_of(r, 0) + 4 * h;
            const int j = f - 27;
            float v = valid ? q.v[rb][r] : 0.f;
            if (uniform) {
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
                if ((lane & 31) == 0 && j >= 0 && j < n_a) atomicAdd(d_a + ray * n_a + j, v);
            } else {
                if (valid && j >= 0 && j < n_a) atomicAdd(d_a + ray * n_a + j, v);
            }
        }
}
