// Shared device-side vocabulary of the NeuralRecon-W hot-path kernels for gfx950 (CDNA4).
//
// Point-per-lane ("swapped operand") MLP formulation
// --------------------------------------------------
// Every MLP on the path (SDF net models/neuconw.py:183-296, colour net :59-170, background NeRF
// models/nerf.py:86-183) is evaluated as   Y^T = W . X^T   with the WEIGHTS as the MFMA A operand
// and the per-point activations as the B operand.  One wave owns 32 points; lane l holds point
// p = l & 31 and "half" h = l >> 5.  A feature vector of 32*RB (zero-padded) features lives in
// the 32x32 MFMA C/D register layout
//
//      v[rb][r]  on lane (p,h)   ==   feature  32*rb + (r&3) + 8*(r>>2) + 4*h   of point p
//
// which is exactly what a 32x32 MFMA writes for output rows = features, columns = points.  The
// next layer consumes the same registers as its B operand without any cross-lane traffic, because
// the K order of the weights is permuted to match when they are packed (ncw_pack.hip):
//
//   bf16  v_mfma_f32_32x32x16_bf16 : k-step s = 2*rb + t takes registers 8t..8t+7 of block rb;
//         logical k = 8h + e  <->  feature 16 s + 4 h + (e&3) + 8 (e>>2)
//   f32   v_mfma_f32_32x32x2_f32   : k-step (rb, r) takes register r of block rb;
//         logical k = h      <->  feature 32 rb + (r&3) + 8 (r>>2) + 4 h
//
// Activations never leave registers between layers; weights stream from L2 in fragment order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// The 16-bit element type.  Every fused-MLP / weight-gradient source file is compiled TWICE (build.py): once with
// ncw_h16 = __bf16 (prec NCW_PREC_BF16) and once with -DNCW_HALF_F16, ncw_h16 = _Float16 (prec NCW_PREC_F16: 10 mantissa
// bits instead of 7 at the same MFMA rate; the backward's cotangents are loss-scaled by the host, see renderer.py).
// In the second compilation every entry point and cross-file helper carries the suffix _f16 (NCW_FN) and the kernels live
// in their own namespace (NCW_NS); the bf16 build's entry points forward prec == NCW_PREC_F16 calls to them.
// (`bf16x8` / `bf16x4` / PrecBF16 keep their names: "the 16-bit fragment / precision".)
#ifdef NCW_HALF_F16
typedef _Float16 ncw_h16;
#define NCW_MFMA_H __builtin_amdgcn_mfma_f32_32x32x16_f16
#define NCW_FN(name) name##_f16
#define NCW_NS ncw_h_f16
#else
typedef __bf16 ncw_h16;
#define NCW_MFMA_H __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define NCW_FN(name) name
#define NCW_NS ncw_h_bf16
#endif
typedef ncw_h16 bf16x8 __attribute__((ext_vector_type(8)));
typedef ncw_h16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NCW_PREC_F32 0
#define NCW_PREC_BF16 1
#define NCW_PREC_F16 2   // forwarded to the _f16 build of the same entry point (which sees it as its 16-bit precision)
// bf16 build only: hand a prec == NCW_PREC_F16 call to the fp16 build of this entry point
#ifdef NCW_HALF_F16
#define NCW_FORWARD_F16(prec, call) do {} while (0)
#else
#define NCW_FORWARD_F16(prec, call) do { if ((prec) == NCW_PREC_F16) return call; } while (0)
#endif

#define NCW_DEV __device__ __forceinline__
#define NCW_HD __host__ __device__ __forceinline__

// feature index (within a 32-block) of C-layout register r on half h
NCW_HD constexpr int ncw_feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

NCW_DEV int ncw_lane() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------------------------
// Precision traits
// ---------------------------------------------------------------------------------------------
struct PrecF32 {
    static constexpr int id = NCW_PREC_F32;
    typedef float welem;   // packed weight element
    typedef float selem;   // stash element
    // elements of packed weights per (in-block rb_in) per out-block: 16 k-steps * 64 lanes * 1
    static constexpr int W_PER_INBLOCK = 16 * 64;
};
struct PrecBF16 {
    static constexpr int id = NCW_PREC_BF16;
    typedef ncw_h16 welem;
    typedef ncw_h16 selem;
    // 2 k-steps * 64 lanes * 8
    static constexpr int W_PER_INBLOCK = 2 * 64 * 8;
};
// Both precisions: a packed [32*RB_OUT x 32*RB_IN] matrix holds RB_IN*RB_OUT*1024 elements, laid
// out [in-block][k-sub-step][out-block][lane][e].  (f32: 16 sub-steps x 1 elem, bf16: 2 x 8.)

// A feature vector in C layout, f32 (the MFMA accumulator itself).
template <int RB>
struct CVec {
    f32x16 v[RB];
};

template <int RB>
NCW_DEV void cvec_zero(CVec<RB>& c) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c.v[i][r] = 0.f;
}

// B-operand form of a feature vector.
template <class P, int RB>
struct Act;
template <int RB>
struct Act<PrecF32, RB> {
    f32x16 v[RB];
};
template <int RB>
struct Act<PrecBF16, RB> {
    bf16x8 f[2 * RB];
};

template <int RB>
NCW_DEV void to_act(Act<PrecF32, RB>& a, const CVec<RB>& c) {
#pragma unroll
    for (int i = 0; i < RB; ++i) a.v[i] = c.v[i];
}
template <int RB>
NCW_DEV void to_act(Act<PrecBF16, RB>& a, const CVec<RB>& c) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) a.f[2 * i + t][e] = (ncw_h16)c.v[i][8 * t + e];
}

// ---------------------------------------------------------------------------------------------
// acc[RB_OUT] += Wp . in   over the first KB_USED in-blocks of `in` (compile-time), weights in
// packed fragment order at `wp` (see ncw_pack.hip).  K_REAL = number of real (non-padding) input
// features: k-steps that only touch padding are skipped at compile time.
// ---------------------------------------------------------------------------------------------
// RB_STRIDE = number of out-blocks of the packed matrix (>= RB_OUT: only the first RB_OUT blocks
// are computed, e.g. the h-part of the transposed skip layer).
template <int RB_IN, int RB_OUT, int K_REAL, int RB_STRIDE = RB_OUT>
NCW_DEV void mma(CVec<RB_OUT>& acc, const Act<PrecF32, RB_IN>& in, const float* __restrict__ wp, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB_IN; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * rb + ncw_feat_of(r, 0) >= K_REAL) continue;  // h=1 feature is even larger
            const float* w = wp + ((size_t)(rb * 16 + r) * RB_STRIDE) * 64 + lane;
#pragma unroll
            for (int ro = 0; ro < RB_OUT; ++ro) {
                float a = w[ro * 64];
                acc.v[ro] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in.v[rb][r], acc.v[ro], 0, 0, 0);
            }
        }
    }
}
template <int RB_IN, int RB_OUT, int K_REAL, int RB_STRIDE = RB_OUT>
NCW_DEV void mma(CVec<RB_OUT>& acc, const Act<PrecBF16, RB_IN>& in, const ncw_h16* __restrict__ wp, int lane) {
#pragma unroll
    for (int s = 0; s < 2 * RB_IN; ++s) {
        if (16 * s >= K_REAL) continue;
        const bf16x8* w = reinterpret_cast<const bf16x8*>(wp) + ((size_t)s * RB_STRIDE) * 64 + lane;
#pragma unroll
        for (int ro = 0; ro < RB_OUT; ++ro) {
            bf16x8 a = w[ro * 64];
            acc.v[ro] = NCW_MFMA_H(a, in.f[s], acc.v[ro], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Weight streaming through LDS (all fused MLP kernels).
//
// The waves of a workgroup run the same layer sequence in lockstep; every packed matrix is streamed
// L2 -> LDS once per WORKGROUP (not once per wave) in chunks of whole in-blocks with
// `global_load_lds_dwordx4` (LDS-DMA: no VGPR round trip; the fragment order of the packed matrix IS
// the lane-linear order the DMA writes, so A fragments are conflict-free ds_read_b128 / b32).
// Two slots: chunk c+1 (or the first chunk of the NEXT matrix, handed in by the caller) is in flight
// while chunk c is consumed; one __syncthreads() per chunk (hipcc emits the s_waitcnt vmcnt(0) that
// retires the DMA in front of the s_barrier).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) void ncw_gvoid;
typedef __attribute__((address_space(3))) void ncw_lvoid;

// LDS pointers are kept in address space 3 explicitly: through a generic pointer hipcc emits
// flat_load (slow LDS aperture path, counted on vmcnt AND lgkmcnt) instead of ds_read_b128.
typedef __attribute__((address_space(3))) char ncw_lchar;
struct WRing {
    ncw_lchar* base;   // two slots: base, base + slot_bytes
    int cur;
    int slot_bytes;
    NCW_DEV ncw_lchar* slot(int i) const { return base + i * slot_bytes; }
};

// cooperative DMA of `bytes` (multiple of 16) from gsrc into an LDS slot by all waves of the WG
NCW_DEV void ring_issue(ncw_lchar* lds_slot, const void* gsrc, int bytes) {
    const int nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int pieces = (bytes + 1023) >> 10;
    const char* g0 = reinterpret_cast<const char*>(gsrc) + lane * 16;
    for (int pc = wave; pc < pieces; pc += nw) {
        if (pc * 1024 + lane * 16 < bytes)
            __builtin_amdgcn_global_load_lds((ncw_gvoid*)(g0 + (size_t)pc * 1024), (ncw_lvoid*)(lds_slot + pc * 1024), 16,
                                             0, 0);
    }
}

// A packed matrix is a linear array of k-"units" (f32: one register r = 2 k-values; bf16: one
// k-step of 16), unit u = in-block * UPB + sub-step, each unit = RB_STRIDE out-blocks x 64 lanes.
template <class P> struct UnitsPerBlock { static constexpr int v = (P::id == NCW_PREC_F32) ? 16 : 2; };
template <class P>
NCW_HD constexpr int ncw_unit_bytes(int rb_stride) {
    return rb_stride * 64 * ((P::id == NCW_PREC_F32) ? 4 : 16);
}
// number of in-blocks that contain real features
NCW_HD constexpr int ncw_nb_used(int rb_in, int k_real) { return (k_real + 31) / 32 < rb_in ? (k_real + 31) / 32 : rb_in; }
// units per chunk
template <class P>
NCW_HD constexpr int ncw_chunk_units(int rb_stride, int n_units, int slot_bytes) {
    int g = slot_bytes / ncw_unit_bytes<P>(rb_stride);
    if (g < 1) g = 1;
    return g < n_units ? g : n_units;
}
// bytes of the first chunk of a matrix (what the previous mma_stream call must prefetch)
template <class P, int RB_IN, int K_REAL, int RB_STRIDE, int SLOT>
NCW_HD constexpr int ncw_first_chunk_bytes() {
    return ncw_chunk_units<P>(RB_STRIDE, ncw_nb_used(RB_IN, K_REAL) * UnitsPerBlock<P>::v, SLOT) *
           ncw_unit_bytes<P>(RB_STRIDE);
}

NCW_DEV void ring_prologue(WRing& ring, const void* w_first, int first_bytes) {
    ring.cur = 0;
    ring_issue(ring.slot(0), w_first, first_bytes);
}

// B operand of unit (rb, sub)
template <int RB_IN>
NCW_DEV float unit_b(const Act<PrecF32, RB_IN>& in, int rb, int sub) { return in.v[rb][sub]; }
template <int RB_IN>
NCW_DEV bf16x8 unit_b(const Act<PrecBF16, RB_IN>& in, int rb, int sub) { return in.f[2 * rb + sub]; }

NCW_DEV f32x16 unit_mfma(float a, float b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
NCW_DEV f32x16 unit_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return NCW_MFMA_H(a, b, c, 0, 0, 0);
}

// first feature touched by unit (rb, sub) on half 0 (the h=1 half is larger): skip test vs K_REAL
template <class P>
NCW_HD constexpr int ncw_unit_first_feature(int rb, int sub) {
    return (P::id == NCW_PREC_F32) ? 32 * rb + ncw_feat_of(sub, 0) : 16 * (2 * rb + sub);
}

// B-operand providers.  mma_stream pulls the B operand of unit (rb, sub) from a provider; prepare(rb)
// is called once, immediately before the first unit of in-block rb.  An eager provider wraps a ready
// Act; a LAZY provider evaluates the previous layer's epilogue (activation, stash stores, bf16
// packing) for just that block, so the VALU work of block rb+1 overlaps the MFMAs of block rb inside
// one wave (MFMA and VALU are separate pipes) instead of idling the matrix pipe for a whole epilogue.
template <class P, int RB_IN_>
struct ActB {
    static constexpr int RB_IN = RB_IN_;
    const Act<P, RB_IN_>& a;
    NCW_DEV explicit ActB(const Act<P, RB_IN_>& a_) : a(a_) {}
    NCW_DEV void prepare(int) {}
    NCW_DEV auto b(int rb, int sub) const { return unit_b<RB_IN_>(a, rb, sub); }
};

// B provider for a feature-axis concatenation [a | g] WITHOUT materialising it (skip connections: the copy
// costs 8 (RA + RG) registers next to 16 RB accumulators -- at W = 512 that alone spills)
template <class P, int RA, int RG>
struct CatB {
    static constexpr int RB_IN = RA + RG;
    const Act<P, RA>& a;
    const Act<P, RG>& g;
    NCW_DEV CatB(const Act<P, RA>& a_, const Act<P, RG>& g_) : a(a_), g(g_) {}
    NCW_DEV void prepare(int) {}
    NCW_DEV auto b(int rb, int sub) const {
        return rb < RA ? unit_b<RA>(a, rb < RA ? rb : 0, sub) : unit_b<RG>(g, rb >= RA ? rb - RA : 0, sub);
    }
};

// acc[RB_OUT] += W . B, W streamed through the LDS ring.  ALL threads of the workgroup must call
// this with identical (uniform) arguments.  w_next/next_bytes: first chunk of the matrix the NEXT
// mma_stream call will consume (nullptr at the end of the kernel).
// PIPE (explicit double-buffering of the weight fragments, +8 fragment registers) pays in the kernels that
// own a whole SIMD (one workgroup per CU, 64 KiB slots); the two-workgroups-per-CU kernels (32 KiB slots) are
// capped at 256 registers -- there the other workgroup's waves hide the LDS latency and PIPE only spills.
// 16 out-blocks (W = 512): the second fragment set alone is 64 registers on top of 256 accumulators: off.
// BP2 (optional second B provider, same shape): acc += W . B + W . B2 in ONE pass of the ring -- every weight fragment read from LDS
// feeds two MFMAs (round 6: the hi and the lo half of an activation pair against W_hi; csrc/ncw_color.hip).
struct NoB2 {};
template <int RB_OUT, int K_REAL, int SLOT, int RB_STRIDE, class P, class BP, bool PIPE = (SLOT != 32768) && (RB_OUT <= 8), class BP2 = NoB2>
NCW_DEV void mma_stream_b(CVec<RB_OUT>& acc, BP& bp, WRing& ring, const typename P::welem* __restrict__ wp,
                          const void* w_next, int next_bytes, int lane, BP2* bp2 = nullptr) {
    constexpr int RB_IN = BP::RB_IN;
    constexpr int UPB = UnitsPerBlock<P>::v;
    constexpr int NU = ncw_nb_used(RB_IN, K_REAL) * UPB;
    constexpr int UB = ncw_unit_bytes<P>(RB_STRIDE);
    constexpr int CU = ncw_chunk_units<P>(RB_STRIDE, NU, SLOT);
    constexpr int NCH = (NU + CU - 1) / CU;
    typedef typename std::conditional<P::id == NCW_PREC_F32, float, bf16x8>::type Frag;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        if (c + 1 < NCH) {
            const int nu = (c + 2) * CU <= NU ? CU : NU - (c + 1) * CU;
            ring_issue(ring.slot(ring.cur ^ 1), reinterpret_cast<const char*>(wp) + (size_t)(c + 1) * CU * UB, nu * UB);
        } else if (w_next != nullptr) {
            ring_issue(ring.slot(ring.cur ^ 1), w_next, next_bytes);
        }
        typedef const __attribute__((address_space(3))) Frag* lfrag_t;
        lfrag_t lw = (lfrag_t)(ring.slot(ring.cur)) + lane;
        // Software pipeline: the RB_OUT weight fragments of the NEXT active unit are read from LDS while the
        // MFMAs of the current unit issue (hipcc otherwise emits ds_read -> s_waitcnt lgkmcnt(0) -> mfma
        // one by one and the matrix pipe idles on LDS latency).  Unit validity is compile-time.
        Frag cur[RB_OUT], nxt[RB_OUT];
        bool loaded = false;
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int q = c * CU + u;
            if (q >= NU) continue;
            const int rb = q / UPB, sub = q % UPB;
            if (sub == 0) bp.prepare(rb);
            if (ncw_unit_first_feature<P>(rb, sub) >= K_REAL) continue;
            if (!loaded) {
#pragma unroll
                for (int ro = 0; ro < RB_OUT; ++ro) cur[ro] = (lw + u * RB_STRIDE * 64)[ro * 64];
                loaded = true;
            }
            int un = -1;  // next active unit of this chunk
#pragma unroll
            for (int v = CU - 1; v > u; --v) {
                const int qv = c * CU + v;
                if (qv < NU && ncw_unit_first_feature<P>(qv / UPB, qv % UPB) < K_REAL) un = v;
            }
            if (PIPE && un >= 0) {
#pragma unroll
                for (int ro = 0; ro < RB_OUT; ++ro) nxt[ro] = (lw + un * RB_STRIDE * 64)[ro * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            const auto b = bp.b(rb, sub);
#pragma unroll
            for (int ro = 0; ro < RB_OUT; ++ro) acc.v[ro] = unit_mfma(cur[ro], b, acc.v[ro]);
            if constexpr (!std::is_same<BP2, NoB2>::value) {
                const auto b2 = bp2->b(rb, sub);
#pragma unroll
                for (int ro = 0; ro < RB_OUT; ++ro) acc.v[ro] = unit_mfma(cur[ro], b2, acc.v[ro]);
            }
            if (PIPE && un >= 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ro = 0; ro < RB_OUT; ++ro) cur[ro] = nxt[ro];
            } else {
                loaded = false;
                // 16 out-blocks: keep hipcc from hoisting the fragment reads of later units above these MFMAs
                // (64 registers per unit on top of 256 accumulators)
                if (RB_OUT > 8) __builtin_amdgcn_sched_barrier(0);
            }
        }
        ring.cur ^= 1;
    }
}

template <int RB_IN, int RB_OUT, int K_REAL, int SLOT, int RB_STRIDE = RB_OUT, class P>
NCW_DEV void mma_stream(CVec<RB_OUT>& acc, const Act<P, RB_IN>& in, WRing& ring, const typename P::welem* __restrict__ wp,
                        const void* w_next, int next_bytes, int lane) {
    ActB<P, RB_IN> bp(in);
    mma_stream_b<RB_OUT, K_REAL, SLOT, RB_STRIDE, P>(acc, bp, ring, wp, w_next, next_bytes, lane);
}
// acc += W . in + W . in2, one pass of the ring (two MFMAs per weight fragment)
template <int RB_IN, int RB_OUT, int K_REAL, int SLOT, int RB_STRIDE = RB_OUT, class P>
NCW_DEV void mma_stream2(CVec<RB_OUT>& acc, const Act<P, RB_IN>& in, const Act<P, RB_IN>& in2, WRing& ring,
                         const typename P::welem* __restrict__ wp, const void* w_next, int next_bytes, int lane) {
    ActB<P, RB_IN> bp(in), bp2(in2);
    mma_stream_b<RB_OUT, K_REAL, SLOT, RB_STRIDE, P, ActB<P, RB_IN>, (SLOT != 32768) && (RB_OUT <= 8), ActB<P, RB_IN>>(
        acc, bp, ring, wp, w_next, next_bytes, lane, &bp2);
}

// concatenation of two activation vectors along the feature axis (skip connections: zero cost)
template <int A, int B>
NCW_DEV void act_concat(Act<PrecF32, A + B>& o, const Act<PrecF32, A>& x, const Act<PrecF32, B>& y) {
#pragma unroll
    for (int i = 0; i < A; ++i) o.v[i] = x.v[i];
#pragma unroll
    for (int i = 0; i < B; ++i) o.v[A + i] = y.v[i];
}
template <int A, int B>
NCW_DEV void act_concat(Act<PrecBF16, A + B>& o, const Act<PrecBF16, A>& x, const Act<PrecBF16, B>& y) {
#pragma unroll
    for (int i = 0; i < 2 * A; ++i) o.f[i] = x.f[i];
#pragma unroll
    for (int i = 0; i < 2 * B; ++i) o.f[2 * A + i] = y.f[i];
}

// number of packed weight elements of a [32*RB_OUT x 32*RB_IN] matrix
NCW_HD constexpr size_t ncw_packed_elems(int rb_out, int rb_in) { return (size_t)rb_out * rb_in * 1024; }

// ---------------------------------------------------------------------------------------------
// bias: packed per out-block as [rb][h][16] f32 (C-layout order)  -> acc initialisation
// ---------------------------------------------------------------------------------------------
template <int RB>
NCW_DEV void load_bias(CVec<RB>& acc, const float* __restrict__ bp, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const f32x4* b4 = reinterpret_cast<const f32x4*>(bp + (rb * 2 + h) * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t = b4[g];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc.v[rb][4 * g + c] = t[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Activation functions
// ---------------------------------------------------------------------------------------------
// Softplus(beta=100) with torch's threshold 20 (models/neuconw.py:261) and its derivative
// sigma(100 z) (exactly 1 above the threshold).  FAST selects hardware exp2/log2.
// max(x, 0) as ONE integer max on the bit pattern (negative floats are negative integers, -0 -> +0): under the default IEEE
// mode `fmaxf(x, 0)` on an MFMA result costs two instructions (the compiler quiets the input with v_max x, x first).  Differs from
// fmaxf only for NaNs (a positive NaN stays a NaN instead of becoming 0 -- it then reaches the non-finite check of the optimiser).
NCW_DEV float ncw_relu(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// A 16-byte fragment of a packed matrix for this lane: uniform base (SALU) + zero-extended 32-bit lane offset = the SGPR-base form
// of global_load (a signed lane index makes the compiler build a 64-bit VGPR address with 2-3 VALU per load).  `frag` = index of
// the 1 KiB fragment (64 lanes x 16 B) in the matrix.
template <class V>
NCW_DEV V ncw_ld_frag(const void* w, size_t frag, int lane) {
    typedef const __attribute__((address_space(1))) V* gp;
    const char* base = (const char*)w + frag * (64 * sizeof(V));
    return *(gp)(base + (unsigned)lane * (unsigned)sizeof(V));
}

template <bool FAST>
NCW_DEV void softplus100(float z, float& y, float& s) {
    if (FAST) {
        // 16-bit kernels: y = max(z, 0) + log2(1 + 2^-|t|) ln2 / 100, t = 100 z log2(e), on the hardware exp2 / log2
        // (4 plain VALU + 2 transcendentals); s = 1 - exp(-100 y) (== sigmoid(100 z)), dead code wherever Softplus' is
        // recomputed from the stashed h.
        // (Round 2 tried a degree-4 polynomial for the correction term -- 7 plain VALU, sdf_infer 7 % faster, the step 0.5 % --
        // and dropped it: its constant tail put a floor of 2.5e-3 under the sigmoid recomputed from the stashed h, which
        // bounded the accuracy of the 16-bit modes.  DESIGN.md 3.1.)
        const float t = z * 144.26950408889634f;  // 100 z log2(e)
        const float w = __builtin_amdgcn_exp2f(-__builtin_fabsf(t));
        const float l = __builtin_amdgcn_logf(1.f + w);  // log2(1 + w)
        y = __builtin_fmaf(l, 0.6931471805599453f * 0.01f, ncw_relu(z));
        s = 1.f - __builtin_amdgcn_exp2f(y * -144.26950408889634f);
    } else {
        const float bz = 100.f * z;
        float e = expf(bz);
        float l = log1pf(e) * 0.01f;
        float sg = 1.f / (1.f + expf(-bz));
        y = bz > 20.f ? z : l;
        s = bz > 20.f ? 1.f : sg;
    }
}

// fp16 hi + lo images of accumulator registers 8 t .. 8 t + 7 (one k-unit of a B operand): x ~= hi + lo to 2^-22.
// lo MUST be the residual of the hi bits that are stored.  Written naively -- h = (half)x; hi = h; lo = (half)(x - (float)h) -- hipcc
// converts twice, v_cvt_pk_f16_f32 for the stored fragment and v_cvt_f16_f32 for the residual, and on gfx950 the two do not round an
// EXACT TIE the same way: the pair is then one whole fp16 ulp off (measured, round 6: one point in ~16 k with 5.5e-5 on the SDF where
// every other point is at 8e-7; profiles/r06/tunits_bisect.log).  The empty asm pins the PACKED pair (one v_cvt_pk_f16_f32 per two
// elements, as hipcc emits for the fragment anyway): one conversion, the residuals read its halves back (v_cvt_f32_f16 / _sdwa).
typedef ncw_h16 ncw_h16x2 __attribute__((ext_vector_type(2)));
// one PAIR: (x0, x1) -> packed hi (ONE v_cvt_pk_f16_f32, pinned) and the two residuals read back from it
NCW_DEV void ncw_split2(float x0, float x1, ncw_h16& h0, ncw_h16& h1, ncw_h16& l0, ncw_h16& l1) {
    ncw_h16x2 p = {(ncw_h16)x0, (ncw_h16)x1};
    asm volatile("" : "+v"(p));
    h0 = p[0];
    h1 = p[1];
    l0 = (ncw_h16)(x0 - (float)p[0]);
    l1 = (ncw_h16)(x1 - (float)p[1]);
}
NCW_DEV void ncw_split8(const f32x16& v, int t, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        ncw_h16 h0, h1, l0, l1;
        ncw_split2(v[8 * t + e], v[8 * t + e + 1], h0, h1, l0, l1);
        hi[e] = h0; hi[e + 1] = h1;
        lo[e] = l0; lo[e + 1] = l1;
    }
}

// Softplus(beta = 100) in "t-units" (round 6; the value-only SDF kernels: sampler queries, sdf(), grid sweep): with
// t = 100 log2(e) z and h' = 100 log2(e) h the activation is h' = log2(1 + 2^t) = max(t, 0) + log2(1 + 2^-|t|) -- no scale in front of
// the exp2, none behind the log2: 3 plain VALU + 2 transcendentals instead of 4 + 2 (`-|t|` is a source modifier of v_exp_f32).
// Because (ln 2 / 100) (100 log2 e) = 1 every hidden matrix is UNCHANGED: t_{l+1} = W h'_l + 144.27 b.  The kernels scale what enters
// the chain -- gamma(x) and the biases, once per point / per launch -- and divide the sdf row's result (models/neuconw.py:261-279).
constexpr float NCW_TU = 144.26950408889634f;  // 100 log2(e)
NCW_DEV float softplus_tu(float t) {
    const float w = __builtin_amdgcn_exp2f(-__builtin_fabsf(t));
    return ncw_relu(t) + __builtin_amdgcn_logf(1.f + w);
}
template <bool TU>
NCW_DEV float softplus_sel(float z) {
    if (TU) return softplus_tu(z);
    float y, s;
    softplus100<true>(z, y, s);
    return y;
}
template <int RB>
NCW_DEV void cvec_scale(CVec<RB>& v, float c) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v.v[rb][r] *= c;
}

template <bool FAST>
NCW_DEV float sigmoidf_(float x) {
    if (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
    return 1.f / (1.f + expf(-x));
}

// ---------------------------------------------------------------------------------------------
// Frequency encoding (models/neuconw.py:7-55) of a d-dimensional point straight into C layout.
// feature f:  f < D -> x[f];  else j = f - D, k = j / (2D), rem = j % (2D):
//             rem < D -> sin(2^k x[rem]),  else cos(2^k x[rem-D]).   f >= D*(2L+1) -> 0.
// ---------------------------------------------------------------------------------------------
template <int D, int L, bool FAST>
NCW_DEV float freq_feature(const float (&x)[D], int f) {
    if (f < D) {
        float v = x[0];
#pragma unroll
        for (int i = 1; i < D; ++i) v = (f == i) ? x[i] : v;
        return v;
    }
    if (f >= D * (2 * L + 1)) return 0.f;
    const int j = f - D;
    const int k = j / (2 * D);
    const int rem = j - k * 2 * D;
    const bool is_cos = rem >= D;
    const int comp = is_cos ? rem - D : rem;
    float v = x[0];
#pragma unroll
    for (int i = 1; i < D; ++i) v = (comp == i) ? x[i] : v;
    const float arg = v * (float)(1 << k);
    if (FAST) {
        const float rev = arg * 0.15915494309189535f;  // revolutions
        return is_cos ? __builtin_amdgcn_cosf(rev) : __builtin_amdgcn_sinf(rev);
    }
    return is_cos ? cosf(arg) : sinf(arg);
}

// derivative of feature f w.r.t. its source component; comp returned through `comp`
template <int D, int L, bool FAST>
NCW_DEV float freq_feature_deriv(const float (&x)[D], int f, int& comp) {
    if (f < D) {
        comp = f;
        return 1.f;
    }
    if (f >= D * (2 * L + 1)) {
        comp = 0;
        return 0.f;
    }
    const int j = f - D;
    const int k = j / (2 * D);
    const int rem = j - k * 2 * D;
    const bool is_cos = rem >= D;
    comp = is_cos ? rem - D : rem;
    float v = x[0];
#pragma unroll
    for (int i = 1; i < D; ++i) v = (comp == i) ? x[i] : v;
    const float fr = (float)(1 << k);
    const float arg = v * fr;
    if (FAST) {
        const float rev = arg * 0.15915494309189535f;
        return is_cos ? -fr * __builtin_amdgcn_sinf(rev) : fr * __builtin_amdgcn_cosf(rev);
    }
    return is_cos ? -fr * sinf(arg) : fr * cosf(arg);
}

template <int RB, int D, int L, bool FAST>
NCW_DEV void freq_encode(CVec<RB>& out, const float (&x)[D], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f0 = 32 * rb + ncw_feat_of(r, 0);
            if (f0 >= D * (2 * L + 1)) {
                out.v[rb][r] = 0.f;
            } else {
                out.v[rb][r] = freq_feature<D, L, FAST>(x, f0 + 4 * h);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Stash: fragment-native layout  [tile32][rb][g = r>>2][lane][4]  (each lane stores 4 consecutive
// features of its point: 8 B bf16 / 16 B f32 per lane, 512 B / 1 KiB per wave-instruction).
// ---------------------------------------------------------------------------------------------
// Timing-experiment hooks (stash stores compiled out / stash loads replaced by constants) live in
// scripts/probes/ncw_exp_hooks.h and exist only in probe libraries built beside the product with NCW_BUILD_TAG
// (build.py defines NCW_PROBE_BUILD for those and for nothing else): the product library cannot be built with them.
#ifdef NCW_PROBE_BUILD
#include "../../scripts/probes/ncw_exp_hooks.h"
#else
#if defined(NCW_EXP_NOSTORE) || defined(NCW_EXP_NOLOAD)
#error "NCW_EXP_* timing hooks are probe-only: build with NCW_BUILD_TAG=<tag> (neuralrecon-w_amd/build.py)"
#endif
#define NCW_EXP_STORE_HOOK()
#define NCW_EXP_LOAD_HOOK(x)
#endif

// Cache policy of the stash traffic (round 4, NOTEBOOK R4.7: 4.20 -> 3.92 ms per step, nothing else changed).  A stash operand
// is written once and read once or twice, milliseconds later, by ANOTHER kernel (the backward, the weight-gradient launch): it
// must not displace what the running kernel re-reads from L2 / the memory-side cache (packed weights, the few stashes listed
// below).  So stash stores and stash loads carry the non-temporal hint BY DEFAULT; the `_keep` variants (default policy) are
// for the two operands the SAME kernel re-reads: h_l of sdf_fwd (its adjoint sweep) and zbar2_l of sdf_bwd (its second pass).
// (Measured: keeping what the NEXT launch reads -- feat, dfeat, the background head's AUX1 block -- in the caches is slower.)
#define NCW_STASH_ST(dst, val) __builtin_nontemporal_store(val, &(dst))
#define NCW_STASH_LD(src) __builtin_nontemporal_load(&(src))

template <int RB>
NCW_DEV void stash_store(float* __restrict__ base, size_t tile, const CVec<RB>& c, int lane) {
    NCW_EXP_STORE_HOOK();
    f32x4* p = reinterpret_cast<f32x4*>(base) + (tile * RB * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) t[c4] = c.v[rb][4 * g + c4];
            NCW_STASH_ST(p[(rb * 4 + g) * 64], t);
        }
}
template <int RB>
NCW_DEV void stash_store(ncw_h16* __restrict__ base, size_t tile, const CVec<RB>& c, int lane) {
    NCW_EXP_STORE_HOOK();
    bf16x4* p = reinterpret_cast<bf16x4*>(base) + (tile * RB * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 t;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) t[c4] = (ncw_h16)c.v[rb][4 * g + c4];
            NCW_STASH_ST(p[(rb * 4 + g) * 64], t);
        }
}
// default cache policy (see above)
template <int RB>
NCW_DEV void stash_store_keep(float* __restrict__ base, size_t tile, const CVec<RB>& c, int lane) {
    NCW_EXP_STORE_HOOK();
    f32x4* p = reinterpret_cast<f32x4*>(base) + (tile * RB * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) t[c4] = c.v[rb][4 * g + c4];
            p[(rb * 4 + g) * 64] = t;
        }
}
template <int RB>
NCW_DEV void stash_store_keep(ncw_h16* __restrict__ base, size_t tile, const CVec<RB>& c, int lane) {
    NCW_EXP_STORE_HOOK();
    bf16x4* p = reinterpret_cast<bf16x4*>(base) + (tile * RB * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 t;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) t[c4] = (ncw_h16)c.v[rb][4 * g + c4];
            p[(rb * 4 + g) * 64] = t;
        }
}
template <int RB>
NCW_DEV void stash_load(CVec<RB>& c, const float* __restrict__ base, size_t tile, int lane) {
    NCW_EXP_LOAD_HOOK(c);
    const f32x4* p = reinterpret_cast<const f32x4*>(base) + (tile * RB * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t = NCW_STASH_LD(p[(rb * 4 + g) * 64]);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) c.v[rb][4 * g + c4] = t[c4];
        }
}
template <int RB>
NCW_DEV void stash_load(CVec<RB>& c, const ncw_h16* __restrict__ base, size_t tile, int lane) {
    NCW_EXP_LOAD_HOOK(c);
    const bf16x4* p = reinterpret_cast<const bf16x4*>(base) + (tile * RB * 4) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 t = NCW_STASH_LD(p[(rb * 4 + g) * 64]);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) c.v[rb][4 * g + c4] = (float)t[c4];
        }
}

// sum of a per-lane value over the two halves of a point (lanes p and p+32)
NCW_DEV float half_pair_sum(float v) { return v + __shfl_xor(v, 32, 64); }

#define NCW_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)
