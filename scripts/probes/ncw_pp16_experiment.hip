// EXPERIMENT RECORD, not part of the library (profiles/r03/pp16_experiment.log).  To rebuild it: copy into
// neuralrecon-w_amd/csrc/ as ncw_pp16.hip, add it to MLP_FILES / F16_FILES in build.py and dispatch ncw_sdf_infer16P_launch from
// sdf_infer_any (ncw_sdf.hip).  8 waves (default): results identical to sdf_infer16.  -DP16_NWAVES=4: TIMING ONLY (its results were
// wrong, not debugged).  NCW_P16_EXP selects the elimination variants.
// Fine-interleaved SDF kernels at W = 512 (the width the reference ships), 16-bit operands: the streamed-weights structure of
// ncw_sdf16.hip (wave w owns output blocks w and w + 8, its A fragments come from the packed matrix in L2 through a small register
// ring, the activations live in LDS as B fragments and are rewritten in place) with the software pipeline of ncw_pp.hip on top.
//
// Why: in ncw_sdf16.hip every wave runs [all MFMAs of a layer] barrier [epilogue of its 2 T blocks] barrier -- bursts.  The issue
// probe (scripts/probes/issue_probe.hip) showed that VALU work only overlaps the matrix pipe when it sits BETWEEN the
// MFMAs in program order, so a burst layer costs MFMA time + epilogue time: fitted (12 + 10.7 T) k cycles per layer against 4.1 T k
// cycles of MFMAs.  Here the 4 tiles of a workgroup form two groups of 2 tiles and every wave pipelines across them:
//
//      segment:   [M(l,g0) | E(l-1,g1)]  bar  [M(l,g1) | E(l,g0)]  bar  [M(l+1,g0) | E(l,g1)]  bar ...
//
// M(l,g): the 128 MFMAs of layer l on group g (32 k-units x 2 blocks x 2 tiles), E: the epilogue (activation, 16-bit packing, LDS and
// stash stores) of the group finished one segment earlier, two accumulator registers between each pair of MFMAs (5 VALU
// per MFMA at this width: inside what the probe found to be free).  One LDS-only barrier per segment; each group buffer ([2 tiles][32
// units][64 lanes][16 B] = 64 KiB) is rewritten in place: E(l,g) overwrites the layer-l inputs of g that M(l,g) finished reading one
// barrier earlier.  Price: the layer's 512 KB weight stream is paid per group (2 tiles) instead of per workgroup (T tiles):
// 512 KB / 56 B/clk = 9.4 k cycles of the CU's L2 port against 8.2 k MFMA cycles per segment -- the kernels sit on the L2 stream.
#include "ncw_mlp.h"

namespace {

#ifndef P16_NWAVES
#define P16_NWAVES 8
#endif
constexpr int P16_WAVES = P16_NWAVES, P16_KU = 32, P16_TILES = 4;
constexpr int P16_NB = 16 / P16_WAVES;  // output blocks per wave: wave + P16_WAVES j
#ifndef P16_DEPTH
#define P16_DEPTH 4
#endif
constexpr int P16_D = P16_DEPTH;  // A-fragment prefetch distance (k-units)
constexpr int P16_RING = 3;       // B-fragment ring slots (one k-unit of both tiles per slot)
constexpr int P16_GRP = 2 * P16_KU * 64;  // fragments of one group buffer

typedef __attribute__((address_space(3))) bf16x8 p16_lfrag;
typedef const __attribute__((address_space(1))) bf16x8* p16_gfrag;

struct P16W { bf16x8 f[P16_D][P16_NB]; };   // register ring: units q .. q + D - 1 of the wave's two output blocks
struct P16Acc { f32x16 v[P16_NB][2]; };     // [block j = wave + 8 j][tile of the group]

// one A fragment: unit u of output block ob.  Uniform base (SALU) + zero-extended 32-bit lane offset = the saddr form of global_load
// (a signed lane index makes the compiler build a 64-bit VGPR address with 2-3 VALU per load)
NCW_DEV bf16x8 p16_ld(const void* w, int rb_stride, int ob, int u, int lane) {
    const char* base = (const char*)w + ((size_t)u * rb_stride + ob) * 1024;
    return *(p16_gfrag)(base + (unsigned)lane * 16u);
}

NCW_DEV void p16_prefetch(P16W& r, const void* w, int rb_stride, int wave, int lane) {
#pragma unroll
    for (int d = 0; d < P16_D; ++d)
#pragma unroll
        for (int j = 0; j < P16_NB; ++j) r.f[d][j] = p16_ld(w, rb_stride, wave + P16_WAVES * j, d, lane);
}

NCW_DEV f32x16 p16_bias(const float* bp, int ob, int lane) {
    CVec<1> b1;
    load_bias(b1, bp + ob * 32, lane);
    return b1.v[0];
}

NCW_DEV f32x16 p16_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// One segment: m += W[blocks wave, wave + 8][.] . in[2 tiles][.] over the 32 k-units of a 512-wide layer (ZERO: m = that product),
// with epi(q, h) -- one accumulator register of the PREVIOUS group's epilogue -- placed after each pair of MFMAs.  Without ZERO the
// caller has put the layer's bias into m (p16_epi_bias: loaded into each accumulator block the moment the epilogue of the segment
// before has consumed it, so it costs no registers).  r holds units 0 .. D-1 of `w` on entry; with NEXT, units 0 .. D-1 of `wn` (the
// matrix of the next segment) on exit: the weight stream never drains.
template <bool ZERO, bool NEXT, int EXP = 0, class EPI>
NCW_DEV void p16_segment(P16Acc& m, P16W& r, const void* w, int stride, const void* wn, int nstride, const p16_lfrag* in, int wave,
                         int lane, EPI&& epi) {
    constexpr int RD = P16_RING - 1;
    bf16x8 b[P16_RING][2];
    const f32x16 zero = p16_zero();
#pragma unroll
    for (int c = 0; c < RD; ++c) { b[c][0] = in[c * 64]; b[c][1] = in[(P16_KU + c) * 64]; }
#pragma unroll
    for (int q = 0; q < P16_KU; ++q) {
        if (!(EXP & 4) && q + RD < P16_KU) { b[(q + RD) % P16_RING][0] = in[(q + RD) * 64]; b[(q + RD) % P16_RING][1] = in[(P16_KU + q + RD) * 64]; }
        bf16x8 a[P16_NB];
#pragma unroll
        for (int j = 0; j < P16_NB; ++j) a[j] = r.f[q % P16_D][j];
#pragma unroll
        for (int j = 0; j < P16_NB; ++j) {
            m.v[j][0] = NCW_MFMA_H(a[j], b[q % P16_RING][0], (ZERO && q == 0) ? zero : m.v[j][0], 0, 0, 0);
            if (!(EXP & 2)) epi(q, 2 * j);
            m.v[j][1] = NCW_MFMA_H(a[j], b[q % P16_RING][1], (ZERO && q == 0) ? zero : m.v[j][1], 0, 0, 0);
            if (!(EXP & 1)) {
                if (q + P16_D < P16_KU) r.f[q % P16_D][j] = p16_ld(w, stride, wave + P16_WAVES * j, q + P16_D, lane);
                else if (NEXT) r.f[q % P16_D][j] = p16_ld(wn, nstride, wave + P16_WAVES * j, q + P16_D - P16_KU, lane);
            }
            if (!(EXP & 2)) epi(q, 2 * j + 1);
        }
        if (EXP & 8) {  // one MFMA at a time: [MFMA, ~5 VALU] x 4 with the LDS reads and the global loads spread between them
#pragma unroll
            for (int i = 0; i < 2 * P16_NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
                if (i == 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
                if (i & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (EXP & 2) {  // keep the accumulators alive without an epilogue
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(m.v[j][t]));
    }
}

// the 3 gamma k-units (32..34) of the forward-orientation skip layer against gbuf ([tile][4 units]; gin = gbuf + first tile of the group)
NCW_DEV void p16_mma_gamma(P16Acc& m, const void* w, int wave, const p16_lfrag* gin, int lane) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int j = 0; j < P16_NB; ++j) {
            const bf16x8 g = p16_ld(w, 16, wave + P16_WAVES * j, P16_KU + q, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) m.v[j][t] = NCW_MFMA_H(g, gin[(t * 4 + q) * 64 + lane], m.v[j][t], 0, 0, 0);
        }
    }
}

// Epilogue step (q, h) of a finished group: accumulator register r = 2 (q & 7) + h of block j = q >> 4, tile t = (q >> 3) & 1 goes
// through f, is packed, and every eighth register one B fragment (k-unit 2 ob + (r >> 3) of tile t) is stored: the 16-bit image of
// registers 8 i .. 8 i + 7 of C-layout block ob IS k-unit 2 ob + i of the next layer (ncw_common.h).
template <class F>
NCW_DEV void p16_epi(int q, int h, const P16Acc& e, bf16x8& frag, p16_lfrag* out, int wave, int lane, F&& f) {
    const int i = q * 2 * P16_NB + h;  // 0 .. 32 NB - 1: [block j][tile t][register r]
    const int j = i >> 5, t = (i >> 4) & 1, r = i & 15;
    frag[r & 7] = (ncw_h16)f(e.v[j][t][r], j, t, r);
    if ((r & 7) == 7) out[(t * P16_KU + 2 * (wave + P16_WAVES * j) + (r >> 3)) * 64 + lane] = frag;
}

// after epilogue step (q, 1) with (q & 7) == 7 the accumulator block (j, t) of e is dead: the bias of the layer that accumulates
// into it next goes there (global, L2-resident; a quarter of a segment or more to land)
NCW_DEV void p16_epi_bias(int q, int h, P16Acc& e, const float* bp, int wave, int lane) {
    const int i = q * 2 * P16_NB + h;
    if ((i & 15) == 15) e.v[i >> 5][(i >> 4) & 1] = p16_bias(bp, wave + P16_WAVES * (i >> 5), lane);
}

NCW_DEV float p16_softplus(float z) {
    float y, s;
    softplus100<true>(z, y, s);
    return y;
}

// abuf: two group buffers (tiles 0, 1 | tiles 2, 3), rewritten in place; gbuf: per tile GU units (gamma / qbar_0: 3, the d_sdf unit: 1)
#define P16_LDS_DECL(GU)                                                                                 \
    __shared__ __attribute__((aligned(16))) char lds[P16_TILES * P16_KU * 1024 + P16_TILES * (GU) * 1024]; \
    p16_lfrag* const abuf = (p16_lfrag*)(ncw_lchar*)lds;                                                 \
    p16_lfrag* const gbuf = abuf + P16_TILES * P16_KU * 64;                                              \
    const int lane = ncw_lane();                                                                         \
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                           \
    const int L = net.n_layers;                                                                          \
    const int64_t tile0 = (int64_t)blockIdx.x * P16_TILES

// ------------------------------------------------------------------------------------------------
// sdf_infer (SDFNetwork.sdf, models/neuconw.py:281-282): gamma -> L-1 Softplus layers -> sdf row
// ------------------------------------------------------------------------------------------------
template <int EXP>
__global__ __launch_bounds__(64 * P16_WAVES) void sdf_infer16P_kernel(NcwSdfNet net, NcwPoints src, int64_t n, float* __restrict__ sdf) {
    P16_LDS_DECL(4);
    const int NL = L - 1;  // Softplus layers
    for (int tw = wave; tw < P16_TILES; tw += P16_WAVES) {  // gamma of tile tw, straight into LDS as k-units 0..2
        int64_t p = (tile0 + tw) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, true>(gam, xs, lane);
        Act<PrecBF16, 2> ga;
        to_act(ga, gam);
#pragma unroll
        for (int q = 0; q < 3; ++q) gbuf[(tw * 4 + q) * 64 + lane] = ga.f[q];
    }
    P16W r;
    P16Acc x, y;  // group 0 accumulates into x, group 1 into y
    bf16x8 frag;
    auto softplus_f = [](float z, int, int, int) { return p16_softplus(z); };
    p16_lfrag* const a0 = abuf;             // group 0
    p16_lfrag* const a1 = abuf + P16_GRP;   // group 1
    {   // layer 0: K = 39 (3 units of gamma) for both groups, then E(0,g0)
        if (NL > 1) p16_prefetch(r, net.w[1], 16, wave, lane);
        ncw_lds_barrier();  // gamma visible
#pragma unroll
        for (int j = 0; j < P16_NB; ++j) {
            bf16x8 w0[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) w0[q] = p16_ld(net.w[0], 16, wave + P16_WAVES * j, q, lane);
            const f32x16 b0 = p16_bias(net.b[0], wave + P16_WAVES * j, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 ax = b0, ay = b0;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    ax = NCW_MFMA_H(w0[q], gbuf[(t * 4 + q) * 64 + lane], ax, 0, 0, 0);
                    ay = NCW_MFMA_H(w0[q], gbuf[((2 + t) * 4 + q) * 64 + lane], ay, 0, 0, 0);
                }
                x.v[j][t] = ax;
                y.v[j][t] = ay;
            }
        }
#pragma unroll
        for (int q = 0; q < P16_KU; ++q)
#pragma unroll
            for (int h = 0; h < 2 * P16_NB; ++h) {
                p16_epi(q, h, x, frag, a0, wave, lane, softplus_f);
                if (NL > 1) p16_epi_bias(q, h, x, net.b[1], wave, lane);
            }
        ncw_lds_barrier();
    }
    for (int l = 1; l < NL; ++l) {
        const bool more = l + 1 < NL;
        {   // [M(l,g0) -> x | E(l-1,g1) <- y], then y takes the bias of layer l
            auto epi = [&](int q, int h) { p16_epi(q, h, y, frag, a1, wave, lane, softplus_f); p16_epi_bias(q, h, y, net.b[l], wave, lane); };
            p16_segment<false, true, EXP>(x, r, net.w[l], 16, net.w[l], 16, a0 + lane, wave, lane, epi);
            if (l == net.skip_layer) p16_mma_gamma(x, net.w[l], wave, gbuf, lane);
            ncw_lds_barrier();
        }
        {   // [M(l,g1) -> y | E(l,g0) <- x], then x takes the bias of layer l + 1.  (One instantiation whether or not a layer follows --
            // after the last one the ring and x take a prefetch nobody uses: two copies of the segment behind a branch make the
            // compiler hoist the common head of all 64 epilogue steps above the branch, 128 live registers.)
            const int ln = more ? l + 1 : l;
            auto epi = [&](int q, int h) { p16_epi(q, h, x, frag, a0, wave, lane, softplus_f); p16_epi_bias(q, h, x, net.b[ln], wave, lane); };
            p16_segment<false, true, EXP>(y, r, net.w[l], 16, net.w[ln], 16, a1 + lane, wave, lane, epi);
            if (l == net.skip_layer) p16_mma_gamma(y, net.w[l], wave, gbuf + 2 * 4 * 64, lane);
            ncw_lds_barrier();
        }
    }
    // drain: E(NL-1, g1)
#pragma unroll
    for (int q = 0; q < P16_KU; ++q)
#pragma unroll
        for (int h = 0; h < 2 * P16_NB; ++h) p16_epi(q, h, y, frag, a1, wave, lane, softplus_f);
    ncw_lds_barrier();
    for (int tw = wave; tw < P16_TILES; tw += P16_WAVES) {  // sdf row
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        const p16_lfrag* in = abuf + (tw >> 1) * P16_GRP + (tw & 1) * P16_KU * 64 + lane;
#pragma unroll
        for (int q = 0; q < P16_KU; ++q) o.v[0] = NCW_MFMA_H(p16_ld(net.w[L - 1], 1, 0, q, lane), in[q * 64], o.v[0], 0, 0, 0);
        const int64_t p = (tile0 + tw) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / net.scale;
    }
}

}  // namespace

int NCW_FN(ncw_sdf_infer16P_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    static const int exp_ = getenv("NCW_P16_EXP") ? atoi(getenv("NCW_P16_EXP")) : 0;  // DEVELOPMENT: timing experiments (wrong results)
    const dim3 grid((unsigned)((tiles + P16_TILES - 1) / P16_TILES)), block(64 * P16_WAVES);
#define P16_EXP_CASE(E) case E: hipLaunchKernelGGL(sdf_infer16P_kernel<E>, grid, block, 0, st, *net, src, n, sdf); break
    switch (exp_) { P16_EXP_CASE(1); P16_EXP_CASE(2); P16_EXP_CASE(3); P16_EXP_CASE(4); P16_EXP_CASE(5); P16_EXP_CASE(7); P16_EXP_CASE(8); default: P16_EXP_CASE(0); }
    NCW_CHECK_LAUNCH();
    return 0;
}
