// Fine-interleaved fused MLP kernels, W = 256 bf16 (round 2 structure of DESIGN.md 3.1).
//
// Starting point: the weights-stationary kernels of ncw_sdf8.hip (8 waves own the 8 output blocks of a layer, the
// activations of the workgroup's 128 points live in LDS as MFMA B fragments).  There every wave runs
// [32 MFMAs -> epilogue of 2 tiles -> 32 MFMAs -> epilogue -> barrier]: bursts.  What the issue hardware does with
// that was measured with scripts/probes/issue_probe.hip (MI355X, cycles per 16 MFMAs 32x32x16 bf16 + 128 v_fma_f32):
//
//      one wave,  MFMAs then VALU (burst)                       1278   (= 520 + 737: no overlap at all)
//      one wave,  MFMA ; 8 VALU ; MFMA ; 8 VALU ...              752   (= the VALU time: the MFMAs are free)
//      two waves on a SIMD, one MFMA-only, one VALU-only        ~2000  for the VALU wave (2 VALU slots per MFMA)
//      two waves on a SIMD, both interleaving MFMA ; 8 VALU      645   per wave-pair step (MFMA pipe 81 % busy)
//      <= 6 VALU (or <= 2 v_exp) per MFMA are hidden completely; v_pk_fma_f32 beside MFMAs costs ~10 cycles each
//
// i.e. a blocked MFMA holds the SIMD's issue port, so VALU work only overlaps the matrix pipe when it sits BETWEEN the
// MFMAs in program order.  (A role split between the two waves of a SIMD -- one on the matrix pipe, one on the VALU, a
// phase apart, as in 8-wave attention kernels -- was built first and measured: M || E costs M + E, 0.173 ms.)
//
// Structure here: the 4 tiles of a workgroup form two groups of 2 tiles, and every wave software-pipelines across
// them: segment k issues the 32 MFMAs of group g of layer l with the epilogue VALU of the PREVIOUS group (Softplus,
// bf16 packing, LDS / stash stores) placed between them, two accumulator registers per pair of MFMAs:
//
//      segment:   [M(l,g0) | E(l-1,g1)]  bar  [M(l,g1) | E(l,g0)]  bar  [M(l+1,g0) | E(l,g1)]  bar ...
//
// One s_barrier per segment (M(l+1,g) starts after every wave's E(l,g)); activation buffers: per group two 32 KiB
// buffers (layer parity), 128 KiB.  The barrier waits for LDS only (lgkmcnt), so the next layer's weight slice
// (global -> registers, issued two loads per MFMA pair, a whole layer ahead) and stash stores stay in flight.
#include <cstdlib>
#include <cstring>

#include "ncw_mlp.h"

namespace {

constexpr int PP_WAVES = 8, PP_TILES = 4;
constexpr int PP_GRP = 2 * 16 * 1024;         // one group buffer: 2 tiles x 16 k-units x 1 KiB
constexpr int PP_XU = 6;                      // extra-input units per tile (gamma(x): 3, gamma(p) / AUX1: 6)
constexpr int PP_BIAS = 12 * 1024;            // bias staging: up to 12 layers x 256 f32

typedef __attribute__((address_space(3))) bf16x8 pp_lfrag;
typedef __attribute__((address_space(3))) f32x4 pp_lf4;

// One k-unit (1 KiB: 64 lanes x 16 B) of a packed matrix, global -> registers
NCW_DEV bf16x8 pp_load_unit(const void* w, int unit_index, int lane) {
    return ncw_ld_frag<bf16x8>(w, (size_t)unit_index, lane);
}

#if defined(NCW_PROBE_BUILD) && defined(NCW_PP_ASM)
// ------------------------------------------------------------------------------------------------
// PROBE BUILDS ONLY (round 6): the weight slices loaded STRAIGHT INTO AGPRs by hand-written
// `global_load_dwordx4 a[..], v, s[..]` (hipcc only ever loads into VGPRs: in round 5's two-blocks-per-wave form every slice load was
// followed by four v_accvgpr_write, +1.6 VALU per MFMA).  The compiler does not count these loads on vmcnt, so the waits are
// hand-placed too: pp_wait_w pins the registers it releases (an MFMA cannot be scheduled above the wait that guards its operand).
// ------------------------------------------------------------------------------------------------
NCW_DEV bf16x8 pp_load_unit_a(const void* w, int unit_index, int lane) {
    bf16x8 r;
    const char* base = (const char*)w + (size_t)unit_index * 1024;
    const unsigned off = (unsigned)lane * 16u;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
// the first 8 units of both blocks have landed when at most the 16 loads issued after them are outstanding
NCW_DEV void pp_wait_w_first_half(bf16x8 (&w)[2][16]) {
    asm volatile("s_waitcnt vmcnt(16)"
                 : "+a"(w[0][0]), "+a"(w[1][0]), "+a"(w[0][1]), "+a"(w[1][1]), "+a"(w[0][2]), "+a"(w[1][2]), "+a"(w[0][3]), "+a"(w[1][3]),
                   "+a"(w[0][4]), "+a"(w[1][4]), "+a"(w[0][5]), "+a"(w[1][5]), "+a"(w[0][6]), "+a"(w[1][6]), "+a"(w[0][7]), "+a"(w[1][7]));
}
NCW_DEV void pp_wait_w_second_half(bf16x8 (&w)[2][16]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+a"(w[0][8]), "+a"(w[1][8]), "+a"(w[0][9]), "+a"(w[1][9]), "+a"(w[0][10]), "+a"(w[1][10]), "+a"(w[0][11]), "+a"(w[1][11]),
                   "+a"(w[0][12]), "+a"(w[1][12]), "+a"(w[0][13]), "+a"(w[1][13]), "+a"(w[0][14]), "+a"(w[1][14]), "+a"(w[0][15]), "+a"(w[1][15]));
}
#endif

template <int NU>
NCW_DEV void pp_load_slice(bf16x8* a, const void* w, int rb_stride, int ob, int u0, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) a[q] = pp_load_unit(w, (u0 + q) * rb_stride + ob, lane);
}

// LDS-only barrier: LDS stores of this phase are complete, vector-memory traffic stays in flight
NCW_DEV void pp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// acc init from the LDS bias staging area: packed bias of one block = [h][16] f32 (C-layout order)
NCW_DEV f32x16 pp_bias(const pp_lf4* bb, int lane) {
    const pp_lf4* p = bb + (lane >> 5) * 4;
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = p[g];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = t[c];
    }
    return v;
}

NCW_DEV void pp_store_units(pp_lfrag* buf, int t, int ob, const f32x16& v, int lane) {
    Act<PrecBF16, 1> o;
    to_act_block<1>(o, 0, v);
    buf[(t * 16 + 2 * ob) * 64 + lane] = o.f[0];
    buf[(t * 16 + 2 * ob + 1) * 64 + lane] = o.f[1];
}

#ifndef PP_SETPRIO
#define PP_SETPRIO 1
#endif
NCW_DEV void pp_prio(int p) {
#if PP_SETPRIO
    if (p) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
}

// Softplus in t-units (ncw_common.h softplus_tu: this kernel is value-only -- gamma and the biases enter scaled by 100 log2 e,
// the sdf row's result is divided by it; the hidden matrices are unchanged)
NCW_DEV float pp_softplus(float t) { return softplus_tu(t); }

// ------------------------------------------------------------------------------------------------
// NB = output blocks per wave: 1 -> 8 waves per workgroup (two per SIMD, <= 256 registers each): the shipped form.  (NB = 2 --
// four 512-register waves, every B fragment feeding two MFMAs -- measured 0.188 vs 0.165 ms in round 2: half of a
// 512-register file is AGPRs, the accumulators land there and every epilogue value costs a v_accvgpr_read.)
// ------------------------------------------------------------------------------------------------
template <int NB> struct PPAcc { f32x16 v[NB][2]; };  // [block of the wave][tile of the group]

// One segment: m = c_init + W[blocks] . h for the two tiles of a group (`w[nb]` = the wave's 16 k-units of block nb,
// `in` = group buffer + lane), with `epi(u)` -- the epilogue of 2 NB accumulator registers of the PREVIOUS group --
// placed after each 2 NB MFMAs.  The B fragments go through a register ring (one k-unit of both tiles per slot,
// PP_RING - 1 slots ahead of their use); with PREFETCH (the layer's LAST segment) unit u of the NEXT layer is fetched
// global -> registers into the registers unit u of this layer has just left.
#define PP_RING 4
// WAITW (the asm probe only): this segment is the first user of a slice that pp_load_unit_a fetched -- wait for its halves by hand
template <bool PREFETCH, int NB, bool WAITW = false, class EPI>
NCW_DEV void pp_segment(PPAcc<NB>& m, const f32x16 (&c_init)[NB], bf16x8 (&w)[NB][16], const pp_lfrag* in,
                        const void* wnext, int nstride, int ob0, int ob_step, int lane, EPI&& epi) {
    constexpr int RD = PP_RING - 1;
    bf16x8 b[PP_RING][2];
#pragma unroll
    for (int c = 0; c < RD; ++c) { b[c][0] = in[c * 64]; b[c][1] = in[(16 + c) * 64]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#if defined(NCW_PROBE_BUILD) && defined(NCW_PP_ASM)
        if constexpr (WAITW && NB == 2) {
            if (u == 0) pp_wait_w_first_half(w);
            if (u == 8) pp_wait_w_second_half(w);
        }
#endif
        if (u + RD < 16)
        { b[(u + RD) % PP_RING][0] = in[(u + RD) * 64]; b[(u + RD) % PP_RING][1] = in[(16 + u + RD) * 64]; }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            m.v[nb][0] = NCW_MFMA_H(w[nb][u], b[u % PP_RING][0], u == 0 ? c_init[nb] : m.v[nb][0], 0, 0, 0);
            m.v[nb][1] = NCW_MFMA_H(w[nb][u], b[u % PP_RING][1], u == 0 ? c_init[nb] : m.v[nb][1], 0, 0, 0);
        }
        if (PREFETCH) {  // unit u of the next layer into the registers unit u has just left (block 0 first: pp_wait_w counts on the order)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#if defined(NCW_PROBE_BUILD) && defined(NCW_PP_ASM)
                if constexpr (NB == 2) w[nb][u] = pp_load_unit_a(wnext, u * nstride + ob0 + nb * ob_step, lane);
                else
#endif
                w[nb][u] = pp_load_unit(wnext, u * nstride + ob0 + nb * ob_step, lane);
            }
        }
        epi(u);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// extra k-units (gamma) from the x buffer: xin = xbuf + (first tile of the group) * XU * 64 + lane
template <int XU, int NB>
NCW_DEV void pp_mma_x(PPAcc<NB>& a, const bf16x8 (&wx)[NB][XU], const pp_lfrag* xin) {
#pragma unroll
    for (int u = 0; u < XU; ++u)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            a.v[nb][0] = NCW_MFMA_H(wx[nb][u], xin[u * 64], a.v[nb][0], 0, 0, 0);
            a.v[nb][1] = NCW_MFMA_H(wx[nb][u], xin[(XU + u) * 64], a.v[nb][1], 0, 0, 0);
        }
}

// Epilogue step u (0..15) of a finished group, per block of the wave: accumulator registers 2u', 2u'+1 of tile
// j = u >> 3 (u' = u & 7) go through `f`, are packed to bf16, and every fourth step one B fragment (k-unit
// 2 ob + (u' >> 2) of tile j) is stored.  The bf16 image of registers 8t..8t+7 of C-layout block ob IS k-unit
// 2 ob + t of the next layer (ncw_common.h).
template <int NB, class F>
NCW_DEV void pp_epi_step(int u, const PPAcc<NB>& e, bf16x8 (&frag)[NB], pp_lfrag* out, int ob0, int ob_step, int lane, F&& f) {
    const int j = u >> 3, r = 2 * (u & 7);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        frag[nb][r & 7] = (ncw_h16)f(e.v[nb][j][r], nb, j, r);
        frag[nb][(r & 7) + 1] = (ncw_h16)f(e.v[nb][j][r + 1], nb, j, r + 1);
        if ((u & 3) == 3) out[(j * 16 + 2 * (ob0 + nb * ob_step) + ((u & 7) >> 2)) * 64 + lane] = frag[nb];
    }
}

// (Round 4's packed-fp16 / polynomial Softplus epilogues -- probe-only variants EPI 1 / 2 of this kernel, NOTEBOOK R4.2, measured in
// profiles/r04/pp_epilogue.log: no gain / accuracy rejected -- were removed in round 6 when the kernel moved to t-units.)
// ------------------------------------------------------------------------------------------------
// SDF inference (SDFNetwork.sdf, models/neuconw.py:281-282): gamma -> L-1 Softplus layers -> sdf row.
// (EPI: kept as a template slot; 0 = the f32 Softplus epilogue in t-units, the only form.)
// ------------------------------------------------------------------------------------------------
template <int NB, int EPI>
__global__ __launch_bounds__(64 * PP_WAVES / NB) void sdf_inferC_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                       float* __restrict__ sdf) {
    constexpr int NW = PP_WAVES / NB;  // waves per workgroup
    __shared__ __attribute__((aligned(16))) char lds[4 * PP_GRP + PP_TILES * 3 * 1024 + PP_BIAS];
    pp_lfrag* const abuf = (pp_lfrag*)(ncw_lchar*)lds;  // [group][parity][tile in group][16 units][64 lanes]
    pp_lfrag* const gbuf = abuf + 4 * PP_GRP / 16;      // gamma: [tile][3 units][64 lanes]
    pp_lf4* const bbuf = (pp_lf4*)(gbuf + PP_TILES * 3 * 64);  // [layer][block][h][16] f32
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ob = wave;  // this wave's blocks: ob + nb * NW
    const int L = net.n_layers, NL = L - 1;
    const int64_t tile0 = (int64_t)blockIdx.x * PP_TILES;
    // ---- biases of the Softplus layers -> LDS (256 f32 per layer) ---------------------------------------------------
    for (int i = threadIdx.x; i < NL * 64; i += 64 * NW) {
        const int l = i >> 6;
        bbuf[i] = reinterpret_cast<const f32x4*>(net.b[l])[i & 63] * NCW_TU;  // t-units
    }
    // ---- gamma of the 4 tiles: tile t by wave t % NW ---------------------------------------------------------------------
    for (int t = wave; t < PP_TILES; t += NW) {
        int64_t p = (tile0 + t) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, true>(gam, xs, lane);
        cvec_scale<2>(gam, NCW_TU);  // t-units: layer 0 and the skip layer's gamma columns see 100 log2(e) gamma
        Act<PrecBF16, 2> ga;
        to_act(ga, gam);
#pragma unroll
        for (int q = 0; q < 3; ++q) gbuf[(t * 3 + q) * 64 + lane] = ga.f[q];
    }
    // weights: ONE register slice of 16 k-units per block; wx = W_0's 3 units, later the skip layer's gamma units
    bf16x8 wa[NB][16], wx[NB][3];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) pp_load_slice<3>(wx[nb], net.w[0], 8, ob + nb * NW, 0, lane);
#if defined(NCW_PROBE_BUILD) && defined(NCW_PP_ASM)
    if constexpr (NB == 2) {  // straight into AGPRs, unit-major like the prefetch (pp_wait_w_* count on the order)
        if (NL > 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wa[nb][u] = pp_load_unit_a(net.w[1], u * 8 + ob + nb * NW, lane);
        }
    } else
#endif
    {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            if (NL > 1) pp_load_slice<16>(wa[nb], net.w[1], 8, ob + nb * NW, 0, lane);
    }
    // accumulators: group 0 always accumulates into x, group 1 into y; while one takes the MFMAs of its group the other --
    // the group finished one segment earlier -- goes through the epilogue: no register copies
    PPAcc<NB> x, y;
    bf16x8 frag[NB];
    f32x16 bias[NB];
    auto read_bias = [&](int l) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bias[nb] = pp_bias(bbuf + (l * 8 + ob + nb * NW) * 8, lane);
    };
#define PP_SEG_END() pp_barrier()
    pp_barrier();
    auto softplus_f = [](float z, int, int, int) { return pp_softplus(z); };
#define PP_EPI(u, acc, outp) pp_epi_step<NB>(u, acc, frag, outp, ob, NW, lane, softplus_f)
    // ---- layer 0 (K = 39: the 3 gamma units): [M(0,g0)] [M(0,g1) | E(0,g0)] ---------------------------------------------
    {
        read_bias(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { x.v[nb][0] = bias[nb]; x.v[nb][1] = bias[nb]; y.v[nb][0] = bias[nb]; y.v[nb][1] = bias[nb]; }
        pp_mma_x<3, NB>(x, wx, gbuf + lane);
        PP_SEG_END();
        pp_mma_x<3, NB>(y, wx, gbuf + 2 * 3 * 64 + lane);
        if (net.skip_layer == 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) pp_load_slice<3>(wx[nb], net.w[1], 8, ob + nb * NW, 16, lane);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) PP_EPI(u, x, abuf + (0 * 2 + 0) * (PP_GRP / 16));
    }
    // ---- hidden layer l >= 1: [M(l,g0) -> x | E(l-1,g1) <- y] [M(l,g1) -> y | E(l,g0) <- x] ------------------------------
    // (the bias block of a segment is read BEFORE the barrier that opens it: no LDS round trip at the segment head)
    read_bias(NL > 1 ? 1 : 0);
    PP_SEG_END();
    for (int l = 1; l < NL; ++l) {
        const bool more = l + 1 < NL;
        {   // g = 0
            const pp_lfrag* in = abuf + (0 * 2 + ((l - 1) & 1)) * (PP_GRP / 16) + lane;
            pp_lfrag* out = abuf + (1 * 2 + ((l - 1) & 1)) * (PP_GRP / 16);  // E(l-1, g1)
            auto epi = [&](int u) { PP_EPI(u, y, out); };
            pp_segment<false, NB, true>(x, bias, wa, in, nullptr, 8, ob, NW, lane, epi);  // (asm probe: first user of the slice)
            if (l == net.skip_layer) pp_mma_x<3, NB>(x, wx, gbuf + lane);
            read_bias(l);
            PP_SEG_END();
        }
        {   // g = 1: the layer's last use of its weight slice -> the next layer's slice takes its registers
            const pp_lfrag* in = abuf + (1 * 2 + ((l - 1) & 1)) * (PP_GRP / 16) + lane;
            pp_lfrag* out = abuf + (0 * 2 + (l & 1)) * (PP_GRP / 16);        // E(l, g0)
            auto epi = [&](int u) { PP_EPI(u, x, out); };
            // ONE code path: the last layer "prefetches" its own slice again (L2 hits, never used).  With two variants of this segment
            // (prefetch / no prefetch behind `if (more)`) hipcc hoists what they share -- the 32 v_exp of the epilogue -- in front of
            // the branch, i.e. out from between the MFMAs (round 6: a block of 32 transcendentals with no MFMA per layer).
            pp_segment<true, NB>(y, bias, wa, in, more ? net.w[l + 1] : net.w[l], 8, ob, NW, lane, epi);
            if (l == net.skip_layer) pp_mma_x<3, NB>(y, wx, gbuf + 2 * 3 * 64 + lane);
            // the skip layer's gamma columns (units 16..18) for the NEXT layer (W_0's units are no longer needed)
            if (more && l + 1 == net.skip_layer) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) pp_load_slice<3>(wx[nb], net.w[l + 1], 8, ob + nb * NW, 16, lane);
            }
            read_bias(more ? l + 1 : l);
            PP_SEG_END();
        }
    }
    // ---- drain: E(NL-1, g1) ----------------------------------------------------------------------------------------------
    {
        pp_lfrag* out = abuf + (1 * 2 + ((NL - 1) & 1)) * (PP_GRP / 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) PP_EPI(u, y, out);
        PP_SEG_END();
    }
    // ---- sdf row: tile t by wave t % NW ------------------------------------------------------------------------------------
    for (int t = wave; t < PP_TILES; t += NW) {
        bf16x8 w1[16];
        pp_load_slice<16>(w1, net.w[L - 1], 1, 0, 0, lane);
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        cvec_scale<1>(o, NCW_TU);  // 100 log2(e) (b + W h) = 100 log2(e) b + W h'
        const pp_lfrag* in = abuf + ((t >> 1) * 2 + ((NL - 1) & 1)) * (PP_GRP / 16) + (t & 1) * 16 * 64 + lane;
#pragma unroll
        for (int u = 0; u < 16; ++u) o.v[0] = NCW_MFMA_H(w1[u], in[u * 64], o.v[0], 0, 0, 0);
        const int64_t p = (tile0 + t) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / (net.scale * NCW_TU);
    }
}

}  // namespace

int NCW_FN(ncw_sdf_inferC_launch)(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    const dim3 grid((unsigned)((tiles + PP_TILES - 1) / PP_TILES));
#if defined(NCW_PROBE_BUILD) && !defined(NCW_HALF_F16)
    // round-5 / round-6 A/B (scripts/diag/pp_nb2.py, probe library only): NCW_PP_NB = 2 -> four 512-register waves, two output blocks each
    // (every B fragment feeds two MFMAs: half the LDS reads); with `-mllvm -amdgpu-mfma-vgpr-form` on this file the accumulators
    // stay in VGPRs and the weight slices go to AGPRs (MFMA srcA)
    static const int nb = getenv("NCW_PP_NB") ? atoi(getenv("NCW_PP_NB")) : 1;
    if (nb == 2) hipLaunchKernelGGL((sdf_inferC_kernel<2, 0>), grid, dim3(64 * PP_WAVES / 2), 0, st, *net, src, n, sdf);
    else hipLaunchKernelGGL((sdf_inferC_kernel<1, 0>), grid, dim3(64 * PP_WAVES), 0, st, *net, src, n, sdf);
#else
    hipLaunchKernelGGL((sdf_inferC_kernel<1, 0>), grid, dim3(64 * PP_WAVES), 0, st, *net, src, n, sdf);
#endif
    NCW_CHECK_LAUNCH();
    return 0;
}
