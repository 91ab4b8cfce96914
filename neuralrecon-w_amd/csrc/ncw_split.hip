// Split-precision SDF VALUE path, W = 256, fp16 build only: sdf_inferS, sdf_fwdS.
//
// NeuS turns the SDF into opacity through sigmoid(sdf * inv_s), inv_s = exp(10 * variance) (models/neuconw.py:173-179,
// rendering/renderer.py:624-632; the sampler uses fixed 512 / 1024, renderer.py:527-533).  inv_s is 20 at initialisation and
// several hundred where NeuS trains, so the 3-9e-4 absolute SDF error of one fp16 rounding per operand is 0.1-0.4 in the
// sigmoid's argument there: measured 2e-2 on the rendered colours at inv_s = 403 against 1.5e-4 for the exact-fp32 mode.
// Every layer contributes equally (scripts/diag/emul16.py: splitting only the last layers, or only weights, or only
// activations, does not help), so the whole value chain gamma -> 8 Softplus layers -> sdf row runs here with BOTH operands
// as fp16 hi + lo pairs:   W h  ~=  W_hi h_hi + W_lo h_hi + W_hi h_lo   (three MFMAs, f32 accumulate; the dropped
// W_lo h_lo term is 2^-22 relative) -- fp32-like SDF values on the fp16 matrix pipe, which these kernels leave 75 % idle
// anyway.  The feature rows, the adjoint sweep (normals) and the whole backward stay plain fp16: their errors are not
// multiplied by inv_s.
//
// Structure: the weights-stationary layout of ncw_sdf8.hip (8 waves own the 8 output blocks of a layer; the 128 points'
// activations live in LDS as B fragments, here [tile][k-unit][hi | lo][64 lanes] = 128 KiB, ONE buffer rewritten in place).
//   * burst chain (ss_value_chain: sdf_fwd, and sdf_infer of nets with more than 8 Softplus layers): every wave keeps its
//     block's 4 tile accumulators until all waves have finished reading the layer input; the layer's weight slice (2 x 64
//     registers, hi + lo) is streamed in two K-halves through two static register sets, each prefetched while the other
//     is in use; two barriers per layer;
//   * pipelined chain (s2_value_chain: sdf_infer): one tile per stage, the epilogue of the previous stage between the
//     MFMAs of the current one, one LDS-only barrier per stage, the whole slice resident (see there).
#include "ncw_mlp.h"

#ifdef NCW_HALF_F16

namespace {

constexpr int SS_WAVES = 8, SS_TILES = 4;
constexpr int SS_ACT = SS_TILES * 16 * 2 * 1024;  // [tile][16 units][hi | lo][64 lanes x 16 B]
constexpr int SS_GAM = SS_TILES * 3 * 2 * 1024;   // gamma: [tile][3 units][hi | lo][64 lanes x 16 B]
constexpr int SB_ACT = SS_TILES * 16 * 1024;      // the adjoint sweep's plain buffers (ncw_sdf8.hip layout), aliased

typedef __attribute__((address_space(3))) bf16x8 ss_lfrag;

NCW_DEV bf16x8 ss_gload(const void* w, size_t unit, int lane) {
    return ncw_ld_frag<bf16x8>(w, (size_t)unit, lane);
}

// units [u0, u0 + NU) of output block ob of a packed matrix (hi) and of its residual matrix (lo)
template <int NU>
NCW_DEV void ss_load_half(bf16x8 (&h)[NU], bf16x8 (&l)[NU], const void* w, const void* wlo, int rb_stride, int ob, int u0, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        h[q] = ss_gload(w, (size_t)(u0 + q) * rb_stride + ob, lane);
        l[q] = ss_gload(wlo, (size_t)(u0 + q) * rb_stride + ob, lane);
    }
}

// fp16 hi + lo images of accumulator registers 8t..8t+7 (one k-unit of the next layer): ncw_common.h ncw_split8 (ONE conversion)
NCW_DEV void ss_split8(const f32x16& v, int t, bf16x8& hi, bf16x8& lo) { ncw_split8(v, t, hi, lo); }

// acc[t] += W[ob, units u0 .. u0 + NU) . in[t, the same units], three MFMAs per unit; tiles in pairs so that consecutive
// MFMAs alternate between two accumulators
template <int NU>
NCW_DEV void ss_mma(f32x16 (&acc)[SS_TILES], const bf16x8 (&wh)[NU], const bf16x8 (&wl)[NU], const ss_lfrag* in, int upt, int u0) {
#pragma unroll
    for (int tp = 0; tp < SS_TILES; tp += 2) {
#pragma unroll
        for (int q = 0; q < NU; ++q) {
            const ss_lfrag* p0 = in + ((tp * upt + u0 + q) * 2) * 64;
            const ss_lfrag* p1 = in + (((tp + 1) * upt + u0 + q) * 2) * 64;
            const bf16x8 bh0 = p0[0], bl0 = p0[64], bh1 = p1[0], bl1 = p1[64];
            acc[tp] = NCW_MFMA_H(wh[q], bh0, acc[tp], 0, 0, 0);
            acc[tp + 1] = NCW_MFMA_H(wh[q], bh1, acc[tp + 1], 0, 0, 0);
            acc[tp] = NCW_MFMA_H(wl[q], bh0, acc[tp], 0, 0, 0);
            acc[tp + 1] = NCW_MFMA_H(wl[q], bh1, acc[tp + 1], 0, 0, 0);
            acc[tp] = NCW_MFMA_H(wh[q], bl0, acc[tp], 0, 0, 0);
            acc[tp + 1] = NCW_MFMA_H(wh[q], bl1, acc[tp + 1], 0, 0, 0);
        }
    }
}

NCW_DEV f32x16 ss_bias(const float* bp, int ob, int lane) {
    CVec<1> b1;
    load_bias(b1, bp + ob * 32, lane);
    return b1.v[0];
}

NCW_DEV void ss_load_sprime(f32x16& sv, const ncw_h16* __restrict__ st_h, size_t tile, int RB, int rb, int lane) {
    stash_load_block(sv, st_h, tile, RB, rb, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = 1.f - __builtin_amdgcn_exp2f(sv[r] * -144.26950408889634f);
}

// ------------------------------------------------------------------------------------------------
// The value chain: gamma (exact sin / cos) -> layers 0 .. L-2 (Softplus, f32 epilogue, hi + lo split) -> sdf row.
// Leaves h_{L-1} (hi | lo) of the 128 points in abuf.  STASH 2: also writes the activation stash of ncw_sdf_fwd
// (gamma, h_1 .. h_{L-1}: the fp16 roundings, i.e. the hi parts -- what the plain kernel stashes); STASH 1 (forward-only
// render): h_l only, the scratch the adjoint sweep of the same kernel re-reads; STASH 0 (sdf_infer): nothing.
// ------------------------------------------------------------------------------------------------
template <int STASH>
NCW_DEV void ss_value_chain(const NcwSdfNet& net, const NcwPoints& src, int64_t n, int64_t tile0, ss_lfrag* abuf, ss_lfrag* gbuf,
                            int lane, int wave, float* __restrict__ sdf, const NcwSdfStash& st) {
    typedef ncw_h16 SE;
    // value-only launches (STASH 0: the sampler's queries, sdf(), the grid sweep) run the hidden layers in t-units (ncw_common.h
    // softplus_tu): gamma and the biases x 100 log2 e, the sdf row's result / it; with a stash the chain stays in the units the
    // backward reads
    constexpr bool TU = STASH == 0;
    const float bscale = TU ? NCW_TU : 1.f;
    const int L = net.n_layers;
    if (wave < SS_TILES) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, false>(gam, xs, lane);  // sinf / cosf: the hardware v_sin / v_cos are not fp32-accurate
        if (STASH == 2) stash_store<2>((SE*)st.gamma, (size_t)(tile0 + wave), gam, lane);
        if (TU) cvec_scale<2>(gam, NCW_TU);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bf16x8 hi, lo;
            ss_split8(gam.v[q >> 1], q & 1, hi, lo);
            gbuf[((wave * 3 + q) * 2 + 0) * 64 + lane] = hi;
            gbuf[((wave * 3 + q) * 2 + 1) * 64 + lane] = lo;
        }
    }
    bf16x8 Ah[8], Al[8], Bh[8], Bl[8];  // the two K-halves of the current layer's slice (hi, lo)
    f32x16 acc[SS_TILES];
    const ss_lfrag* const ain = abuf + lane;
    const ss_lfrag* const gin = gbuf + lane;
    // hidden-layer epilogue: Softplus, stash, hi / lo fragments of this wave's block into abuf (in place)
    auto epilogue = [&](int l_out) {
        ncw_lds_barrier();  // every wave has finished reading the layer input
#pragma unroll
        for (int t = 0; t < SS_TILES; ++t) {
            f32x16 yv;
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = softplus_sel<TU>(acc[t][r]);
            if (STASH >= 1) stash_store_block_keep((SE*)st.h[l_out], (size_t)(tile0 + t), 8, wave, yv, lane);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                bf16x8 hi, lo;
                ss_split8(yv, tt, hi, lo);
                abuf[((t * 16 + 2 * wave + tt) * 2 + 0) * 64 + lane] = hi;
                abuf[((t * 16 + 2 * wave + tt) * 2 + 1) * 64 + lane] = lo;
            }
        }
        ncw_lds_barrier();  // the layer output is complete
    };
    // ---- layer 0: K = 39, the 3 gamma units ------------------------------------------------------------------
    {
        bf16x8 w0h[3], w0l[3];
        ss_load_half<3>(w0h, w0l, net.w[0], net.w_lo[0], 8, wave, 0, lane);
        if (L - 1 > 1) ss_load_half<8>(Ah, Al, net.w[1], net.w_lo[1], 8, wave, 0, lane);
        const f32x16 bias = ss_bias(net.b[0], wave, lane) * bscale;
#pragma unroll
        for (int t = 0; t < SS_TILES; ++t) acc[t] = bias;
        ncw_lds_barrier();  // gamma visible
        ss_mma<3>(acc, w0h, w0l, gin, 3, 0);
        epilogue(1);
    }
    // ---- hidden layers 1 .. L-2 ------------------------------------------------------------------------------
    for (int l = 1; l < L - 1; ++l) {
        const f32x16 bias = ss_bias(net.b[l], wave, lane) * bscale;
#pragma unroll
        for (int t = 0; t < SS_TILES; ++t) acc[t] = bias;
        ss_load_half<8>(Bh, Bl, net.w[l], net.w_lo[l], 8, wave, 8, lane);  // second half: lands during the first
        ss_mma<8>(acc, Ah, Al, ain, 16, 0);
        if (l + 1 < L - 1) ss_load_half<8>(Ah, Al, net.w[l + 1], net.w_lo[l + 1], 8, wave, 0, lane);  // next layer, first half
        ss_mma<8>(acc, Bh, Bl, ain, 16, 8);
        if (l == net.skip_layer) {  // the gamma columns: units 16..18 (once per launch: loaded just in time)
            bf16x8 wgh[3], wgl[3];
            ss_load_half<3>(wgh, wgl, net.w[l], net.w_lo[l], 8, wave, 16, lane);
            ss_mma<3>(acc, wgh, wgl, gin, 3, 0);
        }
        epilogue(l + 1);
    }
    // ---- sdf row (1 output block), tile t by wave t ------------------------------------------------------------
    if (wave < SS_TILES) {
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        if (TU) cvec_scale<1>(o, NCW_TU);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 wh[8], wl[8];
            ss_load_half<8>(wh, wl, net.w[L - 1], net.w_lo[L - 1], 1, 0, 8 * half, lane);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const ss_lfrag* p = ain + ((wave * 16 + 8 * half + q) * 2) * 64;
                const bf16x8 bh = p[0], bl = p[64];
                o.v[0] = NCW_MFMA_H(wh[q], bh, o.v[0], 0, 0, 0);
                o.v[0] = NCW_MFMA_H(wl[q], bh, o.v[0], 0, 0, 0);
                o.v[0] = NCW_MFMA_H(wh[q], bl, o.v[0], 0, 0, 0);
            }
        }
        const int64_t p = (tile0 + wave) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / (net.scale * bscale);
    }
}

// ------------------------------------------------------------------------------------------------
// Version 2 of the value chain: the fine-interleaved pipeline of ncw_pp.hip with split operands, one TILE
// per pipeline stage.  Segment (l, t) issues the 48 MFMAs of tile t of layer l (16 k-units x {hi.hi, lo.hi, hi.lo}) with
// the epilogue of the PREVIOUS segment's finished accumulator (Softplus, hi / lo split, LDS + stash stores) between them,
// three MFMAs and one accumulator register per step; ONE LDS-only barrier per segment.  A tile's activation region
// ([16 units][hi | lo], 32 KiB) is rewritten IN PLACE: within a segment it is either read (its MFMAs) or written (its
// epilogue), never both.  The layer's weight slice (16 units hi + 16 units lo = 128 registers) stays resident for the four
// tiles and is reloaded in place during the layer's last segment; two 16-register accumulators alternate.
// ------------------------------------------------------------------------------------------------
constexpr int S2_TILE = 16 * 2 * 1024;     // one tile region

constexpr int S2_BIAS = 8 * 1024;          // bias staging: up to 8 Softplus layers x 256 f32 (128 + 24 + 8 KiB = all of the LDS)
typedef __attribute__((address_space(3))) f32x4 ss_lf4;

NCW_DEV void s2_barrier() {  // LDS stores of this phase complete; vector-memory traffic stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

NCW_DEV f32x16 s2_bias(const ss_lf4* bb, int lane) {
    const ss_lf4* p = bb + (lane >> 5) * 4;
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = p[g];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = t[c];
    }
    return v;
}

// extra k-units (gamma: NU = 3) of one tile: xin = gbuf + tile * NU * 2 * 64 + lane
template <int NU>
NCW_DEV void s2_mma_x(f32x16& a, const bf16x8* wh, const bf16x8* wl, const ss_lfrag* xin) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bf16x8 bh = xin[(u * 2) * 64], bl = xin[(u * 2 + 1) * 64];
        a = NCW_MFMA_H(wh[u], bh, a, 0, 0, 0);
        a = NCW_MFMA_H(wl[u], bh, a, 0, 0, 0);
        a = NCW_MFMA_H(wh[u], bl, a, 0, 0, 0);
    }
}

// One segment: m += W[ob] . h of one tile (in = tile region + lane; m holds the bias on entry), epi(u) after the three MFMAs
// of unit u.  PREFETCH: unit u of the NEXT layer goes into the registers unit u has just left.
constexpr int S2_RING = 3;  // depths 2 / 3 / 4 measured the same (0.38 ms per 131k points on one box): not LDS latency
template <bool PREFETCH, class EPI>
NCW_DEV void s2_segment(f32x16& m, bf16x8 (&wh)[16], bf16x8 (&wl)[16], const ss_lfrag* in, const void* wn, const void* wn_lo, int ob,
                        int lane, EPI&& epi) {
    // two independent accumulator chains (a dependent MFMA waits for its predecessor's result): m takes the hi.hi terms, c the
    // two cross terms -- which also keeps the small terms from being absorbed one by one into a large partial sum
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    constexpr int RD = S2_RING - 1;  // the B fragments {hi, lo} of a k-unit are read RD units ahead of their use
    bf16x8 b[S2_RING][2];
#pragma unroll
    for (int q = 0; q < RD; ++q) { b[q][0] = in[(q * 2) * 64]; b[q][1] = in[(q * 2 + 1) * 64]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u + RD < 16) {
            b[(u + RD) % S2_RING][0] = in[((u + RD) * 2) * 64];
            b[(u + RD) % S2_RING][1] = in[((u + RD) * 2 + 1) * 64];
        }
        m = NCW_MFMA_H(wh[u], b[u % S2_RING][0], m, 0, 0, 0);
        c = NCW_MFMA_H(wl[u], b[u % S2_RING][0], c, 0, 0, 0);
        c = NCW_MFMA_H(wh[u], b[u % S2_RING][1], c, 0, 0, 0);
        if (PREFETCH) {
            wh[u] = ss_gload(wn, (size_t)u * 8 + ob, lane);
            wl[u] = ss_gload(wn_lo, (size_t)u * 8 + ob, lane);
        }
        epi(u);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) m[r] += c[r];
}

// Epilogue step u (0..15) of a finished tile accumulator: register u goes through Softplus and is split into fp16 hi + lo;
// every fourth step one 8-byte stash piece (4 consecutive features), every eighth step one k-unit (hi fragment + lo fragment)
// of the next layer's input, in place.
template <bool STASH>
NCW_DEV void s2_epi_step(int u, const f32x16& e, bf16x8& fh, bf16x8& fl, float& ycarry, ss_lfrag* out, int ob, int lane,
                         ncw_h16* __restrict__ st_h, size_t tile) {
    const float y = softplus_sel<!STASH>(e[u]);  // (value-only launches: t-units, see ss_value_chain)
    // hi / lo in PAIRS (even step: keep y; odd step: one packed conversion, both residuals from it -- ncw_common.h ncw_split2)
    if ((u & 1) == 0) {
        ycarry = y;
    } else {
        ncw_h16 h0, h1, l0, l1;
        ncw_split2(ycarry, y, h0, h1, l0, l1);
        fh[(u & 7) - 1] = h0; fh[u & 7] = h1;
        fl[(u & 7) - 1] = l0; fl[u & 7] = l1;
    }
    if (STASH && (u & 3) == 3) {  // registers 4g .. 4g+3, g = u >> 2: one [tile][block][g][lane][4] piece of the stash
        bf16x4 t;
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = fh[(u & 7) - 3 + c];
        reinterpret_cast<bf16x4*>(st_h)[((tile * 8 + ob) * 4 + (u >> 2)) * 64 + lane] = t;
    }
    if ((u & 7) == 7) {
        const int unit = 2 * ob + (u >> 3);
        out[(unit * 2 + 0) * 64 + lane] = fh;
        out[(unit * 2 + 1) * 64 + lane] = fl;
    }
}

template <bool STASH>
NCW_DEV void s2_value_chain(const NcwSdfNet& net, const NcwPoints& src, int64_t n, int64_t tile0, ss_lfrag* abuf, ss_lfrag* gbuf,
                            ss_lf4* bbuf, int lane, int wave, float* __restrict__ sdf, const NcwSdfStash& st) {
    typedef ncw_h16 SE;
    const int L = net.n_layers, NL = L - 1;
    const int ob = wave;
    constexpr bool TU = !STASH;  // value-only: t-units (ss_value_chain)
    const float bscale = TU ? NCW_TU : 1.f;
    for (int i = threadIdx.x; i < NL * 64; i += 64 * SS_WAVES) bbuf[i] = reinterpret_cast<const f32x4*>(net.b[i >> 6])[i & 63] * bscale;
    if (wave < SS_TILES) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        if (p >= n) p = n - 1;
        float xs[3];
        load_point(src, p, xs, ray);
        xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
        CVec<2> gam;
        freq_encode<2, 3, 6, false>(gam, xs, lane);  // sinf / cosf: the hardware v_sin / v_cos are not fp32-accurate
        if (STASH == 2) stash_store<2>((SE*)st.gamma, (size_t)(tile0 + wave), gam, lane);
        if (TU) cvec_scale<2>(gam, NCW_TU);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bf16x8 hi, lo;
            ss_split8(gam.v[q >> 1], q & 1, hi, lo);
            gbuf[((wave * 3 + q) * 2 + 0) * 64 + lane] = hi;
            gbuf[((wave * 3 + q) * 2 + 1) * 64 + lane] = lo;
        }
    }
    bf16x8 wh[16], wl[16];
    f32x16 x, y;  // tiles 0, 2 accumulate into x, tiles 1, 3 into y
    bf16x8 fh, fl;
    float yc = 0.f;  // s2_epi_step's carry between an even and the following odd step
    auto region = [&](int t) { return abuf + t * (S2_TILE / 16); };
    auto gin = [&](int t) { return gbuf + t * 3 * 2 * 64 + lane; };
    auto bias_of = [&](int l) { return s2_bias(bbuf + (l * 8 + ob) * 8, lane); };
    // ---- layer 0 (K = 39: the 3 gamma units) for the four tiles, then E(0, t = 0..2); E(0, 3) rides on segment (1, 0) --------
    {
        ss_load_half<3>(reinterpret_cast<bf16x8(&)[3]>(wh[0]), reinterpret_cast<bf16x8(&)[3]>(wl[0]), net.w[0], net.w_lo[0], 8, ob, 0, lane);
        s2_barrier();  // gamma + biases visible
        x = bias_of(0);
        y = x;
        s2_mma_x<3>(x, wh, wl, gin(0));
        s2_mma_x<3>(y, wh, wl, gin(1));
#pragma unroll
        for (int u = 0; u < 16; ++u) s2_epi_step<STASH>(u, x, fh, fl, yc, region(0), ob, lane, (SE*)st.h[1], (size_t)tile0);
#pragma unroll
        for (int u = 0; u < 16; ++u) s2_epi_step<STASH>(u, y, fh, fl, yc, region(1), ob, lane, (SE*)st.h[1], (size_t)(tile0 + 1));
        x = bias_of(0);
        y = x;
        s2_mma_x<3>(x, wh, wl, gin(2));
        s2_mma_x<3>(y, wh, wl, gin(3));
        // the first hidden layer's slice (its latency hides behind the epilogue)
        if (NL > 1) ss_load_half<16>(wh, wl, net.w[1], net.w_lo[1], 8, ob, 0, lane);
#pragma unroll
        for (int u = 0; u < 16; ++u) s2_epi_step<STASH>(u, x, fh, fl, yc, region(2), ob, lane, (SE*)st.h[1], (size_t)(tile0 + 2));
        s2_barrier();
    }
    // ---- hidden layer l >= 1, tile t: [M(l, t) | E(previous segment)] bar ---------------------------------------------------
    for (int l = 1; l < NL; ++l) {
        const bool more = l + 1 < NL, skip = (l == net.skip_layer);
        bf16x8 gh[3], gl[3];
        if (skip) ss_load_half<3>(gh, gl, net.w[l], net.w_lo[l], 8, ob, 16, lane);  // the gamma columns: units 16..18
        {   // t = 0 -> x;  E(l-1, 3) <- y
            x = bias_of(l);
            auto epi = [&](int u) { s2_epi_step<STASH>(u, y, fh, fl, yc, region(3), ob, lane, (SE*)st.h[l], (size_t)(tile0 + 3)); };
            s2_segment<false>(x, wh, wl, region(0) + lane, nullptr, nullptr, ob, lane, epi);
            if (skip) s2_mma_x<3>(x, gh, gl, gin(0));
            s2_barrier();
        }
        {   // t = 1 -> y;  E(l, 0) <- x
            y = bias_of(l);
            auto epi = [&](int u) { s2_epi_step<STASH>(u, x, fh, fl, yc, region(0), ob, lane, (SE*)st.h[l + 1], (size_t)tile0); };
            s2_segment<false>(y, wh, wl, region(1) + lane, nullptr, nullptr, ob, lane, epi);
            if (skip) s2_mma_x<3>(y, gh, gl, gin(1));
            s2_barrier();
        }
        {   // t = 2 -> x;  E(l, 1) <- y
            x = bias_of(l);
            auto epi = [&](int u) { s2_epi_step<STASH>(u, y, fh, fl, yc, region(1), ob, lane, (SE*)st.h[l + 1], (size_t)(tile0 + 1)); };
            s2_segment<false>(x, wh, wl, region(2) + lane, nullptr, nullptr, ob, lane, epi);
            if (skip) s2_mma_x<3>(x, gh, gl, gin(2));
            s2_barrier();
        }
        {   // t = 3 -> y;  E(l, 2) <- x; the layer's last use of its weight slice -> the next layer's takes its registers
            y = bias_of(l);
            auto epi = [&](int u) { s2_epi_step<STASH>(u, x, fh, fl, yc, region(2), ob, lane, (SE*)st.h[l + 1], (size_t)(tile0 + 2)); };
            // (ONE code path -- the last layer reloads its own slice, unused: two variants behind `if (more)` let hipcc hoist the
            // epilogue's v_exp in front of the branch, out from between the MFMAs; csrc/ncw_pp.hip)
            s2_segment<true>(y, wh, wl, region(3) + lane, more ? net.w[l + 1] : net.w[l], more ? net.w_lo[l + 1] : net.w_lo[l], ob, lane, epi);
            if (skip) s2_mma_x<3>(y, gh, gl, gin(3));
            s2_barrier();
        }
    }
    // ---- drain: E(NL-1, 3) --------------------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < 16; ++u) s2_epi_step<STASH>(u, y, fh, fl, yc, region(3), ob, lane, (SE*)st.h[NL], (size_t)(tile0 + 3));
    s2_barrier();
    // ---- sdf row (1 output block), tile t by wave t -------------------------------------------------------------------
    if (wave < SS_TILES) {
        CVec<1> o;
        load_bias(o, net.b[L - 1], lane);
        if (TU) cvec_scale<1>(o, NCW_TU);
        const ss_lfrag* in = region(wave) + lane;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 vh[8], vl[8];
            ss_load_half<8>(vh, vl, net.w[L - 1], net.w_lo[L - 1], 1, 0, 8 * half, lane);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const ss_lfrag* p = in + ((8 * half + q) * 2) * 64;
                const bf16x8 bh = p[0], bl = p[64];
                o.v[0] = NCW_MFMA_H(vh[q], bh, o.v[0], 0, 0, 0);
                o.v[0] = NCW_MFMA_H(vl[q], bh, o.v[0], 0, 0, 0);
                o.v[0] = NCW_MFMA_H(vh[q], bl, o.v[0], 0, 0, 0);
            }
        }
        const int64_t p = (tile0 + wave) * 32 + (lane & 31);
        if (p < n && lane < 32) sdf[p] = o.v[0][0] / (net.scale * bscale);
    }
}

__global__ __launch_bounds__(64 * SS_WAVES) void sdf_inferS2_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                   float* __restrict__ sdf) {
    __shared__ __attribute__((aligned(16))) char lds[SS_ACT + SS_GAM + S2_BIAS];
    ss_lfrag* const abuf = (ss_lfrag*)(ncw_lchar*)lds;
    ss_lfrag* const gbuf = abuf + SS_ACT / 16;
    ss_lf4* const bbuf = (ss_lf4*)(gbuf + SS_GAM / 16);
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    NcwSdfStash none = {};
    s2_value_chain<false>(net, src, n, (int64_t)blockIdx.x * SS_TILES, abuf, gbuf, bbuf, lane, wave, sdf, none);
}

__global__ __launch_bounds__(64 * SS_WAVES) void sdf_inferS_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                  float* __restrict__ sdf) {
    __shared__ __attribute__((aligned(16))) char lds[SS_ACT + SS_GAM];
    ss_lfrag* const abuf = (ss_lfrag*)(ncw_lchar*)lds;
    ss_lfrag* const gbuf = abuf + SS_ACT / 16;
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    NcwSdfStash none = {};
    ss_value_chain<0>(net, src, n, (int64_t)blockIdx.x * SS_TILES, abuf, gbuf, lane, wave, sdf, none);
}

// ------------------------------------------------------------------------------------------------
// sdf_fwd: the split value chain (with the activation stash), then -- plain fp16, exactly the arithmetic of
// sdf_fwdB_kernel (ncw_sdf8.hip) -- the feature rows, the analytic adjoint sweep with the t_l stash and grad = J_gamma^T g_gamma.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) bf16x8 sb_lfrag;

template <int NU>
NCW_DEV void sb_load_slice(bf16x8* a, const void* w, int rb_stride, int ob, int u0, int lane) {
#pragma unroll
    for (int q = 0; q < NU; ++q) a[q] = ss_gload(w, (size_t)(u0 + q) * rb_stride + ob, lane);
}

NCW_DEV void sb_store_units(sb_lfrag* buf, int t, int ob, const f32x16& v, int lane) {
    Act<PrecBF16, 1> o;
    to_act_block<1>(o, 0, v);
    buf[(t * 16 + 2 * ob) * 64 + lane] = o.f[0];
    buf[(t * 16 + 2 * ob + 1) * 64 + lane] = o.f[1];
}

// TRAIN = false: the forward-only render (validation / novel views / vertex colours, rendering/renderer.py:785-916 under
// no_grad): the same arithmetic bit for bit, but of the stash only h_l (re-read by the adjoint sweep below) and feat (the
// colour network's input) are written -- not gamma, not t_l.
template <bool TRAIN>
__global__ __launch_bounds__(64 * SS_WAVES) void sdf_fwdS_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                float* __restrict__ sdf, float* __restrict__ grad,
                                                                NcwSdfStash st) {
    typedef ncw_h16 SE;
    __shared__ __attribute__((aligned(16))) char lds[SS_ACT + SS_GAM];
    ss_lfrag* const sbuf = (ss_lfrag*)(ncw_lchar*)lds;
    ss_lfrag* const gbuf = sbuf + SS_ACT / 16;
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int L = net.n_layers;
    const int64_t tile0 = (int64_t)blockIdx.x * SS_TILES;
    const int jb = wave & 1, jt = wave >> 1;  // this wave's gamma job of the adjoint sweep: block jb of tile jt
    ss_value_chain<(TRAIN ? 2 : 1)>(net, src, n, tile0, sbuf, gbuf, lane, wave, sdf, st);
    // ---- feature rows (plain fp16: the colour network reads them from the fp16 stash): W_feat . h_hi ----------------
    bf16x8 wa[16], wb[16];
    {
        sb_load_slice<16>(wa, net.w_feat, 8, wave, 0, lane);
        const bf16x8 wt1 = ss_gload(net.wt[L - 1], (size_t)wave, lane);  // W_{L-1}^T: unit 0, block = wave
        if (L - 2 >= 1) sb_load_slice<16>(wb, net.wt[L - 2], (L - 2 == net.skip_layer) ? 10 : 8, wave, 0, lane);
        const f32x16 bias = ss_bias(net.b_feat, wave, lane);
        const ss_lfrag* const ain = sbuf + lane;
#pragma unroll
        for (int tp = 0; tp < SS_TILES; tp += 2) {
            f32x16 acc0 = bias, acc1 = bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], ain[((tp * 16 + q) * 2) * 64], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], ain[(((tp + 1) * 16 + q) * 2) * 64], acc1, 0, 0, 0);
            }
            stash_store_block((SE*)st.feat, (size_t)(tile0 + tp), 8, wave, acc0, lane);
            stash_store_block((SE*)st.feat, (size_t)(tile0 + tp + 1), 8, wave, acc1, lane);
        }
        // ---- adjoint start: a_{L-2} = W_{L-1}^T e_0 (the same for every point); t_{L-2} = a * phi'(z_{L-2}) ----------
        bf16x8 e0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) e0f[e] = (ncw_h16)0.f;
        e0f[0] = (ncw_h16)(lane < 32 ? 1.f : 0.f);
        f32x16 a0;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.f;
        a0 = NCW_MFMA_H(wt1, e0f, a0, 0, 0, 0);
        ncw_lds_barrier();  // every wave is done with the split buffer (feature rows, sdf row): the plain buffers alias it
        sb_lfrag* out = (sb_lfrag*)sbuf;  // abuf0
#pragma unroll
        for (int t = 0; t < SS_TILES; ++t) {
            f32x16 sv;
            ss_load_sprime(sv, (const SE*)st.h[L - 1], (size_t)(tile0 + t), 8, wave, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] *= a0[r];
            if (TRAIN) stash_store_block((SE*)st.t[L - 2], (size_t)(tile0 + t), 8, wave, sv, lane);
            sb_store_units(out, t, wave, sv, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
    }
    sb_lfrag* const abuf0 = (sb_lfrag*)sbuf;
    sb_lfrag* const abuf1 = abuf0 + SB_ACT / 16;
    int cur = 0;  // t_{L-2} lives in abuf0
    // ---- adjoint layers l = L-2 .. 1: t_{l-1} = (W_l^T t_l) * phi'(z_{l-1});  wa = slice of wt[l] ---------------
    f32x16 gg;
#pragma unroll
    for (int r = 0; r < 16; ++r) gg[r] = 0.f;
    for (int l = L - 2; l >= 1; --l) {
        const bool skip = (l == net.skip_layer);
        if (!skip && l - 1 >= 1) sb_load_slice<16>(wb, net.wt[l - 1], (l - 1 == net.skip_layer) ? 10 : 8, wave, 0, lane);
        ncw_lds_barrier();  // t_l complete in abuf[cur]
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SS_TILES; tp += 2) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wa[q], in[(tp * 16 + q) * 64 + lane], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], in[((tp + 1) * 16 + q) * 64 + lane], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& acc = j ? acc1 : acc0;
                f32x16 sv;
                ss_load_sprime(sv, (const SE*)st.h[l], (size_t)(tile0 + tp + j), 8, wave, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] *= acc[r];
                if (TRAIN) stash_store_block((SE*)st.t[l - 1], (size_t)(tile0 + tp + j), 8, wave, sv, lane);
                sb_store_units(out, tp + j, wave, sv, lane);
            }
        }
        if (skip) {  // the gamma columns of the transposed skip layer: out-blocks 8, 9 (one (block, tile) job per wave)
            sb_load_slice<16>(wb, net.wt[l], 10, 8 + jb, 0, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) gg = NCW_MFMA_H(wb[q], in[(jt * 16 + q) * 64 + lane], gg, 0, 0, 0);
            if (l - 1 >= 1) sb_load_slice<16>(wb, net.wt[l - 1], (l - 1 == net.skip_layer) ? 10 : 8, wave, 0, lane);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = wb[q];
        cur ^= 1;
    }
    // ---- adjoint layer 0: g_gamma += W_0^T t_0 (2 out-blocks), then grad = J_gamma^T g_gamma -----------------------
    sb_load_slice<16>(wb, net.wt[0], 2, jb, 0, lane);
    ncw_lds_barrier();
    {
        const sb_lfrag* in = cur ? abuf1 : abuf0;
#pragma unroll
        for (int q = 0; q < 16; ++q) gg = NCW_MFMA_H(wb[q], in[(jt * 16 + q) * 64 + lane], gg, 0, 0, 0);
    }
    int64_t p = (tile0 + jt) * 32 + (lane & 31), ray;
    const bool valid = p < n;
    if (!valid) p = n - 1;
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f0 = 32 * jb + ncw_feat_of(r, 0);
        if (f0 >= 39) continue;  // (block 1 holds features 32..38 only)
        int comp;
        const float dv = freq_feature_deriv<3, 6, true>(xs, f0 + 4 * h, comp);
        const float c = gg[r] * dv;
        nx += comp == 0 ? c : 0.f;
        ny += comp == 1 ? c : 0.f;
        nz += comp == 2 ? c : 0.f;
    }
    nx = half_pair_sum(nx); ny = half_pair_sum(ny); nz = half_pair_sum(nz);
    // combine the two blocks of a tile (waves 2 jt and 2 jt + 1) through LDS (the gamma region is free now)
    typedef __attribute__((address_space(3))) float lfloat;
    lfloat* part = (lfloat*)gbuf;
    if (jb == 1 && lane < 32) {
        part[(jt * 32 + lane) * 3 + 0] = nx; part[(jt * 32 + lane) * 3 + 1] = ny; part[(jt * 32 + lane) * 3 + 2] = nz;
    }
    ncw_lds_barrier();
    if (jb == 0 && lane < 32 && valid) {
        grad[p * 3 + 0] = nx + part[(jt * 32 + lane) * 3 + 0];
        grad[p * 3 + 1] = ny + part[(jt * 32 + lane) * 3 + 1];
        grad[p * 3 + 2] = nz + part[(jt * 32 + lane) * 3 + 2];
    }
}

// ------------------------------------------------------------------------------------------------
// sdf_fwd with the ADJOINT SWEEP'S WEIGHTS AS hi + lo PAIRS (NcwSdfNet.wt_lo; round 5).  The normals n = grad sdf leave this
// kernel for the compositor, which multiplies their component along the ray by dist * inv_s inside the sigmoid
// (rendering/renderer.py:600-632): on TRAINED weights (inv_s 403) the single-rounded fp16 adjoint sweep -- 4e-4 on the normals --
// was what put single rays of the timed batch above 1e-4 (scripts/diag/emul_timed_batch.py: 6 of 256 rays -> the sweep's weight
// rounding alone; with W^T = W_hi^T + W_lo^T the worst rays drop 1.5e-4 -> 3.7e-5, t_l may stay single fp16).
// Structure: the value chain and the feature rows as in sdf_fwdS_kernel; in the sweep every B fragment read from LDS feeds FOUR
// MFMAs (hi and lo slice x two tiles); the layer's two 16-unit slices (128 registers) are reloaded IN PLACE for the next layer
// during the layer's last tile pair (unit q of wt[l-1] into the registers unit q of wt[l] has just left), so there is no second
// prefetch set.  The stash t_l, the LDS layout and everything ncw_sdf_bwd reads are unchanged (forward only).
// ------------------------------------------------------------------------------------------------
template <bool TRAIN>
__global__ __launch_bounds__(64 * SS_WAVES) void sdf_fwdSA_kernel(NcwSdfNet net, NcwPoints src, int64_t n,
                                                                 float* __restrict__ sdf, float* __restrict__ grad,
                                                                 NcwSdfStash st) {
    typedef ncw_h16 SE;
    __shared__ __attribute__((aligned(16))) char lds[SS_ACT + SS_GAM];
    ss_lfrag* const sbuf = (ss_lfrag*)(ncw_lchar*)lds;
    ss_lfrag* const gbuf = sbuf + SS_ACT / 16;
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int L = net.n_layers;
    const int64_t tile0 = (int64_t)blockIdx.x * SS_TILES;
    const int jb = wave & 1, jt = wave >> 1;  // this wave's gamma job of the adjoint sweep: block jb of tile jt
    ss_value_chain<(TRAIN ? 2 : 1)>(net, src, n, tile0, sbuf, gbuf, lane, wave, sdf, st);
    auto stride_of = [&](int l) { return l == net.skip_layer ? 10 : (l == 0 ? 2 : 8); };
    bf16x8 wa[16], wl[16];  // hi / lo slice (output block = wave) of the current adjoint layer's transposed matrix
    // ---- feature rows (plain fp16: the colour network reads them from the fp16 stash): W_feat . h_hi ----------------
    {
        sb_load_slice<16>(wl, net.w_feat, 8, wave, 0, lane);  // (wl is free until the sweep starts)
        const bf16x8 wt1 = ss_gload(net.wt[L - 1], (size_t)wave, lane);      // W_{L-1}^T: unit 0, block = wave
        const bf16x8 wt1l = ss_gload(net.wt_lo[L - 1], (size_t)wave, lane);  // ... and its residual
        if (L - 2 >= 1) sb_load_slice<16>(wa, net.wt[L - 2], stride_of(L - 2), wave, 0, lane);
        const f32x16 bias = ss_bias(net.b_feat, wave, lane);
        const ss_lfrag* const ain = sbuf + lane;
#pragma unroll
        for (int tp = 0; tp < SS_TILES; tp += 2) {
            f32x16 acc0 = bias, acc1 = bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = NCW_MFMA_H(wl[q], ain[((tp * 16 + q) * 2) * 64], acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wl[q], ain[(((tp + 1) * 16 + q) * 2) * 64], acc1, 0, 0, 0);
            }
            stash_store_block((SE*)st.feat, (size_t)(tile0 + tp), 8, wave, acc0, lane);
            stash_store_block((SE*)st.feat, (size_t)(tile0 + tp + 1), 8, wave, acc1, lane);
        }
        if (L - 2 >= 1) sb_load_slice<16>(wl, net.wt_lo[L - 2], stride_of(L - 2), wave, 0, lane);
        // ---- adjoint start: a_{L-2} = W_{L-1}^T e_0 (hi + lo; the same for every point); t_{L-2} = a * phi'(z_{L-2}) ----
        bf16x8 e0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) e0f[e] = (ncw_h16)0.f;
        e0f[0] = (ncw_h16)(lane < 32 ? 1.f : 0.f);
        f32x16 a0;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.f;
        a0 = NCW_MFMA_H(wt1, e0f, a0, 0, 0, 0);
        a0 = NCW_MFMA_H(wt1l, e0f, a0, 0, 0, 0);
        ncw_lds_barrier();  // every wave is done with the split buffer (feature rows, sdf row): the plain buffers alias it
        sb_lfrag* out = (sb_lfrag*)sbuf;  // abuf0
#pragma unroll
        for (int t = 0; t < SS_TILES; ++t) {
            f32x16 sv;
            ss_load_sprime(sv, (const SE*)st.h[L - 1], (size_t)(tile0 + t), 8, wave, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] *= a0[r];
            if (TRAIN) stash_store_block((SE*)st.t[L - 2], (size_t)(tile0 + t), 8, wave, sv, lane);
            sb_store_units(out, t, wave, sv, lane);
        }
    }
    sb_lfrag* const abuf0 = (sb_lfrag*)sbuf;
    sb_lfrag* const abuf1 = abuf0 + SB_ACT / 16;
    int cur = 0;  // t_{L-2} lives in abuf0
    f32x16 gg;
#pragma unroll
    for (int r = 0; r < 16; ++r) gg[r] = 0.f;
    // gamma job: gg += (W^T)[block ob of `w` / `wlo`] . t (tile jt), hi + lo, in two 8-unit chunks (64 temporary registers)
    auto gamma_job = [&](const void* w, const void* wlo, int stride, int ob, const sb_lfrag* in) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 th[8], tl[8];
            sb_load_slice<8>(th, w, stride, ob, 8 * c, lane);
            sb_load_slice<8>(tl, wlo, stride, ob, 8 * c, lane);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bf16x8 b = in[(jt * 16 + 8 * c + q) * 64 + lane];
                gg = NCW_MFMA_H(th[q], b, gg, 0, 0, 0);
                gg = NCW_MFMA_H(tl[q], b, gg, 0, 0, 0);
            }
        }
    };
    // ---- adjoint layers l = L-2 .. 1: t_{l-1} = ((W_hi^T + W_lo^T) t_l) * phi'(z_{l-1}) ------------------------------------
    for (int l = L - 2; l >= 1; --l) {
        const bool skip = (l == net.skip_layer);
        const bool more = l - 1 >= 1;           // a further 8-block layer follows: its slices replace this layer's in place
        const void* nh = more ? net.wt[l - 1] : nullptr;
        const void* nl = more ? net.wt_lo[l - 1] : nullptr;
        const int nstride = more ? stride_of(l - 1) : 8;
        ncw_lds_barrier();  // t_l complete in abuf[cur]
        const sb_lfrag* in = cur ? abuf1 : abuf0;
        sb_lfrag* out = cur ? abuf0 : abuf1;
#pragma unroll
        for (int tp = 0; tp < SS_TILES; tp += 2) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const bf16x8 b0 = in[(tp * 16 + q) * 64 + lane], b1 = in[((tp + 1) * 16 + q) * 64 + lane];
                acc0 = NCW_MFMA_H(wa[q], b0, acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wa[q], b1, acc1, 0, 0, 0);
                acc0 = NCW_MFMA_H(wl[q], b0, acc0, 0, 0, 0);
                acc1 = NCW_MFMA_H(wl[q], b1, acc1, 0, 0, 0);
                if (tp == SS_TILES - 2 && more) {  // the last use of unit q in this layer: the next layer's unit q takes its registers
                    wa[q] = ss_gload(nh, (size_t)q * nstride + wave, lane);
                    wl[q] = ss_gload(nl, (size_t)q * nstride + wave, lane);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& acc = j ? acc1 : acc0;
                f32x16 sv;
                ss_load_sprime(sv, (const SE*)st.h[l], (size_t)(tile0 + tp + j), 8, wave, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] *= acc[r];
                if (TRAIN) stash_store_block((SE*)st.t[l - 1], (size_t)(tile0 + tp + j), 8, wave, sv, lane);
                sb_store_units(out, tp + j, wave, sv, lane);
            }
        }
        if (skip) gamma_job(net.wt[l], net.wt_lo[l], 10, 8 + jb, in);  // the gamma columns of the transposed skip layer: out-blocks 8, 9
        cur ^= 1;
    }
    // ---- adjoint layer 0: g_gamma += W_0^T t_0 (2 out-blocks), then grad = J_gamma^T g_gamma -----------------------
    ncw_lds_barrier();
    gamma_job(net.wt[0], net.wt_lo[0], 2, jb, cur ? abuf1 : abuf0);
    int64_t p = (tile0 + jt) * 32 + (lane & 31), ray;
    const bool valid = p < n;
    if (!valid) p = n - 1;
    float xs[3];
    load_point(src, p, xs, ray);
    xs[0] *= net.scale; xs[1] *= net.scale; xs[2] *= net.scale;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f0 = 32 * jb + ncw_feat_of(r, 0);
        if (f0 >= 39) continue;  // (block 1 holds features 32..38 only)
        int comp;
        const float dv = freq_feature_deriv<3, 6, false>(xs, f0 + 4 * h, comp);  // sinf / cosf like the value chain's gamma
        const float c = gg[r] * dv;
        nx += comp == 0 ? c : 0.f;
        ny += comp == 1 ? c : 0.f;
        nz += comp == 2 ? c : 0.f;
    }
    nx = half_pair_sum(nx); ny = half_pair_sum(ny); nz = half_pair_sum(nz);
    // combine the two blocks of a tile (waves 2 jt and 2 jt + 1) through LDS (the gamma region is free now)
    typedef __attribute__((address_space(3))) float lfloat;
    lfloat* part = (lfloat*)gbuf;
    if (jb == 1 && lane < 32) {
        part[(jt * 32 + lane) * 3 + 0] = nx; part[(jt * 32 + lane) * 3 + 1] = ny; part[(jt * 32 + lane) * 3 + 2] = nz;
    }
    ncw_lds_barrier();
    if (jb == 0 && lane < 32 && valid) {
        grad[p * 3 + 0] = nx + part[(jt * 32 + lane) * 3 + 0];
        grad[p * 3 + 1] = ny + part[(jt * 32 + lane) * 3 + 1];
        grad[p * 3 + 2] = nz + part[(jt * 32 + lane) * 3 + 2];
    }
}

// ------------------------------------------------------------------------------------------------
// Background NeRF, FORWARD REFINEMENT in split precision (round 5): density / raw rgb of a SELECTION of points (NcwPoints
// mode 4: the samples the compositor can use, ncw_bg_select) re-evaluated with every operand of every product as an fp16 hi + lo
// pair -- gamma_10(p4), weights and hidden activations: W x ~= W_hi x_hi + W_lo x_hi + W_hi x_lo, f32 accumulate -- and
// written over the plain-fp16 outputs of ncw_nerf_fwd at those samples.  Why: on TRAINED weights the rays whose colour is all
// background (weights_sum ~ 0: sky) carry the plain-fp16 NeRF's error undiluted -- the worst rays of the timed batch, 1.1e-4 to
// 1.6e-4 on two of four seeds, and only ALL three operand groups as hi + lo pairs remove it (scripts/diag/emul_timed_batch.py
// --candidates: 1.6e-4 -> 4e-8; weights alone: no change).  The selection is ~7 % of the background samples (the rest is
// multiplied by 1 - inside_sphere = 0 in the compositor, rendering/renderer.py:693-708), so three MFMAs per product cost less here
// than hi + lo weights would on every sample.  Forward only: stash, ncw_nerf_bwd and the weight gradients are those of the plain
// launch.  The appearance head's view-direction / appearance-code columns come from the per-ray fp32 rows of ncw_aux_ray_bias.
// Structure: 8 waves, 2 tiles (64 points) per workgroup; wave w owns output block w of a 256-wide layer (both tiles), block w & 3
// of tile w >> 2 in the 128-wide head; activations [tile][k-unit][hi | lo] in LDS, rewritten in place; weight slices hi + lo in two
// K-halves of 8 units.
// ------------------------------------------------------------------------------------------------
constexpr int NS_WAVES = 8, NS_TILES = 2;
constexpr int NS_ACT = NS_TILES * 16 * 2 * 1024;  // [tile][16 units][hi | lo][64 lanes x 16 B]
constexpr int NS_X = NS_TILES * 6 * 2 * 1024;     // gamma_10(p4): 84 features = 6 units, hi | lo

NCW_DEV void ns_inverted_sphere(const float (&x)[3], float (&p4)[4]) {  // renderer.py:181-186
    float r = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    r = fminf(fmaxf(r, 1.0f), 1e10f);
    p4[0] = x[0] / r; p4[1] = x[1] / r; p4[2] = x[2] / r; p4[3] = 1.0f / r;
}

// acc[t] += W[ob, units u0 .. u0 + NU) . in[t, units iu0 .. iu0 + NU)] (upt units per tile in `in`), three MFMAs per unit
template <int NU>
NCW_DEV void ns_mma(f32x16 (&acc)[NS_TILES], const bf16x8 (&wh)[NU], const bf16x8 (&wl)[NU], const ss_lfrag* in, int upt, int iu0) {
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        const ss_lfrag* p0 = in + ((0 * upt + iu0 + q) * 2) * 64;
        const ss_lfrag* p1 = in + ((1 * upt + iu0 + q) * 2) * 64;
        const bf16x8 bh0 = p0[0], bl0 = p0[64], bh1 = p1[0], bl1 = p1[64];
        acc[0] = NCW_MFMA_H(wh[q], bh0, acc[0], 0, 0, 0);
        acc[1] = NCW_MFMA_H(wh[q], bh1, acc[1], 0, 0, 0);
        acc[0] = NCW_MFMA_H(wl[q], bh0, acc[0], 0, 0, 0);
        acc[1] = NCW_MFMA_H(wl[q], bh1, acc[1], 0, 0, 0);
        acc[0] = NCW_MFMA_H(wh[q], bl0, acc[0], 0, 0, 0);
        acc[1] = NCW_MFMA_H(wh[q], bl1, acc[1], 0, 0, 0);
    }
}
// one tile only
template <int NU>
NCW_DEV void ns_mma1(f32x16& acc, const bf16x8 (&wh)[NU], const bf16x8 (&wl)[NU], const ss_lfrag* in, int t, int upt, int iu0) {
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        const ss_lfrag* p0 = in + ((t * upt + iu0 + q) * 2) * 64;
        const bf16x8 bh = p0[0], bl = p0[64];
        acc = NCW_MFMA_H(wh[q], bh, acc, 0, 0, 0);
        acc = NCW_MFMA_H(wl[q], bh, acc, 0, 0, 0);
        acc = NCW_MFMA_H(wh[q], bl, acc, 0, 0, 0);
    }
}

// v (one output block of tile t, f32) -> hi / lo fragments of k-units 2 ob, 2 ob + 1 of the activation buffer
NCW_DEV void ns_store(ss_lfrag* abuf, int t, int ob, const f32x16& v, int lane) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        bf16x8 hi, lo;
        ss_split8(v, tt, hi, lo);
        abuf[((t * 16 + 2 * ob + tt) * 2 + 0) * 64 + lane] = hi;
        abuf[((t * 16 + 2 * ob + tt) * 2 + 1) * 64 + lane] = lo;
    }
}

__global__ __launch_bounds__(64 * NS_WAVES) void nerf_refineS_kernel(NcwNerfNet net, NcwPoints src, int64_t n, const float* __restrict__ aux_bias,
                                                                    float* __restrict__ density, float* __restrict__ rgb) {
    __shared__ __attribute__((aligned(16))) char lds[NS_ACT + NS_X + NS_TILES * 32 * 4];
    ss_lfrag* const abuf = (ss_lfrag*)(ncw_lchar*)lds;
    ss_lfrag* const xbuf = abuf + NS_ACT / 16;
    typedef __attribute__((address_space(3))) int ns_lint;
    ns_lint* const rbuf = (ns_lint*)(xbuf + NS_X / 16);
    const int lane = ncw_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile0 = (int64_t)blockIdx.x * NS_TILES;
    const int D = net.D;
    n = points_count(src, n);            // the selection's size, read on the device
    if (tile0 * 32 >= n) return;         // (uniform: before any barrier)
    int64_t pp = 0;
    bool pvalid = false;
    if (wave < NS_TILES) {
        int64_t p = (tile0 + wave) * 32 + (lane & 31), ray;
        pvalid = p < n;
        if (!pvalid) p = n - 1;
        pp = point_slot(src, p);         // density / rgb are addressed by the ray sample
        float xs[3], p4[4];
        load_point(src, p, xs, ray);
        ns_inverted_sphere(xs, p4);
        if (lane < 32) rbuf[wave * 32 + lane] = (int)ray;
        CVec<3> gp;
        freq_encode<3, 4, 10, false>(gp, p4, lane);  // sinf / cosf: frequencies up to 2^9
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            bf16x8 hi, lo;
            ss_split8(gp.v[q >> 1], q & 1, hi, lo);
            xbuf[((wave * 6 + q) * 2 + 0) * 64 + lane] = hi;
            xbuf[((wave * 6 + q) * 2 + 1) * 64 + lane] = lo;
        }
    }
    bf16x8 Ah[8], Al[8], Bh[8], Bl[8];
    f32x16 acc[NS_TILES];
    const ss_lfrag* const ain = abuf + lane;
    const ss_lfrag* const xin = xbuf + lane;
    auto relu_store = [&](int ob) {  // ReLU, hi / lo fragments of this wave's block into abuf (in place)
        ncw_lds_barrier();  // every wave has finished reading the layer input
#pragma unroll
        for (int t = 0; t < NS_TILES; ++t) {
            f32x16 y;
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = ncw_relu(acc[t][r]);
            ns_store(abuf, t, ob, y, lane);
        }
        ncw_lds_barrier();  // the layer output is complete
    };
    // ---- trunk layer 0 (K = 84: the 6 units of gamma(p)) ---------------------------------------------------------
    {
        bf16x8 w0h[6], w0l[6];
        ss_load_half<6>(w0h, w0l, net.w_p[0], net.w_p_lo[0], 8, wave, 0, lane);
        if (D > 1) ss_load_half<8>(Ah, Al, net.w_p[1], net.w_p_lo[1], 8, wave, 0, lane);
        const f32x16 bias = ss_bias(net.b_p[0], wave, lane);
        acc[0] = bias; acc[1] = bias;
        ncw_lds_barrier();  // gamma(p) visible
        ns_mma<6>(acc, w0h, w0l, xin, 6, 0);
        relu_store(wave);
    }
    // ---- trunk layers 1 .. D-1 -----------------------------------------------------------------------------------------
    for (int i = 1; i < D; ++i) {
        const f32x16 bias = ss_bias(net.b_p[i], wave, lane);
        acc[0] = bias; acc[1] = bias;
        ss_load_half<8>(Bh, Bl, net.w_p[i], net.w_p_lo[i], 8, wave, 8, lane);  // second half: lands during the first
        ns_mma<8>(acc, Ah, Al, ain, 16, 0);
        {   // first half of the NEXT matrix (trunk layer i + 1, or the feature layer)
            const void* nh = i + 1 < D ? net.w_p[i + 1] : net.w_feat;
            const void* nl = i + 1 < D ? net.w_p_lo[i + 1] : net.w_feat_lo;
            ss_load_half<8>(Ah, Al, nh, nl, 8, wave, 0, lane);
        }
        ns_mma<8>(acc, Bh, Bl, ain, 16, 8);
        if (i == net.skip + 1) {  // the gamma(p) columns: units 16..21 (once per launch)
            bf16x8 wgh[6], wgl[6];
            ss_load_half<6>(wgh, wgl, net.w_p[i], net.w_p_lo[i], 8, wave, 16, lane);
            ns_mma<6>(acc, wgh, wgl, xin, 6, 0);
        }
        relu_store(wave);
    }
    // ---- density (1 output block; tile t by wave t) and the feature layer (Ah / Al = first half of w_feat) ----------------
    if (wave < NS_TILES) {
        CVec<1> o;
        load_bias(o, net.b_alpha, lane);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 wh[8], wl[8];
            ss_load_half<8>(wh, wl, net.w_alpha, net.w_alpha_lo, 1, 0, 8 * half, lane);
            ns_mma1<8>(o.v[0], wh, wl, ain, wave, 16, 8 * half);
        }
        if (pvalid && lane < 32) density[pp] = o.v[0][0];
    }
    {
        const f32x16 bias = ss_bias(net.b_feat, wave, lane);
        acc[0] = bias; acc[1] = bias;
        ss_load_half<8>(Bh, Bl, net.w_feat, net.w_feat_lo, 8, wave, 8, lane);
        ns_mma<8>(acc, Ah, Al, ain, 16, 0);
        ns_mma<8>(acc, Bh, Bl, ain, 16, 8);
        ncw_lds_barrier();  // every wave (and the density row) has read h_D
#pragma unroll
        for (int t = 0; t < NS_TILES; ++t) ns_store(abuf, t, wave, acc[t], lane);  // no activation (nerf.py:171)
        ncw_lds_barrier();
    }
    // ---- appearance head (nerf.py:131-139,173-174): 128 wide = 4 output blocks; wave w: block w & 3 of tile w >> 2 -----
    const int hb = wave & 3, ht = wave >> 2;
    f32x16 hacc;
    {
        hacc = ss_bias(net.b_a[0], hb, lane);
        // the view-direction / appearance-code columns of static_linear_0: per-ray fp32 rows of ncw_aux_ray_bias
        ncw_add_ray_bias_block(hacc, aux_bias + (size_t)rbuf[ht * 32 + (lane & 31)] * 128, hb, lane);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 wh[8], wl[8];
            ss_load_half<8>(wh, wl, net.w_a[0], net.w_a_lo[0], 4, hb, 8 * half, lane);
            ns_mma1<8>(hacc, wh, wl, ain, ht, 16, 8 * half);
        }
        ncw_lds_barrier();
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = ncw_relu(hacc[r]);
        ns_store(abuf, ht, hb, y, lane);  // units 0..7 of tile ht
        ncw_lds_barrier();
    }
    for (int i = 1; i < net.n_head; ++i) {
        hacc = ss_bias(net.b_a[i], hb, lane);
        bf16x8 wh[8], wl[8];
        ss_load_half<8>(wh, wl, net.w_a[i], net.w_a_lo[i], 4, hb, 0, lane);
        ns_mma1<8>(hacc, wh, wl, ain, ht, 16, 0);
        ncw_lds_barrier();
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = ncw_relu(hacc[r]);
        ns_store(abuf, ht, hb, y, lane);
        ncw_lds_barrier();
    }
    // ---- raw rgb (nerf.py:181): tile t by wave t -----------------------------------------------------------------------
    if (wave < NS_TILES) {
        CVec<1> o;
        load_bias(o, net.b_rgb, lane);
        bf16x8 wh[8], wl[8];
        ss_load_half<8>(wh, wl, net.w_rgb, net.w_rgb_lo, 1, 0, 0, lane);
        ns_mma1<8>(o.v[0], wh, wl, ain, wave, 16, 0);
        if (pvalid && lane < 32) {
            rgb[pp * 3 + 0] = o.v[0][0];
            rgb[pp * 3 + 1] = o.v[0][1];
            rgb[pp * 3 + 2] = o.v[0][2];
        }
    }
}

}  // namespace

// Measured per 131,072 points on MI355X (scripts/diag/split_check.py; plain fp16 kernels: 0.160 / 0.558 ms):
//   sdf_infer: pipelined value chain 0.351-0.361 ms, burst 0.399-0.412 ms -> pipelined (burst for nets with more than 8
//              Softplus layers: the bias staging of the pipelined kernel fills the LDS);
//   sdf_fwd:   burst 0.770-0.776 ms, pipelined 0.800-0.839 ms (its stash stores sit in the MFMA stream and share vmcnt with
//              the weight prefetch) -> burst.
int ncw_sdf_inferS_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    const dim3 grid((unsigned)((tiles + SS_TILES - 1) / SS_TILES));
    if (net->n_layers - 1 <= 8) {
        hipLaunchKernelGGL(sdf_inferS2_kernel, grid, dim3(64 * SS_WAVES), 0, st, *net, src, n, sdf);
    } else hipLaunchKernelGGL(sdf_inferS_kernel, grid, dim3(64 * SS_WAVES), 0, st, *net, src, n, sdf);
    NCW_CHECK_LAUNCH();
    return 0;
}

int ncw_sdf_fwdS_launch_f16(const NcwSdfNet* net, const NcwPoints& src, int64_t n, float* sdf, float* grad,
                            const NcwSdfStash& stash, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    const dim3 grid((unsigned)((tiles + SS_TILES - 1) / SS_TILES));
    bool adj = net->n_layers >= 3;  // the adjoint sweep with hi + lo weights: every transposed residual must be there
    for (int l = 0; l < net->n_layers; ++l) adj = adj && net->wt_lo[l] != nullptr;
    const bool render = stash.t[0] == nullptr;  // forward-only render (include/neuconw_hip.h, NcwSdfStash)
    if (adj) {
        if (render) hipLaunchKernelGGL(sdf_fwdSA_kernel<false>, grid, dim3(64 * SS_WAVES), 0, st, *net, src, n, sdf, grad, stash);
        else hipLaunchKernelGGL(sdf_fwdSA_kernel<true>, grid, dim3(64 * SS_WAVES), 0, st, *net, src, n, sdf, grad, stash);
    } else {
        if (render) hipLaunchKernelGGL(sdf_fwdS_kernel<false>, grid, dim3(64 * SS_WAVES), 0, st, *net, src, n, sdf, grad, stash);
        else hipLaunchKernelGGL(sdf_fwdS_kernel<true>, grid, dim3(64 * SS_WAVES), 0, st, *net, src, n, sdf, grad, stash);
    }
    NCW_CHECK_LAUNCH();
    return 0;
}

int ncw_nerf_refineS_launch_f16(const NcwNerfNet* net, const NcwPoints& src, int64_t n, const float* aux_bias, float* density,
                                float* rgb, hipStream_t st) {
    const int64_t tiles = (n + 31) / 32;
    hipLaunchKernelGGL(nerf_refineS_kernel, dim3((unsigned)((tiles + NS_TILES - 1) / NS_TILES)), dim3(64 * NS_WAVES), 0, st, *net, src,
                       n, aux_bias, density, rgb);
    NCW_CHECK_LAUNCH();
    return 0;
}

#endif  // NCW_HALF_F16
