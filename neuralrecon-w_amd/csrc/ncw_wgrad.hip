// Weight-gradient GEMMs:  dense[i][j] += sum_points X[p][i] * Y[p][j]   (+ dbias[i] += sum_p X[p][i])
// for every Linear of every network, batched in ONE launch from a device descriptor table.
//
// X, Y are activation stashes in fragment-native layout [tile32][rb][g][lane][4] (point per lane).
// The reduction index of these GEMMs is the POINT, so both operands must be transposed to
// "feature per lane, points along K"; that happens once per chunk through LDS (XT[f][p]), after
// which A and B fragments are plain 16-byte (bf16) / 4-byte (f32) LDS reads, conflict-free by row
// padding.  A workgroup owns a 128x128 output quadrant (4 waves x 2x2 blocks of 32x32) for one
// K-slice of the points; K-slices combine with f32 atomics (16 slices x 256 KB per product, tiny).
//
// Replaces the autograd-generated `mm` / `addmm` backward GEMMs of every F.linear on the path
// (models/neuconw.py:269-278,130-170; models/nerf.py:156-182) including the second-order products
// t_l qbar_l^T of SURVEY 8a-2.
#include "ncw_mlp.h"

namespace NCW_NS {

template <class P> struct WgT;
template <> struct WgT<PrecBF16> {
    static constexpr int CH = 64;       // points per chunk
    static constexpr int LDT = 64 + 8;  // row stride in elements (144 B: 16-B aligned, conflict-free b128)
};
template <> struct WgT<PrecF32> {
    static constexpr int CH = 32;
    static constexpr int LDT = 32 + 1;
};

__device__ __forceinline__ int wg_find(const int32_t* __restrict__ prefix, int n, int v) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// stage `nb` feature blocks (starting at block b0 of an rb-block stash) of the chunk's tiles into T[f][p]
template <class P>
NCW_DEV void stage_transposed(typename P::selem* T, const typename P::selem* __restrict__ src, int rb, int b0, int nb,
                              int64_t tile0, int ntile, int tid) {
    typedef typename P::selem SE;
    constexpr int LDT = WgT<P>::LDT;
    const int items = ntile * nb * 4;  // (tile, block, g) triples, one wave-load each
    const int wave = tid >> 6, lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    for (int it = wave; it < items; it += 4) {
        const int g = it & 3;
        const int b = (it >> 2) % nb;
        const int tp = (it >> 2) / nb;
        const SE* s = src + ((((size_t)(tile0 + tp) * rb + (b0 + b)) * 4 + g) * 64 + lane) * 4;
        SE v[4];
        if (sizeof(SE) == 4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(s);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (SE)t[c];
        } else {
            bf16x4 t = *reinterpret_cast<const bf16x4*>(s);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (SE)t[c];
        }
        const int f = b * 32 + 8 * g + 4 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) T[(f + c) * LDT + tp * 32 + p] = v[c];
    }
}

// ordered (run-to-run reproducible) split-K: slab of one workgroup = 128 x 128 partial + 128 bias partials
constexpr int WG_SLAB = 128 * 128 + 128;

template <class P>
__global__ __launch_bounds__(256) void wgrad_kernel(const NcwWgradDesc* __restrict__ descs,
                                                    const int32_t* __restrict__ prefix, int n_desc, int ksplit,
                                                    int64_t ntiles, float* __restrict__ partials) {
    typedef typename P::selem SE;
    constexpr int CH = WgT<P>::CH, LDT = WgT<P>::LDT;
    __shared__ __attribute__((aligned(16))) SE XT[128 * LDT];
    __shared__ __attribute__((aligned(16))) SE YT[128 * LDT];
    const int d = wg_find(prefix, n_desc, blockIdx.x);
    const NcwWgradDesc D = descs[d];
    const int local = blockIdx.x - prefix[d];
    const int quad = local / ksplit, ks = local - quad * ksplit;
    const int nqj = (D.rby + 3) >> 2;
    const int qi = quad / nqj, qj = quad - qi * nqj;
    const int nbi = min(4, D.rbx - 4 * qi), nbj = min(4, D.rby - 4 * qj);
    // a product sized on the device (dead-background elimination, NcwPoints mode 4): its K-slices divide the tiles that
    // exist; slices past the end contribute zero slabs (ordered mode) / nothing (atomics)
    if (D.n_points_dev != nullptr) ntiles = min(ntiles, ((int64_t)D.n_points_dev[0] + 31) / 32);
    const int64_t tpk = (ntiles + ksplit - 1) / ksplit;
    const int64_t t_begin = (int64_t)ks * tpk, t_end = min(t_begin + tpk, ntiles);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (D.dbias != nullptr) && (qj == 0) && (tid < 32 * nbi);
    const bool act_i0 = 2 * wi < nbi, act_i1 = 2 * wi + 1 < nbi, act_j0 = 2 * wj < nbj, act_j1 = 2 * wj + 1 < nbj;
    const int fi = lane & 31, kh = lane >> 5;
    for (int64_t t0 = t_begin; t0 < t_end; t0 += CH / 32) {
        const int ntile = (int)min((int64_t)(CH / 32), t_end - t0);
        __syncthreads();
        stage_transposed<P>(XT, (const SE*)D.x, D.rbx, 4 * qi, nbi, t0, ntile, tid);
        stage_transposed<P>(YT, (const SE*)D.y, D.rby, 4 * qj, nbj, t0, ntile, tid);
        __syncthreads();
        const int npts = ntile * 32;
        if (do_bias) {
            float s = 0.f;
            for (int q = 0; q < npts; ++q) s += (float)XT[tid * LDT + q];
            bsum += s;
        }
        if (act_i0 && act_j0) {
            if (P::id == NCW_PREC_BF16) {
                for (int kk = 0; kk < npts / 16; ++kk) {
                    const int ko = kk * 16 + 8 * kh;
                    bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&XT[((2 * wi) * 32 + fi) * LDT + ko]);
                    bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&YT[((2 * wj) * 32 + fi) * LDT + ko]);
                    bf16x8 a1 = a0, b1 = b0;
                    if (act_i1) a1 = *reinterpret_cast<const bf16x8*>(&XT[((2 * wi + 1) * 32 + fi) * LDT + ko]);
                    if (act_j1) b1 = *reinterpret_cast<const bf16x8*>(&YT[((2 * wj + 1) * 32 + fi) * LDT + ko]);
                    acc[0][0] = NCW_MFMA_H(a0, b0, acc[0][0], 0, 0, 0);
                    if (act_j1) acc[0][1] = NCW_MFMA_H(a0, b1, acc[0][1], 0, 0, 0);
                    if (act_i1) acc[1][0] = NCW_MFMA_H(a1, b0, acc[1][0], 0, 0, 0);
                    if (act_i1 && act_j1) acc[1][1] = NCW_MFMA_H(a1, b1, acc[1][1], 0, 0, 0);
                }
            } else {
                for (int kk = 0; kk < npts / 2; ++kk) {
                    const int ko = kk * 2 + kh;
                    float a0 = (float)XT[((2 * wi) * 32 + fi) * LDT + ko];
                    float b0 = (float)YT[((2 * wj) * 32 + fi) * LDT + ko];
                    float a1 = a0, b1 = b0;
                    if (act_i1) a1 = (float)XT[((2 * wi + 1) * 32 + fi) * LDT + ko];
                    if (act_j1) b1 = (float)YT[((2 * wj + 1) * 32 + fi) * LDT + ko];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    if (act_j1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    if (act_i1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    if (act_i1 && act_j1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
        }
    }
    // ---- ordered mode: this K-slice's 128 x 128 partial (+ 128 bias sums) goes to its own slab; the slices are
    // combined in slice order by wgrad_reduce_kernel, so the result does not depend on the arrival order ----------
    if (partials != nullptr) {
        float* slab = partials + (size_t)blockIdx.x * WG_SLAB;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ib = 2 * wi + a, jb = 2 * wj + b;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(ib * 32 + ncw_feat_of(r, lane >> 5)) * 128 + jb * 32 + (lane & 31)] = acc[a][b][r];
            }
        if (tid < 128) slab[128 * 128 + tid] = do_bias ? bsum : 0.f;
        return;
    }
    // ---- epilogue: f32 atomics into the dense gradient (row = X feature, col = Y feature) ------------
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ib = 2 * wi + a, jb = 2 * wj + b;
            if (ib >= nbi || jb >= nbj) continue;
            const int col = (4 * qj + jb) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (4 * qi + ib) * 32 + ncw_feat_of(r, lane >> 5);
                atomicAdd(&D.dense[(size_t)row * D.ld + col], acc[a][b][r]);
            }
        }
    if (do_bias) atomicAdd(&D.dbias[4 * qi * 32 + tid], bsum);
}

// One workgroup per (product, 128 x 128 quadrant): sums the quadrant's K-slice slabs IN SLICE ORDER and adds the
// total to the dense gradient.  A dense element receives at most two such totals per step (the first-order product
// z-bar_l (x) h_{l-1} and the second-order product t_l (x) q-bar_{l-1} of the same Linear) on top of the zero fill, and
// 0 + a + b == 0 + b + a exactly, so the f32 atomics here cannot introduce an order dependence.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const NcwWgradDesc* __restrict__ descs,
                                                           const int32_t* __restrict__ prefix, int n_desc, int ksplit,
                                                           const float* __restrict__ partials) {
    const int d = wg_find(prefix, n_desc, blockIdx.x);
    const int local = blockIdx.x - prefix[d];
    const int quad = local / ksplit, ks = local - quad * ksplit;
    if (ks != 0) return;
    const NcwWgradDesc D = descs[d];
    const int nqj = (D.rby + 3) >> 2;
    const int qi = quad / nqj, qj = quad - qi * nqj;
    const int nbi = min(4, D.rbx - 4 * qi), nbj = min(4, D.rby - 4 * qj);
    const float* slab0 = partials + (size_t)blockIdx.x * WG_SLAB;
    for (int e = threadIdx.x; e < 128 * 128; e += 256) {
        const int row = e >> 7, col = e & 127;
        if (row >= 32 * nbi || col >= 32 * nbj) continue;
        float s = 0.f;
        for (int k = 0; k < ksplit; ++k) s += slab0[(size_t)k * WG_SLAB + e];
        atomicAdd(&D.dense[(size_t)(4 * qi * 32 + row) * D.ld + 4 * qj * 32 + col], s);
    }
    if (D.dbias != nullptr && qj == 0 && threadIdx.x < 32 * nbi) {
        float s = 0.f;
        for (int k = 0; k < ksplit; ++k) s += slab0[(size_t)k * WG_SLAB + 128 * 128 + threadIdx.x];
        atomicAdd(&D.dbias[4 * qi * 32 + threadIdx.x], s);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 fast path: no scatter-transpose.  The stash bytes of a tile are copied VERBATIM into LDS and the MFMA
// operands "8 points of one feature" are produced by the hardware transpose read ds_read_b64_tr_b16:
//   within a 16-lane group, out[lane i][j] = src[lane 4j + i/4][i % 4]      (probed on gfx950,
//   scripts/probes/tr16_probe.hip), so with source lane s pointing at (point p0 + s/4, features
//   f0 + 4 (s%4) ..+3) lane i receives feature f0 + i of points p0..p0+3.
// The transpose read goes through the compiler-tracked builtin (the compiler counts lgkmcnt and schedules
// the reads against the MFMAs; an inline-asm version raced: hipcc copied the asm's destination registers
// before the data had landed).
// ------------------------------------------------------------------------------------------------
typedef short ncw_s16x4 __attribute__((ext_vector_type(4)));
typedef short ncw_s16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// Staging: the stash bytes go global -> LDS with global_load_lds_dwordx4 (no
// staging registers, no ds_write pass), FOUR one-tile buffers rotate so that three tiles (96 KiB per CU)
// are always in flight -- HBM latency is hidden by depth, not by occupancy.  The DMA writes lane-linear
// 1 KiB pieces, so the bank-conflict fix is a SOURCE-side swizzle: inside each 256-byte row r (32 points
// x 8 B) the 16-byte slots are rotated by (r & 3) * 64 bytes, which puts the four rows a transpose-read
// group touches on disjoint banks.  Only DMA ops use vmcnt inside the loop, so the waits are counted
// (s_waitcnt vmcnt(16): two younger tiles stay in flight) around a raw s_barrier.
// ------------------------------------------------------------------------------------------------
#ifndef NCW_WGRAD_NBUF
#define NCW_WGRAD_NBUF 4
#endif
template <int XB, int YB, int NBUF>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const NcwWgradDesc* __restrict__ descs,
                                                        const int32_t* __restrict__ prefix, int n_desc, int ksplit,
                                                        int64_t ntiles) {
    constexpr int WI = XB / 2, WJ = YB / 2;
    constexpr int NBLK = XB + YB;
    constexpr int BUF = NBLK * 2048;          // one tile of all blocks
    constexpr int PPW = NBLK * 2 / 4;         // 1 KiB pieces per wave per tile
    __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF];
    const int d = wg_find(prefix, n_desc, blockIdx.x);
    const NcwWgradDesc D = descs[d];
    const int local = blockIdx.x - prefix[d];
    if (D.ksplit > 0) ksplit = D.ksplit;                    // per-product split (load balancing)
    if (D.n_points > 0) ntiles = (D.n_points + 31) / 32;    // per-product point count (merged launches)
    if (D.n_points_dev != nullptr) {                        // a selection sized on the device (NcwPoints mode 4):
        const int64_t nt = ((int64_t)D.n_points_dev[0] + 31) / 32;  // the K-slices re-divide the tiles that exist
        ntiles = nt < ntiles ? nt : ntiles;
    }
    const int quad = local / ksplit, ks = local - quad * ksplit;
    const int nqj = (D.rby + YB - 1) / YB;
    const int qi = quad / nqj, qj = quad - qi * nqj;
    const int nbi = min(XB, D.rbx - XB * qi), nbj = min(YB, D.rby - YB * qj);
    const int64_t tpk = (ntiles + ksplit - 1) / ksplit;
    const int64_t t_begin = (int64_t)ks * tpk, t_end = min(t_begin + tpk, ntiles);
    if (t_begin >= t_end) return;  // uniform: whole workgroup
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const int wi = uw >> 1, wj = uw & 1;
    f32x16 acc[WI][WJ];
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
        for (int b = 0; b < WJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[WI];
#pragma unroll
    for (int a = 0; a < WI; ++a) bsum[a] = 0.f;
    const bool do_bias = (D.dbias != nullptr) && (qj == 0) && (wj == 0);
    const char* xg = (const char*)D.x;
    const char* yg = (const char*)D.y;
    // ---- DMA issue: piece pc (0 .. 2*NBLK) = block pc/2, rows 4*(pc&1) .. +3; lane L fills slot L ------------
    const int prow = lane >> 4, pslot = lane & 15;  // row within the piece, 16-byte slot within the row
    auto issue_tile = [&](int64_t tile_raw, int bufi) {
        const int64_t tile = min(tile_raw, t_end - 1);  // past the end: harmless re-load (keeps vmcnt counts uniform)
        ncw_lchar* lbuf = (ncw_lchar*)lds + bufi * BUF;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int pc = uw + 4 * k;
            const int blk = pc >> 1, r0 = 4 * (pc & 1);
            const bool isx = blk < XB;
            const int bsel = isx ? min(blk, nbi - 1) : min(blk - XB, nbj - 1);
            const char* base = isx ? xg : yg;
            const int64_t rb = isx ? D.rbx : D.rby;
            const int64_t b0 = isx ? XB * qi : YB * qj;
            const char* ub = base + (tile * rb + b0 + bsel) * 2048 + r0 * 256;   // scalar
            const int r = r0 + prow;                                                // row inside the block
            const unsigned src = (unsigned)(prow * 256 + ((pslot * 16 - (r & 3) * 64) & 255));
            // issued as asm so the compiler does not see an LDS write it would fence with vmcnt(0) before every
            // LDS read; the counted waits below are the synchronisation
            const unsigned long long ubi = (unsigned long long)ub;
            const unsigned ulo = __builtin_amdgcn_readfirstlane((unsigned)ubi);
            const unsigned uhi = __builtin_amdgcn_readfirstlane((unsigned)(ubi >> 32));
            const unsigned long long ubu = ((unsigned long long)uhi << 32) | ulo;
            const unsigned loff = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lbuf + blk * 2048 + r0 * 256));
            // `nt`: every stash operand is read exactly once per product (ncw_common.h "Cache policy of the stash traffic")
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                         :: "v"(src), "s"(ubu), "s"(loff) : "memory");
        }
    };
    // ---- transpose-read addressing (see wgrad_bf16_kernel): source lane s of 16-lane group q ---------------
    const int q = lane >> 4, s = lane & 15;
    const int kh = q >> 1, fhalf = q & 1;
    const int fsrc = 16 * fhalf + 4 * (s & 3);
    const int rr = 2 * (fsrc >> 3) + ((fsrc >> 2) & 1);  // row inside the block
    const int psrc = 8 * kh + (s >> 2);
    unsigned toff[2][2];  // [k-step within the tile][4-point half]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            toff[kk][hf] = (unsigned)(rr * 256 + (((psrc + 16 * kk + 4 * hf) * 8 + (rr & 3) * 64) & 255));
    auto frag = [&](const ncw_lchar* blockp, int kk) -> bf16x8 {
        typedef __attribute__((address_space(3))) ncw_s16x4* lp;
        const ncw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(blockp + toff[kk][0]));
        const ncw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(blockp + toff[kk][1]));
        const ncw_s16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, w);
    };
    // ---- pipeline -------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) issue_tile(t_begin + i, i);
    int bufi = 0;
    for (int64_t t = t_begin; t < t_end; ++t) {
        // tiles t .. t+NBUF-2 are in flight (PPW DMA ops each, issued in that order): retire tile t only
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        // refill the buffer read in the previous iteration (everyone passed the barrier)
        issue_tile(t + NBUF - 1, bufi == 0 ? NBUF - 1 : bufi - 1);
        const ncw_lchar* bufp = (const ncw_lchar*)lds + bufi * BUF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[WI], bfr[WJ];
#pragma unroll
            for (int a = 0; a < WI; ++a) af[a] = frag(bufp + (WI * wi + a) * 2048, kk);
#pragma unroll
            for (int b = 0; b < WJ; ++b) bfr[b] = frag(bufp + (XB + WJ * wj + b) * 2048, kk);
#pragma unroll
            for (int a = 0; a < WI; ++a) {
                float sacc = 0.f;
#ifdef NCW_HALF_F16
                typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
                const h16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const h16x2 pr = {af[a][2 * e], af[a][2 * e + 1]};
                    sacc = __builtin_amdgcn_fdot2(pr, ones, sacc, false);
                }
#else
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 wv = __builtin_bit_cast(u32x4, af[a]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    sacc += __builtin_bit_cast(float, wv[e] << 16) + __builtin_bit_cast(float, wv[e] & 0xffff0000u);
#endif
                bsum[a] += sacc;
#pragma unroll
                for (int b = 0; b < WJ; ++b)
                    acc[a][b] = NCW_MFMA_H(af[a], bfr[b], acc[a][b], 0, 0, 0);
            }
        }
        bufi = bufi + 1 == NBUF ? 0 : bufi + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy tail loads before LDS goes away
    // ---- epilogue ---------------------------------------------------------------------------------------
    typedef __attribute__((address_space(1))) float* gfp;
#pragma unroll
    for (int a = 0; a < WI; ++a) {
        const int ib = WI * wi + a;
        if (ib >= nbi) continue;
#pragma unroll
        for (int b = 0; b < WJ; ++b) {
            const int jb = WJ * wj + b;
            if (jb >= nbj) continue;
            const int col = (YB * qj + jb) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (XB * qi + ib) * 32 + ncw_feat_of(r, lane >> 5);
#if defined(NCW_PROBE_BUILD) && defined(NCW_EXP_WGRAD_STORE)
                // TIMING PROBE ONLY (wrong results): what the split-K flush would cost as plain stores instead of f32 atomics
                __builtin_nontemporal_store(acc[a][b][r], (float*)&D.dense[(size_t)row * D.ld + col]);
#else
                atomicAdd((float*)&D.dense[(size_t)row * D.ld + col], acc[a][b][r]);
#endif
            }
        }
        if (do_bias) {
            const float tot = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (lane < 32) atomicAdd(&D.dbias[(XB * qi + ib) * 32 + lane], tot);
        }
    }
}

}  // namespace NCW_NS
using namespace NCW_NS;

// tile: 0 = 128 x 256 (X x Y features per workgroup), 1 = 256 x 256 (each stash element is read once per
// product; for products with more than 4 X blocks).  bf16 only; the f32 kernel ignores it.
extern "C" int NCW_FN(ncw_wgrad_tiled)(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
                               int tile, int64_t n_points, void* stream) {
    if (n_desc <= 0 || total_wgs <= 0 || n_points <= 0) return 0;
    if (ksplit < 1 || (tile != 0 && tile != 1)) return NCW_E_BADARG;
    const int64_t ntiles = (n_points + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (tile == 0)
        hipLaunchKernelGGL((wgrad_dma_kernel<4, 8, 4>), dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, ntiles);
    else
        hipLaunchKernelGGL((wgrad_dma_kernel<8, 8, NCW_WGRAD_NBUF>), dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, ntiles);
    NCW_CHECK_LAUNCH();
    return 0;
}

#ifndef NCW_HALF_F16
extern "C" int ncw_wgrad_f16(const NcwWgradDesc*, const int32_t*, int, int, int, int, int64_t, void*);
extern "C" int ncw_wgrad_ordered_f16(const NcwWgradDesc*, const int32_t*, int, int, int, int, int64_t, float*, void*);
#endif

extern "C" int NCW_FN(ncw_wgrad)(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
                                 int prec, int64_t n_points, void* stream) {
    NCW_FORWARD_F16(prec, ncw_wgrad_f16(descs, wg_prefix, n_desc, total_wgs, ksplit, NCW_PREC_BF16, n_points, stream));
    if (n_desc <= 0 || total_wgs <= 0 || n_points <= 0) return 0;
    if (ksplit < 1) return NCW_E_BADARG;
    const int64_t ntiles = (n_points + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (prec == NCW_PREC_BF16)
        hipLaunchKernelGGL(wgrad_kernel<PrecBF16>, dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, ntiles,
                           (float*)nullptr);
    else if (prec == NCW_PREC_F32)
        hipLaunchKernelGGL(wgrad_kernel<PrecF32>, dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, ntiles,
                           (float*)nullptr);
    else return NCW_E_BADARG;
    NCW_CHECK_LAUNCH();
    return 0;
}

#ifndef NCW_HALF_F16
extern "C" int64_t ncw_wgrad_ordered_scratch_floats(int total_wgs) { return (int64_t)total_wgs * WG_SLAB; }
#endif

extern "C" int NCW_FN(ncw_wgrad_ordered)(const NcwWgradDesc* descs, const int32_t* wg_prefix, int n_desc, int total_wgs, int ksplit,
                                         int prec, int64_t n_points, float* partials, void* stream) {
    NCW_FORWARD_F16(prec, ncw_wgrad_ordered_f16(descs, wg_prefix, n_desc, total_wgs, ksplit, NCW_PREC_BF16, n_points, partials, stream));
    if (n_desc <= 0 || total_wgs <= 0 || n_points <= 0) return 0;
    if (ksplit < 1 || partials == nullptr) return NCW_E_BADARG;
    const int64_t ntiles = (n_points + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (prec == NCW_PREC_BF16)
        hipLaunchKernelGGL(wgrad_kernel<PrecBF16>, dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, ntiles,
                           partials);
    else if (prec == NCW_PREC_F32)
        hipLaunchKernelGGL(wgrad_kernel<PrecF32>, dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, ntiles,
                           partials);
    else return NCW_E_BADARG;
    NCW_CHECK_LAUNCH();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(total_wgs), dim3(256), 0, st, descs, wg_prefix, n_desc, ksplit, partials);
    NCW_CHECK_LAUNCH();
    return 0;
}
