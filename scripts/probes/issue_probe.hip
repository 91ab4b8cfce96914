// Issue-port probe for gfx950: how do MFMA (32x32x16 bf16) and VALU instructions of one wave / of two waves on the
// same SIMD share issue?  Each wave runs ITER iterations of a fixed asm body and reports cycles per iteration
// (s_memtime).  Roles per wave are given on the command line.
//   body M      : 16 MFMAs (two accumulator chains)
//   body V<k>   : 16 x k v_fma_f32 (8 independent chains)
//   body I<k>   : 16 x (MFMA ; k v_fma_f32)         -- one wave interleaving
//   body P<k>   : 16 x (MFMA ; k/2 v_pk_fma_f32)
//   body T<k>   : 16 x (MFMA ; k v_exp_f32)
// build: hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

#define MF0 "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
#define MF1 "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
#define VF(a) "v_fma_f32 %" #a ", %12, %13, %" #a "\n"
#define VP(a) "v_pk_fma_f32 %" #a ", %14, %15, %" #a "\n"
#define VE(a) "v_exp_f32 %" #a ", %" #a "\n"
#define VK(a) "v_fmaak_f32 %" #a ", %" #a ", %12, 0x3f8ccccd\n"
#define VM(a) "v_min_f32_e64 %" #a ", |%" #a "|, %13\n"
#define DR "ds_read_b128 %20, %21\n"
#define DW "s_waitcnt lgkmcnt(3)\n"

// operands: 0,1 acc; 2,3 a,b frags; 4..11 eight scalars; 12,13 consts; 14,15 f32x2 consts; 16..19 four f32x2 accumulators
#define BODY_ARGS : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(k0), "v"(k1), "v"(p0), "v"(p1), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(dr), "v"(laddr)

template <int MODE, int K>
__device__ __forceinline__ void body(f32x16& c0, f32x16& c1, bf16x8 a, bf16x8 b, float& x0, float& x1, float& x2, float& x3,
                                     float& x4, float& x5, float& x6, float& x7, float k0, float k1, f32x2 p0, f32x2 p1,
                                     f32x2& q0, f32x2& q1, f32x2& q2, f32x2& q3, bf16x8& dr, unsigned laddr) {
    // NOTE: the x / q operands are declared as inputs but modified by the asm: fine for a timing probe (values unused)
#define V1 VF(4)
#define V2 VF(4) VF(5)
#define V3 VF(4) VF(5) VF(6)
#define V4 VF(4) VF(5) VF(6) VF(7)
#define V5 V4 VF(8)
#define V6 V4 VF(8) VF(9)
#define V8 V4 VF(8) VF(9) VF(10) VF(11)
#define V10 V8 VF(4) VF(5)
#define V12 V8 V4
#define V16 V8 V8
#define P1 VP(16)
#define P2 VP(16) VP(17)
#define P3 VP(16) VP(17) VP(18)
#define P4 VP(16) VP(17) VP(18) VP(19)
#define E1 VE(4)
#define E2 VE(4) VE(5)
#define E4 VE(4) VE(5) VE(6) VE(7)
    if constexpr (MODE == 0) { asm volatile(REP8(MF0 MF1) BODY_ARGS); }
    else if constexpr (MODE == 1) {  // VALU only: 16 x K
        if constexpr (K == 4) asm volatile(REP16(V4) BODY_ARGS);
        else if constexpr (K == 8) asm volatile(REP16(V8) BODY_ARGS);
        else if constexpr (K == 16) asm volatile(REP16(V16) BODY_ARGS);
    } else if constexpr (MODE == 2) {  // interleave MFMA + K fma
        if constexpr (K == 0) asm volatile(REP8(MF0 MF1) BODY_ARGS);
        else if constexpr (K == 2) asm volatile(REP8(MF0 V2 MF1 V2) BODY_ARGS);
        else if constexpr (K == 4) asm volatile(REP8(MF0 V4 MF1 V4) BODY_ARGS);
        else if constexpr (K == 5) asm volatile(REP8(MF0 V5 MF1 V5) BODY_ARGS);
        else if constexpr (K == 6) asm volatile(REP8(MF0 V6 MF1 V6) BODY_ARGS);
        else if constexpr (K == 8) asm volatile(REP8(MF0 V8 MF1 V8) BODY_ARGS);
        else if constexpr (K == 10) asm volatile(REP8(MF0 V10 MF1 V10) BODY_ARGS);
        else if constexpr (K == 12) asm volatile(REP8(MF0 V12 MF1 V12) BODY_ARGS);
        else if constexpr (K == 16) asm volatile(REP8(MF0 V16 MF1 V16) BODY_ARGS);
    } else if constexpr (MODE == 3) {  // interleave MFMA + K/2 pk_fma
        if constexpr (K == 4) asm volatile(REP8(MF0 P2 MF1 P2) BODY_ARGS);
        else if constexpr (K == 8) asm volatile(REP8(MF0 P4 MF1 P4) BODY_ARGS);
    } else if constexpr (MODE == 4) {  // pk only: 16 x K/2
        if constexpr (K == 8) asm volatile(REP16(P4) BODY_ARGS);
    } else if constexpr (MODE == 5) {  // interleave MFMA + K exp
        if constexpr (K == 1) asm volatile(REP8(MF0 E1 MF1 E1) BODY_ARGS);
        else if constexpr (K == 2) asm volatile(REP8(MF0 E2 MF1 E2) BODY_ARGS);
    } else if constexpr (MODE == 6) {  // exp only: 16 x K
        if constexpr (K == 4) asm volatile(REP16(E4) BODY_ARGS);
    } else if constexpr (MODE == 8) {  // MFMA + 8 literal fmaak
#define K4 VK(4) VK(5) VK(6) VK(7)
#define K8 K4 VK(8) VK(9) VK(10) VK(11)
        if constexpr (K == 8) asm volatile(REP8(MF0 K8 MF1 K8) BODY_ARGS);
        else if constexpr (K == 4) asm volatile(REP8(MF0 K4 MF1 K4) BODY_ARGS);
    } else if constexpr (MODE == 9) {  // MFMA + ds_read_b128 + 8 fma (+ counted wait)
        if constexpr (K == 8) asm volatile(REP8(MF0 DR V8 MF1 DR DW V8) BODY_ARGS);
        else if constexpr (K == 0) asm volatile(REP8(MF0 DR MF1 DR DW) BODY_ARGS);
    } else if constexpr (MODE == 10) {  // the kernel's epilogue mix per MFMA: min, 4 fmaak, max(as fma), add(as fma), + ds_read
        asm volatile(REP8(MF0 DR VM(4) VK(4) VK(4) VK(4) VK(4) VF(5) VF(6) MF1 DR DW VM(7) VK(7) VK(7) VK(7) VK(7) VF(8) VF(9)) BODY_ARGS);
    } else if constexpr (MODE == 7) {  // bursts: 16 MFMA then 16 x K fma (what a non-interleaved kernel does)
        if constexpr (K == 8) asm volatile(REP8(MF0 MF1) REP16(V8) BODY_ARGS);
    }
}

struct Role { int mode, k; };

template <int MODE, int K>
__device__ void run(int iters, unsigned long long* out, int wave, float* sink, unsigned laddr) {
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
    float x0 = 1, x1 = 2, x2 = 3, x3 = 4, x4 = 5, x5 = 6, x6 = 7, x7 = 8, k0 = 0.999f, k1 = 0.001f;
    bf16x8 dr = a;
    f32x2 p0 = {0.999f, 0.999f}, p1 = {0.001f, 0.002f}, q0 = {1, 2}, q1 = {3, 4}, q2 = {5, 6}, q3 = {7, 8};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) body<MODE, K>(c0, c1, a, b, x0, x1, x2, x3, x4, x5, x6, x7, k0, k1, p0, p1, q0, q1, q2, q3, dr, laddr);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    float s = c0[0] + c1[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + q0[0] + q1[1] + q2[0] + q3[1] + (float)dr[0];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

#define CASE(M, K) if (r.mode == M && r.k == K) { run<M, K>(iters, out, wave, sink, laddr); return; }
__global__ __launch_bounds__(512) void probe(const Role* roles, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    const Role r = roles[wave];
    __shared__ char ldsbuf[16384];
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsbuf + threadIdx.x * 16;
    CASE(0, 0) CASE(1, 4) CASE(1, 8) CASE(1, 16)
    CASE(2, 0) CASE(2, 2) CASE(2, 4) CASE(2, 5) CASE(2, 6) CASE(2, 8) CASE(2, 10) CASE(2, 12) CASE(2, 16)
    CASE(3, 4) CASE(3, 8) CASE(4, 8) CASE(5, 1) CASE(5, 2) CASE(6, 4) CASE(7, 8) CASE(8, 8) CASE(8, 4) CASE(9, 8) CASE(9, 0) CASE(10, 0)
    // idle role (mode < 0): still joins the barrier
    __syncthreads();
}

static Role parse(const char* s) {
    Role r{-1, 0};
    if (!strcmp(s, "-")) return r;
    const char c = s[0];
    const int k = atoi(s + 1);
    r.k = k;
    r.mode = c == 'M' ? 0 : c == 'V' ? 1 : c == 'I' ? 2 : c == 'P' ? 3 : c == 'Q' ? 4 : c == 'T' ? 5 : c == 'E' ? 6 : c == 'B' ? 7 : c == 'K' ? 8 : c == 'D' ? 9 : c == 'X' ? 10 : -1;
    return r;
}

int main(int argc, char** argv) {
    // usage: issue_probe <nwaves> role0 role1 ...   (missing roles = idle)
    const int iters = 2000;
    std::vector<std::vector<std::string>> configs;
    if (argc > 1) {
        std::vector<std::string> c;
        for (int i = 1; i < argc; ++i) c.push_back(argv[i]);
        configs.push_back(c);
    } else {
        // waves 0..3 land on SIMDs 0..3, waves 4..7 are their partners
        const char* base[][8] = {
            {"M0", "-", "-", "-", "-", "-", "-", "-"},
            {"V8", "-", "-", "-", "-", "-", "-", "-"},
            {"Q8", "-", "-", "-", "-", "-", "-", "-"},
            {"E4", "-", "-", "-", "-", "-", "-", "-"},
            {"I2", "-", "-", "-", "-", "-", "-", "-"}, {"I4", "-", "-", "-", "-", "-", "-", "-"},
            {"I5", "-", "-", "-", "-", "-", "-", "-"}, {"I6", "-", "-", "-", "-", "-", "-", "-"},
            {"I8", "-", "-", "-", "-", "-", "-", "-"}, {"I10", "-", "-", "-", "-", "-", "-", "-"},
            {"I12", "-", "-", "-", "-", "-", "-", "-"}, {"I16", "-", "-", "-", "-", "-", "-", "-"},
            {"P4", "-", "-", "-", "-", "-", "-", "-"}, {"P8", "-", "-", "-", "-", "-", "-", "-"},
            {"T1", "-", "-", "-", "-", "-", "-", "-"}, {"T2", "-", "-", "-", "-", "-", "-", "-"},
            {"B8", "-", "-", "-", "-", "-", "-", "-"},
            {"M0", "-", "-", "-", "M0", "-", "-", "-"},   // two MFMA waves on one SIMD
            {"M0", "-", "-", "-", "V8", "-", "-", "-"},   // MFMA wave + VALU wave on one SIMD
            {"M0", "-", "-", "-", "V16", "-", "-", "-"},
            {"M0", "V8", "-", "-", "-", "-", "-", "-"},   // MFMA wave + VALU wave on DIFFERENT SIMDs
            {"I4", "-", "-", "-", "I4", "-", "-", "-"},   // two interleaving waves on one SIMD
            {"I8", "-", "-", "-", "I8", "-", "-", "-"},
            {"I10", "-", "-", "-", "I10", "-", "-", "-"},
            {"I16", "-", "-", "-", "I16", "-", "-", "-"},
            {"B8", "-", "-", "-", "B8", "-", "-", "-"},   // two burst waves on one SIMD (the round-1 structure)
            {"I2", "-", "-", "-", "V8", "-", "-", "-"},
            {"I4", "-", "-", "-", "V8", "-", "-", "-"},
            {"I8", "I8", "I8", "I8", "I8", "I8", "I8", "I8"},
        };
        for (auto& b : base) configs.push_back(std::vector<std::string>(b, b + 8));
    }
    Role* d_roles; unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_roles, 8 * sizeof(Role)); hipMalloc(&d_out, 8 * sizeof(unsigned long long)); hipMalloc(&d_sink, 512 * 4);
    for (auto& c : configs) {
        Role h[8];
        for (int i = 0; i < 8; ++i) h[i] = i < (int)c.size() ? parse(c[i].c_str()) : Role{-1, 0};
        hipMemcpy(d_roles, h, sizeof(h), hipMemcpyHostToDevice);
        hipMemset(d_out, 0, 8 * sizeof(unsigned long long));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d_roles, iters, d_out, d_sink);
        hipDeviceSynchronize();
        unsigned long long o[8];
        hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        printf("roles:");
        for (int i = 0; i < 8; ++i) printf(" %-4s", i < (int)c.size() ? c[i].c_str() : "-");
        printf(" | cycles per body (16 MFMA slots):");
        for (int i = 0; i < 8; ++i) if (h[i].mode >= 0) printf(" w%d=%.0f", i, (double)o[i] / iters);
        printf("\n");
    }
    return 0;
}
