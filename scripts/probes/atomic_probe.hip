// How fast are f32 atomics into a small set of hot matrices on gfx950?  The question behind "compute the weight gradients where
// the cotangents live": every workgroup would flush a 256 x 256 f32 partial (256 KB) per layer per 128 points into one of ~25
// dense matrices (6.4 MB in total).  Round 2 measured ONE row (agent-scope atomics into arenas shared by all 8 XCDs: 1.32 TB/s).
// Round 5 adds what that row left open:
//   * XCD-PRIVATE arenas: 8 copies of the matrices, a workgroup flushes into the copy of ITS XCD (hardware XCC_ID, or
//     blockIdx % 8), so an address is only ever touched from one L2 -- then WORKGROUP-scope atomics (executed in that L2, no
//     cross-XCD coherence traffic) are sufficient; the 8 copies are summed by the consumer (8 x 6.4 MB of reads);
//   * layer-major order: all workgroups flush into the SAME matrix at a time (256 KB per arena hot in L2) instead of 25;
//   * packed bf16 atomics (global_atomic_pk_add_bf16: half the bytes per element -- accuracy aside);
//   * two atomics in flight per lane.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_probe.hip -o atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef short bf16x2_t __attribute__((ext_vector_type(2)));  // two bf16 bit patterns (the builtin's operand type)

__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

// arena: 0 = one shared set of matrices; 1 = private per blockIdx % 8; 2 = private per hardware XCC_ID
// op: 0 agent-scope f32 atomic, 1 plain store, 2 workgroup-scope f32 atomic, 3 two atomics in flight, 4 pk_add_bf16 (agent),
//     5 pk_add_bf16 (workgroup scope)
template <int OP>
__global__ __launch_bounds__(256) void flush_kernel(float* dense, int n_mats, int reps, int arena, int hot) {
    const int tid = threadIdx.x;
    const int a = arena == 0 ? 0 : (arena == 1 ? (int)(blockIdx.x & 7) : xcc_id());
    float* base = dense + (size_t)a * n_mats * 65536;
    for (int r = 0; r < reps; ++r) {
        // hot = 1: layer-major -- every workgroup is on matrix r at the same time; 0: spread over the n_mats matrices
        float* m = base + (size_t)(hot ? (r % n_mats) : ((blockIdx.x + r) % n_mats)) * 65536;
        if (OP == 4 || OP == 5) {
            bf16x2_t* mb = reinterpret_cast<bf16x2_t*>(m);  // 65536 bf16 pairs would be 2 matrices: flush 32768 pairs = 64 Ki elements
            for (int i = tid; i < 32768; i += 256) {
                bf16x2_t v;
                v[0] = (short)0x3A83; v[1] = (short)0x3B03;  // bf16(1e-3), bf16(2e-3)
                if (OP == 4) __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16x2_t*)(mb + i), v);
                else asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(mb + i), "v"(v) : "memory");  // no scope bits: CU / workgroup scope
            }
        } else if (OP == 3) {
            for (int i = tid; i < 32768; i += 256) {
                atomicAdd(m + i, 1e-3f);
                atomicAdd(m + 32768 + i, 2e-3f);
            }
        } else {
            for (int i = tid; i < 65536; i += 256) {
                const float v = 1e-3f * (float)(i & 7);
                if (OP == 0) atomicAdd(m + i, v);
                else if (OP == 1) m[i] = v;
                else __hip_atomic_fetch_add(m + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

__global__ void xcc_hist_kernel(int* hist, int* mismatch) {
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        atomicAdd(hist + x, 1);
        if (x != (int)(blockIdx.x & 7)) atomicAdd(mismatch, 1);
    }
}

template <int OP>
static float run(float* d, int n_mats, int wgs, int reps, int arena, int hot) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(flush_kernel<OP>, dim3(wgs), dim3(256), 0, 0, d, n_mats, reps, arena, hot);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const int n_mats = 25, wgs = 1024, reps = 8;
    float* d;
    hipMalloc(&d, (size_t)8 * n_mats * 65536 * 4);
    hipMemset(d, 0, (size_t)8 * n_mats * 65536 * 4);
    int* h;
    hipMalloc(&h, 17 * sizeof(int));
    hipMemset(h, 0, 17 * sizeof(int));
    hipLaunchKernelGGL(xcc_hist_kernel, dim3(wgs), dim3(64), 0, 0, h, h + 16);
    int hh[17];
    hipMemcpy(hh, h, sizeof(hh), hipMemcpyDeviceToHost);
    printf("XCC_ID histogram of %d workgroups:", wgs);
    for (int i = 0; i < 8; ++i) printf(" %d", hh[i]);
    printf("   (workgroups whose XCC_ID != blockIdx %% 8: %d)\n", hh[16]);
    const double bytes = (double)wgs * reps * 65536 * 4;  // f32-equivalent bytes of partial tiles flushed
    const char* ops[] = {"agent-scope f32 atomicAdd", "plain stores (reference)", "workgroup-scope f32 atomic", "2 atomics in flight / lane",
                         "pk_add_bf16 (builtin)", "pk_add_bf16 (asm, no sc bits)"};
    const char* arenas[] = {"shared arena", "private: blockIdx % 8", "private: XCC_ID"};
    printf("%-30s %-24s %-12s %9s %9s\n", "operation", "arena", "order", "ms", "TB/s(f32-equivalent)");
    for (int hot = 0; hot < 2; ++hot)
        for (int arena = 0; arena < 3; ++arena)
            for (int op = 0; op < 6; ++op) {
                float ms = 0.f;
                switch (op) {
                    case 0: ms = run<0>(d, n_mats, wgs, reps, arena, hot); break;
                    case 1: ms = run<1>(d, n_mats, wgs, reps, arena, hot); break;
                    case 2: ms = run<2>(d, n_mats, wgs, reps, arena, hot); break;
                    case 3: ms = run<3>(d, n_mats, wgs, reps, arena, hot); break;
                    case 4: ms = run<4>(d, n_mats, wgs, reps, arena, hot); break;
                    default: ms = run<5>(d, n_mats, wgs, reps, arena, hot); break;
                }
                printf("%-30s %-24s %-12s %9.3f %9.2f\n", ops[op], arenas[arena], hot ? "layer-major" : "25 matrices", ms,
                       bytes / ms / 1e9);
            }
    return 0;
}
