// How fast are device-scope f32 atomics into a small set of hot matrices on gfx950?  The question behind "compute the
// weight gradients where the cotangents live": every workgroup would flush a 256 x 256 f32 partial (256 KB) per layer
// per 128 points into one of ~25 dense matrices (6.4 MB in total, shared by all 256 CUs / 8 XCDs).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_probe.hip -o atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void flush_kernel(float* dense, int n_mats, int reps, int mode) {
    // workgroup b adds a 256 KB tile to matrix (b + rep) % n_mats, `reps` times
    const int tid = threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        float* m = dense + (size_t)((blockIdx.x + r) % n_mats) * 65536;
        for (int i = tid; i < 65536; i += 256) {
            const float v = 1e-3f * (float)(i & 7);
            if (mode == 0) atomicAdd(m + i, v);                       // f32 atomic add (no return)
            else if (mode == 1) m[i] = v;                             // plain store (same bytes, for reference)
            else __hip_atomic_fetch_add(m + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__global__ __launch_bounds__(256) void flush4_kernel(float* dense, int n_mats, int reps) {
    // the same with packed atomics where available: 2 x f32 per instruction (global_atomic_pk_add_f32 does not exist;
    // this variant issues two independent atomics per lane per iteration for ILP)
    const int tid = threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        float* m = dense + (size_t)((blockIdx.x + r) % n_mats) * 65536;
        for (int i = tid; i < 32768; i += 256) {
            atomicAdd(m + i, 1e-3f);
            atomicAdd(m + 32768 + i, 2e-3f);
        }
    }
}

int main() {
    const int n_mats = 25, wgs = 1024, reps = 8;
    float* d;
    hipMalloc(&d, (size_t)n_mats * 65536 * 4);
    hipMemset(d, 0, (size_t)n_mats * 65536 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)wgs * reps * 65536 * 4;
    for (int mode = 0; mode < 4; ++mode) {
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(e0);
            if (mode < 3) hipLaunchKernelGGL(flush_kernel, dim3(wgs), dim3(256), 0, 0, d, n_mats, reps, mode);
            else hipLaunchKernelGGL(flush4_kernel, dim3(wgs), dim3(256), 0, 0, d, n_mats, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const char* names[] = {"agent-scope f32 atomicAdd", "plain stores", "workgroup-scope f32 atomic", "2 atomics in flight per lane"};
        printf("%-32s %8.3f ms for %.1f GB of partial-tile flushes -> %.2f TB/s\n", names[mode], ms, bytes / 1e9, bytes / ms / 1e9);
    }
    return 0;
}
