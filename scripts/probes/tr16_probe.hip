// Probe of ds_read_b64_tr_b16 lane/element mapping on gfx950 (prints out[lane][j] given LDS[e] = e).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned short u16;
__global__ void probe(unsigned long long* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) u16 lds[4096];
    int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (u16)i;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)lds + lane * stride_bytes;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane] = v;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    for (int stride : {8, 32, 64}) {
        probe<<<1, 64>>>(d, stride);
        unsigned long long h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes per lane (lane l reads its 4 b16 at element l*%d):\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %4llu", (h[l] >> (16 * j)) & 0xffff);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
