#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out) {
    if ((threadIdx.x & 63) == 0) {
        int hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = hw;
    }
}
int main() {
    int *d; int h[64];
    hipMalloc(&d, 4 * 64);
    for (int threads : {1024, 512, 256}) {
        probe<<<2, threads>>>(d); hipDeviceSynchronize(); hipMemcpy(h, d, 4 * 64, hipMemcpyDeviceToHost);
        printf("block of %d threads: wave -> (simd, wave_id, cu): ", threads);
        for (int w = 0; w < 2 * threads / 64; w++) printf("(%d,%d,%d) ", (h[w] >> 4) & 3, h[w] & 15, (h[w] >> 8) & 15);
        printf("\n");
    }
}
