// Dev microbenchmark: dependent-issue latency of the fp64 instructions the kernels are made of (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define N 2048
template <int OP> __global__ void chain(double *out, double a, double b, unsigned long long *cyc)
{
    double x = a + threadIdx.x * 1e-9, y = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        if (OP == 0) x = fma(x, y, y);
        else if (OP == 1) x = x + y;
        else if (OP == 2) x = x * y;
        else if (OP == 3) x = __builtin_amdgcn_rcp(x);
        else if (OP == 4) x = sqrt(x + 2.0);
        else if (OP == 5) x = x / y;
        else if (OP == 6) x = atan2(x, y);
        else if (OP == 7) x = rint(x * 1e4) / 1e4 + 0.1;
        else if (OP == 8) { float f = (float)x; f = fmaf(f, 1.0001f, 0.5f); x = f; }
        else if (OP == 9) x = sin(x);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> __global__ void chain2(double *out, double a, double b, unsigned long long *cyc)
{   // two independent chains in one wave (ILP 2)
    double x = a + threadIdx.x * 1e-9, y = b, z = a * 1.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; i++) { x = fma(x, y, y); z = fma(z, y, y); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + z;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h[4096];
    hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cyc, 8 * 4096);
    const char *names[] = {"fma_f64", "add_f64", "mul_f64", "rcp_f64", "sqrt_f64(+add)", "div_f64", "atan2_f64", "round4_np(+add)", "f64->f32 fma ->f64", "sin_f64"};
#define RUN(OP, blocks) { chain<OP><<<blocks, 64>>>(out, 0.7, 0.9999, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost); \
    double s = 0; for (int i = 0; i < blocks; i++) s += h[i]; printf("%-22s blocks=%5d  %.1f cycles/op\n", names[OP], blocks, s / blocks / N); }
    for (int blocks : {1, 1024, 2048, 4096}) {
        RUN(0, blocks) RUN(1, blocks) RUN(2, blocks) RUN(3, blocks) RUN(4, blocks) RUN(5, blocks) RUN(6, blocks) RUN(7, blocks) RUN(8, blocks) RUN(9, blocks)
        chain2<0><<<blocks, 64>>>(out, 0.7, 0.9999, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; i++) s += h[i]; printf("%-22s blocks=%5d  %.1f cycles/iter (2 fma)\n", "fma_f64 x2 ILP", blocks, s / blocks / N);
    }
    return 0;
}
