// Dev microbenchmark: dependent-access latency of the ways a wavefront can read "ring slot idx" -- what the lead
// "ring in registers instead of LDS"  would trade: each access depends on the previous one's result.
//   0 LDS read at a wave-uniform index + v_readfirstlane      (today's c.xy[idx] for a uniform idx)
//   1 v_readlane of a register-resident ring                    (one vertex per lane)
//   2 LDS read at a per-lane index                              (today's job-lane gathers)
//   3 ds_bpermute of a register-resident ring                   (per-lane gather without an LDS image)
//   4 DPP wave shift (neighbour i - 1)                          (ring neighbours)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_access tools/ubench_access.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP> __global__ void chain(double *out, int n, unsigned long long *cyc)
{
    __shared__ double ring[64];
    const int lane = threadIdx.x & 63;
    ring[lane] = (double)((lane * 7 + 3) % n);
    double reg = ring[lane];
    __syncthreads();
    int idx = lane % n;
    double acc = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
    for (int i = 0; i < N; i++) {
        if (OP == 0) {
            const int u = __builtin_amdgcn_readfirstlane(idx);
            const double v = ring[u];
            idx = (int)v; acc += v;
        } else if (OP == 1) {
            const int u = __builtin_amdgcn_readfirstlane(idx);
            const int lo = __builtin_amdgcn_readlane(__double2loint(reg), u), hi = __builtin_amdgcn_readlane(__double2hiint(reg), u);
            const double v = __hiloint2double(hi, lo);
            idx = (int)v; acc += v;
        } else if (OP == 2) {
            const double v = ring[idx];
            idx = (int)v; acc += v;
        } else if (OP == 3) {
            const int lo = __builtin_amdgcn_ds_bpermute(idx * 4, __double2loint(reg)), hi = __builtin_amdgcn_ds_bpermute(idx * 4, __double2hiint(reg));
            const double v = __hiloint2double(hi, lo);
            idx = (int)v; acc += v;
        } else {
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(reg), 0x138, 0xf, 0xf, false);   // wave_shr:1
            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(reg), 0x138, 0xf, 0xf, false);
            reg = __hiloint2double(hi, lo) + 1.0; acc += reg;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h[4096];
    hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cyc, 8 * 4096);
    const char *names[] = {"LDS uniform idx + readfirstlane", "v_readlane x2 (register ring)", "LDS per-lane idx", "ds_bpermute x2 (register ring)", "DPP wave_shr x2 + add"};
#define RUN(OP, blocks) { chain<OP><<<blocks, 64>>>(out, 30, cyc); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost); \
    double s = 0; for (int i = 0; i < blocks; i++) s += h[i]; printf("%-34s blocks=%5d  %.1f cycles per dependent access\n", names[OP], blocks, s / blocks / N); }
    for (int blocks : {1, 1024, 4096}) { RUN(0, blocks) RUN(1, blocks) RUN(2, blocks) RUN(3, blocks) RUN(4, blocks) }
    return 0;
}
