// Dev tool: calibrates rocprofv3's FETCH_SIZE for THIS library's access pattern (MI355X_MICROARCH.md: on gfx950 the
// counter reads 1/2 of the bytes of 16-B-per-lane streams; other widths are uncalibrated).  One wavefront per "env"
// issues exactly the loads of load_env() -- per lane one double2 (16 B), one int32, one double, one int32 for the first
// `cap` lanes, plus the wave-uniform 64-byte record, the 32-byte counters, 12 bytes of action and 18 floats of cached
// observation -- over a footprint far beyond L2 + Infinity Cache, so every byte comes from HBM and
//     factor = bytes_requested / (FETCH_SIZE * 1024)
// is what one counted KiB stands for in this mix.   build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_fetch tools/calib_fetch.hip
// run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/calib_fetch 1048576 32      (tools/calib_fetch.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(64) Rec { int v[16]; };
struct alignas(32) Cnt { unsigned long long v[4]; };

__global__ void __launch_bounds__(64) k_touch(int cap, const double2 *xy, const int *id, const double *key, const int *st,
                                               const Rec *rec, const Cnt *cnt, const float *act, const float *obs, double *sink)
{
    const int env = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)env * cap;
    double acc = 0.0;
    const Rec r = rec[env];
    const Cnt c = cnt[env];
    const float a0 = act[3 * (size_t)env], a1 = act[3 * (size_t)env + 1], a2 = act[3 * (size_t)env + 2];
    if (lane < cap) {
        const double2 p = xy[base + lane];
        acc += p.x + p.y + (double)id[base + lane] + key[base + lane] + (double)st[base + lane];
    }
    if (lane < 18) acc += obs[(size_t)env * 18 + lane];
    acc += (double)r.v[lane & 15] + (double)c.v[lane & 3] + a0 + a1 + a2;
    if (acc == 1.2345e300) sink[0] = acc;   // never true: keeps the loads alive
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1u << 20);
    const int cap = argc > 2 ? atoi(argv[2]) : 32;
    double2 *xy; int *id, *st; double *key, *sink; Rec *rec; Cnt *cnt; float *act, *obs;
    hipMalloc(&xy, n * cap * sizeof(double2)); hipMalloc(&id, n * cap * 4); hipMalloc(&key, n * cap * 8); hipMalloc(&st, n * cap * 4);
    hipMalloc(&rec, n * sizeof(Rec)); hipMalloc(&cnt, n * sizeof(Cnt)); hipMalloc(&act, n * 12); hipMalloc(&obs, n * 72); hipMalloc(&sink, 8);
    hipMemset(xy, 1, n * cap * sizeof(double2)); hipMemset(id, 1, n * cap * 4); hipMemset(key, 1, n * cap * 8); hipMemset(st, 1, n * cap * 4);
    hipMemset(rec, 1, n * sizeof(Rec)); hipMemset(cnt, 1, n * sizeof(Cnt)); hipMemset(act, 1, n * 12); hipMemset(obs, 1, n * 72);
    hipDeviceSynchronize();
    const int lanes = cap < 64 ? cap : 64;
    const double bytes = (double)n * (lanes * 32.0 + 64 + 32 + 12 + 72);
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(k_touch, dim3((unsigned)n), dim3(64), 0, 0, cap, xy, id, key, st, rec, cnt, act, obs, sink);
        hipDeviceSynchronize();
    }
    printf("{\"envs\": %zu, \"cap\": %d, \"bytes_requested_per_launch\": %.0f, \"footprint_MB\": %.1f}\n", n, cap, bytes, bytes / 1e6);
    return 0;
}
