// Dev microbenchmark: what does handing a pending extraction to ANOTHER compute unit cost?
//
// 256 workgroups of 16 waves, one per CU (the k_step_group<16> launch shape).  Producer workgroups (even blocks) do what a
// busy CU would: lane 0 of a wave writes a 320-byte job record (the Handoff + Decision of csrc/meshenv_kernels.h) to global
// memory, fences, and publishes it with one atomic ticket on a global queue.  Consumer workgroups (odd blocks) do what an
// idle CU would: one wave polls the queue tail (s_sleep between polls), claims a job, reads the record, then stages that
// env's ring from HBM into LDS exactly as load_env does (record 64 B + 32 slots x 28 B) -- the ring was last touched by the
// producer's XCD, not the consumer's.  s_memrealtime stamps (100 MHz) at: job published | job seen by the consumer | record
// read | ring staged.  The producer and consumer of a pair sit on different XCDs (consecutive workgroups are dealt
// round-robin over the 8 XCDs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_handoff tools/ubench_handoff.hip && tools/ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

struct alignas(64) Job { int env; int pad[15]; double payload[32]; };   // 320 B

__global__ void __launch_bounds__(1024)
k_handoff(Job *jobs, unsigned *tail, unsigned *head, const double2 *ring_xy, const int *ring_id, const double *ring_key,
          const int *ring_stamp, const double *scal, unsigned long long *stamps, int n_pairs, int spin_sleep)
{
    extern __shared__ double2 lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // modes 0..2: producer = block 2p, consumer = 2p + 1 (consecutive workgroups sit on different XCDs: release / acquire at
    // agent scope, i.e. an L2 write-back on one side and an invalidate on the other).  mode 3: producer = block b with
    // (b & 8) == 0, consumer = b + 8 -- the SAME XCD, one shared L2 -- and the record travels as relaxed agent-scope atomics
    // (performed at the L2, no write-back, no invalidate): the cheapest hand-over the hardware offers between two CUs.
    const bool same_xcd = spin_sleep == 3;
    const int pair = same_xcd ? ((blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) : (blockIdx.x >> 1);
    const bool producer = same_xcd ? (blockIdx.x & 8) == 0 : (blockIdx.x & 1) == 0;
    if (wave != 0) return;   // the other 15 waves have finished their step
    if (producer) {
        // the env this CU's wave would have updated: touch its ring first (it sits in THIS XCD's L2, like after phase 1)
        const int env = pair;
        double acc = 0;
        if (lane < 32) acc = ring_xy[env * 32 + lane].x + ring_key[env * 32 + lane];
        acc += __shfl_xor(acc, 1);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        Job &j = jobs[pair];                                        // pair-private mailbox (a shared queue adds one atomic)
        if (same_xcd) {
            if (lane < 32) __hip_atomic_store(&j.payload[lane], acc + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) __hip_atomic_store(&j.env, env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                          // the stores have reached the L2
            if (lane == 0) {
                stamps[pair * 8 + 0] = t0;
                stamps[pair * 8 + 1] = __builtin_amdgcn_s_memrealtime();
                __hip_atomic_store(&j.pad[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        if (lane < 32) j.payload[lane] = acc + lane;                // the record, one store per lane
        if (lane == 0) j.env = env;
        __threadfence();                                            // record visible before the publication
        if (lane == 0) {
            stamps[pair * 8 + 0] = t0;
            stamps[pair * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            __hip_atomic_store(&j.pad[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (spin_sleep == 2) atomicAdd(tail, 1u);               // + the ticket of a shared queue
        }
    } else {
        unsigned long long t_seen = 0;
        if (lane == 0) {
            if (same_xcd) {
                while (__hip_atomic_load(&jobs[pair].pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(4);
            } else {
                while (__hip_atomic_load(&jobs[pair].pad[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    if (spin_sleep) __builtin_amdgcn_s_sleep(4);
                }
            }
            if (spin_sleep == 2) atomicAdd(tail + 1, 1u);           // + the claim of a shared queue
            t_seen = __builtin_amdgcn_s_memrealtime();
        }
        __builtin_amdgcn_wave_barrier();
        Job &j = jobs[pair];
        const int env = __builtin_amdgcn_readfirstlane(same_xcd ? __hip_atomic_load(&j.env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : j.env);
        double v = lane < 32 ? (same_xcd ? __hip_atomic_load(&j.payload[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : j.payload[lane]) : 0.0;
        v += __shfl_xor(v, 1);
        const unsigned long long t_rec = __builtin_amdgcn_s_memrealtime();
        // stage the env like load_env: record + the four ring arrays, one burst
        const double sc = scal[env * 8 + (lane & 7)];
        double2 xy = make_double2(0, 0); int id = 0, st = 0; double key = 0;
        if (lane < 32) { xy = ring_xy[env * 32 + lane]; id = ring_id[env * 32 + lane]; key = ring_key[env * 32 + lane]; st = ring_stamp[env * 32 + lane]; }
        lds[lane] = make_double2(xy.x + sc + key, xy.y + id + st + v);
        __builtin_amdgcn_wave_barrier();
        const double chk = lds[lane ^ 1].x;
        const unsigned long long t_ring = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            const int p = env;   // pair that produced the job
            stamps[p * 8 + 2] = t_seen; stamps[p * 8 + 3] = t_rec; stamps[p * 8 + 4] = t_ring;
            stamps[p * 8 + 5] = (unsigned long long)(chk != 12345.0);
        }
    }
}

__global__ void k_busy(double *out, int iters)
{
    double x = threadIdx.x * 1e-9 + 0.7;
    for (int i = 0; i < iters; i++) x = fma(x, 0.9999, 0.0001);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main()
{
    const int blocks = 256, pairs = blocks / 2, n_env = 4096;
    Job *jobs; unsigned *ctl; double2 *xy; int *id, *st; double *key, *scal; unsigned long long *stamps;
    hipMalloc(&jobs, sizeof(Job) * pairs); hipMalloc(&ctl, 64);
    hipMalloc(&xy, 16 * 32 * n_env); hipMalloc(&id, 4 * 32 * n_env); hipMalloc(&st, 4 * 32 * n_env); hipMalloc(&key, 8 * 32 * n_env);
    hipMalloc(&scal, 64 * n_env); hipMalloc(&stamps, 8 * 8 * pairs);
    hipMemset(xy, 0, 16 * 32 * n_env); hipMemset(id, 0, 4 * 32 * n_env); hipMemset(st, 0, 4 * 32 * n_env); hipMemset(key, 0, 8 * 32 * n_env); hipMemset(scal, 0, 64 * n_env);
    double *busy_out; hipMalloc(&busy_out, 8 * 1024 * 1024);
    // clock warm-up: an idle MI355X sits at its lowest clock and these launches are microseconds long
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k_busy, dim3(1024), dim3(1024), 0, 0, busy_out, 200000);
    hipDeviceSynchronize();
    for (int sleep = 0; sleep < 4; sleep++) {
        std::vector<double> pub, seen, rec, ring;
        for (int rep = 0; rep < 60; rep++) {
            hipMemset(jobs, 0, sizeof(Job) * pairs); hipMemset(ctl, 0, 64); hipMemset(stamps, 0, 8 * 8 * pairs);
            hipLaunchKernelGGL(k_busy, dim3(1024), dim3(1024), 0, 0, busy_out, 20000);   // keeps the clock up between the short launches
            hipLaunchKernelGGL(k_handoff, dim3(blocks), dim3(1024), 4096, 0, jobs, ctl, ctl + 4, xy, id, key, st, scal, stamps, pairs, sleep);
            hipDeviceSynchronize();
            if (rep < 10) continue;
            std::vector<unsigned long long> h(8 * pairs);
            hipMemcpy(h.data(), stamps, 8 * 8 * pairs, hipMemcpyDeviceToHost);
            for (int p = 0; p < pairs; p++) {
                const unsigned long long *s = &h[8 * p];
                if (!s[2] || s[2] < s[1]) continue;   // consumer was already waiting is the case of interest: seen after published
                pub.push_back((s[1] - s[0]) * 0.01); seen.push_back((s[2] - s[1]) * 0.01);
                rec.push_back((s[3] - s[2]) * 0.01); ring.push_back((s[4] - s[3]) * 0.01);
            }
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? -1.0 : v[v.size() / 2]; };
        auto p90 = [](std::vector<double> &v) { return v.empty() ? -1.0 : v[v.size() * 9 / 10]; };
        printf("poll %s: samples %zu | write record + fence %.2f us | published -> seen by the consumer %.2f us (p90 %.2f) | read record %.2f us | stage ring from the other XCD %.2f us (p90 %.2f) | total %.2f us\n",
               sleep == 0 ? "busy        " : (sleep == 1 ? "with s_sleep" : (sleep == 2 ? "sleep+ticket" : "same XCD, relaxed atomics")), pub.size(), med(pub), med(seen), p90(seen), med(rec), med(ring), p90(ring),
               med(pub) + med(seen) + med(rec) + med(ring));
    }
    return 0;
}
