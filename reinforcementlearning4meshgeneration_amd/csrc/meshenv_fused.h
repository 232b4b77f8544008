// meshenv_fused.h -- one launch per vector step of the closed RL loop: the environment step of the CU-group kernel and the
// SAC actor's forward for the SAME 16 environments, back to back in one workgroup.
//
// The policy-in-the-loop path is obs -> actor -> actions -> step -> obs: two dependent launches per vector step
// (k_actor_forward 8.1 us + k_step_group 17.6 us on d1, plus the gap between them).  Both kernels already map 16
// environments to one workgroup on one CU, so the second can simply continue where the first ends: the workgroup's 16
// wavefronts run step_group_body (csrc/meshenv_kernels.h), each leaving the observation its step ends with in the actor's
// LDS input row as well, request the actor's first weights, meet at a barrier, and the first eight wavefronts run
// actor_forward_tile (csrc/meshenv_actor.h), producing the actions of the NEXT step.  Same device functions
// as the two-launch path, so the results are bit-identical to it (tests/test_gpu_actor.py); what is saved is the second
// kernel's launch ramp and the inter-kernel gap.
#pragma once

#include "meshenv_actor.h"
#include "meshenv_kernels.h"

namespace meshenv {

struct GroupActorArgs {
    GroupArgs g;             // FIRST: late_outs() reads the step's output pointers at GroupArgs' offsets of the kernarg segment
    ActorWeights W;
    float *actions_next;     // [n][3]: the policy's actions for the next step (must not alias g.actions)
    float *eps_out;          // [n][3] exploration noise drawn, nullable
    unsigned long long seed, counter;
    int sample;              // 1: in-kernel Philox noise (SAC's stochastic actor), 0: the mean action
    int pad;
};

__host__ __device__ __forceinline__ size_t group_actor_lds_bytes(int cap)
{
    return group_lds_bytes(cap, 16) + sizeof(float) * kActorLdsFloats;
}

template <bool kDefaultParams, bool kSmall = false>
__global__ void __launch_bounds__(64 * 16)
k_step_group_actor(GroupActorArgs A)
{
    extern __shared__ double2 smem[];
    float *lds = (float *)((char *)smem + group_lds_bytes(A.g.cap, 16));
    // every wave writes the observation its step ends with into the actor's input row as well (finish_and_store)
    step_group_body<16, kDefaultParams, false, kSmall>(A.g, lds);   // (returns for every wave: nothing exits before the barrier)
    ActorHead hd;
    actor_request_weights(hd, A.W, threadIdx.x, true);   // all three layers: in flight while the workgroup's slowest wave finishes
    __syncthreads();
    actor_forward_tile(A.W, hd, A.g.S.n_envs, blockIdx.x * kActEnvs, nullptr, nullptr, A.actions_next, A.sample, A.seed, A.counter,
                       A.eps_out, lds, threadIdx.x, 64 * 16);
}

// T vector steps of the closed loop in ONE launch.  The workgroups never talk to each other -- a workgroup owns 16
// environments and runs their policy -- so nothing forces the grid back to the host between steps: each workgroup loops
// [step its 16 envs -> barrier -> actor forward -> barrier] T times on its own.  What that buys: one launch ramp / drain per
// T steps instead of per step, and the workgroups drift apart, so the launch no longer ends, every step, with the CU that
// happens to hold the most extractions (tools/balance_bound.py prices that tail at ~1.8 us per step) -- the imbalance
// averages out over T steps.  Outputs are [T][n]-shaped histories (slice t = what meshenv_step_actor would have written at
// step t), actions [T + 1][n][3]: slice 0 is the input, slice t + 1 the policy's answer to the observations of step t; the
// noise counter advances by one per step, so the stream equals T calls of meshenv_step_actor with counter, counter + 1, ...
// Environment state goes through HBM between steps exactly as between launches (the same load / store code), ordered by
// the workgroup barrier: results are bit-identical to T single-step launches (tests/test_gpu_actor.py).
struct GroupActorArgsT {
    GroupArgs g;             // FIRST (late_outs); g.outs = slice 0 of the [T][n] histories, g.actions = slice 0 of actions
    ActorWeights W;
    float *eps_out;          // [T][n][3] nullable
    unsigned long long seed, counter;
    int sample;
    int T;
    unsigned long long *dbg; // nullable: [grid][T][4] s_memrealtime stamps per workgroup and step (tools/tsteps_timeline.py)
};

// (all three weight layers requested up front: requesting only the first spills 117 VGPRs instead of 77)
template <bool kDefaultParams>
__global__ void __launch_bounds__(64 * 16)
k_step_group_actor_T(GroupActorArgsT A)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (the address-space-4 reads below have no host meaning)
    extern __shared__ double2 smem[];
    const int T = A.T;
    for (int t = 0; t < T; t++) {
        // Nothing of one iteration may be carried to the next in registers: the step body derives long tables from the
        // thread id and the arguments, and hoisted out of the loop they do not fit the 128 VGPRs of a 16-wave workgroup
        // (inlined as it stands: 209 VGPRs spilled, 768 B of scratch per lane).  So the thread id, the step index and the
        // kernel-argument pointer are made opaque once per iteration: everything derived from them -- the arguments
        // included -- is loaded / recomputed inside the step and dies with it, as in the one-step kernel.
        int tid = (int)threadIdx.x, tt = t;
        KernArgPtr ka = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(tt), "+s"(ka));
        typedef const __attribute__((address_space(4))) GroupActorArgsT *aptr;
        const aptr Ap = (aptr)ka;
        float *lds = (float *)((char *)smem + group_lds_bytes(Ap->g.cap, 16));
        unsigned long long *dbg = Ap->dbg;
        if (dbg && tid == 0) dbg[((size_t)blockIdx.x * T + tt) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        {
            GroupArgs g = Ap->g;
            const size_t n = (size_t)g.S.n_envs;
            g.actions += (size_t)tt * n * 3;
            g.step0 += (unsigned long long)tt;
            step_group_body<16, kDefaultParams>(g, lds, tt, ka, tid);
        }
        asm volatile("" : "+s"(ka));
        const aptr Bp = (aptr)ka;
        const ActorWeights W = Bp->W;
        ActorHead hd;
        actor_request_weights(hd, W, tid, true);
        if (dbg && tid == 0) dbg[((size_t)blockIdx.x * T + tt) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (dbg && tid == 0) dbg[((size_t)blockIdx.x * T + tt) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        const size_t n = (size_t)Bp->g.S.n_envs;
        float *eps = Bp->eps_out;
        actor_forward_tile(W, hd, (int)n, blockIdx.x * kActEnvs, nullptr, nullptr, const_cast<float *>(Bp->g.actions) + (size_t)(tt + 1) * n * 3,
                           Bp->sample, Bp->seed, Bp->counter + (unsigned long long)tt, eps ? eps + (size_t)tt * n * 3 : nullptr, lds,
                           tid, 64 * 16);
        __syncthreads();   // the actions of step t + 1 and the environments' state are written: visible to the whole workgroup
        if (dbg && tid == 0) dbg[((size_t)blockIdx.x * T + tt) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace meshenv
