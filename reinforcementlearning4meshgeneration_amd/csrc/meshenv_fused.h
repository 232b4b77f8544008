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

template <bool kDefaultParams>
__global__ void __launch_bounds__(64 * 16)
k_step_group_actor(GroupActorArgs A)
{
    extern __shared__ double2 smem[];
    float *lds = (float *)((char *)smem + group_lds_bytes(A.g.cap, 16));
    // every wave writes the observation its step ends with into the actor's input row as well (finish_and_store)
    step_group_body<16, kDefaultParams>(A.g, lds);   // (returns for every wave: nothing exits before the barrier)
    ActorHead hd;
    actor_request_weights(hd, A.W, threadIdx.x, true);   // all three layers: in flight while the workgroup's slowest wave finishes
    __syncthreads();
    actor_forward_tile(A.W, hd, A.g.S.n_envs, blockIdx.x * kActEnvs, nullptr, nullptr, A.actions_next, A.sample, A.seed, A.counter,
                       A.eps_out, lds, threadIdx.x, 64 * 16);
}

}  // namespace meshenv
