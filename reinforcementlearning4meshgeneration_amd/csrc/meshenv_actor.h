// meshenv_actor.h -- fused SAC actor forward (the caller of the hot path).
//
// The reference trains Stable-Baselines3 SAC with MlpPolicy, ReLU, net_arch [128, 128, 128]
// (rl/baselines/RL_Mesh.py:183-196).  SB3's actor is latent = MLP(obs); mean = Linear(latent), log_std =
// Linear(latent) clamped to [-20, 2]; action = tanh(mean + exp(log_std) * eps), rescaled from [-1, 1] to the action
// Box.  Run eagerly that is ~25 small launches per vector step (~90 us), more than four times the environment
// step itself; this kernel is the whole forward in one launch so that observations and actions never leave the GPU
// and the rollout loop stays at two launches per vector step.
//
// fp32 FMA (no MFMA: 4096 x 128 x 128 is 0.3 GFLOP per step, launch-latency territory), weights transposed to
// [in][out] at upload so that lane j reads W[k][j] coalesced, activations transposed in LDS ([k][env]) so that one
// ds_read_b128 feeds four environments.  32 environments per 256-thread workgroup: thread t computes 4 neurons x 4 envs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace meshenv {

constexpr int kActIn = 18, kActHid = 128, kActOut = 3, kActEnvs = 32;

struct ActorWeights {
    const float *w1t, *b1;   // [18][128], [128]
    const float *w2t, *b2;   // [128][128], [128]
    const float *w3t, *b3;   // [128][128], [128]
    const float *wht, *bh;   // heads: [128][8] (mu0..2, log_std0..2, 0, 0), [8]
    float low[3], high[3];
};

// one hidden layer: y[j][e] = relu(b[j] + sum_k W[k][j] * x[k][e]); thread t: neurons 4*(t&31).., envs 4*(t>>5)..
template <int K>
__device__ __forceinline__ void actor_layer(const float *__restrict__ wt, const float *__restrict__ b,
                                            const float *xT /*[K][32]*/, float *yT /*[128][32]*/)
{
    const int t = threadIdx.x, jg = t & 31, eg = t >> 5;
    float acc[4][4];
    const float4 bv = *reinterpret_cast<const float4 *>(b + 4 * jg);
#pragma unroll
    for (int e = 0; e < 4; e++) { acc[0][e] = bv.x; acc[1][e] = bv.y; acc[2][e] = bv.z; acc[3][e] = bv.w; }
#pragma unroll 4
    for (int k = 0; k < K; k++) {
        const float4 w = *reinterpret_cast<const float4 *>(wt + (size_t)k * kActHid + 4 * jg);
        const float4 x = *reinterpret_cast<const float4 *>(xT + k * kActEnvs + 4 * eg);
        acc[0][0] = fmaf(w.x, x.x, acc[0][0]); acc[0][1] = fmaf(w.x, x.y, acc[0][1]); acc[0][2] = fmaf(w.x, x.z, acc[0][2]); acc[0][3] = fmaf(w.x, x.w, acc[0][3]);
        acc[1][0] = fmaf(w.y, x.x, acc[1][0]); acc[1][1] = fmaf(w.y, x.y, acc[1][1]); acc[1][2] = fmaf(w.y, x.z, acc[1][2]); acc[1][3] = fmaf(w.y, x.w, acc[1][3]);
        acc[2][0] = fmaf(w.z, x.x, acc[2][0]); acc[2][1] = fmaf(w.z, x.y, acc[2][1]); acc[2][2] = fmaf(w.z, x.z, acc[2][2]); acc[2][3] = fmaf(w.z, x.w, acc[2][3]);
        acc[3][0] = fmaf(w.w, x.x, acc[3][0]); acc[3][1] = fmaf(w.w, x.y, acc[3][1]); acc[3][2] = fmaf(w.w, x.z, acc[3][2]); acc[3][3] = fmaf(w.w, x.w, acc[3][3]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float4 o;
        o.x = fmaxf(acc[j][0], 0.0f); o.y = fmaxf(acc[j][1], 0.0f); o.z = fmaxf(acc[j][2], 0.0f); o.w = fmaxf(acc[j][3], 0.0f);
        *reinterpret_cast<float4 *>(yT + (4 * jg + j) * kActEnvs + 4 * eg) = o;
    }
}

__global__ void __launch_bounds__(256)
k_actor_forward(ActorWeights W, int n, const float *__restrict__ obs, const float *__restrict__ noise,
                float *__restrict__ actions)
{
    __shared__ __attribute__((aligned(16))) float bufA[kActHid * kActEnvs];
    __shared__ __attribute__((aligned(16))) float bufB[kActHid * kActEnvs];
    const int t = threadIdx.x;
    const int env0 = blockIdx.x * kActEnvs;
    // observations -> bufB as [k][env] (zero for envs past n)
    for (int i = t; i < kActIn * kActEnvs; i += 256) {
        const int e = i / kActIn, k = i - e * kActIn;
        bufB[k * kActEnvs + e] = (env0 + e < n) ? obs[(size_t)(env0 + e) * kActIn + k] : 0.0f;
    }
    __syncthreads();
    actor_layer<kActIn>(W.w1t, W.b1, bufB, bufA);
    __syncthreads();
    actor_layer<kActHid>(W.w2t, W.b2, bufA, bufB);
    __syncthreads();
    actor_layer<kActHid>(W.w3t, W.b3, bufB, bufA);
    __syncthreads();
    // heads: thread t < 192 -> (env e = t / 6, output o = t % 6): mu0..2, log_std0..2
    float v = 0.0f;
    const int e = t / 6, o = t - 6 * e;
    if (t < 6 * kActEnvs) {
        v = W.bh[o];
        for (int k = 0; k < kActHid; k++) v = fmaf(W.wht[k * 8 + o], bufA[k * kActEnvs + e], v);
    }
    __syncthreads();
    if (t < 6 * kActEnvs) bufB[e * 8 + o] = v;
    __syncthreads();
    if (t < 3 * kActEnvs) {
        const int ee = t / 3, a = t - 3 * ee;
        if (env0 + ee < n) {
            float mu = bufB[ee * 8 + a];
            if (noise) {
                const float ls = fminf(fmaxf(bufB[ee * 8 + 3 + a], -20.0f), 2.0f);
                mu += expf(ls) * noise[(size_t)(env0 + ee) * 3 + a];
            }
            const float sq = tanhf(mu);
            actions[(size_t)(env0 + ee) * 3 + a] = W.low[a] + 0.5f * (sq + 1.0f) * (W.high[a] - W.low[a]);
        }
    }
}

}  // namespace meshenv
