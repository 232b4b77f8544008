// meshenv_actor.h -- fused SAC actor forward (the caller of the hot path).
//
// The reference trains Stable-Baselines3 SAC with MlpPolicy, ReLU, net_arch [128, 128, 128]
// (rl/baselines/RL_Mesh.py:183-196).  SB3's actor is latent = MLP(obs); mean = Linear(latent), log_std =
// Linear(latent) clamped to [-20, 2]; action = tanh(mean + exp(log_std) * eps), rescaled from [-1, 1] to the action
// Box.  Run eagerly that is ~25 small launches per vector step (~90 us), more than four times the environment
// step itself; this kernel is the whole forward in one launch so that observations and actions never leave the GPU
// and the rollout loop stays at two launches per vector step.
//
// GEMM-shaped, so it runs on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate -- bit-for-bit a
// fmaf chain, no reduced precision).  16 environments per 512-thread workgroup (4096 envs -> 256 workgroups, one per
// CU, eight wavefronts = two per SIMD); wave w owns neurons [16w, 16w + 16) as ONE 16x16 output tile whose k-loop is
// split over two accumulators (even / odd k-groups, summed at the end: two independent chains cover the 40-cycle
// dependent MFMA latency).  The B operand (weights) of a whole layer lives in registers: 32 floats per lane, packed on
// the host in exactly the per-lane order so that a wave loads them with eight coalesced 1 KB requests, issued one layer
// ahead of their use.  (Four waves with two tiles each: 8.5 us per launch; eight waves: measured in DESIGN.md section 7.)  The A operand (activations) goes through LDS in a k-permuted layout
// ([env][k & 3][k >> 2], row stride 132 floats) so that one conflict-free ds_read_b128 feeds four MFMA k-steps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace meshenv {

constexpr int kActIn = 18, kActHid = 128, kActOut = 3;
constexpr int kActEnvs = 16;    // environments per workgroup = MFMA M
constexpr int kActInPad = 32;   // layer-1 K padded to a multiple of 16
constexpr int kActStride = 132; // LDS row stride in floats: 132 mod 64 = 4 -> 16 lanes x 16 B hit 64 distinct banks

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Packed weight of a layer with K inputs: [tile 8][K/16][lane 64][4] (tile = wave); element j of lane l in group g is
// W[k = 4 * (4 g + j) + (l >> 4)][n = 16 tile + (l & 15)]  (meshenv_actor_load builds it).
struct ActorWeights {
    const float *w1p, *b1;  // K = 32 (18 padded), [128]
    const float *w2p, *b2;  // K = 128
    const float *w3p, *b3;  // K = 128
    const float *whp, *bh;  // heads: one 16-wide tile [8][64][4] (n = mu0..2, log_std0..2, zeros), [16]
    float low[3], high[3];
};

// Philox4x32-10 (Salmon et al., SC'11) keyed by the caller's seed, counter = (env, draw counter): the exploration
// noise of SAC's actor without a separate random-number launch.  Returns four uniform 32-bit words.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4])
{
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// standard normal number `which` (0..2) of environment `env` at draw `counter`: Box-Muller on two Philox words
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t counter, uint32_t env, int which)
{
    uint32_t r[4];
    philox4x32(env, (uint32_t)counter, (uint32_t)(counter >> 32), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const uint32_t a = which < 2 ? r[0] : r[2], b = which < 2 ? r[1] : r[3];
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.2831853071795864f * u2, &sn, &cs);
    return rad * (which == 1 ? sn : cs);
}

constexpr int kActWaves = 8;    // wavefronts per workgroup = 16-neuron output tiles of a hidden layer

template <int G>  // G = K / 16 float4 groups per tile
struct LayerRegs {
    f32x4 w[G];
};

template <int G>
__device__ __forceinline__ void load_layer(LayerRegs<G> &r, const float *__restrict__ wp, int wave, int lane)
{
#pragma unroll
    for (int g = 0; g < G; g++)
        r.w[g] = *reinterpret_cast<const f32x4 *>(wp + (((size_t)wave * G + g) * 64 + lane) * 4);
}

// y = relu(W x + b) for the wave's 16 neurons; x: LDS [16][kActStride] in the permuted layout of a K-input layer
// (position (k & 3) * (K / 4) + (k >> 2)); y: the 128-input layout of the next layer
template <int G>
__device__ __forceinline__ void actor_layer(const LayerRegs<G> &r, const float *__restrict__ bias, const float *x, float *y,
                                            int wave, int lane)
{
    const int e = lane & 15, q = lane >> 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float *xr = x + e * kActStride + q * (4 * G);
#pragma unroll
    for (int g = 0; g < G; g += 2) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(xr + 4 * g);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(xr + 4 * g + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], r.w[g][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], r.w[g + 1][j], acc1, 0, 0, 0);
        }
    }
    // D[row = 4 (lane >> 4) + reg][col = lane & 15]
    const int n0 = 16 * wave + e;
    const float b0 = bias[n0];
    const int p0 = (n0 & 3) * 32 + (n0 >> 2);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        float *row = y + (4 * q + reg) * kActStride;
        row[p0] = fmaxf((acc0[reg] + acc1[reg]) + b0, 0.0f);
    }
}

// LDS of one actor tile: two activation buffers + the exploration noise
constexpr int kActorLdsFloats = 2 * kActEnvs * kActStride + kActEnvs * 4;

// the first two layers' weights of a wavefront, requested ahead of the forward (k_step_group_actor: before the barrier
// that ends the environment step, so that they arrive while the workgroup's slowest wave finishes)
struct ActorHead {
    LayerRegs<2> r1;
    LayerRegs<8> r2;
    LayerRegs<8> r3;
    bool have_r3;
};

// (all3: the third layer as well -- pays where the requester then waits at a barrier anyway, k_step_group_actor; in the
// standalone kernel one layer ahead is faster: 8.11 against 8.27 us)
__device__ __forceinline__ void actor_request_weights(ActorHead &hd, const ActorWeights &W, int t, const bool all3 = false)
{
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar also when t is an opaque value
    hd.have_r3 = all3;
    if (wave < kActWaves) {
        load_layer<2>(hd.r1, W.w1p, wave, lane);
        load_layer<8>(hd.r2, W.w2p, wave, lane);
        if (all3) load_layer<8>(hd.r3, W.w3p, wave, lane);
    }
}

// The forward of the 16 envs env0 .. env0 + 15 by the first kActWaves wavefronts of a workgroup (`t` = thread index in the
// workgroup; every thread of the workgroup must call: the function contains workgroup barriers, wavefronts past
// kActWaves only take part in those).  lds: kActorLdsFloats floats, 16-byte aligned.  obs == nullptr: the first-layer
// input rows (lds[e * kActStride + (k & 3) * 8 + (k >> 2)], k < 18) were written by the caller; the padding is zeroed here.
__device__ __forceinline__ void actor_forward_tile(const ActorWeights &W, ActorHead &hd, int n, int env0,
                                                   const float *__restrict__ obs, const float *__restrict__ noise,
                                                   float *__restrict__ actions, int sample, uint64_t seed, uint64_t counter,
                                                   float *__restrict__ eps_out, float *lds, int t, int n_threads)
{
    float *bufA = lds, *bufB = lds + kActEnvs * kActStride, *eps_lds = lds + 2 * kActEnvs * kActStride;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (k_step_group_actor_T passes an opaque thread id)
    const bool worker = wave < kActWaves;
    LayerRegs<2> &r1 = hd.r1;
    LayerRegs<8> &r2 = hd.r2;
    LayerRegs<8> &r3 = hd.r3;
    // exploration noise, drawn while the weights are in flight: wave w draws row 4 q + w of every 4-row group
    if (wave < 4 && (lane & 15) < kActOut && (noise || sample)) {
        const int row = 4 * (lane >> 4) + wave, env = env0 + row;
        float eps = 0.0f;
        if (env < n) eps = sample ? philox_normal(seed, counter, (uint32_t)env, lane & 15) : noise[(size_t)env * 3 + (lane & 15)];
        eps_lds[row * 4 + (lane & 15)] = eps;
    }
    // observations -> bufA in the K = 32 layout (zero for k >= 18 and for envs past n)
    for (int i = t; i < kActEnvs * kActInPad; i += n_threads) {
        const int e = i >> 5, k = i & 31;
        if (obs) {
            const float v = (k < kActIn && env0 + e < n) ? obs[(size_t)(env0 + e) * kActIn + k] : 0.0f;
            bufA[e * kActStride + (k & 3) * (kActInPad / 4) + (k >> 2)] = v;
        } else if (k >= kActIn || env0 + e >= n) {
            bufA[e * kActStride + (k & 3) * (kActInPad / 4) + (k >> 2)] = 0.0f;
        }
    }
    __syncthreads();
    if (worker) {
        actor_layer<2>(r1, W.b1, bufA, bufB, wave, lane);
        if (!hd.have_r3) load_layer<8>(r3, W.w3p, wave, lane);   // one layer ahead
    }
    __syncthreads();
    if (worker) actor_layer<8>(r2, W.b2, bufB, bufA, wave, lane);
    f32x4 wh[8];
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < 8; g++) wh[g] = *reinterpret_cast<const f32x4 *>(W.whp + ((size_t)g * 64 + lane) * 4);
    }
    __syncthreads();
    if (worker) actor_layer<8>(r3, W.b3, bufA, bufB, wave, lane);
    __syncthreads();
    if (wave != 0) return;
    // heads on wave 0: one 16-wide tile, two accumulators over even / odd k-groups
    const int e = lane & 15, q = lane >> 4;
    f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
    const float *xr = bufB + e * kActStride + q * 32;
#pragma unroll
    for (int g = 0; g < 8; g += 2) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(xr + 4 * g);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(xr + 4 * g + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], wh[g][j], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], wh[g + 1][j], h1, 0, 0, 0);
        }
    }
    const float bh = W.bh[e];
    // lane (col e, rows 4 q + reg): col a < 3 is mu_a, col 3 + a its log_std
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const float v = h0[reg] + h1[reg] + bh;
        const float ls_raw = __shfl(v, lane + 3, 64);
        const int env = env0 + 4 * q + reg;
        if (e < kActOut && env < n) {
            float mu = v;
            if (noise || sample) {
                const float eps = eps_lds[(4 * q + reg) * 4 + e];
                if (eps_out) eps_out[(size_t)env * 3 + e] = eps;
                const float ls = fminf(fmaxf(ls_raw, -20.0f), 2.0f);
                mu += expf(ls) * eps;
            }
            const float sq = tanhf(mu);
            actions[(size_t)env * 3 + e] = W.low[e] + 0.5f * (sq + 1.0f) * (W.high[e] - W.low[e]);
        }
    }
}

__global__ void __launch_bounds__(64 * kActWaves)
k_actor_forward(ActorWeights W, int n, const float *__restrict__ obs, const float *__restrict__ noise,
                float *__restrict__ actions, int sample, uint64_t seed, uint64_t counter, float *__restrict__ eps_out)
{
    __shared__ __attribute__((aligned(16))) float lds[kActorLdsFloats];
    ActorHead hd;
    actor_request_weights(hd, W, threadIdx.x);
    actor_forward_tile(W, hd, n, blockIdx.x * kActEnvs, obs, noise, actions, sample, seed, counter, eps_out, lds, threadIdx.x,
                       64 * kActWaves);
}

}  // namespace meshenv
