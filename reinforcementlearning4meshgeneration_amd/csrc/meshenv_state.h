// meshenv_state.h -- HBM layout of the vectorised BoudaryEnv state (shared by kernels and the C-ABI host code).
//
// Structure-of-arrays over environments with a UNIFORM ring stride `cap` (max ring length of the batch,
// rounded up to 16): environment e owns slots [e*cap, e*cap + n0) of every ring array.  A wavefront can
// therefore issue all of its loads (scalars, action, ring coords/ids/keys/stamps) in one go, with no
// dependent offset lookup -- one HBM round trip per step -- and stages its ring with coalesced 16-byte loads.
// 288 GB of HBM make the padding irrelevant (65536 envs x 640 slots x 32 B = 1.3 GB).
// Domains (the initial polygons) live in a separate table together with everything reset() computes for
// them, so reset is a copy.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace meshenv {

constexpr int kObsDim = 18;
constexpr int kNotCand = INT32_MIN;  // ring_stamp value of a slot that is not in candidate_vertices

// one 64-byte record per environment (uniform, lane-broadcast load)
struct alignas(64) EnvScalars {
    int32_t n;         // len(updated_boundary.vertices)
    int32_t ref;       // ring index of current_point_environment.reference_point (-1: None)
    int32_t n_elem;    // len(generated_meshes)
    int32_t n_new;     // new vertices appended to boundary.vertices this episode
    int32_t counter;   // candidate insertion stamp counter
    int32_t dom;       // domain index
    // the two fields a rejected action changes, adjacent: such a step writes these 8 bytes, not the record
    int32_t failed;    // failed_num
    int32_t status;    // MESHENV_ST_* bits
    double bl;         // current_point_environment.base_length
    double area;       // current_area
    double ct, st;     // cos / sin of the action frame angle 2*pi - atan2(right - ref) (D:69-73), per state
};

// Work counters of one env (roofline accounting).  The ring length n only changes on a valid extraction or a reset, so
// the sum of n over steps is kept lazily: sum_n covers the steps before last_change, the host adds
// n_current * (steps_done - last_change).  Rejected actions (88 % of the steps) do not touch the record.
struct alignas(32) EnvCounters {
    unsigned long long last_change;  // index of the first step the current ring length applies to
    unsigned long long valid, sum_n, sum_n_valid;
};

// per-domain constants + the scalars reset() derives
struct alignas(64) DomConst {
    double orig_area;  // Boundary2D.poly_area()
    double min_area;   // estimated_area_range[0] ** 2
    double crit_area;  // estimated_area_range[1] ** 2
    double bl;         // base_length of the first observation
    double ct, st;     // action frame of the first observation
    int32_t off;       // first slot in the dom_* arrays
    int32_t n0;        // ring length
    int32_t ref;       // first reference vertex
    int32_t pad;
};

struct Params {
    double radius, max_ref_angle, w0, w1, min_degree, max_degree, same_eps, ray_length;
    int fail_limit, log_cap;
};

// The reference's constants (rl/boundary_env.py:26-56, general/mesh.py:26-28, components.py:721-722).  Handles created
// with exactly these values run kernel instantiations in which they are literals (kDefaultParams): eight doubles leave
// the scalar registers and x / radius becomes an exact multiplication.
constexpr double kDefPi = 3.141592653589793;
constexpr double kDefRadius = 4.0, kDefMaxRefAngle = kDefPi * 0.972, kDefKeyLambda = 0.618;
constexpr double kDefMinDegree = 0.01 * kDefPi, kDefMaxDegree = 0.99 * kDefPi, kDefSameEps = 0.001, kDefRayLength = 10000.0;

__host__ __device__ inline void apply_default_params(Params &p)
{
    p.radius = kDefRadius; p.max_ref_angle = kDefMaxRefAngle; p.w0 = kDefKeyLambda; p.w1 = 1 - kDefKeyLambda;
    p.min_degree = kDefMinDegree; p.max_degree = kDefMaxDegree; p.same_eps = kDefSameEps; p.ray_length = kDefRayLength;
}

constexpr int kNewBit = 0x40000000;  // ring_id of the k-th vertex created this episode = kNewBit | k

struct LastEpisode {
    int32_t n_elem, n_new;    // elements / created vertices of the archived episode
    int32_t flags;            // bit 0: is_complete (ring <= 5), bit 1: log overflow
    int32_t episodes;         // episodes archived so far
};

// Pointers only the rare paths need (reset, element / vertex logging, the quality report).  They live in device memory
// behind DevState::cold and are fetched where they are used: as kernel arguments they would sit in -- or be spilled from --
// the scalar registers of every wave of every step.
struct DevCold {
    const double2 *dom_xy;    // [sum n0]
    double *dom_key;          // [sum n0] reset-time candidate keys
    int32_t *dom_stamp;       // [sum n0] -index, or kNotCand
    float *dom_obs;           // [D][18] first observation
    // ---- logs (generated_meshes / boundary.vertices), optional
    // Two halves per env: the running episode writes half (status >> 4) & 1; a reset that ends an episode with
    // elements flips the bit, so the finished mesh (what the reference's eval callback reads from
    // env.generated_meshes before it resets, CustomizeCallback.py:131-133) stays readable under auto-reset.
    int32_t *log_quads;       // [E][2][log_cap][4]
    double2 *log_vxy;         // [E][2][log_cap]
    LastEpisode *last_ep;     // [E] extent of the archived half
    int64_t pad;
};

struct DevState {
    // ---- domain table
    int n_domains;
    DomConst *dom;            // [D]
    const DevCold *cold;      // device copy of the rarely used pointers
    // ---- environments
    int n_envs;
    int cap;                  // ring stride
    double2 *ring_xy;         // [E][cap] ring coordinates (x, y)
    int32_t *ring_id;         // [E][cap] ring slot -> global vertex id
    double *ring_key;         // [E][cap] cached candidate key per slot
    int32_t *ring_stamp;      // [E][cap] insertion stamp per slot (kNotCand: not a candidate)
    EnvScalars *scal;         // [E]
    EnvCounters *cnt;         // [E]
    float *obs_cache;         // [E][18] observation of the current state
    float *msg;               // optional [E][21] packed (obs | reward | done | complete) float32 output, NULL = off
#ifdef MESHENV_STAMPS
    unsigned long long *dbg;  // [E][16] diagnostic build: in-kernel timeline stamps
#endif
    Params prm;
};

// The cold pointers, loaded at the point of use (the empty asm keeps the compiler from hoisting the loads into the
// kernel prologue, where they would occupy scalar registers on every path).
__device__ __forceinline__ DevCold load_cold(const DevState &S)
{
    const DevCold *p = S.cold;
    asm volatile("" : "+s"(p));
    return *p;
}

}  // namespace meshenv
