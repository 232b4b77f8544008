// meshenv_state.h -- HBM layout of the vectorised BoudaryEnv state (shared by kernels and the C-ABI host code).
//
// Structure-of-arrays over environments; each environment owns one contiguous ring segment
// [env_off[e], env_off[e] + n0) in the flat ring arrays, so a wavefront stages its ring with fully
// coalesced 16-byte loads.  Domains (the initial polygons) live in a separate table together with
// everything reset() computes for them, so reset is a copy.
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
    int32_t failed;    // failed_num
    int32_t n_new;     // new vertices appended to boundary.vertices this episode
    int32_t counter;   // candidate insertion stamp counter
    int32_t status;    // MESHENV_ST_* bits
    int32_t dom;       // domain index
    double bl;         // current_point_environment.base_length
    double area;       // current_area
    double pad[2];
};

struct alignas(32) EnvCounters {
    unsigned long long steps, valid, sum_n, sum_n_valid;
};

// per-domain constants
struct alignas(32) DomConst {
    double orig_area;  // Boundary2D.poly_area()
    double min_area;   // estimated_area_range[0] ** 2
    double crit_area;  // estimated_area_range[1] ** 2
    double pad;
};

struct Params {
    double radius, max_ref_angle, w0, w1, min_degree, max_degree, same_eps, ray_length;
    int fail_limit, log_cap;
};

struct DevState {
    // ---- domain table
    int n_domains;
    const int32_t *dom_off;   // [D+1]
    const double2 *dom_xy;    // [sum n0]
    double *dom_key;          // [sum n0] reset-time candidate keys
    int32_t *dom_stamp;       // [sum n0] -index, or kNotCand
    const DomConst *dom_const;  // [D]
    float *dom_obs;           // [D][18] first observation
    int32_t *dom_ref;         // [D]
    double *dom_bl;           // [D]
    // ---- environments
    int n_envs;
    const int32_t *env_off;   // [E+1]
    double2 *ring_xy;         // ring coordinates (x, y)
    int32_t *ring_id;         // ring slot -> global vertex id
    double *ring_key;         // cached candidate key per slot
    int32_t *ring_stamp;      // insertion stamp per slot (kNotCand: not a candidate)
    EnvScalars *scal;         // [E]
    EnvCounters *cnt;         // [E]
    float *obs_cache;         // [E][18] observation of the current state
    // ---- logs (generated_meshes / boundary.vertices), optional
    int32_t *log_quads;       // [E][log_cap][4]
    double2 *log_vxy;         // [E][log_cap]
    Params prm;
};

}  // namespace meshenv
