// meshenv_kernels.h -- wavefront-per-environment HIP kernels of the BoudaryEnv hot path (gfx950 / CDNA4).
//
// Execution model: one 64-lane wavefront (= one 64-thread workgroup) owns one environment for the whole
// launch.
//   * HBM -> LDS: scalars, action and the ring arrays (coords, ids, candidate keys, stamps) are requested in
//     one burst at kernel entry (uniform ring stride, no dependent offset lookup), 16 B per lane, coalesced.
//   * O(n) passes (point-in-polygon, boundary intersection scan, observation scan, boundary-quality scan,
//     candidate selection) run one ring vertex per lane and are combined with ballots and DPP scans that
//     carry the reference's sequential first-wins order as (value, traversal order) pairs.
//   * O(1) geometry with several independent atan2/sincos evaluations (quad corner angles, candidate keys,
//     observation window, action frame) is batched into "job lanes": each stage evaluates all of its
//     transcendentals in one pass over a few lanes and broadcasts with v_readlane, so the dependent chain of a
//     valid extraction is 5 atan2 stages + 1 sincos stage instead of ~25 serial calls.
//   * wave-uniform control flow (rule type, valid / invalid action) never diverges inside a wave, and a
//     single-wave workgroup makes wave_sync() a wave-local LDS fence.
// No MFMA: this is branchy fp64 geometry, not a contraction.
#pragma once

#include "meshenv_geom.h"
#include "meshenv_libm.h"
#include "meshenv_state.h"

namespace meshenv {

constexpr int kStNoReference = 1;
constexpr int kStLogOverflow = 2;
// internal memo bits: the rule -1 / rule +1 quad of the CURRENT state is known to be rejected (quad validity and the
// boundary intersection test are pure functions of the state, so a repeated attempt needs no geometry at all);
// cleared whenever the state changes (extraction, reset)
constexpr int kStRm1Bad = 4, kStRp1Bad = 8;
constexpr int kNvMeta = 8;   // int32 words of move() bookkeeping per env: episode counter | last_not_valid_points: first id, last id, length, its episode
constexpr int kStLogHalf = 16;  // which half of the element / vertex log the running episode writes

// scratch area appended to the LDS ring arrays
#ifdef MESHENV_STAMPS
#define MESHENV_STAMP(c, k) do { if ((c).lane == 0) (c).sc->stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MESHENV_STAMP(c, k) do { } while (0)
#endif

#ifdef MESHENV_SPEC_STATS
// diagnostic build only (tools/spec_stats.py): event counts and per-env timestamps (100 MHz ticks) of the last launch
__device__ unsigned long long g_spec_count[16];
__device__ unsigned long long g_spec_time[65536 * 12];
#if MESHENV_SPEC_STATS == 2   // counts (global atomics: they distort the timeline, so timestamps come from a build of their own)
#define SPEC_COUNT(k) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_spec_count[k], 1ULL); } while (0)
#else
#define SPEC_COUNT(k) do { } while (0)
#endif
#define SPEC_TIME(env, k) do { if ((threadIdx.x & 63) == 0 && (env) < 65536) g_spec_time[(size_t)(env) * 12 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SPEC_NOW() __builtin_amdgcn_s_memrealtime()
#define SPEC_TIME_AT(env, k, t) do { if ((threadIdx.x & 63) == 0 && (env) < 65536) g_spec_time[(size_t)(env) * 12 + (k)] = (t); } while (0)
#else
#define SPEC_NOW() 0ULL
#define SPEC_TIME_AT(env, k, t) do { (void)(t); } while (0)
#define SPEC_COUNT(k) do { } while (0)
#define SPEC_TIME(env, k) do { } while (0)
#endif

struct Scratch {
    double2 q[4];       // quad vertices
    double ang[4];      // quad corner angles
    double tmp[16];     // job-lane results handed between stages
    double tmp2[8];     // quad edge / diagonal lengths
    float robs[18];     // observation rows before the final float32 rounding
    int ipad[2];
    unsigned long long red[6];   // first-wins minima of the observation scan (find_next_state, LDS atomics)
#ifdef MESHENV_STAMPS
    unsigned long long stamps[16];
#endif
};

struct Ctx {
    // LDS views
    double2 *xy;
    double *key;
    int32_t *stamp;
    int32_t *id;
    double *ang_ord;   // observation scan: clockwise angle per traversal position
    int32_t *list;     // compaction list of the boundary intersection scan
    Scratch *sc;
    // wave-uniform
    int lane, env;
    size_t base;  // env * cap
    int n, ref, n_elem, failed, n_new, counter, status, dom;
    double bl, area, ct, st;
    bool ring_dirty;
    bool tie = false;  // angles on a 1e-4 rounding boundary take glibc's atan2 (move() / smoothing kernels; geom.h, atan2_x_nc)
    // lane-private
    float obs;  // lanes 0..17: current observation
};

__device__ __forceinline__ P2 ldp(const Ctx &c, int i)
{
    const double2 v = c.xy[i];
    return mkp(v.x, v.y);
}

// Fence between LDS phases of ONE wavefront (lanes hand data to each other through LDS).  A wave's LDS
// instructions execute in issue order, so no hardware wait is needed: this only stops the compiler from moving
// or caching LDS accesses across the hand-over.  (Workgroup barriers are spelled __syncthreads() and only used by
// k_step_group between its check and update phases.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// bounded spin on an LDS word another wavefront of the workgroup sets (all waves of a workgroup are co-resident, so the
// setter always runs; the bound only keeps a logic error from hanging the GPU)
__device__ __forceinline__ int spin_until_nonzero(volatile int *w)
{
    int v = *w;
    for (int it = 0; v == 0 && it < (1 << 22); it++) {
        __builtin_amdgcn_s_sleep(1);
        v = *w;
    }
    return v;
}

// Python list index wrap for i in [-n, 2n)
// i mod n for i in [-n, 2n): as unsigned numbers exactly one of i, i + n, i - n lies in [0, n) and it is the smallest
// (the other two are "negative", i.e. huge, or >= n) -- two adds and one v_min3_u32 instead of two compares, two adds and
// two selects.
__device__ __forceinline__ int wrapi(int i, int n)
{
    const unsigned a = (unsigned)i, b = (unsigned)(i + n), c = (unsigned)(i - n);
    const unsigned m = a < b ? a : b;
    return (int)(m < c ? m : c);
}

__host__ __device__ __forceinline__ size_t lds_bytes_for(int cap)
{
    return ((size_t)cap * (sizeof(double2) + 2 * sizeof(double) + 3 * sizeof(int32_t)) + sizeof(Scratch) + 64 + 15) / 16 * 16;
}

__device__ __forceinline__ void carve_lds(Ctx &c, void *smem, int cap)
{
    // cap is a multiple of 16, so every array stays 16-byte aligned
    c.xy = (double2 *)smem;
    c.key = (double *)(c.xy + cap);
    c.ang_ord = c.key + cap;
    c.stamp = (int32_t *)(c.ang_ord + cap);
    c.id = c.stamp + cap;
    c.list = c.id + cap;
    c.sc = (Scratch *)(c.list + cap);
}

// ------------------------------------------------------------------------------------------ load / store

// What load_env requests from HBM, held in registers until it is written to LDS (k_step_spec puts its one workgroup barrier
// between the two halves, so that the barrier overlaps the memory round trip).
struct EnvLoad {
    EnvScalars s;
    double2 v_xy, w_xy;
    double v_key, w_key;
    int v_id, v_st, w_id, w_st;
    float obs;
};

// (with_keys = false: the candidate keys and stamps stay in HBM -- only a step that extracts an element reads them (the
// candidate patch of env_apply and the selection that follows), see load_keys)
// (lane_in: the T-step closed-loop kernel hands its per-iteration opaque lane id down; -1 = the hardware's)
__device__ __forceinline__ EnvLoad load_env_issue(const DevState &S, int env, const bool with_keys = true, const int lane_in = -1)
{
    EnvLoad L;
    const int lane = lane_in >= 0 ? lane_in : lane_id();
    const size_t base = (size_t)env * S.cap;
    // everything below is independent: one HBM round trip
    L.s = S.scal[env];
    const int first = S.cap < 64 ? S.cap : 64;
    // per-env base pointers are wave-uniform (SGPR pairs); the lane index stays a 32-bit offset
    const double2 *gxy = S.ring_xy + base;
    const int32_t *gid = S.ring_id + base;
    const double *gkey = S.ring_key + base;
    const int32_t *gst = S.ring_stamp + base;
    const unsigned ul = (unsigned)lane;
    L.v_xy = make_double2(0, 0); L.w_xy = make_double2(0, 0);
    L.v_key = 0; L.w_key = 0;
    L.v_id = 0; L.v_st = kNotCand; L.w_id = 0; L.w_st = kNotCand;
    if (lane < first) {
        L.v_xy = gxy[ul];
        L.v_id = gid[ul];
        if (with_keys) {
            L.v_key = gkey[ul];
            L.v_st = gst[ul];
        }
    }
    // rings longer than 64: the second chunk is requested in the same burst, up to the ring STRIDE (known from the
    // kernel arguments) rather than the ring length (known only once the record has arrived) -- no second round trip
    const bool second = S.cap > 64 && (int)(64u + ul) < S.cap;
    if (second) {
        L.w_xy = gxy[64u + ul];
        L.w_id = gid[64u + ul];
        if (with_keys) {
            L.w_key = gkey[64u + ul];
            L.w_st = gst[64u + ul];
        }
    }
    L.obs = lane < kObsDim ? S.obs_cache[(size_t)env * kObsDim + lane] : 0.0f;
    return L;
}

// (lds_cap > 0: the ring slots this env's LDS region holds, when it is smaller than the HBM stride S.cap -- the CU-group
// kernel packs the rings of mixed domains by their own lengths, see step_group_body; 0 = S.cap)
__device__ __forceinline__ void load_env_commit(Ctx &c, const DevState &S, int env, const EnvLoad &L, const bool with_keys = true,
                                                const int lane_in = -1, const int lds_cap = 0)
{
    const int lane = lane_in >= 0 ? lane_in : lane_id();
    c.lane = lane;
    c.env = env;
    c.base = (size_t)env * S.cap;
    const EnvScalars &s = L.s;
    const int lcap = lds_cap > 0 ? lds_cap : S.cap;
    const int first = lcap < 64 ? lcap : 64;
    const double2 *gxy = S.ring_xy + c.base;
    const int32_t *gid = S.ring_id + c.base;
    const double *gkey = S.ring_key + c.base;
    const int32_t *gst = S.ring_stamp + c.base;
    const unsigned ul = (unsigned)lane;
    const bool second = lcap > 64 && (int)(64u + ul) < lcap;
    c.obs = L.obs;
    c.n = uniform_i32(s.n); c.ref = uniform_i32(s.ref); c.n_elem = uniform_i32(s.n_elem);
    c.failed = uniform_i32(s.failed); c.n_new = uniform_i32(s.n_new); c.counter = uniform_i32(s.counter);
    c.status = uniform_i32(s.status); c.dom = uniform_i32(s.dom);
    c.bl = uniform_f64(s.bl); c.area = uniform_f64(s.area); c.ct = uniform_f64(s.ct); c.st = uniform_f64(s.st);
    c.ring_dirty = false;
    if (lane < first) {
        c.xy[lane] = L.v_xy;
        c.id[lane] = L.v_id;
        if (with_keys) {
            c.key[lane] = L.v_key;
            c.stamp[lane] = L.v_st;
        }
    }
    if (second) {
        c.xy[64u + ul] = L.w_xy;
        c.id[64u + ul] = L.w_id;
        if (with_keys) {
            c.key[64u + ul] = L.w_key;
            c.stamp[64u + ul] = L.w_st;
        }
    }
    for (unsigned i = 128u + ul; i < (unsigned)c.n; i += 64u) {
        c.xy[i] = gxy[i];
        c.id[i] = gid[i];
        if (with_keys) {
            c.key[i] = gkey[i];
            c.stamp[i] = gst[i];
        }
    }
    wave_sync();
}

__device__ __forceinline__ void load_env(Ctx &c, const DevState &S, int env, const bool with_keys = true, const int lane_in = -1,
                                         const int lds_cap = 0)
{
    const EnvLoad L = load_env_issue(S, env, with_keys, lane_in);
    load_env_commit(c, S, env, L, with_keys, lane_in, lds_cap);
}

// the candidate keys and stamps of a ring staged without them (load_env(.., false)), before an extraction
__device__ __forceinline__ void load_keys(Ctx &c, const DevState &S)
{
    const double *gkey = S.ring_key + c.base;
    const int32_t *gst = S.ring_stamp + c.base;
    for (unsigned i = (unsigned)c.lane; i < (unsigned)c.n; i += 64u) {
        c.key[i] = gkey[i];
        c.stamp[i] = gst[i];
    }
    wave_sync();
}

// ring arrays LDS -> HBM (slots [0, n))
__device__ __forceinline__ void store_ring(const Ctx &c, const DevState &S)
{
    double2 *gxy = S.ring_xy + c.base;
    int32_t *gid = S.ring_id + c.base;
    double *gkey = S.ring_key + c.base;
    int32_t *gst = S.ring_stamp + c.base;
    for (unsigned i = (unsigned)c.lane; i < (unsigned)c.n; i += 64u) {
        gxy[i] = c.xy[i];
        gid[i] = c.id[i];
        gkey[i] = c.key[i];
        gst[i] = c.stamp[i];
    }
}

// (ring_elsewhere: the ring arrays of this step are written back by the helper wavefront of the CU-group kernel)
__device__ __forceinline__ void store_env(Ctx &c, const DevState &S, bool ring_elsewhere = false)
{
    wave_sync();
    if (c.ring_dirty) {
        if (!ring_elsewhere) store_ring(c, S);
        if (c.lane < kObsDim) S.obs_cache[(size_t)c.env * kObsDim + c.lane] = c.obs;
    }
    if (c.lane == 0) {
        if (c.ring_dirty) {
            EnvScalars s;
            s.n = c.n; s.ref = c.ref; s.n_elem = c.n_elem; s.failed = c.failed; s.n_new = c.n_new;
            s.counter = c.counter; s.status = c.status; s.dom = c.dom; s.bl = c.bl; s.area = c.area;
            s.ct = c.ct; s.st = c.st;
            S.scal[c.env] = s;
        } else {  // a rejected action: only failed_num and the status bits moved
            int2 *fs = reinterpret_cast<int2 *>(&S.scal[c.env].failed);
            *fs = make_int2(c.failed, c.status);
        }
    }
}

// ------------------------------------------------------------------------------------------ candidates (a5)

// find_reference_point, M:269-290: head of the list ordered by (key asc, insertion desc)
__device__ __forceinline__ int select_reference(const Ctx &c)
{
    double bk = kInf;
    int bs = kNotCand, bi = -1;
    for (int i = c.lane; i < c.n; i += 64) {
        const int st = c.stamp[i];
        if (st != kNotCand) {
            const double k = c.key[i];
            if (k < bk || (k == bk && st > bs)) { bk = k; bs = st; bi = i; }
        }
    }
    // (key asc, stamp desc): stamps are unique among candidates, so the winner is found in two plain reductions --
    // the smallest key, then the largest stamp among the lanes that hold it -- and located with one ballot
    const double kmin = wave_min_f64(bk);
    if (!(kmin < kInf)) return -1;
    const int smax = wave_max_i32(bk == kmin ? bs : kNotCand);
    const unsigned long long hit = __ballot(bk == kmin && bs == smax);
    return lane_i32(bi, (int)__ffsll((long long)hit) - 1);
}

// find_reference_point with not_valid_points (M:284-288, the move() API): the first list entry that is not within
// 0.001 of a listed point (is_vertex_inside_list, M:428-433).  nv = the env's list (LDS), n_nv its length.
__device__ __forceinline__ int select_reference_nv(const Ctx &c, const double2 *nv, int n_nv)
{
    double bk = kInf;
    int bs = kNotCand, bi = -1;
    for (int i = c.lane; i < c.n; i += 64) {
        const int st = c.stamp[i];
        if (st != kNotCand) {
            const P2 v = ldp(c, i);
            bool listed = false;
            for (int k = 0; k < n_nv; k++) listed = listed || dist(mkp(nv[k].x, nv[k].y), v) < 0.001;
            const double k = c.key[i];
            if (!listed && (k < bk || (k == bk && st > bs))) { bk = k; bs = st; bi = i; }
        }
    }
    const double kmin = wave_min_f64(bk);
    if (!(kmin < kInf)) return -1;
    const int smax = wave_max_i32(bk == kmin ? bs : kNotCand);
    const unsigned long long hit = __ballot(bk == kmin && bs == smax);
    return lane_i32(bi, (int)__ffsll((long long)hit) - 1);
}

// the two angles of MeshGeneration.check_boundary_point (M:202-231) for ring slot `index`:
// which = 0 -> cw(v; ring[index+1], ring[index-1]); which = 1 -> cw(v; ring[index+2], ring[index-2])
__device__ __forceinline__ void key_angle_terms(const Ctx &c, int index, int which, double &cc, double &dd)
{
    const int n = c.n, o = which + 1;
    cw_terms(ldp(c, index), ldp(c, wrapi(index + o, n)), ldp(c, wrapi(index - o, n)), cc, dd);
}

// key = degrees(0.618*a0 + 0.382*a1), None when a0 >= 0.972*pi or a0 == 0
__device__ __forceinline__ bool key_from_angles(const Params &p, double a0, double a1, double &out)
{
    if (a0 >= p.max_ref_angle || a0 == 0.0) return false;
    double sum_angle = 0.0;
    sum_angle += a0 * p.w0;
    sum_angle += a1 * p.w1;
    out = sum_angle * (180.0 / kPi);  // math.degrees
    return true;
}

// ------------------------------------------------------------------------------------------ observation (a6) + boundary quality

// Boundary-quality request fused into find_next_state: the O(n) scan of compute_boundary_quality (M:329-382)
// shares the observation scan's pass over the ring, its distances share stage A, its minimum shares the
// reduction stage -- the two are independent chains, so the scheduler overlaps them.
struct BqArgs {
    int mode;    // 0 none (reset), 1 new vertex at ring slot a, 2 kept vertices at ring slots a, b (quad order)
    int a, b;
    double ang0, ang1;  // clockwise boundary angles (job lanes of the quad pass)
    double q_ang0, q_ang2;  // quad corner angles 0 and 2; their sines are evaluated in stage B
    double half01, half23;  // 0.5*e0*e1 and 0.5*e2*e3 of Mesh.compute_area (C:935-950)
    double mesh_area;   // out; current_area is reduced by it before the observation is built (B:200)
    double b_reward;    // out
    bool skip;          // the boundary-quality term is computed elsewhere (CU-group kernel: on a helper wavefront)
};

// find_next_state, B:504-571 -> PointEnvironment.get_neighbors C:1073-1082 + get_radius_points C:1184-1282.
// Updates c.ref, c.bl, c.ct/st (action frame of the new state), c.obs (lanes 0..17), c.status; fills bq outputs.
// (is_static / nv: compile-time constants at every call; the move() API passes PointEnvironment(static=True) and its
// not_valid_points list, C:1213-1218, M:284-288)
// (lr: the move() / smoothing kernels only -- squares and the bisector's sines / cosines as the reference's libm gives
// them, csrc/meshenv_libm.h; the step kernels pass none and compile to what they were)
__device__ __forceinline__ double dist_lr(const LibmRef lr, P2 a, P2 b)
{
    if (lr.tab == nullptr || !lr.exact) return dist(a, b);
    return sqrt_pos(pow2_nc(a.x - b.x) + pow2_nc(a.y - b.y));
}

__device__ __forceinline__ int nf_compact(Ctx &c, bool near, int i, int count);   // (below: survivors of a 64-lane chunk -> c.list)

__device__ __forceinline__ void find_next_state(Ctx &c, const DevState &S, BqArgs &bq, const bool is_static = false,
                                                const double2 *nv = nullptr, int n_nv = 0, const LibmRef lr = LibmRef{nullptr, 0})
{
    const Params &p = S.prm;
    const int lane = c.lane, n = c.n;
    wave_sync();
    const int idx = nv ? select_reference_nv(c, nv, n_nv) : select_reference(c);
    c.ref = idx;
    MESHENV_STAMP(c, 9);
    const bool none = idx < 0;  // the reference returns None here; the reward terms are still needed
    const int idc = none ? 0 : idx;
    const int i_right = wrapi(idc - 1, n), i_left = wrapi(idc + 1, n);
    const double2 ref_v = c.xy[idc], right_v = c.xy[i_right];
    const P2 ref = mkp(uniform_f64(ref_v.x), uniform_f64(ref_v.y));
    const P2 right = mkp(uniform_f64(right_v.x), uniform_f64(right_v.y));
    const double radius = p.radius;

    // ---- stage A.  dist jobs: lanes 0..5 window edge j of base_length, lanes 8..14 boundary-quality
    //      distances, lanes 16..21 reference->neighbour distances.  atan2 jobs: lanes 0..5 the six neighbour
    //      rows (lane 0 -> rotation_angle, lane 3 -> theta), lane 6 the action frame (D:69).
    //   neighbour lane j < 3 : right side i = j     -> vertex idx-1-j
    //   neighbour lane j >= 3: left side  i = j - 3 -> vertex idx+1+(j-3)
    // Index selection is select-only (no lane-divergent branches): every lane evaluates one dist() and one set
    // of atan2 terms, lanes without a job use slot 0 / a dummy argument.
    const int nbr = lane < 3 ? wrapi(idc - 1 - lane, n) : wrapi(idc + 1 + (lane - 3), n);  // valid for lane < 6
    const int nbr16 = (lane - 16) < 3 ? wrapi(idc - 1 - (lane - 16), n) : wrapi(idc + 1 + (lane - 19), n);
    const int bqa = bq.a, bqlo = bq.a < bq.b ? bq.a : bq.b;
    int ia = 0, ib = 0;
    {
        const bool j0 = lane < 6, j16 = lane >= 16 && lane < 22;
        const bool m1 = bq.mode == 1 && !bq.skip, m2 = bq.mode == 2 && !bq.skip;
        const int k = lane - 10;
        const bool jwin = lane >= 10 && lane < (m1 ? 14 : 15) && (m1 || m2);
        const int wbase = m1 ? bqa : bqlo;
        ia = j0 ? wrapi(idc + 2 - lane, n) : ia;
        ib = j0 ? wrapi(idc + 3 - lane, n) : ib;
        ia = j16 ? idc : ia;
        ib = j16 ? nbr16 : ib;
        ia = jwin ? wrapi(wbase - 2 + k, n) : ia;
        ib = jwin ? wrapi(wbase - 1 + k, n) : ib;
        ia = (lane == 8 || lane == 9) && (m1 || m2) ? bqa : ia;
        ib = lane == 8 ? (m1 ? wrapi(bqa + 1, n) : (m2 ? bq.b : ib)) : ib;
        ib = (lane == 9 && m1) ? wrapi(bqa - 1, n) : ib;
    }
    const double dv = dist_lr(lr, ldp(c, ia), ldp(c, ib));
    // atan2 jobs.  lanes 1..5: cw(ref; v, right); lane 0: cw(ref; right, ref + (1, 0)); lane 6: atan2(right - ref);
    // lanes 7..63: the clockwise angle of traversal position ord = lane - 7 of the observation scan (stage C only
    // evaluates atan2 itself for rings longer than 58), so stages A and C share one transcendental pass.
    constexpr int kScanLanes = 57;
    const int ord_a = lane - 7;
    const bool scanjob = lane >= 7 && ord_a < n - 1;
    double jy, jx;
    {
        const int vsel = lane < 6 ? nbr : (scanjob ? wrapi(idc - 1 - ord_a, n) : 0);
        const P2 v = ldp(c, vsel);
        const P2 p1 = lane == 0 ? right : v;
        const P2 p2 = lane == 0 ? mkp(ref.x + 1, ref.y) : right;
        cw_terms(ref, p1, p2, jy, jx);
        jy = lane == 6 ? right.y - ref.y : jy;
        jx = lane == 6 ? right.x - ref.x : jx;
    }
    double jt = 0.0;
    if (lane < 7 || scanjob) jt = atan2_sel(c.tie, jy, jx);
    const double na = cw_finish(jt);
    if (scanjob) c.ang_ord[ord_a] = na;
    // base_length = round(sum of the 6 window edges / 6, 4), summed in the reference's order
    double sum = lane_f64(dv, 0);
    sum += lane_f64(dv, 1);
    sum += lane_f64(dv, 2);
    sum += lane_f64(dv, 3);
    sum += lane_f64(dv, 4);
    sum += lane_f64(dv, 5);
    const double bl = uniform_f64(round4_py(sum / 6));
    const double target_length = uniform_f64(bl * radius);
    const double rot = lane_f64(na, 0);
    const double theta = lane_f64(na, 3);
    const double thd = 2 * kPi - lane_f64(jt, 6);
    const double clipmax = uniform_f64(theta + kPi / 2);
    // boundary-quality distances
    const double bq_d0 = lane_f64(dv, 8), bq_d1 = lane_f64(dv, 9);
    double bq_sum = lane_f64(dv, 10);
    bq_sum += lane_f64(dv, 11);
    bq_sum += lane_f64(dv, 12);
    bq_sum += lane_f64(dv, 13);
    const double bq_sum5 = bq_sum + lane_f64(dv, 14);
    const double dst = uniform_f64(bq_d0 + bq_d1);  // mode 1: d(new, next) + d(new, prev)
    const double ndraw = __hiloint2double(__shfl(__double2hiint(dv), (lane & 7) + 16, 64), __shfl(__double2loint(dv), (lane & 7) + 16, 64));
    MESHENV_STAMP(c, 10);

    // ---- stage B: sincos jobs: lane 0 theta/2, lane 1 rotation_angle, lane 2 action frame,
    //      lanes 3, 4 the two quad corners of Mesh.compute_area
    double sj = 0.0, cj = 1.0;
    if (lane < 5) {  // one exec-masked call; the argument is a select chain
        const double arg = lane == 0 ? theta / 2 : lane == 1 ? rot : lane == 2 ? thd : lane == 3 ? bq.q_ang0 : bq.q_ang2;
        const SinCos sc = sincos_small_nc(arg);
        sj = sc.s;
        cj = sc.c;
    }
    if (lr.tab != nullptr && lane < 2) {
        // theta and the rotation angle are quantised: their sines / cosines from the host libm's table (lane 0: theta / 2)
        const int q = min(max(angle_q(lane == 0 ? theta : rot), 0), kFtQ - 1);
        const double *e = lr.tab + (lane == 0 ? kFtSinCosHalf : kFtSinCosFull) + 2 * q;
        sj = e[0];
        cj = e[1];
    }
    if (bq.mode != 0) {
        bq.mesh_area = uniform_f64(bq.half01 * lane_f64(sj, 3) + bq.half23 * lane_f64(sj, 4));
        c.area -= bq.mesh_area;
    }
    const double area_ratio = c.area / S.dom[c.dom].orig_area;
    // bisector segment ref -> p_s (Vertex.rotate about the origin, C:146-160)
    const double px = target_length * lane_f64(cj, 0), py = target_length * lane_f64(sj, 0);
    const double cr = lane_f64(cj, 1), sr = lane_f64(sj, 1);
    const double qx = (0.0 + cr * px) - sr * py;
    const double qy = (0.0 + sr * px) + cr * py;
    const double ux = uniform_f64((ref.x + qx) - ref.x), uy = uniform_f64((ref.y + qy) - ref.y);  // u = p_s - ref
    MESHENV_STAMP(c, 11);
    wave_sync();

    // ---- stage C: O(n) scan, traversal order ord = 0..n-2 <-> ring index idx-1-ord (C:1239-1267), fused with the
    //      near-vertex scan of compute_boundary_quality (ring index = base + lane)
    const double third = uniform_f64(theta / 3);
    const u64 kSlotInit = ((u64)0x3f800000u << 32) | 0x7fffffffu;  // (1.0f, no vertex)
    u64 k0 = kSlotInit, k1 = kSlotInit, k2 = kSlotInit;
    double rbest = 1.0;
    int rord = 0x7fffffff;
    double m_d = kInf;
    int carry = 0;
    const int bqi = bq.a;
    const int w1 = wrapi(bqi + 1, n), w2 = wrapi(bqi + 2, n), w3 = wrapi(bqi - 1, n), w4 = wrapi(bqi - 2, n);
    const bool bq_scan = bq.mode == 1 && !bq.skip;
    const P2 add_v = ldp(c, bq_scan ? bqi : 0);
    const DistThr dst_thr = dist_thr(dst);
    // Rings of several 64-vertex chunks (d1 / d2 / d3: 120 / 196 / 272 vertices): an EXACT pre-filter, the survivors
    // compacted (c.list, in traversal order), and the scan itself -- its clockwise angle, the fan-slot test, the bisector's
    // intersection quotients -- once over the survivors instead of once per chunk.  What a position can contribute:
    //  * a fan slot needs d(ref, v) < target_length, hence |v.x - ref.x| < target_length (1 + 2^-52);
    //  * a hit of the bisector needs 0 < s < 1 and 0 < h < 1 AS COMPUTED, with h = fl(fl(r + fl(s u)) / w) in the x
    //    coordinate (r = ref.x - v.x, w = b.x - v.x, u = p_s.x - ref.x) in the general and the w.y == 0 branch: then
    //    fl(r + fl(s u)) lies between 0 and w up to a rounding, i.e. ref.x + s u lies in the edge's x-extent widened by
    //    4 ulp of (|r| + |u|), and with 0 < s < 1 it also lies in ref.x +- |u|, |u| <= target_length (1 + 2^-50); in the
    //    w.x == 0 branch s = fl((v.x - ref.x) / u) in (0, 1) says the same of v.x = b.x directly.
    // So a position whose edge has no x in common with the slab ref.x +- W, W = target_length (1 + 1e-9) + 1e-9, changes
    // nothing, whatever its (possibly ill-conditioned) quotients evaluate to; NaN coordinates fail every comparison in
    // both forms.  -DMESHENV_NO_FILTERS builds the unfiltered scan.  (Rings of at most 65 vertices: every position is
    // scanned, in place -- `count` = n - 1 positions, position j IS traversal order j.)
#ifdef MESHENV_NO_FILTERS
    const bool long_ring = false;
#else
    const bool long_ring = n - 1 > 64;
#endif
    int count = n - 1;
    if (long_ring) {
        const double W = target_length * (1 + 1e-9) + 1e-9;
        const double slab_lo = ref.x - W, slab_hi = ref.x + W;
        int m = 0;
        for (int base = 0; base < n - 1; base += 64) {
            const int ord = base + lane;
            const bool in = ord < n - 1;
            const int ii = wrapi(idc - 1 - (in ? ord : 0), n);
            const P2 v = ldp(c, ii), b = ldp(c, wrapi(ii + 1, n));
            const double xlo = v.x < b.x ? v.x : b.x, xhi = v.x < b.x ? b.x : v.x;
            m = nf_compact(c, in && xlo <= slab_hi && xhi >= slab_lo, ord, m);
        }
        count = m;
        wave_sync();
    }
    // the clockwise angles of the scanned positions stage A had no lane for (rings longer than 58), in full 64-lane passes
    // of their own: a 120-vertex ring costs one more transcendental pass, not two
    for (int a0 = long_ring ? 0 : kScanLanes; a0 < count; a0 += 64) {
        const bool in_list = a0 + lane < count;
        const int o = long_ring ? c.list[in_list ? a0 + lane : 0] : a0 + lane;
        const bool in = in_list && o >= kScanLanes;
        double cc, dd;
        cw_terms(ref, ldp(c, wrapi(idc - 1 - (in ? o : 0), n)), right, cc, dd);
#ifdef MESHENV_NO_FILTERS
        const double a = cw_sel(c.tie, cc, dd);
#else
        bool need_exact;   // all of these angles are quantised: fast form, exact for the pass when a lane sits in a guard band
        double a = cw_fast(cc, dd, need_exact);
        if (__ballot(need_exact && in) != 0ULL) a = cw_sel(c.tie, cc, dd);
#endif
        if (in) c.ang_ord[o] = a;
    }
    wave_sync();
    for (int base = 0; base < n; base += 64) {
        // (2) boundary-quality scan (mode 1) of ring index base + lane: added(i) = near(i) && !added(i-1), M:355-357
        if (bq_scan) {
            const int i = base + lane;
            bool near = false;
            if (i < n && !(i == bqi || i == w1 || i == w2 || i == w3 || i == w4)) near = dist_lt(add_v, ldp(c, i), dst_thr);
            const unsigned long long m = __ballot(near);
            bool added = false;
            if (near) {
                const unsigned long long zeros_below = ~m & ((1ULL << lane) - 1ULL);
                if (zeros_below != 0ULL) {
                    const int pz = 63 - __clzll((long long)zeros_below);
                    added = ((lane - pz - 1) & 1) == 0;
                } else {
                    added = (lane & 1) == carry;
                }
            }
            if (added) {
                const double d = seg_point_distance(ldp(c, wrapi(i + 1, n)), ldp(c, i), add_v);
                m_d = d < m_d ? d : m_d;
            }
            carry = (int)((__ballot(added) >> 63) & 1ULL);
        }
        if (base >= count) continue;   // (wave-uniform: a pre-filtered ring has fewer positions to scan than vertices)
        // (1) observation scan of scanned position base + lane
        const bool in_range = base + lane < count;
        const int ord = long_ring ? c.list[in_range ? base + lane : 0] : base + lane;
        const int ii = wrapi(idc - 1 - (in_range ? ord : 0), n);
        const P2 v = ldp(c, ii);
        const double angle = in_range ? c.ang_ord[ord] : 0.0;
        const bool live = in_range && ii != i_right && ii != i_left && angle != 0.0;
        if (live) {
            const double d = dist_lr(lr, ref, v);
            const double kf = angle / third;
            if (kf < 3.0 && d < target_length) {
                const int k = (int)kf;
                const float cnd = (float)((d / radius) / bl);
                const u64 key = ((u64)__float_as_uint(cnd) << 32) | (unsigned)ord;  // keys order by (value, traversal order): '<' keeps the first
                if (k == 0) k0 = key < k0 ? key : k0;
                else if (k == 1) k1 = key < k1 ? key : k1;
                else k2 = key < k2 ? key : k2;
            }
            // Segment(ref, p_s).intersection_vertex(Segment(ring[i], ring[i+1])), C:649-668.  The three branches of
            // the reference (w.y == 0 / w.x == 0 / general) are one instruction stream with selected operands, so an
            // axis-aligned domain does not pay for all three in turn:  s = num / den,  h = (a + s * b) / cden
            const P2 b = ldp(c, wrapi(ii + 1, n));
            const double wx = b.x - v.x, wy = b.y - v.y;
            const bool hy = wy == 0.0, hx = !hy && wx == 0.0;
            const double rx = ref.x - v.x, ry = ref.y - v.y;
            const double num_g = rx / wx - ry / wy, den_g = uy / wy - ux / wx;
            const double num = hy ? (v.y - ref.y) : (hx ? (v.x - ref.x) : num_g);
            const double den = hy ? uy : (hx ? ux : den_g);
            const bool have = !((hy && uy == 0.0) || (hx && ux == 0.0));
            const double s = num / den;
            const double h = hx ? (ry + s * uy) / wy : (rx + s * ux) / wx;
            if (have && 0.0 < s && s < 1.0 && 0.0 < h && h < 1.0) {
                const double val = (dist_lr(lr, ref, mkp(ref.x + s * ux, ref.y + s * uy)) / radius) / bl;
                if (val < rbest) { rbest = val; rord = ord; }
            }
        }
    }
    const bool ang_all = !long_ring;   // c.ang_ord holds the angle of every traversal position (false: of the scanned ones only)
    MESHENV_STAMP(c, 12);
    // ---- reductions: first-wins minima as packed (value bits, order) keys, one interleaved DPP scan
    u64 rk = (u64)__double_as_longlong(rbest), mk = (u64)__double_as_longlong(m_d);
    const u64 my_rk = rk;
    // Only the few lanes whose vertex lies inside the fan (or whose edge the bisector hits) hold anything but the identity:
    // they take the minima with LDS atomics -- five DPP reductions of 64-bit keys are ~170 instructions on the one
    // wavefront whose issue slot the whole extraction runs on, this is ~40 and three LDS round trips.
    u64 *red = c.sc->red;
    const u64 rk_init = (u64)__double_as_longlong(1.0), mk_init = (u64)__double_as_longlong(kInf);
    if (lane < 6) red[lane] = lane < 3 ? kSlotInit : (lane == 3 ? rk_init : (lane == 4 ? mk_init : ~0ULL));
    wave_sync();
    if (k0 != kSlotInit) atomicMin(&red[0], k0);
    if (k1 != kSlotInit) atomicMin(&red[1], k1);
    if (k2 != kSlotInit) atomicMin(&red[2], k2);
    if (rk != rk_init) atomicMin(&red[3], rk);
    if (mk != mk_init) atomicMin(&red[4], mk);
    wave_sync();
    k0 = red[0]; k1 = red[1]; k2 = red[2]; rk = red[3]; mk = red[4];
    if (my_rk == rk && rk != rk_init) atomicMin(&red[5], (u64)(unsigned)rord);   // earliest among equal hits
    wave_sync();
    const unsigned ro = (unsigned)red[5];
    rbest = __longlong_as_double((long long)rk);
    m_d = __longlong_as_double((long long)mk);
    const float s0 = __uint_as_float((unsigned)(k0 >> 32)), s1 = __uint_as_float((unsigned)(k1 >> 32)), s2 = __uint_as_float((unsigned)(k2 >> 32));
    const int o0 = (int)(unsigned)k0, o1 = (int)(unsigned)k1, o2 = (int)(unsigned)k2;
    const bool use_hit = rbest < 1.0 && (float)rbest < s1;
    MESHENV_STAMP(c, 13);
    wave_sync();

    // ---- observation rows
    if (lane < 6) {
        const int row = lane < 3 ? lane : 8 - (lane - 3);
        float v1;
        if (lane == 0) v1 = is_static ? 0.0f : (float)area_ratio;
        else if (lane == 3) v1 = (float)theta;
        else if (lane < 3) v1 = (float)(na < kPi ? na : fmax(na, 1.5 * kPi) - 2 * kPi);
        else v1 = (float)fmin(na, clipmax);
        c.sc->robs[2 * row] = (float)((ndraw / radius) / bl);
        c.sc->robs[2 * row + 1] = v1;
    } else if (lane >= 8 && lane < 11) {
        // fan rows: slot winners, or the three vertices around the edge the bisector hits (C:1269-1279)
        const int j = lane - 8;
        const float sw = j == 0 ? s0 : (j == 1 ? s1 : s2);
        const int ow = j == 0 ? o0 : (j == 1 ? o1 : o2);
        float fd = 1.0f;
        float fa = (float)fmin(((2 * j + 1) * theta) / 6, clipmax);  // default [1, clip((2j+1)*theta/6)]
        if (use_hit) {
            const int ov = (int)ro + 1 - j;  // ring[_i - 1 + j] in traversal order
            const P2 hv = ldp(c, wrapi(idc - 1 - ov, n));
            fd = (float)((dist_lr(lr, ref, hv) / radius) / bl);
            // (pre-filtered scan: the two neighbours of the hit edge's first vertex need not have been scanned -- positions
            // -1 and n - 1 are the reference vertex's neighbours and itself, whose angles the reference takes as well)
            if (ang_all) {
                fa = (float)c.ang_ord[ov];
            } else {
                double cc, dd;
                cw_terms(ref, hv, right, cc, dd);
                fa = (float)cw_sel(c.tie, cc, dd);
            }
        } else if (sw < 1.0f) {
            fd = sw;
            fa = (float)fmin(c.ang_ord[ow], clipmax);
        }
        c.sc->robs[6 + 2 * j] = fd;
        c.sc->robs[7 + 2 * j] = fa;
    }
    // ---- boundary quality (M:329-382 / M:392-426)
    if (bq.mode != 0 && !bq.skip) {
        double amin = 1e300;
        bool have = false;
        if (bq.ang0 < kPi / 3) { amin = bq.ang0; have = true; }
        if (bq.ang1 < kPi / 3) { amin = bq.ang1 < amin ? bq.ang1 : amin; have = true; }
        const double q1 = have ? 3 * amin / kPi : 1.0;
        if (bq.mode == 1) {
            const double targt_len = dst / 2;
            const double mean_dist = bq_sum / 4;
            const double smoothness = (targt_len < mean_dist ? targt_len : mean_dist) / (targt_len > mean_dist ? targt_len : mean_dist);
            double q2 = 1.0;
            if (m_d < 1e299) q2 = (m_d < 0.5 * dst) ? m_d / (0.5 * dst) : 1.0;
            bq.b_reward = cbrt(smoothness * q1 * q2);  // math.pow(x, 1/3)
        } else {
            const double targt_len = bq_d0;
            const double mean_dist = bq_sum5 / 5;
            const double smoothness = (targt_len < mean_dist ? targt_len : mean_dist) / (targt_len > mean_dist ? targt_len : mean_dist);
            bq.b_reward = sqrt(q1 * smoothness);  // math.pow(angle_quality * smoothness, 1/2)
        }
    }
    wave_sync();
    if (none) {
        c.status |= kStNoReference;
        c.obs = 0.0f;
    } else {
        c.status &= ~kStNoReference;
        c.bl = bl;
        c.ct = lane_f64(cj, 2);
        c.st = lane_f64(sj, 2);
        if (lane < kObsDim) c.obs = round4_npf(c.sc->robs[lane]);
    }
}

// ------------------------------------------------------------------------------------------ ring passes of the checks (a7, a8, a10 filter)

// is_point_inside_area -> calculate_crossing_segments, M:539-546, 47-102.  One ring edge per lane, ballot parity.
// Orientation tests only use sign and zero-ness of round(dy, 4), so the scaled roundings are used.
__device__ __forceinline__ int nf_compact(Ctx &c, bool near, int i, int count);

// one edge (ring[ic], ring[ic - 1]) of the crossing count: M:47-102 for that edge
__device__ __forceinline__ bool edge_counted(const Ctx &c, P2 p, P2 far, int ic, bool in)
{
    const int n = c.n;
    const int im1 = wrapi(ic - 1, n);
    const P2 vi = ldp(c, ic), vm = ldp(c, im1);
    const bool np_i = (c.id[ic] & kNewBit) != 0, np_m = (c.id[im1] & kNewBit) != 0;
    const double dy = vi.y - vm.y;
    const double orientation = (np_i || np_m) ? round4_np_scaled(dy) : round4_py_scaled(dy);
    bool counted = false;
    // is_cross is a pure conjunction; the ray-side test is the selective one, so it goes first
    if (in && orientation != 0.0 && straddle(p, far, vi, vm) && straddle(vi, vm, p, far)) {
        const bool on_i = round4_np_scaled(vi.y - p.y) == 0.0, on_m = round4_np_scaled(vm.y - p.y) == 0.0;
        // neighbour edge: the next one when the ray passes through vertex i, the previous one when through i-1
        const int ia = on_i ? wrapi(ic + 1, n) : im1, ib = on_i ? ic : wrapi(ic - 2, n);
        const double dyo = ldp(c, ia).y - ldp(c, ib).y;
        const double other = ((c.id[ia] & kNewBit) != 0 || (c.id[ib] & kNewBit) != 0) ? round4_np_scaled(dyo) : round4_py_scaled(dyo);
        const bool keep = other != 0.0 && !(other * orientation < 0.0);
        counted = on_i ? (keep && orientation < 0.0) : (on_m ? (keep && !(orientation < 0.0)) : true);
    }
    return counted;
}

__device__ __forceinline__ bool point_inside(Ctx &c, const Params &prm, P2 p)
{
    const int n = c.n;
    const P2 far = mkp(prm.ray_length, p.y);
    int parity = 0;
#ifndef MESHENV_NO_FILTERS
    if (n > 64) {
        // Rings of several 64-vertex chunks: an exact pre-filter per edge, the survivors compacted (c.list), and the
        // crossing test itself once over the survivors instead of once per chunk.  An edge whose endpoints lie strictly on
        // the same side of the ray's line cannot be counted: with a = vi.y - p.y, b = vm.y - p.y of equal sign and
        // |a| > 2e-3 |vi.x - p.x|, |b| > 2e-3 |vm.x - p.x| (and > 1e-9), both collinearity terms of
        // straddle(p, far, vi, vm) fail sin_rounds_to_zero's first test (|c| > 1e-3 |d|: c = -a (L - p.x), d = dx (L - p.x))
        // and its cross products -(L - p.x) a and -(L - p.x) b have the same sign and cannot underflow: straddle is False,
        // is_cross is False, the edge adds nothing.  Identical results by construction (the filter only drops edges the full
        // test rejects); -DMESHENV_NO_FILTERS builds the unfiltered form.
        int count = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + c.lane;
            const bool in = i < n;
            const int ic = in ? i : 0;
            const P2 vi = ldp(c, ic), vm = ldp(c, wrapi(ic - 1, n));
            const double a = vi.y - p.y, b = vm.y - p.y;
            const double ma = fmax(2e-3 * fabs(vi.x - p.x), 1e-9), mb = fmax(2e-3 * fabs(vm.x - p.x), 1e-9);
            const bool clear = (a > ma && b > mb) || (a < -ma && b < -mb);
            count = nf_compact(c, in && !clear, i, count);
        }
        wave_sync();
        for (int j0 = 0; j0 < count; j0 += 64) {
            const int j = j0 + c.lane;
            const bool in = j < count;
            const int ic = c.list[in ? j : 0];
            parity ^= __popcll(__ballot(edge_counted(c, p, far, ic, in))) & 1;
        }
        wave_sync();
        return parity != 0;
    }
#endif
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + c.lane;
        const bool in = i < n;
        parity ^= __popcll(__ballot(edge_counted(c, p, far, in ? i : 0, in))) & 1;
    }
    return parity != 0;
}

// distance filter of check_intersection_with_boundary (M:511-513): ring vertices outside the quad that are
// nearer to the reference vertex than the farthest quad vertex
struct NearFilter {
    P2 ref;
    DistThr max_dist;
    int mp0, mp1, mp2, mp3;  // ring slots of the quad vertices (-1: not on the ring)
};

__device__ __forceinline__ bool nf_near(const NearFilter &f, int i, P2 v)
{
    return !(i == f.mp0 || i == f.mp1 || i == f.mp2 || i == f.mp3) && dist_lt(f.ref, v, f.max_dist);
}

// survivors of one 64-vertex chunk go to the LDS list, in ring order
__device__ __forceinline__ int nf_compact(Ctx &c, bool near, int i, int count)
{
    const unsigned long long m = __ballot(near);
    if (near) c.list[count + __popcll(m & ((1ULL << c.lane) - 1ULL))] = i;
    return count + __popcll(m);
}

// One pass over the ring vertices: the distance filter (list in LDS, returns its length) and, for a rule-0
// candidate point p (same_eps > 0), find_same_point (B:599-602; only existence is used).
__device__ __forceinline__ int near_filter_pass(Ctx &c, const NearFilter &f, P2 p, double same_eps, bool &same)
{
    int count = 0;
    bool any_same = false;
    const DistThr same_thr = dist_thr(same_eps);
    for (int i0 = 0; i0 < c.n; i0 += 64) {
        const int i = i0 + c.lane;
        const bool in = i < c.n;
        const P2 v = ldp(c, in ? i : 0);
        if (same_eps > 0.0) any_same = any_same || (__ballot(in && dist_lt(v, p, same_thr)) != 0ULL);
        count = nf_compact(c, in && nf_near(f, i, v), i, count);
    }
    same = any_same;
    return count;
}

// ------------------------------------------------------------------------------------------ quad pass (a9 + speculative a5/a12 jobs)

// Description of the ring AFTER the extraction, addressed without touching LDS (the quad may still be
// rejected): new-vertex rule: slot `lo` holds new_point, everything else unchanged; removal rules: slots
// lo < hi are deleted and the ring is compacted.
struct VRing {
    int n;        // length after the update
    int lo, hi;   // removal: deleted slots (old indices); new vertex: lo = slot of the new vertex, hi = -1
    bool is_new;
    P2 new_point;
};

__device__ __forceinline__ P2 vr_at(const Ctx &c, const VRing &r, int j)
{
    // select-only: the slot index is remapped for the removal rules, the new vertex is patched in afterwards
    int o = j + ((!r.is_new && j >= r.lo) ? 1 : 0);
    o += (!r.is_new && o >= r.hi) ? 1 : 0;
    const P2 p = ldp(c, o);
    const bool patch = r.is_new && j == r.lo;
    return mkp(patch ? r.new_point.x : p.x, patch ? r.new_point.y : p.y);
}

// Mesh.is_valid(0), C:730-749 + segments_crossed C:806-818, plus every other evaluation that only depends on
// the quad and the post-update ring, batched so that a valid extraction pays ONE atan2 stage here:
//   straddle jobs  lanes 0..3   the four Segment.straddle calls of segments_crossed
//   dist jobs      lanes 0..5   quad edges e0..e3 and diagonals d02, d13 (reward)         -> tmp2[0..5]
//   atan2 jobs     lanes 0..3   quad corner angles                                       -> ang[0..3]
//                  lanes 4..11  the two key angles of the four ref_neighbors (M:202-231) -> tmp[0..7]
//                  lanes 12,13  boundary-quality angles (M:333-340 / M:400-407)          -> tmp[8..9]
// pk = packed ring slots of ref_neighbors (4 x 16 bit... passed as four ints), bc0/bc1 = centres of the
// boundary-quality angles on the post-update ring.
// (no_reject: the speculative update wave of k_step_spec evaluates the job lanes only -- no validity decision, no
// straddle jobs; a compile-time constant at each call)
__device__ __forceinline__ bool quad_pass(Ctx &c, const Params &prm, const VRing &vr, int p0, int p1, int p2, int p3,
                                          int bc0, int bc1, bool have_pre, double pre_qy, double pre_qx,
                                          const bool no_reject = false)
{
    const int lane = c.lane;
    const double2 *q = c.sc->q;
    // (0) exact early reject.  Corner angle a = cw(m_i; m_i+1, m_i-1) with atan2 terms (cc, dd): cc > 0 gives
    //     theta = -atan2 < 0 -> a = round(2pi + theta) >= pi; cc == +-0 gives a in {0, pi, 2pi} (all rounded): every
    //     case violates 0.01pi <= a <= 0.99pi, so only cc < 0 can pass Mesh.is_valid -- no atan2 needed to fail.
    //     (have_pre: env_check already evaluated -- and passed -- this test for the new-vertex quad; its terms are reused)
    double qy = -1.0, qx = 1.0;
    if (have_pre) {
        qy = lane < 4 ? pre_qy : qy;
        qx = lane < 4 ? pre_qx : qx;
    } else {
        if (lane < 4) {
            const double2 a = q[lane], b = q[(lane + 1) & 3], d = q[(lane + 3) & 3];
            cw_terms(mkp(a.x, a.y), mkp(b.x, b.y), mkp(d.x, d.y), qy, qx);
        }
        if (!no_reject && __ballot(!(qy < 0.0)) != 0ULL && prm.min_degree > 0.0 && prm.max_degree < kPi) return false;
    }
    // (a) segments_crossed: is_cross(m0m1, m2m3) || is_cross(m0m3, m1m2); 4 independent straddles
    bool sres = false;
    if (!no_reject && lane < 4) {
        // lane:        0        1        2        3
        // self    (m0,m1)  (m2,m3)  (m0,m3)  (m1,m2)
        // other   (m2,m3)  (m0,m1)  (m1,m2)  (m0,m3)
        const int a = (0x1020 >> (4 * lane)) & 3, b = (0x2331 >> (4 * lane)) & 3;
        const int cc = (0x0102 >> (4 * lane)) & 3, d = (0x3213 >> (4 * lane)) & 3;
        sres = straddle(mkp(q[a].x, q[a].y), mkp(q[b].x, q[b].y), mkp(q[cc].x, q[cc].y), mkp(q[d].x, q[d].y));
    }
    const unsigned sm = (unsigned)__ballot(sres);
    if (((sm & 3u) == 3u) || ((sm & 12u) == 12u)) return false;
    // (b) reward distances
    if (lane < 6) {
        // e0=(0,3) e1=(1,0) e2=(2,1) e3=(3,2) d02=(0,2) d13=(1,3)
        const int a = (0x103210 >> (4 * lane)) & 3, b = (0x322103 >> (4 * lane)) & 3;
        c.sc->tmp2[lane] = dist(mkp(q[a].x, q[a].y), mkp(q[b].x, q[b].y));
    }
    // (c) atan2 jobs: lanes 0..3 reuse the corner terms of (0); lanes 4..13 address the post-update ring
    double jy = qy, jx = qx;
    {
        const int j = lane - 4;
        const int k = j >> 1, o = (j & 1) + 1;
        const int pos = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
        const bool keyjob = lane >= 4 && lane < 12, bqjob = lane == 12 || lane == 13;
        const int ctr = keyjob ? pos : (lane == 12 ? bc0 : bc1);
        const int off = keyjob ? o : 1;
        const bool job = keyjob || bqjob;
        const int ci = job ? ctr : 0, i1 = job ? wrapi(ctr + off, vr.n) : 0, i2 = job ? wrapi(ctr - off, vr.n) : 0;
        double ty, tx;
        cw_terms(vr_at(c, vr, ci), vr_at(c, vr, i1), vr_at(c, vr, i2), ty, tx);
        jy = job ? ty : jy;
        jx = job ? tx : jx;
    }
    bool bad = false;
    if (lane < 14) {
        // every angle of this stage is quantised before use: the fast form, the exact one for the whole stage when any
        // lane sits in a guard band
#ifdef MESHENV_NO_FILTERS
        const double a = cw_sel(c.tie, jy, jx);
#else
        bool need_exact;
        double a = cw_fast(jy, jx, need_exact);
        if (__ballot(need_exact) != 0ULL) a = cw_sel(c.tie, jy, jx);
#endif
        if (lane < 4) {
            c.sc->ang[lane] = a;
            bad = a > prm.max_degree || a < prm.min_degree;
        } else {
            c.sc->tmp[lane - 4] = a;
        }
    }
    const bool any_bad = __ballot(bad) != 0ULL;
    wave_sync();
    return no_reject || !any_bad;
}

// check_intersection_with_boundary, M:510-530, after the distance filter: `count` surviving ring vertices are in
// c.list; one (vertex, quad edge, ring neighbour) segment test per lane.  r = position of the reference vertex in
// the quad (c.sc->q).
__device__ __forceinline__ bool cross_jobs(Ctx &c, int count, int mp0, int mp1, int mp2, int mp3, int r)
{
    if (count == 0) return false;
    const int n = c.n, lane = c.lane;
    wave_sync();
    const double2 *q = c.sc->q;
    const double2 qa = q[(r + 3) & 3], qb = q[(r + 2) & 3], qc = q[(r + 1) & 3];
    const P2 c0a = mkp(qa.x, qa.y), c0b = mkp(qb.x, qb.y);  // (m[r-1], m[r-2])
    const P2 c1b = mkp(qc.x, qc.y);                          // (m[r-2], m[r-3])
    bool any_hit = false;
    const int total = 4 * count;
    for (int j0 = 0; j0 < total && !any_hit; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < total;
        const int i = c.list[in ? (j >> 2) : 0];
        const int nb = (j & 2) ? wrapi(i + 1, n) : wrapi(i - 1, n);
        const bool use = in && !(nb == mp0 || nb == mp1 || nb == mp2 || nb == mp3);
        const P2 ea = (j & 1) ? c0b : c0a, eb = (j & 1) ? c1b : c0b;
        const bool hit = use && is_cross(ea, eb, ldp(c, i), ldp(c, nb));
        any_hit = __ballot(hit) != 0ULL;
    }
    wave_sync();
    return any_hit;
}

// max distance from the reference vertex to the other quad vertices (M:510), quad given as four points
__device__ __forceinline__ double quad_max_dist(int lane, P2 ref, P2 a, P2 b, P2 cpt)
{
    const P2 o = lane == 0 ? a : (lane == 1 ? b : cpt);
    const double d = dist(ref, o);
    const double d0 = lane_f64(d, 0), d1 = lane_f64(d, 1), d2 = lane_f64(d, 2);
    double m = d0 > d1 ? d0 : d1;
    return d2 > m ? d2 : m;
}

// ------------------------------------------------------------------------------------------ episode control

// reset(): copy the domain's precomputed reset state (B:67-84 computes only per-domain constants)
__device__ __forceinline__ void reset_from_domain(Ctx &c, const DevState &S)
{
    // archive the finished episode's log: flip the log half (c holds the pre-reset n, n_elem, n_new, status)
    int half = c.status & kStLogHalf;
    if (S.prm.log_cap > 0 && c.n_elem > 0) {
        if (c.lane == 0) {
            LastEpisode le;
            le.n_elem = c.n_elem;
            le.n_new = c.n_new;
            le.flags = (c.n <= 5 ? 1 : 0) | (c.status & kStLogOverflow);
            LastEpisode *last_ep = load_cold(S).last_ep;
            le.episodes = last_ep[c.env].episodes + 1;
            last_ep[c.env] = le;
        }
        half ^= kStLogHalf;
    }
    const DomConst dc = S.dom[c.dom];
    const DevCold cold = load_cold(S);
    const int doff = uniform_i32(dc.off), n0 = uniform_i32(dc.n0);
    wave_sync();
    for (int i = c.lane; i < n0; i += 64) {
        c.xy[i] = cold.dom_xy[doff + i];
        c.id[i] = i;
        c.key[i] = cold.dom_key[doff + i];
        c.stamp[i] = cold.dom_stamp[doff + i];
    }
    c.n = n0;
    c.ref = uniform_i32(dc.ref);
    c.bl = uniform_f64(dc.bl);
    c.ct = uniform_f64(dc.ct);
    c.st = uniform_f64(dc.st);
    c.area = uniform_f64(dc.orig_area);
    c.n_elem = 0; c.failed = 0; c.n_new = 0; c.counter = 0;
    c.status = (c.ref < 0 ? kStNoReference : 0) | half;
    c.obs = c.lane < kObsDim ? cold.dom_obs[(size_t)c.dom * kObsDim + c.lane] : 0.0f;
    c.ring_dirty = true;
    wave_sync();
}

__device__ __forceinline__ void log_quad(Ctx &c, const DevState &S, int g0, int g1, int g2, int g3)
{
    const int cap = S.prm.log_cap;
    if (c.n_elem < cap) {
        if (c.lane == 0) {
            int32_t *dst = load_cold(S).log_quads + (((size_t)c.env * 2 + ((c.status >> 4) & 1)) * cap + c.n_elem) * 4;
            dst[0] = g0; dst[1] = g1; dst[2] = g2; dst[3] = g3;
        }
    } else if (cap > 0) {
        c.status |= kStLogOverflow;
    }
    c.n_elem += 1;
}

// The quad of an action and the ring update_boundary (M:575-648) would leave behind, as slot arithmetic on the current
// ring: a function of (rule / new vertex, reference slot, ring length) alone.
struct QuadPlan {
    int mp0, mp1, mp2, mp3, r;   // ring slots of the quad vertices (mp0 = -1: the new vertex), position of the reference vertex
    int p0, p1, p2, p3;          // ref_neighbors as post-update ring slots
    int t0, t1;                  // post-update slots of the kept quad vertices
    int bc0, bc1;                // centres of the boundary-quality angles on the post-update ring
    VRing vr;
};

__device__ __forceinline__ QuadPlan plan_quad(bool new_vertex, int rule, int index, int n, P2 new_point)
{
    QuadPlan q;
    if (new_vertex) {  // [new, i-1, i, i+1], B:177-182
        q.mp0 = -1; q.mp1 = wrapi(index - 1, n); q.mp2 = index; q.mp3 = wrapi(index + 1, n); q.r = 2;
    } else if (rule == -1) {  // [i-1, i, i+1, i+2], B:147-153
        q.mp0 = wrapi(index - 1, n); q.mp1 = index; q.mp2 = wrapi(index + 1, n); q.mp3 = wrapi(index + 2, n); q.r = 1;
    } else {  // [i-2, i-1, i, i+1], B:156-162
        q.mp0 = wrapi(index - 2, n); q.mp1 = wrapi(index - 1, n); q.mp2 = index; q.mp3 = wrapi(index + 1, n); q.r = 2;
    }
    q.vr.is_new = new_vertex;
    q.vr.new_point = new_point;
    q.t0 = 0; q.t1 = 0;
    if (new_vertex) {
        q.vr.n = n; q.vr.lo = index; q.vr.hi = -1;
        q.p0 = wrapi(index + 1, n); q.p1 = wrapi(index - 1, n); q.p2 = wrapi(index + 2, n); q.p3 = wrapi(index - 2, n);
        q.bc0 = wrapi(index + 1, n); q.bc1 = wrapi(index - 1, n);
    } else {
        q.vr.lo = q.mp1 < q.mp2 ? q.mp1 : q.mp2; q.vr.hi = q.mp1 < q.mp2 ? q.mp2 : q.mp1; q.vr.n = n - 2;
        q.t0 = q.mp0 - (q.mp0 > q.vr.lo ? 1 : 0) - (q.mp0 > q.vr.hi ? 1 : 0);
        q.t1 = q.mp3 - (q.mp3 > q.vr.lo ? 1 : 0) - (q.mp3 > q.vr.hi ? 1 : 0);
        const int id = q.t0 > q.t1 ? q.t0 : q.t1, nn = n - 2;
        q.p0 = wrapi(id, nn); q.p1 = wrapi(id - 1, nn); q.p2 = wrapi(id + 1, nn); q.p3 = wrapi(id - 2, nn);
        q.bc0 = q.t0; q.bc1 = q.t1;
    }
    return q;
}

// k_step_spec: what the checking wavefront of an env posts (in LDS) once the action has survived the cheap exact tests,
// so that another wavefront of the workgroup can run the extraction speculatively while the checks finish.
struct alignas(16) SpecJob {
    int state;        // 0 none / withdrawn, 1 posted, 2 claimed by an update wave
    int verdict;      // 0 pending, 1 commit, 2 discard
    int upd_done;     // update wave: the post-update ring (buffer B) is final, the reward may be computed from it
    int helper_done;  // checking wave: the reward has been computed, an auto-reset may overwrite buffer B
    int what;         // new_vertex | (rule + 1) << 1 | index << 3
    int n;            // ring length
    int env, pad;
    double npx, npy;  // the candidate point
};

struct SpecHook {
    SpecJob *job;               // this env's job record
    int *n_nopost;              // workgroup counter of owners that can no longer post (the pollers' exit condition)
    bool posted;
    bool stale;                 // the posted quad is not the one the checks went on with (find_same_point hit)
};

__device__ __forceinline__ void spec_post(SpecHook *hk, const Ctx &c, bool new_vertex, int rule, int index, P2 new_point)
{
    SpecJob *j = hk->job;
    if (c.lane == 0) {
        // (everything else the update wave needs -- the env's 64-byte record and its work counters -- it reads from HBM,
        // where they stay unchanged until a commit)
        *(int4 *)&j->verdict = make_int4(0, 0, 0, (new_vertex ? 1 : 0) | ((rule + 1) << 1) | (index << 3));
        j->n = c.n;
        j->env = c.env;
        *(double2 *)&j->npx = make_double2(new_point.x, new_point.y);
    }
    wave_sync();  // a wave's LDS stores complete in order: the record is in place before the flag
    if (c.lane == 0) {
        *(volatile int *)&j->state = 1;
        atomicAdd(hk->n_nopost, 1);   // after the flag: a poller that sees the count also sees the post
    }
    hk->posted = true;
    // from here on this wave is on the launch's critical path (nine posts in ten end in an extraction): it takes the SIMD's
    // issue slots ahead of the waves that are still rejecting their actions (arbitration is by priority, then age)
    __builtin_amdgcn_s_setprio(2);
    SPEC_COUNT(0);
    SPEC_TIME(c.env, 1);
}

struct StepResult {
    double reward;
    int done, complete;
    bool valid;
};

// Outcome of the checks of one step(): everything the update needs, so that the checks and the update can run on
// different wavefronts (k_step_group hands it over through LDS).
struct Decision {
    int ok;             // the action extracts a valid element
    int new_vertex;     // rule 0 with a new vertex (else the quad's two interior ring vertices are removed)
    int index;          // ring slot of the reference vertex
    int mp0, mp1, mp2, mp3;  // ring slots of the quad vertices before the update (mp0 = -1: the new vertex)
    int p0, p1, p2, p3;      // ref_neighbors as ring slots after the update
    int t0, t1;              // slots of the kept quad vertices after the update
    int lo, hi;              // removed slots (lo < hi)
    int done, no_reference;
    double reward;           // reward accumulated so far (penalties, terminal bonus of B:141-143)
    P2 new_point;
};

// Checks of step(action), B:113-191: action decode, rule selection, point-in-polygon / same-point, quad validity,
// boundary intersection.  a0 = rule type, (a1, a2) = candidate point in the local frame.  On success the quad,
// its corner angles / edge lengths and the speculative candidate-key and boundary-quality angles are in c.sc.
// (pre_reject: evaluate the quad's corner test before the point-in-polygon pass; a compile-time constant at each call)
// (is_move: the move() API, B:265-326 -- the point comes as (radius fraction, angle) Python floats in mv_r / mv_a, the
// rule from mv_type against TYPE_THRESHOLD = 0.3, and there is no find_same_point; a compile-time constant per call)
// (hook != nullptr: k_step_spec -- every quad takes the exact corner-sign test up front and, once it has passed, the
// action is posted for a speculative update on another wavefront; a compile-time null everywhere else)
// (fuse_passes: the three ring passes of a rule-0 point as one, see below; a compile-time constant per call)
__device__ __forceinline__ Decision env_check(Ctx &c, const DevState &S, float a0, float a1, float a2, const bool pre_reject,
                                              const bool is_move = false, double mv_r = 0.0, double mv_a = 0.0,
                                              double mv_type = 0.5, SpecHook *hook = nullptr, const bool fuse_passes = false)
{
    const Params &prm = S.prm;
    const int lane = c.lane;
    Decision d;
    d.ok = 0; d.new_vertex = 0; d.done = 0; d.no_reference = 0; d.reward = 0.0;
    d.index = c.ref;
    d.mp0 = d.mp1 = d.mp2 = d.mp3 = 0; d.p0 = d.p1 = d.p2 = d.p3 = 0; d.t0 = d.t1 = 0; d.lo = d.hi = 0;
    d.new_point = mkp(0.0, 0.0);
    const int index = c.ref;
    const int n = c.n;

    if (index < 0) {
        // no reference vertex (the reference's find_next_state returned None and its next step() would
        // raise): end the episode as truncated
        d.reward = -1.0;
        d.done = 1;
        d.no_reference = 1;
        return d;
    }
    if (n <= 5) {
        d.reward = 10.0;  // B:141-143
        d.done = 1;
        return d;
    }
    int mp0, mp1, mp2, mp3, r;
    bool new_vertex = false, have_filter = false, have_pre = false;
    double pre_qy = 0.0, pre_qx = 0.0;
    int near_count = 0;
    P2 new_point = mkp(0.0, 0.0);
    int rule;
    const bool rule_m1 = is_move ? (mv_type <= 0.3) : (a0 <= -0.5f);
    const bool rule_p1 = is_move ? (mv_type >= 1 - 0.3) : (a0 >= 0.5f);
    if (rule_m1) rule = -1;
    else if (rule_p1) rule = 1;
    else {
        rule = 0;
        // action_2_point -> detransformation, B:616-625, B:98-106, D:67-83; cos/sin of the frame angle are
        // per-state values computed with the observation
        double px = (double)round4_npf(a1), py = (double)round4_npf(a2);
        if (is_move) {  // B:266-269: (bl * radius * r) * cos / sin(angle), rounded to 6 places, scale 1 (B:105)
            const SinCos sc = sincos_nc(mv_a);
            const double br = (c.bl * prm.radius) * mv_r;
            px = round6_py(br * sc.c);
            py = round6_py(br * sc.s);
        }
        const P2 p0 = ldp(c, index);
        double ox = c.ct * px + c.st * py;
        double oy = -c.st * px + c.ct * py;
        ox *= is_move ? 1.0 : c.bl;
        oy *= is_move ? 1.0 : c.bl;
        ox += p0.x;
        oy += p0.y;
        new_point = mkp(uniform_f64(round4_np(ox)), uniform_f64(round4_np(oy)));
        MESHENV_STAMP(c, 1);
        // Exact early rejection before the point-in-polygon pass.  Unless the point coincides with a ring vertex
        // (find_same_point, checked conservatively on squared distances), the only quad this action can build is
        // [new, i-1, i, i+1]; if one of its corners has a non-negative cross term its angle cannot lie in
        // [0.01 pi, 0.99 pi] (see quad_pass) and the step fails whether or not the point is inside -- with the same
        // reward.  Four of five uniform-random rule-0 actions end here, ~30 instructions in.  Used by the rollout kernel
        // only (+7 %): one step per launch gains nothing in the latency regime (the slowest chains are valid steps,
        // which only pay for the test) and loses 4 % at 65 536 envs (measured, tools/ab_all.sh).
        if (pre_reject && prm.min_degree > 0.0 && prm.max_degree < kPi) {
            // the quad goes to the scratch slots quad_pass reads; lane k evaluates corner k: (self, next, previous)
            wave_sync();
            if (lane < 4) {
                const int mp = lane == 1 ? wrapi(index - 1, n) : lane == 2 ? index : wrapi(index + 1, n);
                c.sc->q[lane] = lane == 0 ? make_double2(new_point.x, new_point.y) : c.xy[mp];
            }
            wave_sync();
            double qy = -1.0, qx = 1.0;
            if (lane < 4) {
                const double2 *q = c.sc->q;
                const double2 a = q[lane], b = q[(lane + 1) & 3], dd = q[(lane + 3) & 3];
                cw_terms(mkp(a.x, a.y), mkp(b.x, b.y), mkp(dd.x, dd.y), qy, qx);
            }
            pre_qy = qy; pre_qx = qx;
            have_pre = true;
            if (__ballot(lane < 4 && !(qy < 0.0)) != 0ULL) {  // the new-vertex quad is invalid; is it the quad at all?
                have_pre = false;
                const double eps2 = prm.same_eps * prm.same_eps * 1.000001;
                bool maybe_same = false;
                for (int base = 0; base < n; base += 64) {
                    const int i = base + lane;
                    bool near = false;
                    if (i < n) {
                        const P2 v = ldp(c, i);
                        const double dx = v.x - new_point.x, dy = v.y - new_point.y;
                        near = dx * dx + dy * dy < eps2;
                    }
                    maybe_same = maybe_same || (__ballot(near) != 0ULL);
                }
                if (!maybe_same) {
                    d.reward += c.n_elem ? -1.0 / c.n_elem : -1.0;
                    return d;
                }
            }
            // the new-vertex quad survived the corner test: hand it out before the point-in-polygon pass
            if (hook && have_pre) spec_post(hook, c, true, 0, index, new_point);
        }
        NearFilter f;
        f.ref = p0;
        f.mp0 = -1; f.mp1 = wrapi(index - 1, n); f.mp2 = index; f.mp3 = wrapi(index + 1, n);
        bool same;
        if (fuse_passes && n <= 64) {
            // One ring pass for the three per-vertex tests of a rule-0 point -- crossing parity (M:539-546), find_same_point
            // and the distance filter of the quad [new, i-1, i, i+1]: their dependency chains are independent, so the
            // slowest check of the workgroup (a valid rule-0 action, which pays for all three) overlaps them instead of
            // running two passes back to back.  A point outside the ring pays ~55 instructions it used not to, which is
            // why only the CU-group kernel (latency regime) asks for it: at 65 536 envs it cost 4 % (tools/ab_all.sh).
            f.max_dist = dist_thr(quad_max_dist(lane, p0, new_point, ldp(c, f.mp1), ldp(c, f.mp3)));
            const bool in = lane < n;
            const int ic = in ? lane : 0;
            const P2 v = ldp(c, ic);
            const bool counted = edge_counted(c, new_point, mkp(prm.ray_length, new_point.y), ic, in);
            const bool sm = !is_move && in && dist_lt(v, new_point, dist_thr(prm.same_eps));
            const bool near = in && nf_near(f, lane, v);
            const bool inside = (__popcll(__ballot(counted)) & 1) != 0;
            same = __ballot(sm) != 0ULL;
            near_count = nf_compact(c, near, lane, 0);
            MESHENV_STAMP(c, 2);
            if (!inside) {
                d.reward += c.n_elem ? -1.0 / c.n_elem : -1.0;
                return d;
            }
        } else
        {
            const bool inside = point_inside(c, prm, new_point);
            MESHENV_STAMP(c, 2);
            if (!inside) {
                d.reward += c.n_elem ? -1.0 / c.n_elem : -1.0;
                return d;
            }
            // second ring pass: find_same_point + the distance filter of the quad [new, i-1, i, i+1]
            f.max_dist = dist_thr(quad_max_dist(lane, p0, new_point, ldp(c, f.mp1), ldp(c, f.mp3)));
            near_count = near_filter_pass(c, f, new_point, is_move ? 0.0 : prm.same_eps, same);
        }
        if (same) {  // existing point: the rule -1 quad, B:168-175 (its own filter pass follows)
            rule = -1;
            if (hook) hook->stale = hook->posted;
        } else {
            new_vertex = true;
            have_filter = true;
        }
    }
    // (move(): a rejection changes the reference vertex without changing the ring, so the memo does not apply there)
    if (!is_move && !new_vertex && (c.status & (rule == -1 ? kStRm1Bad : kStRp1Bad))) {
        d.reward += c.n_elem ? -1.0 / c.n_elem : -1.0;  // B:248, outcome remembered from an earlier attempt on this state
        return d;
    }
    const QuadPlan qp = plan_quad(new_vertex, rule, index, n, new_point);
    mp0 = qp.mp0; mp1 = qp.mp1; mp2 = qp.mp2; mp3 = qp.mp3; r = qp.r;
    const VRing vr = qp.vr;
    const int p0 = qp.p0, p1 = qp.p1, p2 = qp.p2, p3 = qp.p3;  // ref_neighbors as post-update ring slots
    const int t0 = qp.t0, t1 = qp.t1;                          // post-update slots of the kept quad vertices
    const int bc0 = qp.bc0, bc1 = qp.bc1;
    if (!(have_pre && new_vertex)) {  // (the early-rejection test of rule 0 has stored this very quad already)
        wave_sync();
        if (lane < 4) {
            const int mp = lane == 0 ? mp0 : lane == 1 ? mp1 : lane == 2 ? mp2 : mp3;
            c.sc->q[lane] = mp < 0 ? make_double2(new_point.x, new_point.y) : c.xy[mp];
        }
        wave_sync();
    }
    MESHENV_STAMP(c, 3);
    bool pre_ok = have_pre && new_vertex;
    if (hook && !hook->posted && !pre_ok && prm.min_degree > 0.0 && prm.max_degree < kPi) {
        // rules -1 / +1: the corner-sign test of quad_pass (0), taken here so that the post comes right after it
        double qy = -1.0, qx = 1.0;
        if (lane < 4) {
            const double2 *q = c.sc->q;
            const double2 a = q[lane], b = q[(lane + 1) & 3], dd = q[(lane + 3) & 3];
            cw_terms(mkp(a.x, a.y), mkp(b.x, b.y), mkp(dd.x, dd.y), qy, qx);
        }
        if (__ballot(lane < 4 && !(qy < 0.0)) != 0ULL) {
            d.reward += c.n_elem ? -1.0 / c.n_elem : -1.0;  // B:248
            if (!new_vertex) c.status |= (rule == -1 ? kStRm1Bad : kStRp1Bad);
            return d;
        }
        pre_qy = qy; pre_qx = qx;
        pre_ok = true;
        spec_post(hook, c, new_vertex, rule, index, new_point);
    }
    bool ok = quad_pass(c, prm, vr, p0, p1, p2, p3, bc0, bc1, pre_ok, pre_qy, pre_qx);
    MESHENV_STAMP(c, 4);
    if (ok) {
        if (!have_filter) {  // rules -1 / +1 (and the same-point case): the filter pass on its own
            const double2 *q = c.sc->q;
            NearFilter f;
            f.ref = mkp(q[r].x, q[r].y);
            f.mp0 = mp0; f.mp1 = mp1; f.mp2 = mp2; f.mp3 = mp3;
            const double2 qa = q[(r + 1) & 3], qb = q[(r + 2) & 3], qc = q[(r + 3) & 3];
            f.max_dist = dist_thr(quad_max_dist(lane, f.ref, mkp(qa.x, qa.y), mkp(qb.x, qb.y), mkp(qc.x, qc.y)));
            bool unused;
            near_count = near_filter_pass(c, f, f.ref, 0.0, unused);
        }
        ok = !cross_jobs(c, near_count, mp0, mp1, mp2, mp3, r);
    }
    MESHENV_STAMP(c, 5);
    if (!ok) {
        d.reward += c.n_elem ? -1.0 / c.n_elem : -1.0;  // B:248
        if (!new_vertex && !is_move) c.status |= (rule == -1 ? kStRm1Bad : kStRp1Bad);
        return d;
    }
    d.ok = 1;
    d.new_vertex = new_vertex ? 1 : 0;
    d.mp0 = mp0; d.mp1 = mp1; d.mp2 = mp2; d.mp3 = mp3;
    d.p0 = p0; d.p1 = p1; d.p2 = p2; d.p3 = p3;
    d.t0 = t0; d.t1 = t1;
    d.lo = vr.lo; d.hi = vr.hi;
    d.new_point = new_point;
    return d;
}

// The extraction itself, B:192-238 + find_next_state: element log, ring update, candidate list patch, reward,
// next observation.  Runs on whichever wavefront holds the env's LDS region (c) and the decision.
// (upd_done != nullptr: CU-group kernel with a helper wavefront -- the reward is computed there from the post-update
// ring; this wave publishes the updated ring through *upd_done and skips every reward-only computation)
// (is_move: move() extracts the element without reward, current_area or failed_num bookkeeping and observes with
// static=True and its not_valid_points, B:328-345)
__device__ __forceinline__ void env_apply(Ctx &c, const DevState &S, Decision &d, volatile int *upd_done = nullptr,
                                          const bool is_move = false, const double2 *nv = nullptr, int n_nv = 0,
                                          const LibmRef lr = LibmRef{nullptr, 0})
{
    const bool has_helper = upd_done != nullptr;
    const Params &prm = S.prm;
    const int lane = c.lane;
    const int n = c.n, index = d.index;
    const bool new_vertex = d.new_vertex != 0;
    const int g0 = d.mp0 < 0 ? (kNewBit | c.n_new) : c.id[d.mp0];
    const int g1 = c.id[d.mp1], g2 = c.id[d.mp2], g3 = c.id[d.mp3];
    log_quad(c, S, g0, g1, g2, g3);  // generated_meshes.append, B:192

    // quad terms of the reward (pre-update coordinates; the Mesh holds the Vertex objects)
    double e_reward;
    BqArgs bq;
    {
        const double e0 = c.sc->tmp2[0], e1 = c.sc->tmp2[1], e2 = c.sc->tmp2[2], e3 = c.sc->tmp2[3];
        const double d02 = c.sc->tmp2[4], d13 = c.sc->tmp2[5];
        const double ang0 = c.sc->ang[0], ang1 = c.sc->ang[1], ang2 = c.sc->ang[2], ang3 = c.sc->ang[3];
        // Mesh.get_quality('robust'), C:873-884
        double mn = e1 < e0 ? e1 : e0;
        mn = e2 < mn ? e2 : mn;
        mn = e3 < mn ? e3 : mn;
        e_reward = 0.0;
        if (!has_helper) {
            const double q1 = sqrt(2.0) * mn / (d13 > d02 ? d13 : d02);
            double amn = ang1 < ang0 ? ang1 : ang0, amx = ang1 > ang0 ? ang1 : ang0;
            amn = ang2 < amn ? ang2 : amn; amx = ang2 > amx ? ang2 : amx;
            amn = ang3 < amn ? ang3 : amn; amx = ang3 > amx ? ang3 : amx;
            e_reward = uniform_f64(sqrt(q1 * (amn / amx)));
        }
        bq.skip = has_helper;
        // Mesh.compute_area, C:935-950, left-to-right: ((0.5*e0)*e1)*sin(c1) + ((0.5*e2)*e3)*sin(c3);
        // corner_1 == corner angle 0, corner_3 == corner angle 2
        bq.half01 = uniform_f64(0.5 * e0 * e1);
        bq.half23 = uniform_f64(0.5 * e2 * e3);
        bq.q_ang0 = ang0;
        bq.q_ang2 = ang2;
        bq.ang0 = c.sc->tmp[8];
        bq.ang1 = c.sc->tmp[9];
    }
    // candidate keys of the four ref_neighbors from the speculative job lanes
    double kk = 0.0;
    bool okk = false;
    int pos = 0;
    if (lane < 4) {
        pos = lane == 0 ? d.p0 : lane == 1 ? d.p1 : lane == 2 ? d.p2 : d.p3;
        okk = key_from_angles(prm, c.sc->tmp[2 * lane], c.sc->tmp[2 * lane + 1], kk);
    }

    // update_boundary, M:575-648
    wave_sync();
    if (new_vertex) {
        if (lane == 0) {
            c.xy[index] = make_double2(d.new_point.x, d.new_point.y);
            c.id[index] = kNewBit | c.n_new;
            c.stamp[index] = kNotCand;
            const int cap = prm.log_cap;
            if (c.n_new < cap) load_cold(S).log_vxy[((size_t)c.env * 2 + ((c.status >> 4) & 1)) * cap + c.n_new] = make_double2(d.new_point.x, d.new_point.y);
        }
        if (prm.log_cap > 0 && c.n_new >= prm.log_cap) c.status |= kStLogOverflow;
        c.n_new += 1;
        bq.mode = 1; bq.a = index; bq.b = 0;
    } else {
        // delete the two interior quad vertices (slots mp1, mp2), compacting the ring in LDS
        const int lo = d.lo, hi = d.hi;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            double2 vxy = make_double2(0, 0);
            double vk = 0; int vs = 0, vid = 0;
            const bool live = i < n && i != lo && i != hi;
            if (live) { vxy = c.xy[i]; vk = c.key[i]; vs = c.stamp[i]; vid = c.id[i]; }
            wave_sync();
            if (live) {
                const int j = i - (i > lo ? 1 : 0) - (i > hi ? 1 : 0);
                c.xy[j] = vxy; c.key[j] = vk; c.stamp[j] = vs; c.id[j] = vid;
            }
            wave_sync();
        }
        c.n = n - 2;
        bq.mode = 2; bq.a = d.t0; bq.b = d.t1;
    }
    wave_sync();
    // remove_reference_candidates(ref_neighbors [+ removed]) then add_reference_candidates in order
    {
        const unsigned long long mk = __ballot(okk);
        if (lane < 4) {
            if (okk) {
                c.key[pos] = kk;
                c.stamp[pos] = c.counter + __popcll(mk & ((2ULL << lane) - 1ULL));
            } else {
                c.stamp[pos] = kNotCand;
            }
        }
        c.counter += __popcll(mk);
    }
    MESHENV_STAMP(c, 6);
    c.ring_dirty = true;
    c.status &= ~(kStRm1Bad | kStRp1Bad);  // new state: forget the memoised rejections
    if (has_helper) {  // ring coordinates and length are final: the helper may read them
        wave_sync();
        if (lane == 0) *upd_done = 1;
    }
    const bool finished = c.n <= 5;  // B:232-238
    if (finished && c.n == 4) log_quad(c, S, c.id[0], c.id[1], c.id[2], c.id[3]);
    // current_area -= mesh_area (B:200) happens inside: the area needs sin(corner angles), a stage-B job
    if (is_move) {
        bq.mode = 0;
        bq.skip = false;
        find_next_state(c, S, bq, true, nv, n_nv, lr);
    } else {
        find_next_state(c, S, bq);
    }
    MESHENV_STAMP(c, 14);
    if (!has_helper && !is_move) {
        const double mesh_area = bq.mesh_area;
        // get_quality(mesh, 2), M:1733-1740
        const double quality = e_reward + 1 * (bq.b_reward - 1);
        // get_speed_penalty, B:434-450
        const DomConst &dc = S.dom[c.dom];
        const double min_area = dc.min_area, crit_area = dc.crit_area;
        double speed = 0.0;
        if (min_area <= mesh_area && mesh_area < crit_area) speed = (mesh_area - crit_area) / (crit_area - min_area);
        else if (mesh_area < min_area) speed = -1.0;
        d.reward += quality + speed;
        if (finished) d.reward += 10.0;
    }
    if (finished) d.done = 1;
}

// The reward of a valid extraction on a helper wavefront of the CU-group kernel: quad quality from the scratch values
// of the checks, element area, boundary quality on the POST-update ring (published by the update wave through
// h.upd_done), speed penalty.  Same operations in the same order as the fused path of env_apply / find_next_state
// (B:192-231, M:329-426, M:1733-1740, B:434-450), which the one-wave kernels keep.  c is carved on the env's LDS region.
__device__ __forceinline__ double reward_on_helper(Ctx &c, const DevState &S, const Decision &d, int n_before, int dom,
                                                   volatile int *upd_done)
{
    const int lane = c.lane;
    const double e0 = c.sc->tmp2[0], e1 = c.sc->tmp2[1], e2 = c.sc->tmp2[2], e3 = c.sc->tmp2[3];
    const double d02 = c.sc->tmp2[4], d13 = c.sc->tmp2[5];
    const double ang0 = c.sc->ang[0], ang1 = c.sc->ang[1], ang2 = c.sc->ang[2], ang3 = c.sc->ang[3];
    const double bq_ang0 = c.sc->tmp[8], bq_ang1 = c.sc->tmp[9];
    double mn = e1 < e0 ? e1 : e0;
    mn = e2 < mn ? e2 : mn;
    mn = e3 < mn ? e3 : mn;
    const double q1e = sqrt(2.0) * mn / (d13 > d02 ? d13 : d02);
    double amn = ang1 < ang0 ? ang1 : ang0, amx = ang1 > ang0 ? ang1 : ang0;
    amn = ang2 < amn ? ang2 : amn; amx = ang2 > amx ? ang2 : amx;
    amn = ang3 < amn ? ang3 : amn; amx = ang3 > amx ? ang3 : amx;
    const double e_reward = uniform_f64(sqrt(q1e * (amn / amx)));
    const double half01 = uniform_f64(0.5 * e0 * e1), half23 = uniform_f64(0.5 * e2 * e3);
    double sj = 0.0;
    if (lane < 2) sj = sincos_small_nc(lane == 0 ? ang0 : ang2).s;  // the sincos of stage B
    const double mesh_area = uniform_f64(half01 * lane_f64(sj, 0) + half23 * lane_f64(sj, 1));
    // speed penalty, B:434-450
    const DomConst &dc = S.dom[dom];
    const double min_area = dc.min_area, crit_area = dc.crit_area;
    double speed = 0.0;
    if (min_area <= mesh_area && mesh_area < crit_area) speed = (mesh_area - crit_area) / (crit_area - min_area);
    else if (mesh_area < min_area) speed = -1.0;

    spin_until_nonzero(upd_done);  // the update wave sets it unconditionally (bounded: see spin_until_nonzero)
    wave_sync();
    const bool m1 = d.new_vertex != 0;
    const int n = m1 ? n_before : n_before - 2;
    c.n = n;
    const int bqa = m1 ? d.index : d.t0, bqb = m1 ? 0 : d.t1;
    const int bqlo = bqa < bqb ? bqa : bqb;
    // boundary-quality distances: the job lanes 8..14 of stage A
    int ia = 0, ib = 0;
    {
        const int k = lane - 10;
        const bool jwin = lane >= 10 && lane < (m1 ? 14 : 15);
        const int wbase = m1 ? bqa : bqlo;
        ia = jwin ? wrapi(wbase - 2 + k, n) : ia;
        ib = jwin ? wrapi(wbase - 1 + k, n) : ib;
        ia = (lane == 8 || lane == 9) ? bqa : ia;
        ib = lane == 8 ? (m1 ? wrapi(bqa + 1, n) : bqb) : ib;
        ib = (lane == 9 && m1) ? wrapi(bqa - 1, n) : ib;
    }
    const double dv = dist(ldp(c, ia), ldp(c, ib));
    const double bq_d0 = lane_f64(dv, 8), bq_d1 = lane_f64(dv, 9);
    double bq_sum = lane_f64(dv, 10);
    bq_sum += lane_f64(dv, 11);
    bq_sum += lane_f64(dv, 12);
    bq_sum += lane_f64(dv, 13);
    const double bq_sum5 = bq_sum + lane_f64(dv, 14);
    const double dst = uniform_f64(bq_d0 + bq_d1);
    double amin = 1e300;
    bool have = false;
    if (bq_ang0 < kPi / 3) { amin = bq_ang0; have = true; }
    if (bq_ang1 < kPi / 3) { amin = bq_ang1 < amin ? bq_ang1 : amin; have = true; }
    const double q1 = have ? 3 * amin / kPi : 1.0;
    double b_reward;
    if (m1) {
        // near-vertex scan of compute_boundary_quality: added(i) = near(i) && !added(i-1), M:355-357
        double m_d = kInf;
        int carry = 0;
        const int bqi = bqa;
        const int w1 = wrapi(bqi + 1, n), w2 = wrapi(bqi + 2, n), w3 = wrapi(bqi - 1, n), w4 = wrapi(bqi - 2, n);
        const P2 add_v = ldp(c, bqi);
        const DistThr dst_thr = dist_thr(dst);
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool near = false;
            if (i < n && !(i == bqi || i == w1 || i == w2 || i == w3 || i == w4)) near = dist_lt(add_v, ldp(c, i), dst_thr);
            const unsigned long long m = __ballot(near);
            bool added = false;
            if (near) {
                const unsigned long long zeros_below = ~m & ((1ULL << lane) - 1ULL);
                if (zeros_below != 0ULL) {
                    const int pz = 63 - __clzll((long long)zeros_below);
                    added = ((lane - pz - 1) & 1) == 0;
                } else {
                    added = (lane & 1) == carry;
                }
            }
            if (added) {
                const double dd = seg_point_distance(ldp(c, wrapi(i + 1, n)), ldp(c, i), add_v);
                m_d = dd < m_d ? dd : m_d;
            }
            carry = (int)((__ballot(added) >> 63) & 1ULL);
        }
        m_d = wave_min_f64(m_d);
        const double targt_len = dst / 2;
        const double mean_dist = bq_sum / 4;
        const double smoothness = (targt_len < mean_dist ? targt_len : mean_dist) / (targt_len > mean_dist ? targt_len : mean_dist);
        double q2 = 1.0;
        if (m_d < 1e299) q2 = (m_d < 0.5 * dst) ? m_d / (0.5 * dst) : 1.0;
        b_reward = cbrt(smoothness * q1 * q2);  // math.pow(x, 1/3)
    } else {
        const double targt_len = bq_d0;
        const double mean_dist = bq_sum5 / 5;
        const double smoothness = (targt_len < mean_dist ? targt_len : mean_dist) / (targt_len > mean_dist ? targt_len : mean_dist);
        b_reward = sqrt(q1 * smoothness);  // math.pow(angle_quality * smoothness, 1/2)
    }
    const double quality = e_reward + 1 * (b_reward - 1);  // get_quality(mesh, 2), M:1733-1740
    double reward = 0.0;
    reward += quality + speed;
    if (n <= 5) reward += 10.0;  // B:232-238
    return reward;
}

// failed_num bookkeeping and the 100-failure truncation, B:253-263
__device__ __forceinline__ StepResult env_finish(Ctx &c, const Params &prm, const Decision &d)
{
    StepResult out;
    out.valid = d.ok != 0;
    int done = d.done, complete = d.no_reference ? 0 : 1;
    if (d.ok) {
        c.failed = 0;
    } else {
        // the observation of an unchanged state is the cached one (find_next_state is a pure function of
        // ring, candidate list and current_area, none of which changed)
        c.failed += 1;
        if (c.failed >= prm.fail_limit) {  // B:256-262
            done = 1;
            complete = 0;
        }
    }
    out.reward = d.reward;
    out.done = done;
    out.complete = complete;
    return out;
}

// step(action), B:113-263, on one wavefront
__device__ __forceinline__ StepResult env_step(Ctx &c, const DevState &S, float a0, float a1, float a2, const bool pre_reject)
{
    Decision d = env_check(c, S, a0, a1, a2, pre_reject);
    if (d.ok) env_apply(c, S, d);
    return env_finish(c, S.prm, d);
}

// ------------------------------------------------------------------------------------------ kernels

// One wave per domain: everything reset() computes (candidate list, reference vertex, first observation).
__global__ void __launch_bounds__(64) k_init_domains(DevState S, int cap)
{
    extern __shared__ double2 smem[];
    Ctx c;
    carve_lds(c, smem, cap);
    const int d = blockIdx.x;
    c.lane = lane_id();
    c.env = -1;
    c.base = 0;
    c.dom = d;
    const DomConst dc = S.dom[d];
    const DevCold cold = load_cold(S);
    const int doff = dc.off;
    c.n = dc.n0;
    c.area = dc.orig_area;
    c.status = 0;
    c.bl = 0.0; c.ct = 1.0; c.st = 0.0;
    c.obs = 0.0f;
    for (int i = c.lane; i < c.n; i += 64) {
        c.xy[i] = cold.dom_xy[doff + i];
        c.id[i] = i;
    }
    wave_sync();
    // find_reference_candidates, M:233-261: stable sort by key -> ties in ring order (stamp = -index)
    for (int i = c.lane; i < c.n; i += 64) {
        double cc, dd, k = 0.0;
        key_angle_terms(c, i, 0, cc, dd);
        const double a0 = cw_exact(cc, dd);
        key_angle_terms(c, i, 1, cc, dd);
        const double a1 = cw_exact(cc, dd);
        const bool ok = key_from_angles(S.prm, a0, a1, k);
        c.key[i] = k;
        c.stamp[i] = ok ? -i : kNotCand;
    }
    BqArgs bq;
    bq.skip = false;
    bq.mode = 0; bq.a = 0; bq.b = 0; bq.ang0 = 0; bq.ang1 = 0; bq.q_ang0 = 0; bq.q_ang2 = 0; bq.half01 = 0; bq.half23 = 0;
    find_next_state(c, S, bq);
    for (int i = c.lane; i < c.n; i += 64) {
        cold.dom_key[doff + i] = c.key[i];
        cold.dom_stamp[doff + i] = c.stamp[i];
    }
    if (c.lane < kObsDim) cold.dom_obs[(size_t)d * kObsDim + c.lane] = c.obs;
    if (c.lane == 0) {
        S.dom[d].ref = c.ref;
        S.dom[d].bl = c.bl;
        S.dom[d].ct = c.ct;
        S.dom[d].st = c.st;
    }
}

__global__ void __launch_bounds__(64) k_reset(DevState S, int cap, const uint8_t *mask, float *obs_out, int first,
                                               unsigned long long step_now, int is_static, int32_t *nv_count,
                                               int32_t *nv_meta)
{
    extern __shared__ double2 smem[];
    Ctx c;
    carve_lds(c, smem, cap);
    const int env = blockIdx.x;
    const bool doit = mask == nullptr || mask[env] != 0;
    c.lane = lane_id();
    if (!doit) {
        if (obs_out && c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = S.obs_cache[(size_t)env * kObsDim + c.lane];
        return;
    }
    c.env = env;
    c.base = (size_t)env * S.cap;
    c.dom = uniform_i32(S.scal[env].dom);
    if (first) {
        c.n = 0; c.n_elem = 0; c.n_new = 0; c.status = 0;
    } else {
        c.n = uniform_i32(S.scal[env].n);
        c.n_elem = uniform_i32(S.scal[env].n_elem);
        c.n_new = uniform_i32(S.scal[env].n_new);
        c.status = uniform_i32(S.scal[env].status);
    }
    const int n_old = c.n;
    reset_from_domain(c, S);
    if (is_static && c.lane == 1) c.obs = 0.0f;   // reset(static=True): row 0 carries 0 instead of the area ratio (C:1213-1218)
    if (nv_count && c.lane == 0) {
        nv_count[env] = 0;                    // self.not_valid_points = [], B:73
        nv_meta[(size_t)env * kNvMeta] += 1;  // a new episode: its generated vertices are new objects (last_not_valid_points, B:416)
    }
    if (c.lane == 0) {
        EnvCounters z;
        if (first) {
            z.last_change = 0; z.valid = 0; z.sum_n = 0; z.sum_n_valid = 0;
        } else {  // the ring length changes here: close the running term of sum_n
            z = S.cnt[env];
            z.sum_n += (unsigned long long)n_old * (step_now - z.last_change);
            z.last_change = step_now;
        }
        S.cnt[env] = z;
    }
    store_env(c, S);
    if (obs_out && c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
}

// One step() of every env (kMulti = false, meshenv_step) or n_steps consecutive steps in one launch
// (kMulti = true, meshenv_rollout).  Two instantiations on purpose: inside the multi-step loop the compiler hoists
// every lane predicate and table of the step body into the loop preheader (hundreds of instructions and SGPR
// spills that a single step would pay for nothing).
// The single by-value argument of k_step.  The one-step instantiation reads what its
// epilogue needs (output pointers, the ring arrays for the write-back) from there a second time, behind an opaque copy of
// the segment pointer, instead of keeping ~20 scalar registers alive -- or spilled to VGPR lanes -- across the whole step.
struct KStepArgs {
    DevState S;
    int cap, n_steps;
    const float *actions;
    float *obs_out;
    double *reward;
    uint8_t *done, *complete;
    float *term_obs;
    int auto_reset;
    unsigned long long step0;
};
template <bool kDefaultParams>
__device__ __forceinline__ KStepArgs late_kstep_args()
{
    typedef const __attribute__((address_space(4))) unsigned long long *qptr;
    qptr q = (qptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));   // not before this point
    static_assert(sizeof(KStepArgs) % 8 == 0, "whole quadwords");
    union { KStepArgs a; unsigned long long w[sizeof(KStepArgs) / 8]; } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(KStepArgs) / 8; i++) u.w[i] = q[i];
    if (kDefaultParams) apply_default_params(u.a.S.prm);
    return u.a;
}

// (five waves per SIMD = at most 96 VGPRs: with four the allocator drifts to 97 and the 65 536-env launch loses 7 % to the
// lost wave; tools/ab_all.sh)
#ifndef MESHENV_STEP_WAVES_PER_SIMD
#define MESHENV_STEP_WAVES_PER_SIMD 5
#endif
// (kTie: the handle has smoothed a front -- its rings hold vertices off the 1e-4 lattice and half-quantum angles, so the
// angles take the tie-breaking atan2 of the move() path, csrc/meshenv_geom.h; the host then steps with these instantiations)
// (kSmall: ring stride <= 64, as in step_group_body -- every ring pass is one 64-lane pass)
// (kPre: the exact early rejection of a rule-0 quad before the point-in-polygon pass (env_check, pre_reject) in the ONE-step
// kernel too.  The T-step kernel always had it; in one step per launch it is a throughput lever only -- the slowest chains of a
// small batch are valid steps, which only pay for the test: 512 / 1024 envs +1 %, 8192 neutral, 65 536 envs -5.4 %, mixed
// 32 768 -5.5 % (tools/ab_big.sh, tools/ab_misc.sh, round 4) -- so the host selects it together with record-first staging)
template <bool kMulti, bool kDefaultParams, bool kTie = false, bool kSmall = false, bool kPre = false>
__global__ void __launch_bounds__(64, MESHENV_STEP_WAVES_PER_SIMD)
k_step(const KStepArgs A)
{
    // (one by-value argument block: its layout IS the kernel-argument segment that late_kstep_args reads again)
    DevState S = A.S;
    const int cap = A.cap, n_steps = A.n_steps;
    if (kSmall) __builtin_assume(cap <= 64 && S.cap <= 64 && cap > 0 && S.cap > 0);   // no second chunk in the loads either
    const float *__restrict__ actions = A.actions;
    float *__restrict__ obs_out = A.obs_out;
    double *__restrict__ reward = A.reward;
    uint8_t *__restrict__ done = A.done, *__restrict__ complete = A.complete;
    float *__restrict__ term_obs = A.term_obs;
    (void)term_obs;   // (only the diagnostic / no-late-argument builds read it from here)
    int auto_reset = A.auto_reset;
    const unsigned long long step0 = A.step0;
    extern __shared__ double2 smem[];
    if (kDefaultParams) apply_default_params(S.prm);
    Ctx c;
    c.tie = kTie;
    carve_lds(c, smem, cap);
    const int env = blockIdx.x;
    const int E = S.n_envs;
    const float *a = actions + (size_t)env * 3;
    float a0 = a[0], a1 = a[1], a2 = a[2];
#ifdef MESHENV_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime();
#endif
    if (!kMulti && (auto_reset & 2)) {
        // Throughput regime (bit 1 of auto_reset, set by the host for >= 16 384 envs): the 64-byte record and the action
        // alone decide a step whose rule -1 / +1 quad is memoised as rejected (~45 % of the steps of a random policy at
        // steady state) -- such a wave never stages its ring: 28 n bytes and the whole load / LDS prologue saved, at the
        // price of one more dependent round trip for the others (hidden by the other waves of the SIMD at this occupancy;
        // at 4096 envs it would sit on the critical path, so the latency kernels do not do this).  Same values as the full
        // path: reward -1 / n_elem (B:248), failed_num + 1, the cached observation of the unchanged state.
        const EnvScalars s0 = S.scal[env];
        const int lane0 = lane_id();
        const float ob = lane0 < kObsDim ? S.obs_cache[(size_t)env * kObsDim + lane0] : 0.0f;
        const int st0 = uniform_i32(s0.status), failed0 = uniform_i32(s0.failed) + 1, ne0 = uniform_i32(s0.n_elem);
        const bool memo = (a0 <= -0.5f && (st0 & kStRm1Bad) != 0) || (a0 >= 0.5f && (st0 & kStRp1Bad) != 0);
        if (memo && uniform_i32(s0.ref) >= 0 && uniform_i32(s0.n) > 5 && failed0 < S.prm.fail_limit) {
            const double rw = ne0 ? -1.0 / ne0 : -1.0;
            if (lane0 < kObsDim) obs_out[(size_t)env * kObsDim + lane0] = ob;
            if (S.msg && lane0 < 21)
                S.msg[(size_t)env * 21 + lane0] = lane0 < kObsDim ? ob : (lane0 == 18 ? (float)rw : (lane0 == 19 ? 0.0f : 1.0f));
            if (lane0 == 0) {
                reward[env] = rw;
                done[env] = 0;
                complete[env] = 1;
                S.scal[env].failed = failed0;
            }
            return;
        }
    }
    // (same regime: the ring is staged without its candidate keys / stamps -- 12 of 32 B per slot -- which only the 11 % of
    // steps that extract an element read; those fetch them behind one more round trip, hidden like the one above)
    const bool light = !kMulti && (auto_reset & 4) != 0;
    auto_reset &= 1;
#ifndef MESHENV_STAMPS
    // (requested with the rest of the state in the T-step instantiation: no round trip at the end; the one-step
    // instantiation reads them in its epilogue, when the step changed the ring)
    EnvCounters cnt0;
    cnt0.last_change = 0; cnt0.valid = 0; cnt0.sum_n = 0; cnt0.sum_n_valid = 0;
    if (kMulti) cnt0 = S.cnt[env];
#endif
    if (light) load_env(c, S, env, false);
    else load_env(c, S, env);
    if (kSmall) __builtin_assume(c.n <= 64 && c.n >= 0);
#ifdef MESHENV_STAMPS
    const unsigned long long stamp_t1 = __builtin_amdgcn_s_memrealtime();
    if (c.lane < 16) c.sc->stamps[c.lane] = 0;
    wave_sync();
#endif
#ifdef MESHENV_STAMPS
    unsigned long long st_valid = 0;
#else
    EnvCounters k = cnt0;
    bool k_dirty = false;
#endif
    double last_reward = 0.0;
    int last_done = 0, last_complete = 0;
    const int T = kMulti ? n_steps : 1;
    for (int t = 0; t < T; t++) {
        if (kMulti) {
            // the lane id is made opaque once per iteration: everything derived from it (predicates, select tables) is
            // then recomputed inside the step instead of being hoisted out of the loop and spilled
            asm volatile("" : "+v"(c.lane));
        }
        if (kMulti && t > 0) {
            const float *at = actions + ((size_t)t * E + env) * 3;
            a0 = at[0]; a1 = at[1]; a2 = at[2];
        }
        if (kSmall) __builtin_assume(c.n <= 64 && c.n >= 0);   // (again per iteration: an extraction / a reset has rewritten n)
        const int n_before = c.n;
        // env_step, with the keys of a ring staged without them fetched between the checks and the update
        Decision d = env_check(c, S, a0, a1, a2, kMulti || kPre);
        if (d.ok) {
            if (light) load_keys(c, S);
            env_apply(c, S, d);
        }
        const StepResult r = env_finish(c, S.prm, d);
#ifndef MESHENV_STAMPS
        if (!kMulti) {   // one step per launch: the epilogue on freshly read arguments (see late_kstep_args)
            const KStepArgs L = late_kstep_args<kDefaultParams>();
            const bool l_auto = (L.auto_reset & 1) != 0;   // (bits 1, 2 of the argument select the staging mode)
            if (r.valid || (r.done && l_auto)) {
                k = L.S.cnt[env];   // (the work counters too: one step in nine needs them, and only here)
                const unsigned long long next = L.step0 + 1ULL;
                k.sum_n += (unsigned long long)n_before * (next - k.last_change);
                k.last_change = next;
                if (r.valid) { k.valid += 1ULL; k.sum_n_valid += (unsigned long long)n_before; }
                k_dirty = true;
            }
            if (c.lane == 0) {
                L.reward[env] = r.reward;
                L.done[env] = (uint8_t)r.done;
                L.complete[env] = (uint8_t)r.complete;
            }
            if (r.done) {
                if (L.term_obs && c.lane < kObsDim) L.term_obs[(size_t)env * kObsDim + c.lane] = c.obs;
                if (l_auto) reset_from_domain(c, L.S);
            }
            if (c.lane < kObsDim) L.obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
            if (L.S.msg && c.lane < 21) {
                const float v = c.lane < kObsDim ? c.obs : (c.lane == 18 ? (float)r.reward : (c.lane == 19 ? (float)r.done : (float)r.complete));
                L.S.msg[(size_t)env * 21 + c.lane] = v;
            }
            store_env(c, L.S);
            if (k_dirty && c.lane == 0) L.S.cnt[env] = k;
            return;
        }
#endif
#ifdef MESHENV_STAMPS
        if (r.valid) st_valid += 1;
#else
        if (r.valid || (r.done && auto_reset)) {  // the ring length changes after this step: close the running term
            const unsigned long long next = step0 + (unsigned long long)t + 1ULL;
            k.sum_n += (unsigned long long)n_before * (next - k.last_change);
            k.last_change = next;
            if (r.valid) { k.valid += 1ULL; k.sum_n_valid += (unsigned long long)n_before; }
            k_dirty = true;
        }
#endif
        const size_t o = (size_t)t * E + env;
        if (c.lane == 0) {
            reward[o] = r.reward;
            done[o] = (uint8_t)r.done;
            complete[o] = (uint8_t)r.complete;
        }
        last_reward = r.reward; last_done = r.done; last_complete = r.complete;
        if (r.done) {
#ifndef MESHENV_STAMPS
            // (T steps per launch: the end of an episode -- one step in ~25 -- reads what it needs afresh; the per-step
            // stores keep their pointers: re-reading those on every step of one wave's serial chain cost 11 %)
            const KStepArgs L = late_kstep_args<kDefaultParams>();
            if (L.term_obs && c.lane < kObsDim) L.term_obs[(size_t)env * kObsDim + c.lane] = c.obs;
            if (L.auto_reset & 1) reset_from_domain(c, L.S);
#else
            if (term_obs && c.lane < kObsDim) term_obs[(size_t)env * kObsDim + c.lane] = c.obs;
            if (auto_reset) reset_from_domain(c, S);
#endif
        }
    }
#ifndef MESHENV_STAMPS
    {   // the epilogue of the T-step launch on freshly read arguments (the one-step instantiation has returned above)
        const KStepArgs L = late_kstep_args<kDefaultParams>();
        if (c.lane < kObsDim) L.obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
        if (L.S.msg && c.lane < 21) {
            const float v = c.lane < kObsDim ? c.obs : (c.lane == 18 ? (float)last_reward : (c.lane == 19 ? (float)last_done : (float)last_complete));
            L.S.msg[(size_t)env * 21 + c.lane] = v;
        }
        store_env(c, L.S);
        if (k_dirty && c.lane == 0) L.S.cnt[env] = k;
        return;
    }
#endif
    if (c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
    if (S.msg && c.lane < 21) {
        const float v = c.lane < kObsDim ? c.obs : (c.lane == 18 ? (float)last_reward : (c.lane == 19 ? (float)last_done : (float)last_complete));
        S.msg[(size_t)env * 21 + c.lane] = v;
    }
    store_env(c, S);
#ifdef MESHENV_STAMPS
    // diagnostic build only: per-wave timeline (100 MHz realtime ticks) instead of the work counters
    if (c.lane == 0) {
        EnvCounters k;
        k.last_change = stamp_t0; k.valid = stamp_t1; k.sum_n = __builtin_amdgcn_s_memrealtime();
        k.sum_n_valid = st_valid | ((unsigned long long)(a0 <= -0.5f ? 1 : (a0 >= 0.5f ? 2 : 0)) << 8);
        S.cnt[env] = k;
    }
    if (c.lane == 0) c.sc->stamps[15] = __builtin_amdgcn_s_memtime() - stamp_c0;
    wave_sync();
    if (c.lane < 16) S.dbg[(size_t)env * 16 + c.lane] = c.sc->stamps[c.lane];
#else
    if (k_dirty && c.lane == 0) S.cnt[env] = k;
#endif
}

// ------------------------------------------------------------------------------------------ move() API

enum { kMoveOk = 0, kMoveNone = 1, kMoveRaises = 2, kMoveNeedsSmoothing = 3, kMoveSmoothRaises = 4 };

__host__ __device__ __forceinline__ size_t move_lds_bytes(int cap) { return lds_bytes_for(cap) + sizeof(double2) * (size_t)cap; }

// BoudaryEnv.move(new_point, type), B:265-432, for every env: one wavefront per env like k_step.  points[e] = (radius
// fraction, angle), types[e] the rule selector.  No reward (the reference returns 0), no current_area / failed_num
// bookkeeping; the reference vertex of a rejected move joins the env's not_valid_points (nv_xy / nv_count, staged in LDS)
// and is skipped by the selection; the observation is the static one.  code[e]: kMoveOk, kMoveNone (observation None),
// kMoveRaises (ring <= 5 or no reference vertex on entry: the reference raises UnboundLocalError; nothing changes),
// kMoveNeedsSmoothing (no selectable vertex left on a ring of more than 4: the reference runs smooth_pave -- not built
// -- and retries; here done = 1, complete = 0 and the caller resets).
__global__ void __launch_bounds__(64) k_move(DevState S, int cap, const double *__restrict__ points,
                                              const double *__restrict__ types, float *__restrict__ obs_out,
                                              uint8_t *__restrict__ done, uint8_t *__restrict__ complete,
                                              uint8_t *__restrict__ code, double2 *__restrict__ nv_xy,
                                              int32_t *__restrict__ nv_count, int32_t *__restrict__ nv_gid, const LibmRef lr)
{
    extern __shared__ double2 smem[];
    Ctx c;
    c.tie = true;   // smoothed fronts hold half-quantum angles
    carve_lds(c, smem, cap);
    double2 *nv = (double2 *)((char *)smem + lds_bytes_for(cap));
    const int env = blockIdx.x;
    const double mv_r = points[2 * (size_t)env], mv_a = points[2 * (size_t)env + 1], mv_type = types[env];
    int n_nv = uniform_i32(nv_count[env]);
    load_env(c, S, env);
    const int lane = c.lane;
    if (c.ref < 0 || c.n <= 5) {
        if (lane < kObsDim) obs_out[(size_t)env * kObsDim + lane] = c.obs;
        if (lane == 0) { done[env] = 0; complete[env] = 0; code[env] = (uint8_t)kMoveRaises; }
        return;
    }
    c.status &= ~(kStRm1Bad | kStRp1Bad);  // the memo of step() is tied to the reference vertex, which a move may change
    double2 *gnv = nv_xy + (size_t)env * cap;
    for (int k = lane; k < n_nv; k += 64) nv[k] = gnv[k];
    wave_sync();
    const double2 refpt = c.xy[c.ref];  // the reference vertex of this move, before any update
    const int refgid = c.id[c.ref];     // ... and its identity (the reference's list holds Vertex objects)
    Decision d = env_check(c, S, 0.0f, 0.0f, 0.0f, false, true, mv_r, mv_a, mv_type);
    if (d.ok) {
        env_apply(c, S, d, nullptr, true, nv, n_nv, lr);  // B:345: the selection still sees the old not_valid_points
        n_nv = 0;                                     // B:365
    } else {
        // B:359-361: `reference_point not in not_valid_points` is object identity (Vertex defines no __eq__): the test is
        // on the vertex ids -- a coincident twin of a listed vertex (self-touching fronts have them) is a different object
        // and is listed too
        bool listed = false;
        const int32_t *gids = nv_gid + (size_t)env * cap;
        for (int k = lane; k < n_nv; k += 64) listed = listed || gids[k] == refgid;
        if (__ballot(listed) == 0ULL) {
            if (lane == 0) {
                nv[n_nv] = refpt;
                gnv[n_nv] = refpt;
                nv_gid[(size_t)env * cap + n_nv] = refgid;
            }
            n_nv += 1;
        }
        wave_sync();
        BqArgs bq;
        bq.skip = false;
        bq.mode = 0; bq.a = 0; bq.b = 0; bq.ang0 = 0; bq.ang1 = 0; bq.q_ang0 = 0; bq.q_ang2 = 0; bq.half01 = 0; bq.half23 = 0;
        find_next_state(c, S, bq, true, nv, n_nv, lr);
        c.ring_dirty = true;  // reference vertex, base length, action frame and observation moved: full record write-back
    }
    // (a rejected move leaves the ring arrays as they are in HBM: only the record and the cached observation go back --
    // until round 4 the 32 B per slot of an unchanged ring were rewritten by 80 % of the moves, 2.5x the kernel's write traffic)
    const bool ring_unchanged = !d.ok;
    const bool none = c.ref < 0;
    const int cd = none ? (c.n > 4 ? kMoveNeedsSmoothing : kMoveNone) : kMoveOk;
    if (lane < kObsDim) obs_out[(size_t)env * kObsDim + lane] = c.obs;
    if (lane == 0) {
        done[env] = (uint8_t)((d.done || cd == kMoveNeedsSmoothing) ? 1 : 0);
        complete[env] = (uint8_t)(c.n <= 4 ? 1 : 0);   // B:403-432
        code[env] = (uint8_t)cd;
        nv_count[env] = n_nv;
    }
    store_env(c, S, ring_unchanged);
}

// ------------------------------------------------------------------------------------------ CU-group step kernel

// What a wavefront leaves in LDS for the wavefront that will run the update of its environment.
struct alignas(32) Handoff {
    int valid;  // the checks passed: an update is pending
    int simd;   // hardware SIMD the checking wave runs on
    int upd_done;     // phase 2: the update wave has written the post-update ring (the helper may read it)
    int helper_done;  // phase 2: the helper has finished reading the ring (an auto-reset may overwrite it)
    int env, n, ref, n_elem, failed, n_new, counter, status, dom;
    int lds_off, lds_cap;   // ragged packing: byte offset and slot count of the env's LDS region (step_group_body)
    int pad;
    double bl, area, ct, st;
    EnvCounters cnt0;
    Decision d;
};

__host__ __device__ __forceinline__ size_t group_lds_bytes(int cap, int G)
{
    return (size_t)G * (lds_bytes_for(cap) + sizeof(Handoff));
}

// Where a step's results go.  In the CU-group kernel these five pointers are read from the kernel-argument segment at
// the END of the step (late_outs): loaded at kernel entry they would be spilled and reloaded by every wave.
struct StepOuts {
    float *obs_out;
    double *reward;
    uint8_t *done;
    uint8_t *complete;
    float *term_obs;
};

// the single by-value argument of k_step_group: its layout IS the kernel-argument segment
struct GroupArgs {
    DevState S;
    StepOuts outs;
    const float *actions;
    unsigned long long step0;  // index of this step since the handle was created (lazy work counters)
    int cap;
    int auto_reset;
    // Ragged LDS packing (nullptr = every env owns lds_bytes_for(cap) bytes): [n_envs] (byte offset of the env's region in
    // its workgroup's LDS, ring slots it holds).  A batch of mixed domains -- d1 / d2 / d3: 120 / 196 / 272 vertices --
    // does not fit sixteen rings of the LONGEST stride into one CU's LDS, but it does fit sixteen rings of their own lengths.
    const int2 *env_lds;
    int ho_off;   // ragged: byte offset of the hand-over blocks (behind the largest workgroup's rings)
    int pad;
};

// tstep / n: the T-steps-per-launch closed loop (csrc/meshenv_fused.h) writes step t of the launch into slice t of
// [T][n]-shaped output histories; every other kernel passes the literal 0 and compiles to what it was.
typedef const __attribute__((address_space(4))) char *KernArgPtr;
__device__ __forceinline__ StepOuts late_outs(const int tstep = 0, const int n_envs = 0, KernArgPtr ka_in = nullptr)
{
    typedef const __attribute__((address_space(4))) char *kptr;
    // (ka_in: a callee has no kernarg pointer of its own -- the out-of-line step of the T-step kernel gets it handed down)
    kptr ka = ka_in ? ka_in : (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));  // not before this point
    typedef const __attribute__((address_space(4))) unsigned long long *qptr;
    qptr q = (qptr)(ka + offsetof(GroupArgs, outs));
    static_assert(sizeof(StepOuts) == 5 * sizeof(unsigned long long), "StepOuts is five pointers");
    StepOuts o;
    o.obs_out = (float *)q[0];
    o.reward = (double *)q[1];
    o.done = (uint8_t *)q[2];
    o.complete = (uint8_t *)q[3];
    o.term_obs = (float *)q[4];
    if (tstep != 0) {
        const size_t off = (size_t)tstep * (size_t)n_envs;
        o.obs_out += off * kObsDim;
        o.reward += off;
        o.done += off;
        o.complete += off;
        if (o.term_obs) o.term_obs += off * kObsDim;
    }
    return o;
}

// The DevState at the head of the kernel-argument segment (GroupArgs and every argument block built on it), read again behind
// an opaque copy of the segment pointer: the write-back of a step needs the ring arrays, the record and counter arrays and
// the domain table, none of which the checks or the update in between touch -- re-reading them costs two scalar loads,
// holding them costs scalar registers (or VGPR lanes) on every path.  prm comes from the caller's copy (literals).
__device__ __forceinline__ DevState late_state(const DevState &S, KernArgPtr ka_in = nullptr)
{
    typedef const __attribute__((address_space(4))) unsigned long long *qptr;
    qptr q = (qptr)(ka_in ? ka_in : (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr());
    asm volatile("" : "+s"(q));   // not before this point
    constexpr unsigned kWords = offsetof(DevState, prm) / 8;
    static_assert(offsetof(DevState, prm) % 8 == 0, "whole quadwords before prm");
    union { DevState s; unsigned long long w[sizeof(DevState) / 8]; } u;
    u.s = S;
#pragma unroll
    for (unsigned i = 0; i < kWords; i++) u.w[i] = q[i];
    return u.s;
}

// reward / flags / observation of one finished step, auto-reset, state write-back, work counters
// (helper_done != nullptr: the reward of this valid step is written by a helper wavefront, which also reads the ring:
// an auto-reset waits for it before it overwrites the ring)
__device__ __forceinline__ void finish_and_store(Ctx &c, const DevState &S_in, const Decision &d, const EnvCounters &cnt0,
                                                 int n_before, int auto_reset, unsigned long long step0,
                                                 volatile int *helper_done = nullptr, float *actor_row = nullptr, const int tstep = 0,
                                                 KernArgPtr ka = nullptr, const bool cnt_late = false)
{
    const StepResult r = env_finish(c, S_in.prm, d);
    const int env = c.env;
    const DevState S = late_state(S_in, ka);
    const StepOuts o = late_outs(tstep, S.n_envs, ka);
    float *__restrict__ obs_out = o.obs_out;
    double *__restrict__ reward = o.reward;
    uint8_t *__restrict__ done = o.done;
    uint8_t *__restrict__ complete = o.complete;
    float *__restrict__ term_obs = o.term_obs;
    const bool has_helper = helper_done != nullptr;
    if (c.lane == 0) {
        if (!has_helper) reward[env] = r.reward;
        done[env] = (uint8_t)r.done;
        complete[env] = (uint8_t)r.complete;
    }
    if (r.done) {
        if (term_obs && c.lane < kObsDim) term_obs[(size_t)env * kObsDim + c.lane] = c.obs;
        if (auto_reset) {
            if (has_helper) {
                spin_until_nonzero(helper_done);  // the helper sets it unconditionally
                wave_sync();
            }
            reset_from_domain(c, S);
        }
    }
    if (c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
    // (k_step_group_actor: the same observation straight into the actor's LDS input row, k-permuted as its first layer
    // reads it -- position (k & 3) * 8 + (k >> 2), csrc/meshenv_actor.h)
    if (actor_row && c.lane < kObsDim) actor_row[(c.lane & 3) * 8 + (c.lane >> 2)] = c.obs;
    if (S.msg && c.lane < 21 && !(has_helper && c.lane == 18)) {  // the exchange message of the multi-GPU path
        const float v = c.lane < kObsDim ? c.obs : (c.lane == 18 ? (float)r.reward : (c.lane == 19 ? (float)r.done : (float)r.complete));
        S.msg[(size_t)env * 21 + c.lane] = v;
    }
    store_env(c, S, has_helper && !r.done);
    if ((r.valid || (r.done && auto_reset)) && c.lane == 0) {  // the ring length changes after this step
        // (cnt_late: a rejected step of the CU-group kernel -- its counters only move when the episode is truncated, so
        // they are read here instead of being held, unused, through the checks of every wave)
        EnvCounters k = cnt_late ? S.cnt[env] : cnt0;
        k.sum_n += (unsigned long long)n_before * (step0 + 1ULL - k.last_change);
        k.last_change = step0 + 1ULL;
        if (r.valid) { k.valid += 1ULL; k.sum_n_valid += (unsigned long long)n_before; }
        S.cnt[env] = k;
    }
}

// One step() of every env with G environments (wavefronts) per workgroup, one workgroup per CU-sized slice.
//
// Why: ~12 % of the steps extract an element and take ~4x longer than a rejected action; with one wave per
// workgroup those long waves land on SIMDs at random, and the launch ends with the SIMD that happens to hold three or
// four of them (measured: 12.2 us alone on a SIMD, 13.3 / 14.8 / 16.8 us with 2 / 3 / 4 -- profiles/, tools/stamps_timeline.py).
// Here every wave runs the CHECKS of its own env (phase 1), the workgroup meets at one barrier, and the pending
// UPDATES are dealt round-robin over the four SIMDs of the CU (phase 2): wave (simd s, r-th on that SIMD) takes the
// (4r+s)-th pending env of the group.  The env's ring never moves -- it is already in the workgroup's LDS -- only
// ~40 scalars are handed over (Handoff).  Environments stay independent: no data is shared between envs.
// (the body as a function: k_step_group is just this; k_step_group_actor, csrc/meshenv_fused.h, appends the policy's forward)
// (actor_in: k_step_group_actor only -- LDS [G][132] floats, the actor's first-layer input; nullptr otherwise)
// (kRagged: the workgroup's LDS is packed by each env's own ring length, GroupArgs::env_lds -- batches of mixed domains only;
// the uniform instantiation holds no offset table, no per-env region lookup and no Handoff fields for them)
// (kSmall: the batch's ring stride is at most 64 slots, so every ring fits one 64-lane pass: the compiler is told n <= 64 and
// drops the chunk loops of every ring pass and the multi-chunk forms behind them -- 8864 -> 7621 static instructions)
template <int G, bool kDefaultParams, bool kRagged = false, bool kSmall = false>
__device__ __forceinline__ void step_group_body(const GroupArgs &A, float *actor_in = nullptr, const int tstep = 0,
                                                KernArgPtr ka = nullptr, const int tid_in = -1)
{
    // (tid_in: the T-step kernel's per-iteration opaque thread id, so that nothing derived from it is hoisted out of its loop)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    extern __shared__ double2 smem[];
    DevState S = A.S;
    const int cap = A.cap, auto_reset = A.auto_reset;
    if (kSmall) __builtin_assume(cap <= 64 && S.cap <= 64 && cap > 0 && S.cap > 0);   // no second chunk in the loads either
    const float *__restrict__ actions = A.actions;
    if (kDefaultParams) apply_default_params(S.prm);
    const int wave = uniform_i32(tid >> 6);  // wave-uniform by construction: keeps env and every address derived from it in SGPRs
    const size_t env_bytes = lds_bytes_for(cap);
    Handoff *ho = (Handoff *)((char *)smem + (kRagged ? (size_t)A.ho_off : (size_t)G * env_bytes));
    const int env = blockIdx.x * G + wave;
    const bool active = env < S.n_envs;
    SPEC_TIME(env, 8);
    int pending = 0;
#ifdef MESHENV_STAMPS
    // diagnostic build (tools/group_timeline.py, tools/group_phase2_timeline.py): dbg[env][0..5] describe the wave
    // that owns env in phase 1, dbg[env][6..15] the phase-2 update of env (whichever wave ran it)
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (active) {
        Ctx c;
        int my_off = (int)((size_t)wave * env_bytes), my_cap = cap;
        if (kRagged) {   // wave-uniform: one scalar load, requested with the state
            const int2 e = A.env_lds[env];
            my_off = uniform_i32(e.x);
            my_cap = uniform_i32(e.y);
        }
        carve_lds(c, (char *)smem + my_off, my_cap);
        const float *a = actions + (size_t)env * 3;
        const float a0 = a[0], a1 = a[1], a2 = a[2];
        EnvCounters cnt0;
        cnt0.last_change = 0; cnt0.valid = 0; cnt0.sum_n = 0; cnt0.sum_n_valid = 0;
        constexpr bool kCntLate = true;
        load_env(c, S, env, true, tid_in >= 0 ? (tid & 63) : -1, kRagged ? my_cap : 0);
        if (kSmall) __builtin_assume(c.n <= 64 && c.n >= 0);
        const int n_before = c.n;
        Decision d = env_check(c, S, a0, a1, a2, false, false, 0.0, 0.0, 0.5, nullptr, true);
        if (!d.ok) {
            finish_and_store(c, S, d, cnt0, n_before, auto_reset, A.step0, nullptr, actor_in ? actor_in + wave * 132 : nullptr, tstep, ka, kCntLate);
        } else {
            pending = 1;
            if (c.lane == 0) {
                Handoff &h = ho[wave];
                h.env = env; h.n = c.n; h.ref = c.ref; h.n_elem = c.n_elem; h.failed = c.failed; h.n_new = c.n_new;
                h.counter = c.counter; h.status = c.status; h.dom = c.dom;
                if (kRagged) { h.lds_off = my_off; h.lds_cap = my_cap; }
                h.bl = c.bl; h.area = c.area; h.ct = c.ct; h.st = c.st;
                h.d = d;
                h.upd_done = 0;
                h.helper_done = 0;
            }
        }
    }
    if ((tid & 63) == 0) {
        ho[wave].valid = pending;
        ho[wave].simd = (int)__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | (1 << 11));  // HW_REG_HW_ID.simd_id
    }
#ifdef MESHENV_STAMPS
    const unsigned long long dbg_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
#ifdef MESHENV_STAMPS
    const unsigned long long dbg_t2 = __builtin_amdgcn_s_memrealtime();
    if (active && (tid & 63) == 0) {
        unsigned long long *o = S.dbg + (size_t)env * 16;
        o[0] = dbg_t0; o[1] = dbg_t1; o[2] = dbg_t2; o[3] = (unsigned long long)pending; o[4] = 0; o[5] = 0;
#ifdef MESHENV_DBG_P1SET   // two of the check's stage stamps per build (tools/phase1_stages.py): 0 = decode, ring pass; 1 = quad, intersections
        {
            Ctx cs;
            carve_lds(cs, (char *)smem + (kRagged ? A.env_lds[env].x : (int)((size_t)wave * env_bytes)), kRagged ? A.env_lds[env].y : cap);
            o[4] = cs.sc->stamps[MESHENV_DBG_P1SET ? 4 : 1];
            o[5] = cs.sc->stamps[MESHENV_DBG_P1SET ? 5 : 2];
        }
#endif
        if (!pending) o[15] = 0;
    }
#endif

    // ---- deal the pending updates over the SIMDs (lane w reads wave w's entry: five ballots, not a 16-step loop)
    const int dl = tid & 63;
    int hv = 0, hs = -1;
    if (dl < G) {
        hv = ho[dl].valid;
        hs = ho[dl].simd & 3;
    }
    const unsigned vmask = (unsigned)__ballot(hv != 0);
    const int m = __popc(vmask);
    if (m == 0) return;
    const unsigned sm0 = (unsigned)__ballot(hs == 0), sm1 = (unsigned)__ballot(hs == 1);
    const unsigned sm2 = (unsigned)__ballot(hs == 2), sm3 = (unsigned)__ballot(hs == 3);
    const unsigned wbit = 1u << wave;
    const int my_simd = (sm1 & wbit) ? 1 : (sm2 & wbit) ? 2 : (sm3 & wbit) ? 3 : 0;
    const unsigned mine = my_simd == 0 ? sm0 : my_simd == 1 ? sm1 : my_simd == 2 ? sm2 : sm3;
    const bool balanced = __popc(sm0) * 4 == G && __popc(sm1) * 4 == G && __popc(sm2) * 4 == G && __popc(sm3) * 4 == G;
    // roles: deal index j < m -> update of the j-th pending env; m <= j < 2m -> reward helper of the (j-m)-th one (only
    // when every update gets a helper: 2m <= G).  The helper runs concurrently on another wavefront, usually an idle SIMD.
    int src = -1, hsrc = -1;
#ifdef MESHENV_NO_HELPER
    const bool helpers = false;
#else
    const bool helpers = balanced && G >= 4 && 2 * m <= G && m <= 8;
#endif
    if (balanced && G >= 4) {
        const int r = __popc(mine & ((1u << wave) - 1u));  // my rank among the waves of my SIMD
        const int j = r * 4 + my_simd;                     // my turn in the deal
        int k = j < m ? j : -1;
        if (helpers && j >= m) {
            // Helpers go to the SIMDs that carry fewer updates.  SIMD s runs ceil((m - s) / 4) updates; the number of
            // helpers it takes, per m = 1..8 (nibble s of the entry): an update counts ~3 helper units, and
            // updates + helpers <= 4 waves per SIMD (G = 16).  With G = 8 (two waves per SIMD) m <= 4 and the entries
            // still fit.  Which helper serves which pending env does not matter: they are numbered SIMD-major.
            // (G = 8: m = 3 -> [0,0,1,2], two waves per SIMD)
            const unsigned long long t_lo = G >= 16 ? 0x1111300011000010ULL : 0x1111210011000010ULL;  // m = 1..4
            const unsigned long long t_hi = 0x2222321133001220ULL;                                    // m = 5..8
            const unsigned hcw = (unsigned)(((m <= 4 ? t_lo : t_hi) >> (16 * ((m - 1) & 3))) & 0xffffULL);
            const int mains_here = (m + 3 - my_simd) >> 2;
            const int t = r - mains_here;
            const int here = (int)((hcw >> (4 * my_simd)) & 15u);
            const int below = my_simd == 0 ? 0 : (int)((hcw & 15u) + (my_simd > 1 ? ((hcw >> 4) & 15u) : 0u) + (my_simd > 2 ? ((hcw >> 8) & 15u) : 0u));
            if (t >= 0 && t < here) k = below + t;
        }
        if (k >= 0) {
            unsigned rest = vmask;
            for (int t = 0; t < k; t++) rest &= rest - 1u;  // drop the k lowest pending waves
            const int which = __ffs((int)rest) - 1;
            if (j < m) src = which;
            else hsrc = which;
        }
    } else if (vmask & (1u << wave)) {
        src = wave;  // waves not spread evenly over the SIMDs: everybody keeps its own env
    }
    if (src < 0 && hsrc < 0) return;

    if (hsrc >= 0) {
        // ---- phase 2, helper: the reward of env ho[hsrc]
        Ctx c;
        Handoff &h = ho[hsrc];
        if (kRagged) carve_lds(c, (char *)smem + uniform_i32(h.lds_off), uniform_i32(h.lds_cap));
        else carve_lds(c, (char *)smem + (size_t)hsrc * env_bytes, cap);
        c.lane = tid & 63;
        Decision d = h.d;
        d.index = uniform_i32(d.index); d.new_vertex = uniform_i32(d.new_vertex);
        d.t0 = uniform_i32(d.t0); d.t1 = uniform_i32(d.t1);
        const int henv = uniform_i32(h.env);
        const double rew = reward_on_helper(c, S, d, uniform_i32(h.n), uniform_i32(h.dom), &h.upd_done);
        wave_sync();
        if (c.n > 5) {  // not the end of the episode: the ring is final, write it back here (off the update wave's path)
            c.env = henv;
            c.base = (size_t)henv * S.cap;
            store_ring(c, late_state(S, ka));
        }
        if (c.lane == 0) {
            *(volatile int *)&h.helper_done = 1;
            const StepOuts o = late_outs(tstep, S.n_envs, ka);
            o.reward[henv] = rew;
            if (S.msg) S.msg[(size_t)henv * 21 + 18] = (float)rew;
        }
        return;
    }

    // ---- phase 2: the update of env ho[src], in place in its LDS region
    Ctx c;
    Handoff &h = ho[src];
    if (kRagged) carve_lds(c, (char *)smem + uniform_i32(h.lds_off), uniform_i32(h.lds_cap));
    else carve_lds(c, (char *)smem + (size_t)src * env_bytes, cap);
    c.lane = tid & 63;
    c.env = uniform_i32(h.env);
    c.base = (size_t)c.env * S.cap;
    c.n = uniform_i32(h.n); c.ref = uniform_i32(h.ref); c.n_elem = uniform_i32(h.n_elem); c.failed = uniform_i32(h.failed);
    if (kSmall) __builtin_assume(c.n <= 64 && c.n >= 0);
    c.n_new = uniform_i32(h.n_new); c.counter = uniform_i32(h.counter); c.status = uniform_i32(h.status); c.dom = uniform_i32(h.dom);
    c.bl = uniform_f64(h.bl); c.area = uniform_f64(h.area); c.ct = uniform_f64(h.ct); c.st = uniform_f64(h.st);
    c.ring_dirty = false;
    c.obs = 0.0f;
    Decision d = h.d;
    d.index = uniform_i32(d.index); d.new_vertex = uniform_i32(d.new_vertex);
    d.mp0 = uniform_i32(d.mp0); d.mp1 = uniform_i32(d.mp1); d.mp2 = uniform_i32(d.mp2); d.mp3 = uniform_i32(d.mp3);
    d.p0 = uniform_i32(d.p0); d.p1 = uniform_i32(d.p1); d.p2 = uniform_i32(d.p2); d.p3 = uniform_i32(d.p3);
    d.t0 = uniform_i32(d.t0); d.t1 = uniform_i32(d.t1); d.lo = uniform_i32(d.lo); d.hi = uniform_i32(d.hi);
    d.ok = 1;
    const EnvCounters cnt0 = S.cnt[c.env];   // requested now, used after the update: the round trip hides behind it
    const int n_before = c.n;
    env_apply(c, S, d, helpers ? &h.upd_done : nullptr);
#ifdef MESHENV_STAMPS
    const unsigned long long dbg_t8 = __builtin_amdgcn_s_memrealtime();
#endif
    finish_and_store(c, S, d, cnt0, n_before, auto_reset, A.step0, helpers ? &h.helper_done : nullptr,
                     actor_in ? actor_in + src * 132 : nullptr, tstep, ka);
#ifdef MESHENV_STAMPS
    if (c.lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        unsigned long long *o = S.dbg + (size_t)c.env * 16;
        o[6] = c.sc->stamps[6]; o[7] = dbg_t2; o[8] = dbg_t8;
        for (int k = 9; k <= 14; k++) o[k] = c.sc->stamps[k];
        o[15] = t_end;
        if (active) {
            unsigned long long *w = S.dbg + (size_t)env * 16;
            w[4] = t_end; w[5] = (unsigned long long)(src + 1);
        }
    }
#endif
}


template <int G, bool kDefaultParams, bool kRagged = false, bool kSmall = false>
__global__ void __launch_bounds__(64 * G)
k_step_group(GroupArgs A)
{
    step_group_body<G, kDefaultParams, kRagged, kSmall>(A);
}

// ------------------------------------------------------------------------------------------ speculative CU-group kernel


__host__ __device__ __forceinline__ size_t spec_lds_bytes(int cap, int G)
{
    return (size_t)G * (2 * lds_bytes_for(cap) + sizeof(SpecJob)) + 64;  // + n_nopost, poller[4]
}

// The extraction of env slot e of the workgroup on a COPY (buffer B) of its ring, started while the checks of that env
// still run on the original (buffer A): quad job lanes, ring update, candidate patch, next observation.  Nothing leaves
// the wavefront before the checking wave's verdict; on "commit" this wave writes the step's results (everything but the
// reward, which the checking wave computes from its own quad values and buffer B).
template <bool kDefaultParams>
__device__ __forceinline__ void spec_update(SpecJob *job, void *region_a, void *region_b, const DevState &S, int cap,
                                            int auto_reset, unsigned long long step0)
{
    wave_sync();  // (compiler) nothing of the job record is read before the claim
    __builtin_amdgcn_s_setprio(3);  // the extraction is the longest dependent chain of the launch
    SPEC_COUNT(1);
    SPEC_TIME(job->env, 2);
    Ctx a, c;
    carve_lds(a, region_a, cap);
    carve_lds(c, region_b, cap);
    c.lane = threadIdx.x & 63;
    const int lane = c.lane;
    c.env = uniform_i32(job->env);
    c.base = (size_t)c.env * S.cap;
    const EnvScalars rec = S.scal[c.env];      // requested now, consumed after the ring copy and the quad stage
    const EnvCounters cnt0 = S.cnt[c.env];
    c.n = uniform_i32(job->n);
    c.ring_dirty = false;
    c.obs = 0.0f;
    const int n = c.n;
    for (int i = lane; i < n; i += 64) {
        c.xy[i] = a.xy[i];
        c.key[i] = a.key[i];
        c.stamp[i] = a.stamp[i];
        c.id[i] = a.id[i];
    }
    const int what = uniform_i32(job->what);
    const bool new_vertex = (what & 1) != 0;
    const int rule = ((what >> 1) & 3) - 1, index = what >> 3;
    const P2 new_point = mkp(uniform_f64(job->npx), uniform_f64(job->npy));
    const QuadPlan qp = plan_quad(new_vertex, rule, index, n, new_point);
    wave_sync();
    if (lane < 4) {
        const int mp = lane == 0 ? qp.mp0 : lane == 1 ? qp.mp1 : lane == 2 ? qp.mp2 : qp.mp3;
        c.sc->q[lane] = mp < 0 ? make_double2(new_point.x, new_point.y) : c.xy[mp];
    }
    wave_sync();
    quad_pass(c, S.prm, qp.vr, qp.p0, qp.p1, qp.p2, qp.p3, qp.bc0, qp.bc1, false, 0.0, 0.0, true);
    if (*(volatile int *)&job->verdict == 2) { SPEC_COUNT(2); return; }
    c.ref = uniform_i32(rec.ref); c.n_elem = uniform_i32(rec.n_elem); c.failed = uniform_i32(rec.failed);
    c.n_new = uniform_i32(rec.n_new); c.counter = uniform_i32(rec.counter); c.status = uniform_i32(rec.status);
    c.dom = uniform_i32(rec.dom);
    c.bl = uniform_f64(rec.bl); c.area = uniform_f64(rec.area); c.ct = uniform_f64(rec.ct); c.st = uniform_f64(rec.st);
    c.status &= ~(kStRm1Bad | kStRp1Bad);
    Decision d;
    d.ok = 1; d.new_vertex = new_vertex ? 1 : 0; d.index = index;
    d.mp0 = qp.mp0; d.mp1 = qp.mp1; d.mp2 = qp.mp2; d.mp3 = qp.mp3;
    d.p0 = qp.p0; d.p1 = qp.p1; d.p2 = qp.p2; d.p3 = qp.p3;
    d.t0 = qp.t0; d.t1 = qp.t1; d.lo = qp.vr.lo; d.hi = qp.vr.hi;
    d.done = 0; d.no_reference = 0; d.reward = 0.0;
    d.new_point = new_point;
    env_apply(c, S, d, &job->upd_done);
    SPEC_TIME(c.env, 3);
    const int verdict = spin_until_nonzero(&job->verdict);
    if (verdict != 1) { SPEC_COUNT(3); return; }
    SPEC_COUNT(4);
    finish_and_store(c, S, d, cnt0, n, auto_reset, step0, &job->helper_done);
    SPEC_TIME(c.env, 5);
}

// One step() of every env, G environments (wavefronts) per workgroup, one workgroup per CU, WITHOUT a workgroup barrier:
//   * every wave runs the checks of its own env on the ring it staged in LDS (buffer A).  An action that survives the
//     cheap exact tests (rule -1 / +1: not memoised as rejected, corner-sign test; new vertex: corner-sign test, before
//     the point-in-polygon pass) is POSTED;
//   * a wave whose own action has been rejected (88 % of them, most within 2-3 us) polls the workgroup's posts, claims one
//     and runs that env's extraction speculatively on a copy of the ring (spec_update) while the owner's checks go on;
//   * the owner's verdict commits or discards it.  On commit the update wave writes the state and the outputs, the owner
//     computes the reward (from its own quad values and the updated copy).  A post nobody claimed is withdrawn by its
//     owner, which then runs the extraction itself.
// The dependent chain of a valid extraction is then  load + cheap tests + extraction  instead of  load + all checks +
// barrier + extraction  (k_step_group), and no wave waits for the slowest check of its workgroup.
template <int G, bool kDefaultParams>
__global__ void __launch_bounds__(64 * G)
k_step_spec(GroupArgs A)
{
    extern __shared__ double2 smem[];
    DevState S = A.S;
    const int cap = A.cap, auto_reset = A.auto_reset;
    const float *__restrict__ actions = A.actions;
    if (kDefaultParams) apply_default_params(S.prm);
    const int wave = uniform_i32((int)(threadIdx.x >> 6));
    const size_t env_bytes = lds_bytes_for(cap);
    char *buf_a = (char *)smem, *buf_b = buf_a + (size_t)G * env_bytes;
    SpecJob *jobs = (SpecJob *)(buf_b + (size_t)G * env_bytes);
    volatile int *n_nopost = (volatile int *)(jobs + G);  // owners that have posted, or have finished without posting
    const int env = blockIdx.x * G + wave;
    const bool active = env < S.n_envs;
    const int lane = threadIdx.x & 63;
    const unsigned long long t_entry = SPEC_NOW();
    // This wave's state is requested from HBM first: nothing may precede these loads that the compiler would take for a
    // store to global memory (an asm statement, with or without a memory clobber, is one) -- the wave-uniform loads (env
    // record, counters, action) are only selected as scalar loads while the kernel is provably clean.
    const int env_ld = active ? env : 0;
    const float *a_ptr = actions + (size_t)env_ld * 3;
    const float a0 = a_ptr[0], a1 = a_ptr[1], a2 = a_ptr[2];
    const EnvCounters cnt0 = S.cnt[env_ld];
    const EnvLoad ld = load_env_issue(S, env_ld);
    if (threadIdx.x < G) {
        *(volatile int *)&jobs[threadIdx.x].state = 0;
        *(volatile int *)&jobs[threadIdx.x].verdict = 0;
    }
    if (threadIdx.x == 0) *n_nopost = 0;
    // The job board is clean before anybody posts or polls: the only workgroup barrier, at t = 0.  In assembly, because
    // __syncthreads() would also drain the vector loads just issued (lgkmcnt covers the LDS stores and the scalar loads,
    // which return with the ring data anyway).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SPEC_TIME_AT(env, 8, t_entry);
    SPEC_TIME(env, 9);

    if (active) {
        Ctx c;
        carve_lds(c, buf_a + (size_t)wave * env_bytes, cap);
        load_env_commit(c, S, env, ld);
        const int n_before = c.n;
        SPEC_TIME(env, 0);
        SpecJob *job = jobs + wave;
        SpecHook hook;
        hook.job = job; hook.n_nopost = (int *)n_nopost; hook.posted = false; hook.stale = false;
        Decision d = env_check(c, S, a0, a1, a2, true, false, 0.0, 0.0, 0.5, &hook);
        // settle the post: 1 -> 0 withdraws it, 2 means an update wave is on it
        bool claimed = false;
        if (hook.posted) {
            int old = 0;
            if (lane == 0) old = atomicCAS((int *)&job->state, 1, 0);
            claimed = uniform_i32(old) == 2;
        }
        const bool commit = claimed && d.ok && !hook.stale;
        SPEC_TIME(env, 4);
        if (d.ok) SPEC_COUNT(5);
        if (d.ok && !commit) SPEC_COUNT(6);
        if (claimed) {
            wave_sync();
            if (lane == 0) *(volatile int *)&job->verdict = commit ? 1 : 2;
        }
        if (!hook.posted && lane == 0) atomicAdd((int *)n_nopost, 1);
        if (commit) {
            // the reward, from this wave's quad values (scratch of buffer A) and the updated ring (buffer B)
            Ctx cb;
            carve_lds(cb, buf_b + (size_t)wave * env_bytes, cap);
            cb.sc = c.sc;
            cb.lane = lane;
            const double rew = reward_on_helper(cb, S, d, n_before, c.dom, &job->upd_done);
            wave_sync();
            if (cb.n > 5) {  // not the end of the episode: the ring is final, written back here (off the update wave's path)
                cb.env = env;
                cb.base = (size_t)env * S.cap;
                store_ring(cb, S);
            }
            if (lane == 0) {
                *(volatile int *)&job->helper_done = 1;
                const StepOuts o = late_outs();
                o.reward[env] = rew;
                if (S.msg) S.msg[(size_t)env * 21 + 18] = (float)rew;
            }
            SPEC_TIME(env, 7);
        } else {
            if (d.ok) env_apply(c, S, d);  // nobody took the post (or there was none): the extraction in place, as k_step does
            finish_and_store(c, S, d, cnt0, n_before, auto_reset, A.step0);
            SPEC_TIME(env, 6);
        }
    } else if (lane == 0) {
        atomicAdd((int *)n_nopost, 1);
    }

    __builtin_amdgcn_s_setprio(0);
    // ---- this wave's own env is done: serve the workgroup's posts until no owner can post any more.  Every idle wave
    //      polls, slowly (~1000 cycles apart: a poll costs issue slots of the SIMD's working waves); a dozen pollers out of
    //      phase still pick a post up within a fraction of a microsecond.
    for (int it = 0; it < (1 << 20); it++) {
        const int settled = *n_nopost;  // read BEFORE the states: a post precedes its owner's count
        int st = 0;
        if (lane < G) st = *(volatile int *)&jobs[lane].state;
        const unsigned posted = (unsigned)__ballot(st == 1);
        if (posted != 0u) {
            // spread the claimers: start looking at this wave's own slot
            const unsigned rot = (posted >> wave) | (posted << (G - wave));
            const int pick = (wave + (__ffs((int)(rot & ((1u << G) - 1u))) - 1)) % G;
            int old = 0;
            if (lane == 0) old = atomicCAS((int *)&jobs[pick].state, 1, 2);
            if (uniform_i32(old) == 1) {
                spec_update<kDefaultParams>(jobs + pick, buf_a + (size_t)pick * env_bytes, buf_b + (size_t)pick * env_bytes, S, cap,
                                            auto_reset, A.step0);
                __builtin_amdgcn_s_setprio(0);
            }
            continue;
        }
        if (settled >= G) break;  // no owner can post any more and nothing is posted
        __builtin_amdgcn_s_sleep(15);  // ~960 cycles between polls
    }
}

}  // namespace meshenv
