// meshenv_kernels.h -- wavefront-per-environment HIP kernels of the BoudaryEnv hot path (gfx950 / CDNA4).
//
// Execution model: one 64-lane wavefront (= one 64-thread workgroup) owns one environment for the whole
// launch.  The ring (coords, ids and, when needed, candidate keys/stamps) is staged into LDS with
// coalesced 16-byte loads; O(n) passes (point-in-polygon, boundary intersection scan, observation scan,
// boundary-quality scan, candidate selection) run one ring vertex per lane and are combined with
// ballots / shuffle reductions that carry the reference's sequential first-wins order as (value, order)
// pairs; O(1) parts (action decode, quad validity, reward) use a handful of lanes for the independent
// atan2 evaluations and are broadcast.  No MFMA: this is branchy fp64 geometry.
//
// A single-wave workgroup makes __syncthreads() a wave-local LDS fence, so phases that hand data between
// lanes through LDS stay cheap, and wave-uniform control flow (rule type, valid / invalid action) never
// diverges inside a wave.
#pragma once

#include "meshenv_geom.h"
#include "meshenv_state.h"

namespace meshenv {

constexpr int kStNoReference = 1;
constexpr int kStLogOverflow = 2;

// scratch area appended to the LDS ring arrays
struct Scratch {
    double2 q[4];       // quad vertices
    double ang[4];      // quad corner angles
    float robs[18];     // observation rows before the final float32 rounding
    int ipad[2];
};

struct Ctx {
    // LDS views
    double2 *xy;
    double *key;
    int32_t *stamp;
    int32_t *id;
    uint8_t *flag;
    Scratch *sc;
    // wave-uniform registers
    int lane, env, off, n0;
    int n, ref, n_elem, failed, n_new, counter, status, dom;
    double bl, area;
    bool keys_loaded, ring_dirty;
    // lane-private
    float obs;  // lanes 0..17: current observation
};

__device__ __forceinline__ P2 ldp(const Ctx &c, int i)
{
    const double2 v = c.xy[i];
    return mkp(v.x, v.y);
}

// Python list index wrap for i in [-n, 2n)
__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

__host__ __device__ __forceinline__ size_t lds_bytes_for(int cap)
{
    return (size_t)cap * (sizeof(double2) + sizeof(double) + 2 * sizeof(int32_t) + 1) + sizeof(Scratch) + 64;
}

__device__ __forceinline__ void carve_lds(Ctx &c, void *smem, int cap)
{
    // cap is a multiple of 16, so every array stays 16-byte aligned
    c.xy = (double2 *)smem;
    c.key = (double *)(c.xy + cap);
    c.stamp = (int32_t *)(c.key + cap);
    c.id = c.stamp + cap;
    c.sc = (Scratch *)(c.id + cap);
    c.flag = (uint8_t *)(c.sc + 1);
}

// ------------------------------------------------------------------------------------------ load / store

__device__ __forceinline__ void load_keys(Ctx &c, const DevState &S)
{
    if (c.keys_loaded) return;
    for (int i = c.lane; i < c.n; i += 64) {
        c.key[i] = S.ring_key[c.off + i];
        c.stamp[i] = S.ring_stamp[c.off + i];
    }
    c.keys_loaded = true;
    __syncthreads();
}

__device__ __forceinline__ void load_env(Ctx &c, const DevState &S, int env, bool with_keys)
{
    c.lane = lane_id();
    c.env = env;
    c.off = S.env_off[env];
    c.n0 = S.env_off[env + 1] - c.off;
    const EnvScalars s = S.scal[env];
    c.n = s.n; c.ref = s.ref; c.n_elem = s.n_elem; c.failed = s.failed; c.n_new = s.n_new;
    c.counter = s.counter; c.status = s.status; c.dom = s.dom; c.bl = s.bl; c.area = s.area;
    c.keys_loaded = false;
    c.ring_dirty = false;
    for (int i = c.lane; i < c.n; i += 64) {
        c.xy[i] = S.ring_xy[c.off + i];
        c.id[i] = S.ring_id[c.off + i];
    }
    c.obs = c.lane < kObsDim ? S.obs_cache[(size_t)env * kObsDim + c.lane] : 0.0f;
    __syncthreads();
    if (with_keys) load_keys(c, S);
}

__device__ __forceinline__ void store_env(Ctx &c, const DevState &S)
{
    __syncthreads();
    if (c.ring_dirty) {
        for (int i = c.lane; i < c.n; i += 64) {
            S.ring_xy[c.off + i] = c.xy[i];
            S.ring_id[c.off + i] = c.id[i];
            S.ring_key[c.off + i] = c.key[i];
            S.ring_stamp[c.off + i] = c.stamp[i];
        }
        if (c.lane < kObsDim) S.obs_cache[(size_t)c.env * kObsDim + c.lane] = c.obs;
    }
    if (c.lane == 0) {
        EnvScalars s;
        s.n = c.n; s.ref = c.ref; s.n_elem = c.n_elem; s.failed = c.failed; s.n_new = c.n_new;
        s.counter = c.counter; s.status = c.status; s.dom = c.dom; s.bl = c.bl; s.area = c.area;
        s.pad[0] = 0; s.pad[1] = 0;
        S.scal[c.env] = s;
    }
}

// ------------------------------------------------------------------------------------------ candidates (a5)

// MeshGeneration.check_boundary_point, M:202-231.  Returns false for None.
__device__ __forceinline__ bool check_boundary_point(const Ctx &c, const Params &p, int index, double &out)
{
    const int n = c.n;
    const P2 v = ldp(c, index);
    const double a0 = cw(v, ldp(c, wrapi(index + 1, n)), ldp(c, wrapi(index - 1, n)));
    if (a0 >= p.max_ref_angle || a0 == 0.0) return false;
    double sum_angle = 0.0;
    sum_angle += a0 * p.w0;
    const double a1 = cw(v, ldp(c, wrapi(index + 2, n)), ldp(c, wrapi(index - 2, n)));
    sum_angle += a1 * p.w1;
    out = sum_angle * (180.0 / kPi);  // math.degrees
    return true;
}

// find_reference_point, M:269-290: head of the list ordered by (key asc, insertion desc)
__device__ __forceinline__ int select_reference(const Ctx &c)
{
    double bk = __longlong_as_double(0x7ff0000000000000LL);  // +inf
    int bs = kNotCand, bi = -1;
    for (int i = c.lane; i < c.n; i += 64) {
        const int st = c.stamp[i];
        if (st != kNotCand) {
            const double k = c.key[i];
            if (k < bk || (k == bk && st > bs)) { bk = k; bs = st; bi = i; }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ok = shfl_xor_f64(bk, m);
        const int os = __shfl_xor(bs, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ok < bk || (ok == bk && os > bs)) { bk = ok; bs = os; bi = oi; }
    }
    return bi;
}

// ------------------------------------------------------------------------------------------ observation (a6)

// find_next_state, B:504-571 -> PointEnvironment.get_neighbors C:1073-1082 + get_radius_points C:1184-1282.
// Needs keys in LDS.  Updates c.ref, c.bl, c.obs (lanes 0..17), c.status.
__device__ __forceinline__ void find_next_state(Ctx &c, const DevState &S)
{
    const Params &p = S.prm;
    const int lane = c.lane, n = c.n;
    __syncthreads();
    const int idx = select_reference(c);
    c.ref = idx;
    if (idx < 0) {  // the reference returns None here
        c.status |= kStNoReference;
        c.obs = 0.0f;
        return;
    }
    c.status &= ~kStNoReference;
    const int i_right = wrapi(idx - 1, n), i_left = wrapi(idx + 1, n);
    const P2 ref = ldp(c, idx), right = ldp(c, i_right);
    const double area_ratio = c.area / S.dom_const[c.dom].orig_area;

    // base_length = round(sum of the 6 window edges / 6, 4): lane j holds edge j of the reference's
    // summation order, the sum itself is sequential
    double ed = 0.0;
    if (lane < 6) ed = dist(ldp(c, wrapi(idx + 2 - lane, n)), ldp(c, wrapi(idx + 3 - lane, n)));
    double sum = bcast_f64(ed, 0);
#pragma unroll
    for (int j = 1; j < 6; j++) sum += bcast_f64(ed, j);
    const double bl = round4_py(sum / 6);
    c.bl = bl;
    const double radius = p.radius;
    const double target_length = bl * radius;

    // lanes 0..5: the six neighbour rows; lane 0 also yields rotation_angle, lane 3 yields theta
    //   lane j < 3 : right side i = j     -> vertex idx-1-j
    //   lane j >= 3: left side  i = j - 3 -> vertex idx+1+(j-3)
    double nd = 0.0, na = 0.0;
    if (lane < 6) {
        const int vi = lane < 3 ? wrapi(idx - 1 - lane, n) : wrapi(idx + 1 + (lane - 3), n);
        const P2 v = ldp(c, vi);
        nd = (dist(ref, v) / radius) / bl;
        na = lane == 0 ? cw(ref, right, mkp(ref.x + 1, ref.y)) : cw(ref, v, right);
    }
    const double rot = bcast_f64(na, 0);
    const double theta = bcast_f64(na, 3);
    const double clipmax = theta + kPi / 2;
    if (lane < 6) {
        const int row = lane < 3 ? lane : 8 - (lane - 3);
        float v1;
        if (lane == 0) v1 = (float)area_ratio;
        else if (lane == 3) v1 = (float)theta;
        else if (lane < 3) v1 = (float)(na < kPi ? na : fmax(na, 1.5 * kPi) - 2 * kPi);
        else v1 = (float)fmin(na, clipmax);
        c.sc->robs[2 * row] = (float)nd;
        c.sc->robs[2 * row + 1] = v1;
    }

    // bisector segment ref -> p_s (Vertex.rotate about the origin, C:146-160)
    const double px = target_length * cos(theta / 2), py = target_length * sin(theta / 2);
    const double cr = cos(rot), sr = sin(rot);
    const double qx = (0.0 + cr * px) - sr * py;
    const double qy = (0.0 + sr * px) + cr * py;
    const double ux = (ref.x + qx) - ref.x, uy = (ref.y + qy) - ref.y;  // u = p_s - ref

    // O(n) scan, traversal order ord = 0..n-2 <-> ring index idx-1-ord (C:1239-1267)
    const double third = theta / 3;
    float s0 = 1.0f, s1 = 1.0f, s2 = 1.0f;  // best normalised distance per fan slot
    int o0 = 0x7fffffff, o1 = 0x7fffffff, o2 = 0x7fffffff;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    double rbest = 1.0;
    int rord = 0x7fffffff;
    for (int ord = lane; ord < n - 1; ord += 64) {
        const int ii = wrapi(idx - 1 - ord, n);
        if (ii == i_right || ii == i_left) continue;
        const P2 v = ldp(c, ii);
        const double d = dist(ref, v);
        const double angle = cw(ref, v, right);
        if (angle == 0.0) continue;
        const double kf = angle / third;
        if (kf < 3.0 && d < target_length) {
            const int k = (int)kf;
            const float cnd = (float)((d / radius) / bl);
            if (k == 0) { if (cnd < s0) { s0 = cnd; o0 = ord; a0 = angle; } }
            else if (k == 1) { if (cnd < s1) { s1 = cnd; o1 = ord; a1 = angle; } }
            else { if (cnd < s2) { s2 = cnd; o2 = ord; a2 = angle; } }
        }
        // Segment(ref, p_s).intersection_vertex(Segment(ring[i], ring[i+1])), C:649-668
        const P2 b = ldp(c, wrapi(ii + 1, n));
        const double wx = b.x - v.x, wy = b.y - v.y;
        double s, h;
        if (wy == 0.0) {
            if (uy == 0.0) continue;
            s = (v.y - ref.y) / uy;
            h = (ref.x - v.x + s * ux) / wx;
        } else if (wx == 0.0) {
            if (ux == 0.0) continue;
            s = (v.x - ref.x) / ux;
            h = (ref.y - v.y + s * uy) / wy;
        } else {
            s = ((ref.x - v.x) / wx - (ref.y - v.y) / wy) / (uy / wy - ux / wx);
            h = (ref.x - v.x + s * ux) / wx;
        }
        if (0.0 < s && s < 1.0 && 0.0 < h && h < 1.0) {
            const double val = (dist(ref, mkp(ref.x + s * ux, ref.y + s * uy)) / radius) / bl;
            if (val < rbest) { rbest = val; rord = ord; }
        }
    }
    // first-wins reductions
    float w0 = s0, w1 = s1, w2 = s2;
    int wo0 = o0, wo1 = o1, wo2 = o2;
    wave_argmin_f32(w0, wo0);
    wave_argmin_f32(w1, wo1);
    wave_argmin_f32(w2, wo2);
    double rb = rbest;
    int ro = rord;
    wave_argmin_f64(rb, ro);
    // defaults of the fan rows: [1, clip((2j+1)*theta/6)]
    float f0d = 1.0f, f1d = 1.0f, f2d = 1.0f;
    float f0a = (float)fmin((1 * theta) / 6, clipmax);
    float f1a = (float)fmin((3 * theta) / 6, clipmax);
    float f2a = (float)fmin((5 * theta) / 6, clipmax);
    if (w0 < 1.0f) { f0d = w0; f0a = (float)fmin(bcast_f64(a0, wo0 & 63), clipmax); }
    if (w1 < 1.0f) { f1d = w1; f1a = (float)fmin(bcast_f64(a1, wo1 & 63), clipmax); }
    if (w2 < 1.0f) { f2d = w2; f2a = (float)fmin(bcast_f64(a2, wo2 & 63), clipmax); }
    if (rb < 1.0 && (float)rb < f1d) {
        // the bisector hits edge (_i, _i+1) closer than the middle fan slot: report ring[_i-1.._i+1]
        const int hit = idx - 1 - ro;  // the reference's loop variable (may be negative)
        double hd = 0.0, ha = 0.0;
        if (lane < 3) {
            const P2 v = ldp(c, wrapi(wrapi(hit, n) + lane - 1, n));
            hd = (dist(ref, v) / radius) / bl;
            ha = cw(ref, v, right);
        }
        f0d = (float)bcast_f64(hd, 0); f0a = (float)bcast_f64(ha, 0);
        f1d = (float)bcast_f64(hd, 1); f1a = (float)bcast_f64(ha, 1);
        f2d = (float)bcast_f64(hd, 2); f2a = (float)bcast_f64(ha, 2);
    }
    if (lane == 0) {
        float *r = c.sc->robs;
        r[6] = f0d; r[7] = f0a; r[8] = f1d; r[9] = f1a; r[10] = f2d; r[11] = f2a;
    }
    __syncthreads();
    if (lane < kObsDim) c.obs = round4_npf(c.sc->robs[lane]);
}

// ------------------------------------------------------------------------------------------ point in polygon (a7)

__device__ __forceinline__ double round4_by(bool is_np, double v) { return is_np ? round4_np(v) : round4_py(v); }

// is_point_inside_area -> calculate_crossing_segments, M:539-546, 47-102.  One ring edge per lane.
__device__ __forceinline__ bool point_inside(const Ctx &c, const Params &prm, P2 p)
{
    const int n = c.n, n0 = c.n0;
    const P2 far = mkp(prm.ray_length, p.y);
    int count = 0;
    for (int i = c.lane; i < n; i += 64) {
        const int im1 = wrapi(i - 1, n);
        const P2 vi = ldp(c, i), vm = ldp(c, im1);
        const bool np_i = c.id[i] >= n0, np_m = c.id[im1] >= n0;
        const double orientation = round4_by(np_i || np_m, vi.y - vm.y);
        if (orientation == 0.0) continue;
        // is_cross is a pure conjunction; the ray-side test is the selective one, so it goes first
        if (!(straddle(p, far, vi, vm) && straddle(vi, vm, p, far))) continue;
        if (round4_np(vi.y - p.y) == 0.0) {
            const int ip1 = wrapi(i + 1, n);
            const double next_o = round4_by(c.id[ip1] >= n0 || np_i, ldp(c, ip1).y - vi.y);
            if (next_o == 0.0) continue;
            if (next_o * orientation < 0.0) continue;
            if (orientation < 0.0) count += 1;
        } else if (round4_np(vm.y - p.y) == 0.0) {
            const int im2 = wrapi(i - 2, n);
            const double pre_o = round4_by(np_m || c.id[im2] >= n0, vm.y - ldp(c, im2).y);
            if (pre_o == 0.0) continue;
            if (pre_o * orientation < 0.0) continue;
            if (orientation < 0.0) continue;
            count += 1;
        } else {
            count += 1;
        }
    }
    return (wave_sum_i32(count) & 1) != 0;
}

// ------------------------------------------------------------------------------------------ quad validity (a9, a10)

// Mesh.is_valid(0), C:730-749 + segments_crossed C:806-818.  Quad in c.sc->q; corner angles land in c.sc->ang.
__device__ __forceinline__ bool quad_is_valid(Ctx &c, const Params &prm)
{
    const double2 *q = c.sc->q;
    const P2 m0 = mkp(q[0].x, q[0].y), m1 = mkp(q[1].x, q[1].y), m2 = mkp(q[2].x, q[2].y), m3 = mkp(q[3].x, q[3].y);
    bool bad = false;
    double deg = 0.0;
    if (c.lane < 4) {
        const double2 a = q[c.lane], b = q[(c.lane + 1) & 3], d = q[(c.lane + 3) & 3];
        deg = cw(mkp(a.x, a.y), mkp(b.x, b.y), mkp(d.x, d.y));
        c.sc->ang[c.lane] = deg;
        bad = deg > prm.max_degree || deg < prm.min_degree;
    }
    const bool crossed = is_cross(m0, m1, m2, m3) || is_cross(m0, m3, m1, m2);
    const bool any_bad = __ballot(bad) != 0ULL;
    __syncthreads();
    return !crossed && !any_bad;
}

// check_intersection_with_boundary, M:510-530.  mpos[k] = ring slot of quad vertex k (-1 = new vertex),
// r = position of the reference vertex in the quad.  One ring vertex per lane.
__device__ __forceinline__ bool intersects_boundary(const Ctx &c, int mp0, int mp1, int mp2, int mp3, int r)
{
    const int n = c.n;
    const double2 *q = c.sc->q;
    const P2 ref = mkp(q[r].x, q[r].y);
    double max_dist = -1.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k != r) {
            const double d = dist(ref, mkp(q[k].x, q[k].y));
            max_dist = d > max_dist ? d : max_dist;
        }
    }
    const double2 qa = q[(r + 3) & 3], qb = q[(r + 2) & 3], qc = q[(r + 1) & 3];
    const P2 c0a = mkp(qa.x, qa.y), c0b = mkp(qb.x, qb.y);  // (m[r-1], m[r-2])
    const P2 c1a = c0b, c1b = mkp(qc.x, qc.y);              // (m[r-2], m[r-3])
    bool hit = false;
    for (int i = c.lane; i < n; i += 64) {
        if (i == mp0 || i == mp1 || i == mp2 || i == mp3) continue;
        const P2 v = ldp(c, i);
        if (!(dist(ref, v) < max_dist)) continue;
        const int ip = wrapi(i - 1, n), in = wrapi(i + 1, n);
        const bool use_p = !(ip == mp0 || ip == mp1 || ip == mp2 || ip == mp3);
        const bool use_n = !(in == mp0 || in == mp1 || in == mp2 || in == mp3);
        const P2 vp = ldp(c, ip), vn = ldp(c, in);
        if (use_p && (is_cross(c0a, c0b, v, vp) || is_cross(c1a, c1b, v, vp))) hit = true;
        if (!hit && use_n && (is_cross(c0a, c0b, v, vn) || is_cross(c1a, c1b, v, vn))) hit = true;
    }
    return __ballot(hit) != 0ULL;
}

// ------------------------------------------------------------------------------------------ reward (a12)

// compute_boundary_quality(add_v), M:329-382; index = ring slot of the new vertex
__device__ __forceinline__ double boundary_quality_new(Ctx &c, int index)
{
    const int n = c.n, lane = c.lane;
    const P2 add_v = ldp(c, index);
    const int w1 = wrapi(index + 1, n), w2 = wrapi(index + 2, n), w3 = wrapi(index - 1, n), w4 = wrapi(index - 2, n);
    // angles at the two ring neighbours of the new vertex (lanes 0, 1)
    double ang = 0.0;
    if (lane < 2) {
        const int ctr = lane == 0 ? w1 : w3;
        ang = cw(ldp(c, ctr), ldp(c, wrapi(ctr + 1, n)), ldp(c, wrapi(ctr - 1, n)));
    }
    const double ang0 = bcast_f64(ang, 0), ang1 = bcast_f64(ang, 1);
    double amin = 1e300;
    bool have = false;
    if (ang0 < kPi / 3) { amin = ang0; have = true; }
    if (ang1 < kPi / 3) { amin = ang1 < amin ? ang1 : amin; have = true; }
    const double q1 = have ? 3 * amin / kPi : 1.0;

    const double dst = dist(add_v, ldp(c, w1)) + dist(add_v, ldp(c, w3));
    // close_vs: vertices outside the 5-window nearer than dst; a vertex is skipped when its predecessor
    // index was appended (M:355-357)  ->  added(i) = near(i) && !added(i-1)
    double m_d = 1e300;
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool near = false;
        if (i < n && !(i == index || i == w1 || i == w2 || i == w3 || i == w4)) near = dist(add_v, ldp(c, i)) < dst;
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
    m_d = wave_min_f64(m_d);
    const double targt_len = dst / 2;
    double ed = 0.0;
    if (lane < 4) ed = dist(ldp(c, wrapi(index - 2 + lane, n)), ldp(c, wrapi(index - 1 + lane, n)));
    double sum = bcast_f64(ed, 0);
    sum += bcast_f64(ed, 1);
    sum += bcast_f64(ed, 2);
    sum += bcast_f64(ed, 3);
    const double mean_dist = sum / 4;
    const double smoothness = (targt_len < mean_dist ? targt_len : mean_dist) / (targt_len > mean_dist ? targt_len : mean_dist);
    double q2 = 1.0;
    if (m_d < 1e299) q2 = (m_d < 0.5 * dst) ? m_d / (0.5 * dst) : 1.0;
    return cbrt(smoothness * q1 * q2);  // math.pow(x, 1/3)
}

// compute_ele_boundary_quality else-branch, M:392-426; t0/t1 = ring slots of the kept quad vertices in quad order
__device__ __forceinline__ double boundary_quality_kept(Ctx &c, int t0, int t1)
{
    const int n = c.n, lane = c.lane;
    double ang = 0.0;
    if (lane < 2) {
        const int ctr = lane == 0 ? t0 : t1;
        ang = cw(ldp(c, ctr), ldp(c, wrapi(ctr + 1, n)), ldp(c, wrapi(ctr - 1, n)));
    }
    const double ang0 = bcast_f64(ang, 0), ang1 = bcast_f64(ang, 1);
    double amin = 1e300;
    bool have = false;
    if (ang0 < kPi / 3) { amin = ang0; have = true; }
    if (ang1 < kPi / 3) { amin = ang1 < amin ? ang1 : amin; have = true; }
    const int index = t0 < t1 ? t0 : t1;
    const double targt_len = dist(ldp(c, t0), ldp(c, t1));
    double ed = 0.0;
    if (lane < 5) ed = dist(ldp(c, wrapi(index - 2 + lane, n)), ldp(c, wrapi(index - 1 + lane, n)));
    double sum = bcast_f64(ed, 0);
#pragma unroll
    for (int j = 1; j < 5; j++) sum += bcast_f64(ed, j);
    const double mean_dist = sum / 5;
    const double smoothness = (targt_len < mean_dist ? targt_len : mean_dist) / (targt_len > mean_dist ? targt_len : mean_dist);
    const double angle_quality = have ? 3 * amin / kPi : 1.0;
    return sqrt(angle_quality * smoothness);  // math.pow(x, 1/2)
}

// ------------------------------------------------------------------------------------------ episode control

// reset(): copy the domain's precomputed reset state (B:67-84 computes only per-domain constants)
__device__ __forceinline__ void reset_from_domain(Ctx &c, const DevState &S)
{
    const int doff = S.dom_off[c.dom];
    __syncthreads();
    for (int i = c.lane; i < c.n0; i += 64) {
        c.xy[i] = S.dom_xy[doff + i];
        c.id[i] = i;
        c.key[i] = S.dom_key[doff + i];
        c.stamp[i] = S.dom_stamp[doff + i];
    }
    c.n = c.n0;
    c.ref = S.dom_ref[c.dom];
    c.bl = S.dom_bl[c.dom];
    c.area = S.dom_const[c.dom].orig_area;
    c.n_elem = 0; c.failed = 0; c.n_new = 0; c.counter = 0;
    c.status = c.ref < 0 ? kStNoReference : 0;
    c.obs = c.lane < kObsDim ? S.dom_obs[(size_t)c.dom * kObsDim + c.lane] : 0.0f;
    c.keys_loaded = true;
    c.ring_dirty = true;
    __syncthreads();
}

__device__ __forceinline__ void log_quad(Ctx &c, const DevState &S, int g0, int g1, int g2, int g3)
{
    const int cap = S.prm.log_cap;
    if (c.n_elem < cap) {
        if (c.lane == 0) {
            int32_t *dst = S.log_quads + ((size_t)c.env * cap + c.n_elem) * 4;
            dst[0] = g0; dst[1] = g1; dst[2] = g2; dst[3] = g3;
        }
    } else if (cap > 0) {
        c.status |= kStLogOverflow;
    }
    c.n_elem += 1;
}

struct StepResult {
    double reward;
    int done, complete;
    bool valid;
};

// step(action), B:113-263.  a0 = rule type, (a1, a2) = candidate point in the local frame.
__device__ __forceinline__ StepResult env_step(Ctx &c, const DevState &S, float a0, float a1, float a2)
{
    const Params &prm = S.prm;
    const int lane = c.lane;
    StepResult out;
    out.valid = false;
    int done = 0;
    bool failed = true;
    double reward = 0.0;
    const int index = c.ref;
    const int n = c.n;

    bool no_reference = false;
    if (index < 0) {
        // no reference vertex (the reference's find_next_state returned None and its next step() would
        // raise): end the episode as truncated
        reward = -1.0;
        done = 1;
        no_reference = true;
    } else if (n <= 5) {
        reward = 10.0;  // B:141-143
        done = 1;
    } else {
        int mp0, mp1, mp2, mp3, r;
        bool have_mesh = true, new_vertex = false;
        P2 new_point = mkp(0.0, 0.0);
        int rule;
        if (a0 <= -0.5f) rule = -1;
        else if (a0 >= 0.5f) rule = 1;
        else {
            rule = 0;
            // action_2_point -> detransformation, B:616-625, B:98-106, D:67-83
            const double px = (double)round4_npf(a1), py = (double)round4_npf(a2);
            const P2 p0 = ldp(c, index), p1 = ldp(c, wrapi(index - 1, n));
            const double theta = 2 * kPi - atan2(p1.y - p0.y, p1.x - p0.x);
            const double ct = cos(theta), st = sin(theta);
            double ox = ct * px + st * py;
            double oy = -st * px + ct * py;
            ox *= c.bl;
            oy *= c.bl;
            ox += p0.x;
            oy += p0.y;
            new_point = mkp(round4_np(ox), round4_np(oy));
            if (point_inside(c, prm, new_point)) {
                // find_same_point, B:599-602: first ring vertex within eps
                int first = 0x7fffffff;
                for (int i = lane; i < n; i += 64)
                    if (i < first && dist(ldp(c, i), new_point) < prm.same_eps) first = i;
                first = wave_min_i32(first);
                if (first != 0x7fffffff) rule = -1;  // existing point: the rule -1 quad, B:168-175
                else new_vertex = true;
            } else {
                reward += c.n_elem ? -1.0 / c.n_elem : -1.0;
                have_mesh = false;
            }
        }
        if (have_mesh) {
            if (new_vertex) {  // [new, i-1, i, i+1], B:177-182
                mp0 = -1; mp1 = wrapi(index - 1, n); mp2 = index; mp3 = wrapi(index + 1, n); r = 2;
            } else if (rule == -1) {  // [i-1, i, i+1, i+2], B:147-153
                mp0 = wrapi(index - 1, n); mp1 = index; mp2 = wrapi(index + 1, n); mp3 = wrapi(index + 2, n); r = 1;
            } else {  // [i-2, i-1, i, i+1], B:156-162
                mp0 = wrapi(index - 2, n); mp1 = wrapi(index - 1, n); mp2 = index; mp3 = wrapi(index + 1, n); r = 2;
            }
            __syncthreads();
            if (lane < 4) {
                const int mp = lane == 0 ? mp0 : lane == 1 ? mp1 : lane == 2 ? mp2 : mp3;
                c.sc->q[lane] = mp < 0 ? make_double2(new_point.x, new_point.y) : c.xy[mp];
            }
            __syncthreads();
            bool ok = quad_is_valid(c, prm);
            if (ok) ok = !intersects_boundary(c, mp0, mp1, mp2, mp3, r);
            if (ok) {
                load_keys(c, S);
                const int g0 = mp0 < 0 ? c.n0 + c.n_new : c.id[mp0];
                const int g1 = c.id[mp1], g2 = c.id[mp2], g3 = c.id[mp3];
                log_quad(c, S, g0, g1, g2, g3);  // generated_meshes.append, B:192

                // quad geometry for the reward, from the pre-update coordinates (Mesh holds the Vertex objects)
                const double2 *q = c.sc->q;
                const P2 m0 = mkp(q[0].x, q[0].y), m1 = mkp(q[1].x, q[1].y), m2 = mkp(q[2].x, q[2].y), m3 = mkp(q[3].x, q[3].y);
                const double e0 = dist(m0, m3), e1 = dist(m1, m0), e2 = dist(m2, m1), e3 = dist(m3, m2);
                const double ang0 = c.sc->ang[0], ang1 = c.sc->ang[1], ang2 = c.sc->ang[2], ang3 = c.sc->ang[3];
                // Mesh.compute_area, C:935-950: corner_1 == corner angle 0, corner_3 == corner angle 2
                const double mesh_area = 0.5 * e0 * e1 * sin(ang0) + 0.5 * e2 * e3 * sin(ang2);
                // Mesh.get_quality('robust'), C:873-884
                double mn = e1 < e0 ? e1 : e0;
                mn = e2 < mn ? e2 : mn;
                mn = e3 < mn ? e3 : mn;
                const double d02 = dist(m0, m2), d13 = dist(m1, m3);
                const double q1 = sqrt(2.0) * mn / (d13 > d02 ? d13 : d02);
                double amn = ang1 < ang0 ? ang1 : ang0, amx = ang1 > ang0 ? ang1 : ang0;
                amn = ang2 < amn ? ang2 : amn; amx = ang2 > amx ? ang2 : amx;
                amn = ang3 < amn ? ang3 : amn; amx = ang3 > amx ? ang3 : amx;
                const double e_reward = sqrt(q1 * (amn / amx));

                // update_boundary, M:575-648
                __syncthreads();
                double b_reward;
                int p0, p1, p2, p3;  // ref_neighbors as ring slots
                if (new_vertex) {
                    const int id = index;
                    if (lane == 0) {
                        c.xy[id] = make_double2(new_point.x, new_point.y);
                        c.id[id] = c.n0 + c.n_new;
                        c.stamp[id] = kNotCand;
                        const int cap = prm.log_cap;
                        if (c.n_new < cap) S.log_vxy[(size_t)c.env * cap + c.n_new] = make_double2(new_point.x, new_point.y);
                    }
                    if (prm.log_cap > 0 && c.n_new >= prm.log_cap) c.status |= kStLogOverflow;
                    c.n_new += 1;
                    p0 = wrapi(id + 1, n); p1 = wrapi(id - 1, n); p2 = wrapi(id + 2, n); p3 = wrapi(id - 2, n);
                } else {
                    // delete the two interior quad vertices (slots mp1, mp2), compacting the ring in LDS
                    const int lo = mp1 < mp2 ? mp1 : mp2, hi = mp1 < mp2 ? mp2 : mp1;
                    const int keep0 = c.id[mp0], keep1 = c.id[mp3];
                    for (int base = 0; base < n; base += 64) {
                        const int i = base + lane;
                        double2 vxy = make_double2(0, 0);
                        double vk = 0; int vs = 0, vid = 0;
                        const bool live = i < n && i != lo && i != hi;
                        if (live) { vxy = c.xy[i]; vk = c.key[i]; vs = c.stamp[i]; vid = c.id[i]; }
                        __syncthreads();
                        if (live) {
                            const int j = i - (i > lo ? 1 : 0) - (i > hi ? 1 : 0);
                            c.xy[j] = vxy; c.key[j] = vk; c.stamp[j] = vs; c.id[j] = vid;
                        }
                        __syncthreads();
                    }
                    c.n = n - 2;
                    // new slots of the kept vertices
                    const int nn = c.n;
                    int t0 = mp0 - (mp0 > lo ? 1 : 0) - (mp0 > hi ? 1 : 0);
                    int t1 = mp3 - (mp3 > lo ? 1 : 0) - (mp3 > hi ? 1 : 0);
                    (void)keep0; (void)keep1;
                    const int id = t0 > t1 ? t0 : t1;
                    p0 = wrapi(id, nn); p1 = wrapi(id - 1, nn); p2 = wrapi(id + 1, nn); p3 = wrapi(id - 2, nn);
                    mp0 = t0; mp3 = t1;
                }
                __syncthreads();
                // remove_reference_candidates(ref_neighbors [+ removed]) then add_reference_candidates in order
                {
                    double k = 0.0;
                    bool okk = false;
                    if (lane < 4) {
                        const int pos = lane == 0 ? p0 : lane == 1 ? p1 : lane == 2 ? p2 : p3;
                        okk = check_boundary_point(c, prm, pos, k);
                    }
                    const unsigned long long mk = __ballot(okk);
                    if (lane < 4) {
                        const int pos = lane == 0 ? p0 : lane == 1 ? p1 : lane == 2 ? p2 : p3;
                        if (okk) {
                            c.key[pos] = k;
                            c.stamp[pos] = c.counter + __popcll(mk & ((2ULL << lane) - 1ULL));
                        } else {
                            c.stamp[pos] = kNotCand;
                        }
                    }
                    c.counter += __popcll(mk);
                }
                __syncthreads();
                b_reward = new_vertex ? boundary_quality_new(c, index) : boundary_quality_kept(c, mp0, mp3);
                c.area -= mesh_area;
                // get_quality(mesh, 2), M:1733-1740
                const double quality = e_reward + 1 * (b_reward - 1);
                // get_speed_penalty, B:434-450
                const DomConst dc = S.dom_const[c.dom];
                double speed = 0.0;
                if (dc.min_area <= mesh_area && mesh_area < dc.crit_area) speed = (mesh_area - dc.crit_area) / (dc.crit_area - dc.min_area);
                else if (mesh_area < dc.min_area) speed = -1.0;
                reward += quality + speed;
                failed = false;
                out.valid = true;
                c.ring_dirty = true;
                if (c.n <= 5) {  // B:232-238
                    reward += 10.0;
                    done = 1;
                    if (c.n == 4) log_quad(c, S, c.id[0], c.id[1], c.id[2], c.id[3]);
                }
                find_next_state(c, S);
            } else {
                reward += c.n_elem ? -1.0 / c.n_elem : -1.0;  // B:248
            }
        }
    }
    int complete = no_reference ? 0 : 1;
    if (!failed) {
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
    out.reward = reward;
    out.done = done;
    out.complete = complete;
    return out;
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
    c.dom = d;
    const int doff = S.dom_off[d];
    c.off = 0;
    c.n0 = S.dom_off[d + 1] - doff;
    c.n = c.n0;
    c.area = S.dom_const[d].orig_area;
    c.status = 0;
    c.keys_loaded = true;
    for (int i = c.lane; i < c.n; i += 64) {
        c.xy[i] = S.dom_xy[doff + i];
        c.id[i] = i;
    }
    __syncthreads();
    // find_reference_candidates, M:233-261: stable sort by key -> ties in ring order (stamp = -index)
    for (int i = c.lane; i < c.n; i += 64) {
        double k = 0.0;
        const bool ok = check_boundary_point(c, S.prm, i, k);
        c.key[i] = k;
        c.stamp[i] = ok ? -i : kNotCand;
    }
    c.bl = 0.0;
    c.obs = 0.0f;
    find_next_state(c, S);
    for (int i = c.lane; i < c.n; i += 64) {
        S.dom_key[doff + i] = c.key[i];
        S.dom_stamp[doff + i] = c.stamp[i];
    }
    if (c.lane < kObsDim) S.dom_obs[(size_t)d * kObsDim + c.lane] = c.obs;
    if (c.lane == 0) {
        S.dom_ref[d] = c.ref;
        S.dom_bl[d] = c.bl;
    }
}

__global__ void __launch_bounds__(64) k_reset(DevState S, int cap, const uint8_t *mask, float *obs_out, int first)
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
    c.off = S.env_off[env];
    c.n0 = S.env_off[env + 1] - c.off;
    c.dom = S.scal[env].dom;
    reset_from_domain(c, S);
    if (first && c.lane == 0) {
        EnvCounters z;
        z.steps = 0; z.valid = 0; z.sum_n = 0; z.sum_n_valid = 0;
        S.cnt[env] = z;
    }
    store_env(c, S);
    if (obs_out && c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
}

// n_steps consecutive steps of every env in one launch (n_steps = 1 is meshenv_step).
__global__ void __launch_bounds__(64)
k_step(DevState S, int cap, int n_steps, const float *__restrict__ actions, float *__restrict__ obs_out,
       double *__restrict__ reward, uint8_t *__restrict__ done, uint8_t *__restrict__ complete,
       float *__restrict__ term_obs, int auto_reset)
{
    extern __shared__ double2 smem[];
    Ctx c;
    carve_lds(c, smem, cap);
    const int env = blockIdx.x;
    const int E = S.n_envs;
    load_env(c, S, env, n_steps > 1);
    unsigned long long st_steps = 0, st_valid = 0, st_sum = 0, st_sumv = 0;
    for (int t = 0; t < n_steps; t++) {
        const float *a = actions + ((size_t)t * E + env) * 3;
        const float a0 = a[0], a1 = a[1], a2 = a[2];
        const int n_before = c.n;
        const StepResult r = env_step(c, S, a0, a1, a2);
        st_steps += 1;
        st_sum += (unsigned long long)n_before;
        if (r.valid) { st_valid += 1; st_sumv += (unsigned long long)n_before; }
        const size_t o = (size_t)t * E + env;
        if (c.lane == 0) {
            reward[o] = r.reward;
            done[o] = (uint8_t)r.done;
            complete[o] = (uint8_t)r.complete;
        }
        if (r.done) {
            if (term_obs && c.lane < kObsDim) term_obs[(size_t)env * kObsDim + c.lane] = c.obs;
            if (auto_reset) reset_from_domain(c, S);
        }
    }
    if (c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
    store_env(c, S);
    if (c.lane == 0) {
        EnvCounters k = S.cnt[env];
        k.steps += st_steps; k.valid += st_valid; k.sum_n += st_sum; k.sum_n_valid += st_sumv;
        S.cnt[env] = k;
    }
}

}  // namespace meshenv
