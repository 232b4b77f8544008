// meshenv_samples.h -- MeshGeneration.extract_samples_2 (general/mesh.py:1438-1489), the data-preparation step of the
// reference's ANN scripts (general/EBRD.py:414, 579; general/post_processing.py:532), for the generated meshes of a whole
// batch of environments.
//
// Every element whose quality (get_quality(element, index), general/mesh.py:1728-1747, index 1 or 5) reaches the
// threshold contributes, for each of its four corners as reference point rp (left / right neighbour l_p / r_p, opposite
// corner = target):
//   * the neighbour paths of n_neighbor vertices leaving l_p and r_p (get_nodes, :1422-1436: depth-first over
//     get_connected_vertices(), never through rp / the other neighbour, never revisiting the path so far);
//   * the radius neighbours of rp in n_radius angular sectors between l_p and r_p (get_radius_neighbors, :1601-1632): per
//     sector the vertices of boundary.vertices strictly inside it, nearest first (stable), within radius * the mean of
//     rp's two edges, minus [rp, l_p, r_p, target], plus one synthetic point on the sector's bisector;
//   * one sample per (right path, sector tuple, left path) except where the target lies on both paths: per point of
//     (right path, sector tuple from the LAST sector to the first, left path reversed) the pair [distance to rp /
//     (base_length * radius), clockwise angle from r_p], base_length = mean length of the 2 n_neighbor path edges; the
//     target in the same frame; the type 1 / 0 / 0.5 (target on the right path / the left path / neither).
// The mesh graph comes from the element log exactly as for the smoothers (build_segment_lists, csrc/meshenv_smooth.h).
// One wavefront per environment, one (element, corner) per lane at a time: the enumeration is nested loops over the
// neighbour lists (no path storage), the per-item sample counts are prefix-summed over the wavefront, and the same kernel
// runs twice -- counts, then rows at host-computed offsets.  Squares follow the reference's libm where that is validated
// (csrc/meshenv_libm.h); the one value class that is not bit-identical to the reference is what passes through cos / sin
// of an UNQUANTISED angle -- the synthetic sector points (<= 2 ulp on their distances) -- see tests/test_gpu_samples.py.
#pragma once

#include "meshenv_smooth.h"

namespace meshenv {

constexpr int kSampMaxNeighbor = 3, kSampMaxRadius = 4, kSampSecMax = 32;

struct SampleParams {
    int n_neighbor, n_radius, index;
    double radius, quality_threshold;
};

// status per env
enum { kSampOk = 0, kSampSkipped = 1, kSampLogOverflow = 2, kSampDegree = 3, kSampSectorOverflow = 4 };

__host__ __device__ __forceinline__ size_t samples_lds_bytes(int ring_cap, int log_cap)
{
    const size_t V = (size_t)ring_cap + log_cap;
    return V * sizeof(double2) + V * kSmoothMaxDeg * 2 + V + (size_t)log_cap + 64;
}

struct SampCtx {
    const double2 *coord;
    const unsigned short *adj;
    const unsigned char *deg;
    LibmRef lr;
};

__device__ __forceinline__ P2 sp(const SampCtx &s, int v)
{
    const double2 c = s.coord[v];
    return mkp(c.x, c.y);
}

// get_quality(element, index) >= threshold: index 1 = compute_element_quality (M:1714-1726) = sqrt(q1 q2) of
// Mesh.get_quality_3 (C:952-972), index 5 = Mesh.get_quality('strong') (C:907-930)
__device__ __forceinline__ double sample_element_quality(const SampCtx &s, const int *v, int index)
{
    P2 p[4];
    for (int i = 0; i < 4; i++) p[i] = sp(s, v[i]);
    double e[4], ang[4];
    for (int i = 0; i < 4; i++) {
        e[i] = dist_lr(s.lr, p[i], p[(i + 3) & 3]);
        ang[i] = cw(p[i], p[(i + 1) & 3], p[(i + 3) & 3]);
    }
    const double area = 0.5 * e[0] * e[1] * sin(ang[0]) + 0.5 * e[2] * e[3] * sin(ang[2]);   // cw(p0; p1, p3), cw(p2; p3, p1)
    double q1 = 0.0;
    if (area > 0) {
        const double ra = sqrt(area);
        double product = 1.0;
        for (int i = 0; i < 4; i++) product *= (ra - e[i] > 0) ? e[i] / ra : 1.0 / (e[i] / ra);
        q1 = pow(product, 0.25);
    }
    double ap = 1.0;
    for (int i = 0; i < 4; i++) ap *= 1 - (fabs(ang[i] * (180.0 / kPi) - 90) / 90);
    const double q2 = ap < 0 ? 0.0 : pow(ap, 0.25);
    if (index == 1) return sqrt(q1 * q2);
    double amin = kInf, amax = -kInf;
    for (int i = 0; i < 4; i++) { const double a = fabs(ang[i]); amin = a < amin ? a : amin; amax = a > amax ? a : amax; }
    return sqrt(q1 * (amin / amax));
}

// One sector of get_radius_neighbors: ids[0 .. return) = the close vertices, nearest first; syn = the synthetic point.
__device__ __forceinline__ int sample_sector(const SampCtx &s, int n_vert, int base, int start, int end_, int target, double radius,
                                             double start_angle, double end_angle, unsigned short *ids, P2 &syn, bool &overflow)
{
    const P2 pb = sp(s, base), ps = sp(s, start), pe = sp(s, end_);
    const double bl = radius * (0.5 * dist_lr(s.lr, pb, ps) + 0.5 * dist_lr(s.lr, pb, pe));
    double ds[kSampSecMax];
    int m = 0;
    for (int v = 0; v < n_vert; v++) {
        if (v == base) continue;                                    // compute_dist: `point is not vertex`
        const P2 pv = sp(s, v);
        const double d = dist_lr(s.lr, pb, pv);
        if (!(d <= bl) || v == start || v == end_ || v == target) continue;   // beyond S_T, or in the exclusion list
        const double a = cw(pb, ps, pv);
        if (!(start_angle < a && a < end_angle)) continue;
        if (m == kSampSecMax) { overflow = true; continue; }
        int at = m;                                                  // sorted(key = distance) is stable: after equal distances
        while (at > 0 && ds[at - 1] > d) { ds[at] = ds[at - 1]; ids[at] = ids[at - 1]; at--; }
        ds[at] = d; ids[at] = (unsigned short)v;
        m++;
    }
    const double a0 = cw(pb, ps, mkp(pb.x + 1, pb.y));
    const double mid = a0 - (start_angle + end_angle) / 2;
    syn = mkp(pb.x + bl * cos(mid), pb.y + bl * sin(mid));
    return m;
}

// the k-th path of N vertices leaving `root` (get_nodes with exclusion {x0, x1}); returns false when k is past the end.
// Enumeration order = the reference's depth-first order, INCLUDING what its shared `path` list does (M:1422-1436: a node is
// appended while len(path) < N and written at position N - layer - 1 afterwards): with N = 3, if the FIRST second-level node
// a0 has no admissible child, the list is still [root, a0] when the next node a1 arrives, so a1 is appended behind a0 --
// its children are filtered against path[:2] = [root, a0] and emitted as [root, a0, child].  From then on the list is full
// and positions are what they should be.  path[0] = root.
__device__ __forceinline__ bool sample_path(const SampCtx &s, int N, int root, int x0, int x1, int k, int *path)
{
    path[0] = root; path[1] = -1; path[2] = -1;
    if (N == 1) return k == 0;
    int seen = 0, ord1 = 0, a0 = -1;
    bool a0_dead = false;
    const int d0 = s.deg[root];
    for (int i = 0; i < d0; i++) {
        const int a = s.adj[(size_t)root * kSmoothMaxDeg + i];
        if (a == x0 || a == x1 || a == root) continue;
        if (N == 2) {
            if (seen == k) { path[1] = a; return true; }
            seen++;
            continue;
        }
        const bool quirk = ord1 == 1 && a0_dead;           // a1 behind a dead-end a0
        const int mid = quirk ? a0 : a, ex = quirk ? a0 : a;
        int kids = 0;
        const int d1 = s.deg[a];
        for (int j = 0; j < d1; j++) {
            const int b = s.adj[(size_t)a * kSmoothMaxDeg + j];
            if (b == x0 || b == x1 || b == root || b == ex) continue;
            kids++;
            if (seen == k) { path[1] = mid; path[2] = b; return true; }
            seen++;
        }
        if (ord1 == 0) { a0 = a; a0_dead = kids == 0; }
        ord1++;
    }
    return false;
}

__device__ __forceinline__ bool path_has(const int *path, int N, int v)
{
    return path[0] == v || (N > 1 && path[1] == v) || (N > 2 && path[2] == v);
}

template <bool kWrite>
__global__ void __launch_bounds__(64)
k_extract_samples(DevState S, int ring_cap, int which, const uint8_t *__restrict__ mask, SampleParams P, LibmRef lr,
                  long long *__restrict__ count_out, unsigned char *__restrict__ status_out, const long long *__restrict__ offsets,
                  double *__restrict__ samples, double *__restrict__ outputs, double *__restrict__ types)
{
    extern __shared__ double2 smem[];
    const int env = blockIdx.x, lane = lane_id();
    if (mask && mask[env] == 0) {
        if (lane == 0 && !kWrite) { count_out[env] = 0; status_out[env] = kSampSkipped; }
        return;
    }
    // the filling launch writes exactly what the counting launch counted: an env that reported an error (count 0) writes nothing
    if (kWrite && (count_out[env] == 0 || status_out[env] != kSampOk)) return;
    const long long row_end = kWrite ? offsets[env + 1] : 0;
    const int log_cap = S.prm.log_cap;
    const EnvScalars sc = S.scal[env];
    const DevCold cold = load_cold(S);
    int n_elem = uniform_i32(sc.n_elem), n_new = uniform_i32(sc.n_new);
    int status = uniform_i32(sc.status), half = (status >> 4) & 1, overflow = status & kStLogOverflow;
    if (which) {
        const LastEpisode le = cold.last_ep[env];
        n_elem = uniform_i32(le.n_elem); n_new = uniform_i32(le.n_new);
        overflow = uniform_i32(le.flags) & 2;
        half ^= 1;
    }
    if (overflow || n_elem > log_cap || n_new > log_cap) {
        if (lane == 0 && !kWrite) { count_out[env] = 0; status_out[env] = kSampLogOverflow; }
        return;
    }
    const DomConst dc = S.dom[uniform_i32(sc.dom)];
    const int doff = uniform_i32(dc.off), n0 = uniform_i32(dc.n0), nv = n0 + n_new;
    const size_t lbase = ((size_t)env * 2 + half) * log_cap;
    const int4 *quads = reinterpret_cast<const int4 *>(cold.log_quads + lbase * 4);
    const double2 *vnew = cold.log_vxy + lbase;
    const int V = ring_cap + log_cap;
    double2 *coord = smem;
    unsigned short *adj = (unsigned short *)(coord + V);
    unsigned char *deg = (unsigned char *)(adj + (size_t)V * kSmoothMaxDeg);
    unsigned char *good = deg + V;                                  // [n_elem] quality >= threshold
    for (int i = lane; i < n0; i += 64) coord[i] = cold.dom_xy[doff + i];
    for (int k = lane; k < n_new; k += 64) coord[n0 + k] = vnew[k];
    const bool too_many = build_segment_lists(quads, n_elem, n0, 0, nv, adj, deg, nullptr);
    if (__ballot(too_many) != 0ULL) {
        if (lane == 0 && !kWrite) { count_out[env] = 0; status_out[env] = kSampDegree; }
        return;
    }
    wave_sync();
    SampCtx s;
    s.coord = coord; s.adj = adj; s.deg = deg; s.lr = lr;
    auto vid = [&](int g) { return (g & kNewBit) ? n0 + (g & ~kNewBit) : g; };
    for (int e = lane; e < n_elem; e += 64) {
        const int4 q = quads[e];
        const int v[4] = {vid(q.x), vid(q.y), vid(q.z), vid(q.w)};
        good[e] = sample_element_quality(s, v, P.index) >= P.quality_threshold ? 1 : 0;
    }
    wave_sync();
    const int N = P.n_neighbor, R = P.n_radius;
    const int row = 2 * (2 * N + R);
    const double two_pi = 6.2832;                                   // round(2 * math.pi, 4)
    long long base = kWrite ? offsets[env] : 0;
    long long total = 0;
    bool sector_overflow = false;
    const int items = 4 * n_elem;
    for (int it0 = 0; it0 < items; it0 += 64) {
        const int it = it0 + lane;
        const bool in = it < items;
        long long cnt = 0;
        int rp = 0, l_p = 0, r_p = 0, target = 0;
        int nR = 0, nL = 0, nRt = 0, nLt = 0;
        unsigned short sec[kSampMaxRadius][kSampSecMax];
        int sec_n[kSampMaxRadius] = {0, 0, 0, 0};
        P2 syn[kSampMaxRadius];
        long long fans = 1;
        bool live = false;
        if (in && good[it >> 2]) {
            live = true;
            const int4 q = quads[it >> 2];
            const int v[4] = {vid(q.x), vid(q.y), vid(q.z), vid(q.w)};
            const int i = it & 3;
            rp = v[i]; l_p = v[(i + 1) & 3]; r_p = v[(i + 3) & 3]; target = v[(i + 2) & 3];
            int path[3];
            for (int k = 0; sample_path(s, N, r_p, rp, l_p, k, path); k++) { nR++; nRt += path_has(path, N, target) ? 1 : 0; }
            for (int k = 0; sample_path(s, N, l_p, rp, r_p, k, path); k++) { nL++; nLt += path_has(path, N, target) ? 1 : 0; }
            const double angle = cw(sp(s, rp), sp(s, l_p), sp(s, r_p));
            for (int j = 1; j <= R; j++) {
                const double sa = ((double)(j - 1) * angle) / (double)R, ea = ((double)j * angle) / (double)R;   // i * angle / N
                sec_n[j - 1] = sample_sector(s, nv, rp, l_p, r_p, target, P.radius, sa, ea, sec[j - 1], syn[j - 1], sector_overflow);
                fans *= (long long)(sec_n[j - 1] + 1);
            }
            cnt = (long long)nR * fans * nL - (long long)nRt * fans * nLt;
        }
        // exclusive prefix of cnt over the lanes
        long long incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const long long up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const long long chunk_total = __shfl(incl, 63, 64);
        if (kWrite && live && cnt > 0) {
            long long at = base + total + (incl - cnt);
            const P2 prp = sp(s, rp), prr = sp(s, r_p), pt = sp(s, target);
            int rr[3], ll[3];
            for (int kr = 0; sample_path(s, N, r_p, rp, l_p, kr, rr); kr++) {
                const bool t_r = path_has(rr, N, target);
                for (long long f = 0; f < fans; f++) {
                    // itertools.product(*reversed(neighbors)): the first sector's choice varies fastest
                    int pick[kSampMaxRadius];
                    long long rem = f;
                    for (int j = 0; j < R; j++) { pick[j] = (int)(rem % (sec_n[j] + 1)); rem /= (sec_n[j] + 1); }
                    for (int kl = 0; sample_path(s, N, l_p, rp, r_p, kl, ll); kl++) {
                        const bool t_l = path_has(ll, N, target);
                        if (t_r && t_l) continue;
                        if (at >= row_end) continue;   // never past this env's slice (the counts and the rows come from the same code)
                        // base_length: ((d(rp, rr0) + sum_rr) + sum_ll) + d(rp, ll0), sums from int 0, / (2 * n_neighbor)
                        double srr = 0.0, sll = 0.0;
                        for (int j = 1; j < N; j++) srr = (j == 1 ? 0.0 : srr) + dist_lr(lr, sp(s, rr[j]), sp(s, rr[j - 1]));
                        for (int j = 1; j < N; j++) sll = (j == 1 ? 0.0 : sll) + dist_lr(lr, sp(s, ll[j]), sp(s, ll[j - 1]));
                        double bl = dist_lr(lr, prp, sp(s, rr[0]));
                        bl = bl + (N > 1 ? srr : 0.0);   // (+ int 0 when the generator is empty)
                        bl = bl + (N > 1 ? sll : 0.0);
                        bl = (bl + dist_lr(lr, prp, sp(s, ll[0]))) / (double)(2 * N);
                        const double scale = bl * P.radius;
                        double *dst = samples + (size_t)at * row;
                        int w = 0;
                        auto put = [&](P2 p) {
                            dst[w++] = dist_lr(lr, prp, p) / scale;
                            dst[w++] = fmod(cw(prp, p, prr), two_pi);
                        };
                        for (int j = 0; j < N; j++) put(sp(s, rr[j]));
                        for (int j = R - 1; j >= 0; j--) put(pick[j] < sec_n[j] ? sp(s, sec[j][pick[j]]) : syn[j]);
                        for (int j = N - 1; j >= 0; j--) put(sp(s, ll[j]));
                        outputs[(size_t)at * 2] = dist_lr(lr, prp, pt) / scale;
                        outputs[(size_t)at * 2 + 1] = fmod(cw(prp, pt, prr), two_pi);
                        types[at] = t_r ? 1.0 : (t_l ? 0.0 : 0.5);
                        at++;
                    }
                }
            }
        }
        total += chunk_total;
    }
    if (!kWrite && lane == 0) {
        count_out[env] = total;
        status_out[env] = (unsigned char)kSampOk;
    }
    if (!kWrite && __ballot(sector_overflow) != 0ULL && lane == 0) {
        count_out[env] = 0;
        status_out[env] = (unsigned char)kSampSectorOverflow;
    }
}

}  // namespace meshenv
