// meshenv_smooth.h -- post-processing smoothing of the generated mesh:
// MeshGeneration.smooth_pave(vertices, current_boundary_vertices, iteration=..., interior=True), general/mesh.py:790-795
//   = smooth_fixed_vertices (M:1258-1288) over the generated vertices that are off the front
//   + find_reference_candidates(target_angle=0) (M:233-261).
// (interior=False additionally runs smooth_current_boundary_3, M:939-1028, on the front itself: not built.)
//
// The reference relaxes on the Vertex.segments graph.  The hot path does not keep that graph; everything the smoother
// reads from it is recoverable from the element log: only generated vertices move (`vertex in self.original_vertices`
// skips the domain's), a generated vertex receives segments only from Mesh.connect_vertices (C:832-837) of the elements
// that contain it -- for i = 0..3 the pair (vertices[i], vertices[i-1]), unless the two are connected already -- and
// get_connected_vertices (C:115-124) lists the other end of every segment in list order.  So the neighbour list of a
// generated vertex = the partners of those pairs over the elements in log order, first occurrence only; for a vertex at
// position p of a quad the two pairs come in the order (p, p-1), (p+1, p) for p < 3 and (0, 3), (3, 2) for p = 3.
//
// The relaxation is Gauss-Seidel in boundary.vertices order (every vertex sees the new position of its predecessors),
// each coordinate accumulated term by term as x += (neighbour.x + vertex.x); the sweeps stop when the running sum of the
// moved vertices' x + y differs from the previous sweep's by <= 0.001, or after `iteration` sweeps.  Bit parity with that
// needs the same additions in the same order with the same operands.  One wavefront per env; inside it a sweep runs level
// by level (k_smooth_interior: level = longest path from below in the list order, so that a level's vertices are mutually
// non-adjacent and see exactly what the serial sweep would have shown them), one lane per vertex of the level, each lane
// summing its own <= 16 neighbours in list order; only the stop rule's running sum is a serial chain (v_readlane per
// vertex).  The graph (16 x uint16 per generated vertex) is built in LDS by one lane per generated vertex scanning the
// element log, and the coordinates of the domain ring and of the generated vertices sit in one LDS array, so that a sweep
// touches no HBM: HBM traffic is the logs once in, the moved coordinates once out.
#pragma once

#include "meshenv_kernels.h"

namespace meshenv {

constexpr int kSmoothMaxDeg = 16;   // neighbours per generated vertex (a quad-mesh vertex has 3-6; more sets code -3)

enum { kSmoothSkipped = -1, kSmoothLogOverflow = -2, kSmoothDegree = -3, kSmoothNotFinished = -4, kSmoothIndexError = -5, kSmoothRaises = -6,
       kSmoothNonFinite = -7 };

// LDS: coord[ring_cap + log_cap] double2 | xy_sum[log_cap] double | adj[log_cap][16] uint16 | lvl[log_cap] uint16 |
//      deg[log_cap] uint8 | front[log_cap] uint8
__host__ __device__ __forceinline__ size_t smooth_lds_bytes(int ring_cap, int log_cap)
{
    return (size_t)(ring_cap + log_cap) * sizeof(double2) + (size_t)log_cap * (8 + kSmoothMaxDeg * 2 + 2 + 2) + 64;
}

// Vertex.segments as neighbour lists, for the vertices first .. first + count - 1 (unified index: domain vertex i -> i,
// k-th generated vertex -> n0 + k), one lane per vertex, every lane scanning the element log in order (all lanes read
// the same 16-byte record).  Row r = vertex first + r: adj[r][0 .. deg[r]), n_mesh[r] (nullable) = elements that contain
// it.  A domain vertex starts with its two ring segments in the order general/mesh.py:1926-1930 creates them.  on_hole[r]
// (nullable) = 1 when the vertex has an edge that is used an odd number of times by (domain ring + elements): such edges
// bound what is not meshed yet, i.e. their vertices ARE the current front -- which makes the front recoverable from the
// logs of an archived episode, whose ring has been reset.  Returns true (per lane) when a vertex has more than
// kSmoothMaxDeg neighbours.
__device__ __forceinline__ bool build_segment_lists(const int4 *__restrict__ quads, int n_elem, int n0, int first, int count,
                                                    unsigned short *adj, unsigned char *deg, unsigned char *n_mesh,
                                                    unsigned char *on_hole = nullptr)
{
    const int lane = lane_id();
    bool too_many = false;
    for (int r0 = 0; r0 < count; r0 += 64) {
        const int r = r0 + lane, u_me = first + r;
        const int me = u_me >= n0 ? (kNewBit | (u_me - n0)) : u_me;
        unsigned short mine[kSmoothMaxDeg] = {};
        unsigned long long uses = 0ULL;   // 4-bit use count per neighbour slot
        int d = 0, nm = 0;
        if (u_me < n0) {  // for i in range(n0): Segment(v[i-1], v[i]) is appended to v[i-1], then to v[i]
            const int prev = u_me == 0 ? n0 - 1 : u_me - 1, next = u_me == n0 - 1 ? 0 : u_me + 1;
            mine[0] = (unsigned short)(u_me == n0 - 1 ? next : prev);
            mine[1] = (unsigned short)(u_me == n0 - 1 ? prev : next);
            d = 2;
            uses = 0x11ULL;
        }
        for (int e = 0; e < n_elem; e++) {
            const int4 q = quads[e];
            const int p = q.x == me ? 0 : (q.y == me ? 1 : (q.z == me ? 2 : (q.w == me ? 3 : -1)));
            if (p < 0 || r >= count) continue;
            nm += 1;
            // pairs (i, i-1) for i = 0..3 that contain position p, in the order of i
            const int first_g = p == 0 ? q.w : (p == 1 ? q.x : (p == 2 ? q.y : q.x));
            const int second_g = p == 0 ? q.y : (p == 1 ? q.z : (p == 2 ? q.w : q.z));
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int g = s == 0 ? first_g : second_g;
                const unsigned short u = (unsigned short)((g & kNewBit) ? n0 + (g & ~kNewBit) : g);
                int at = -1;
#pragma unroll
                for (int j = 0; j < kSmoothMaxDeg; j++) at = (at < 0 && j < d && mine[j] == u) ? j : at;
                if (at < 0) {
                    if (d < kSmoothMaxDeg) {
#pragma unroll
                        for (int j = 0; j < kSmoothMaxDeg; j++)
                            if (j == d) mine[j] = u;
                        at = d;
                        d += 1;
                    } else {
                        too_many = true;
                    }
                }
                if (at >= 0) uses += 1ULL << (4 * at);
            }
        }
        if (r < count) {
#pragma unroll
            for (int j = 0; j < kSmoothMaxDeg; j++) adj[(size_t)r * kSmoothMaxDeg + j] = mine[j];
            deg[r] = (unsigned char)d;
            if (n_mesh) n_mesh[r] = (unsigned char)(nm > 255 ? 255 : nm);
            if (on_hole) on_hole[r] = (uses & 0x1111111111111111ULL) != 0ULL ? 1 : 0;
        }
    }
    return too_many;
}

// One wavefront per env.  sweeps_out[env]: sweeps made (>= 1 when a vertex could move; 1 with nothing to move, as the
// reference's loop), or kSmoothSkipped (masked out), kSmoothLogOverflow (graph incomplete: nothing changed),
// kSmoothDegree.  diff_out[env]: the last |sum - previous sum| (the number the reference prints).
__global__ void __launch_bounds__(64)
k_smooth_interior(DevState S, int ring_cap, int which, const uint8_t *__restrict__ mask, const int32_t *__restrict__ gate,
                  int iteration, int32_t *__restrict__ sweeps_out, double *__restrict__ diff_out)
{
    extern __shared__ double2 smem[];
    const int env = blockIdx.x, lane = lane_id();
    if (mask && mask[env] == 0) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothSkipped;
        return;
    }
    if (gate && gate[env] < 0) {   // the front smoother (k_smooth_front) refused or raised: smooth_pave stops there
        if (lane == 0 && sweeps_out) sweeps_out[env] = gate[env];
        return;
    }
    const int log_cap = S.prm.log_cap;
    const EnvScalars sc = S.scal[env];
    const DevCold cold = load_cold(S);
    int n = uniform_i32(sc.n), n_elem = uniform_i32(sc.n_elem), n_new = uniform_i32(sc.n_new);
    const int status = uniform_i32(sc.status);
    bool overflow = (status & kStLogOverflow) != 0;
    int half = (status >> 4) & 1;
    if (which) {   // the archived episode (meshenv_get_last_episode): its front is read off the logs, see build_segment_lists
        const LastEpisode le = cold.last_ep[env];
        if (uniform_i32(le.episodes) <= 0) {
            if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothNotFinished;
            return;
        }
        n_elem = uniform_i32(le.n_elem); n_new = uniform_i32(le.n_new);
        overflow = (uniform_i32(le.flags) & 2) != 0;
        half ^= 1;
        n = 0;
    }
    if (overflow || n_elem > log_cap || n_new > log_cap) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothLogOverflow;
        return;
    }
    const DomConst dc = S.dom[uniform_i32(sc.dom)];
    const int doff = uniform_i32(dc.off), n0 = uniform_i32(dc.n0);
    const size_t lbase = ((size_t)env * 2 + half) * log_cap;
    const int4 *quads = reinterpret_cast<const int4 *>(cold.log_quads + lbase * 4);
    double2 *vnew = cold.log_vxy + lbase;

    double2 *coord = smem;                                              // [0, n0): domain ring, [n0, n0 + n_new): generated
    double *xy_sum = (double *)(coord + ring_cap + log_cap);             // x + y of a moved vertex (the stop rule's terms)
    unsigned short *adj = (unsigned short *)(xy_sum + log_cap);
    unsigned short *lvl = adj + (size_t)log_cap * kSmoothMaxDeg;         // 0: does not move
    unsigned char *deg = (unsigned char *)(lvl + log_cap);
    unsigned char *front = deg + log_cap;

    for (int i = lane; i < n0; i += 64) coord[i] = cold.dom_xy[doff + i];
    for (int k = lane; k < n_new; k += 64) {
        coord[n0 + k] = vnew[k];
        front[k] = 0;
    }
    wave_sync();
    // `[v for v in vertices if v not in current_boundary_vertices]`, M:794
    const int32_t *rid = S.ring_id + (size_t)env * S.cap;
    for (int i = lane; i < n; i += 64) {
        const int g = rid[i];
        if (g & kNewBit) front[g & ~kNewBit] = 1;
    }
    // the neighbour lists: one lane per generated vertex, the element log read by all lanes at the same address
    const bool too_many = build_segment_lists(quads, n_elem, n0, n0, n_new, adj, deg, nullptr, which ? front : nullptr);
    if (__ballot(too_many) != 0ULL) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothDegree;
        return;
    }
    wave_sync();
    // Levels of the Gauss-Seidel order.  In list order vertex v reads the NEW position of an adjacent u < v and the OLD one
    // of an adjacent u > v; with level(v) = 1 + max level(u) over the movable neighbours u < v (longest path from below) no
    // two vertices of a level are adjacent, lower levels are finished and higher ones untouched when a level runs -- so a
    // level can be relaxed by one lane per vertex with exactly the values the serial sweep would have read.  (Monotone
    // fixpoint iteration from 1: the order of the updates inside a pass does not matter.)
    for (int k = lane; k < n_new; k += 64) lvl[k] = (front[k] == 0 && deg[k] != 0) ? 1 : 0;
    wave_sync();
    for (int pass = 0; pass <= n_new; pass++) {
        bool changed = false;
        for (int k = lane; k < n_new; k += 64) {
            const int cur = lvl[k];
            if (cur == 0) continue;
            int m = 0;
            const int d = deg[k];
            for (int j = 0; j < d; j++) {
                const int u = (int)adj[(size_t)k * kSmoothMaxDeg + j] - n0;
                if (u >= 0 && u < k) { const int lu = lvl[u]; m = lu > m ? lu : m; }
            }
            if (m + 1 != cur) { lvl[k] = (unsigned short)(m + 1); changed = true; }
        }
        wave_sync();
        if (__ballot(changed) == 0ULL) break;
    }
    int my_max = 0;
    for (int k = lane; k < n_new; k += 64) my_max = lvl[k] > my_max ? lvl[k] : my_max;
    const int n_levels = wave_max_i32(my_max);
    // smooth_fixed_vertices, M:1258-1288
    double sum_coordinates = 0.0, diffs = 100.0;
    int it = 0;
    while (diffs > 0.001 && it < iteration) {
        it += 1;
        for (int l = 1; l <= n_levels; l++) {
            for (int k = lane; k < n_new; k += 64) {
                if (lvl[k] != l) continue;
                const int d = deg[k];
                const double2 v = coord[n0 + k];
                double x = 0.0, y = 0.0;   // the reference starts from an int 0
                for (int j = 0; j < d; j++) {
                    const double2 c = coord[adj[(size_t)k * kSmoothMaxDeg + j]];
                    x += c.x + v.x;
                    y += c.y + v.y;
                }
                const double nx = x / (double)(2 * d), ny = y / (double)(2 * d);
                coord[n0 + k] = make_double2(nx, ny);   // read by no vertex of this level
                xy_sum[k] = nx + ny;
            }
            wave_sync();
        }
        // new_sum_coordinates += vertex.x + vertex.y in list order; a vertex that does not move adds nothing (an exact
        // + 0.0 here: the running sum starts at +0 and can never become -0)
        double new_sum = 0.0;
        for (int k0 = 0; k0 < n_new; k0 += 64) {
            const int k = k0 + lane;
            const double sv = (k < n_new && lvl[k] != 0) ? xy_sum[k] : 0.0;
            const int m = n_new - k0 < 64 ? n_new - k0 : 64;
            for (int j = 0; j < m; j++) new_sum += lane_f64(sv, j);
        }
        diffs = fabs(new_sum - sum_coordinates);
        sum_coordinates = new_sum;
    }
    for (int k = lane; k < n_new; k += 64)
        if (front[k] == 0 && deg[k] != 0) vnew[k] = coord[n0 + k];
    if (lane == 0) {
        if (sweeps_out) sweeps_out[env] = it;
        if (diff_out) diff_out[env] = diffs;
    }
}

// ------------------------------------------------------------------------------------------ finished meshes: smooth()
//
// MeshGeneration.smooth(boundary.vertices, iteration=...), general/mesh.py:1290-1392 -- what general/EBRD.py:391 runs once
// an episode has ended with a front of <= 5 vertices.  Every generated vertex, in list order, by the number of elements
// that contain it: 1 -> (two neighbours) the 4th-vertex estimate from their common neighbour, else a 0.999 pull towards
// each neighbour in turn; 2 -> (one interior + two front neighbours) the mean of two 4th-vertex estimates, else the pull;
// otherwise the Laplacian step of smooth_fixed_vertices.  Front vertices move too, the stop rule sums x + y over the whole
// vertex list.  The estimate branches read two-hop neighbourhoods and the front, so the sweep is run in list order by the
// whole wavefront (wave-uniform control flow, LDS broadcast reads; the Laplacian gather and the front scan use the
// lanes); the graph holds the domain vertices' lists as well (get_common_vertex walks them).

__host__ __device__ __forceinline__ size_t smooth_final_lds_bytes(int ring_cap, int log_cap)
{
    const size_t V = (size_t)ring_cap + log_cap;
    return V * sizeof(double2) + V * kSmoothMaxDeg * 2 + (size_t)ring_cap * 2 + V * 3 + 64;
}

struct SmoothGraph {
    const double2 *coord;
    const unsigned short *adj;
    const unsigned char *deg;
};

__device__ __forceinline__ bool graph_has(const SmoothGraph &g, int v, int w)
{
    const int d = uniform_i32((int)g.deg[v]);
    bool have = false;
    for (int j = 0; j < d; j++) have = have || (int)g.adj[(size_t)v * kSmoothMaxDeg + j] == w;
    return have;
}

// Boundary2D.compute_dist([v for v in a.get_common_vertex(b) if v is not skip], point)[0][0], C:162-165,369-376: of the
// neighbours of a (in a's list order) that are also neighbours of b, the nearest to `point` (first among equals); -1: none
__device__ __forceinline__ int nearest_common(const SmoothGraph &g, int a, int b, int skip, int point)
{
    const int da = uniform_i32((int)g.deg[a]);
    const double2 pp = g.coord[point];
    int best = -1;
    double bd = 0.0;
    for (int j = 0; j < da; j++) {
        const int w = uniform_i32((int)g.adj[(size_t)a * kSmoothMaxDeg + j]);
        if (w == skip || !graph_has(g, b, w)) continue;
        const double2 cw_ = g.coord[w];
        const double dd = dist(mkp(pp.x, pp.y), mkp(cw_.x, cw_.y));
        if (best < 0 || dd < bd) { best = w; bd = dd; }
    }
    return best;
}

// Mesh.estimate_4th_vertex, C:980-990 -> Segment.get_ray_segment / build_ray, C:588-610
__device__ __forceinline__ double2 estimate_4th_vertex(double2 origin, double2 left, double2 right, double factor,
                                                       bool has_suggest, double suggest_dist)
{
    const P2 o = mkp(origin.x, origin.y);
    double distance = (dist(o, mkp(left.x, left.y)) + dist(o, mkp(right.x, right.y))) * factor;
    if (has_suggest) {
        const double lim = 0.6 * suggest_dist;
        distance = lim < distance ? lim : distance;
    }
    const double r1x = (origin.x + origin.x) / 2, r1y = (origin.y + origin.y) / 2;
    const double r2x = (left.x + right.x) / 2, r2y = (left.y + right.y) / 2;
    const SinCos sc = sincos_nc(atan2_nc(r2y - r1y, r2x - r1x));
    return make_double2(r1x + distance * sc.c, r1y + distance * sc.s);
}

// One wavefront per env whose episode has ended complete (front <= 5).  which = 0: the running episode, not yet reset
// (step with auto_reset = 0); which = 1: the archived episode (the other half of the logs: what auto-reset left behind;
// its front -- smooth() only asks which vertices are on it and how far they are -- is recovered from the logs, see
// build_segment_lists).  Codes as k_smooth_interior plus kSmoothNotFinished (front > 5 / nothing archived / archived
// episode truncated: untouched) and kSmoothIndexError (the reference raises IndexError at M:1311 / 1344 / 1348 on an
// empty common-neighbour list: untouched).
__global__ void __launch_bounds__(64)
k_smooth_final(DevState S, int ring_cap, int which, const uint8_t *__restrict__ mask, int iteration, double lr_1, double lr_2,
               int32_t *__restrict__ sweeps_out, double *__restrict__ diff_out)
{
    extern __shared__ double2 smem[];
    const int env = blockIdx.x, lane = lane_id();
    if (mask && mask[env] == 0) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothSkipped;
        return;
    }
    const int log_cap = S.prm.log_cap;
    const EnvScalars sc = S.scal[env];
    const DevCold cold = load_cold(S);
    int n = uniform_i32(sc.n), n_elem = uniform_i32(sc.n_elem), n_new = uniform_i32(sc.n_new);
    const int status = uniform_i32(sc.status);
    bool overflow = (status & kStLogOverflow) != 0, finished = n <= 5;
    int half = (status >> 4) & 1;
    if (which) {
        const LastEpisode le = cold.last_ep[env];
        n_elem = uniform_i32(le.n_elem); n_new = uniform_i32(le.n_new);
        overflow = (uniform_i32(le.flags) & 2) != 0;
        finished = uniform_i32(le.episodes) > 0 && (uniform_i32(le.flags) & 1) != 0;
        half ^= 1;
        n = 0;   // the front comes from the logs
    }
    if (overflow || n_elem > log_cap || n_new > log_cap || !finished) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = !finished ? kSmoothNotFinished : kSmoothLogOverflow;
        return;
    }
    const DomConst dc = S.dom[uniform_i32(sc.dom)];
    const int doff = uniform_i32(dc.off), n0 = uniform_i32(dc.n0), nv = n0 + n_new;
    const size_t lbase = ((size_t)env * 2 + half) * log_cap;
    const int4 *quads = reinterpret_cast<const int4 *>(cold.log_quads + lbase * 4);
    double2 *vnew = cold.log_vxy + lbase;

    const int V = ring_cap + log_cap;
    double2 *coord = smem;
    unsigned short *adj = (unsigned short *)(coord + V);
    unsigned short *ringu = adj + (size_t)V * kSmoothMaxDeg;     // the front as unified vertex indices
    unsigned char *deg = (unsigned char *)(ringu + ring_cap);
    unsigned char *n_mesh = deg + V;
    unsigned char *front = n_mesh + V;

    for (int i = lane; i < n0; i += 64) coord[i] = cold.dom_xy[doff + i];
    for (int k = lane; k < n_new; k += 64) coord[n0 + k] = vnew[k];
    for (int u = lane; u < nv; u += 64) front[u] = 0;
    wave_sync();
    const int32_t *rid = S.ring_id + (size_t)env * S.cap;
    for (int i = lane; i < n; i += 64) {
        const int g = rid[i];
        const int u = (g & kNewBit) ? n0 + (g & ~kNewBit) : g;
        ringu[i] = (unsigned short)u;
        front[u] = 1;
    }
    const bool too_many = build_segment_lists(quads, n_elem, n0, 0, nv, adj, deg, n_mesh, which ? front : nullptr);
    if (__ballot(too_many) != 0ULL) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothDegree;
        return;
    }
    wave_sync();
    if (which) {
        // the front of the archived episode: the vertices on edges used an odd number of times; none left (a front of 4
        // was closed by the last element, B:235-238) -> that element's four vertices
        int cnt = 0;
        for (int u0 = 0; u0 < nv; u0 += 64) {
            const int u = u0 + lane;
            const bool f = u < nv && front[u] != 0;
            const unsigned long long m = __ballot(f);
            if (f) {
                const int pos = cnt + __popcll(m & ((1ULL << lane) - 1ULL));
                if (pos < ring_cap) ringu[pos] = (unsigned short)u;
            }
            cnt += __popcll(m);
        }
        if (cnt == 0 && n_elem > 0) {
            const int4 q = quads[n_elem - 1];
            const int gq[4] = {q.x, q.y, q.z, q.w};
            if (lane < 4) {
                const int gg = lane == 0 ? gq[0] : (lane == 1 ? gq[1] : (lane == 2 ? gq[2] : gq[3]));
                const int u = (gg & kNewBit) ? n0 + (gg & ~kNewBit) : gg;
                ringu[lane] = (unsigned short)u;
                front[u] = 1;
            }
            cnt = 4;
        }
        n = cnt < ring_cap ? cnt : ring_cap;
        wave_sync();
    }
    SmoothGraph g;
    g.coord = coord; g.adj = adj; g.deg = deg;
    // the stop rule's sum runs over the domain ring first: that prefix never changes
    double sum_domain = 0.0;
    for (int i0 = 0; i0 < n0; i0 += 64) {
        const int i = i0 + lane;
        double sv = 0.0;
        if (i < n0) { const double2 c = coord[i]; sv = c.x + c.y; }
        const int m = n0 - i0 < 64 ? n0 - i0 : 64;
        for (int j = 0; j < m; j++) sum_domain += lane_f64(sv, j);
    }
    double sum_coordinates = 0.0, diffs = 100.0;
    int it = 0;
    bool index_error = false;
    while (!index_error && diffs > 0.001 && it < iteration) {
        it += 1;
        for (int k = n0; k < nv && !index_error; k++) {
            const int d = uniform_i32((int)deg[k]), nm = uniform_i32((int)n_mesh[k]);
            const unsigned short *row = adj + (size_t)k * kSmoothMaxDeg;
            bool pull = false;
            if (nm == 1) {
                if (d == 2) {
                    const int c0 = uniform_i32((int)row[0]), c1 = uniform_i32((int)row[1]);
                    const int origin = nearest_common(g, c0, c1, k, k);
                    if (origin < 0) { index_error = true; break; }
                    // compute_dist([front vertices that are neither neighbours of k nor k], origin)[0], M:1313-1318
                    const double2 po = coord[origin];
                    double bd = kInf;
                    int bi = 0x7fffffff;
                    for (int i = lane; i < n; i += 64) {
                        const int w = ringu[i];
                        if (w == c0 || w == c1 || w == k || w == origin) continue;
                        const double2 pw = coord[w];
                        const double dd = dist(mkp(po.x, po.y), mkp(pw.x, pw.y));
                        if (dd < bd) { bd = dd; bi = i; }
                    }
                    const double dmin = wave_min_f64(bd);
                    const int imin = -wave_max_i32(bd == dmin && bi != 0x7fffffff ? -bi : INT32_MIN + 1);
                    if (imin != 0x7fffffff) {   // `if not len(p_dist): continue`
                        const double2 nw = estimate_4th_vertex(po, coord[c0], coord[c1], 0.5, true, dmin);
                        if (lane == 0) coord[k] = nw;
                    }
                } else {
                    pull = true;
                }
            } else if (nm == 2) {
                int ub0 = -1, ub1 = -1, in0 = -1, nu = 0, ni = 0;
                for (int j = 0; j < d; j++) {
                    const int w = uniform_i32((int)row[j]);
                    if (uniform_i32((int)front[w]) != 0) { if (nu == 0) ub0 = w; else if (nu == 1) ub1 = w; nu += 1; }
                    else { if (ni == 0) in0 = w; ni += 1; }
                }
                if (ni == 1 && nu == 2) {
                    const int q1 = nearest_common(g, in0, ub0, k, k);
                    const int q2 = nearest_common(g, in0, ub1, k, k);
                    if (q1 < 0 || q2 < 0) { index_error = true; break; }
                    const double2 e1 = estimate_4th_vertex(coord[q1], coord[ub0], coord[in0], 0.7, false, 0.0);
                    const double2 e2 = estimate_4th_vertex(coord[q2], coord[ub1], coord[in0], 0.7, false, 0.0);
                    if (lane == 0) coord[k] = make_double2((e1.x + e2.x) / 2, (e1.y + e2.y) / 2);
                } else {
                    pull = true;
                }
            } else if (d != 0) {
                const double2 v = coord[k];
                double tx = 0.0, ty = 0.0;
                if (lane < d) {
                    const double2 c = coord[row[lane]];
                    tx = c.x + v.x;
                    ty = c.y + v.y;
                }
                double x = 0.0 + lane_f64(tx, 0), y = 0.0 + lane_f64(ty, 0);   // the reference starts from an int 0
                for (int j = 1; j < d; j++) {
                    x += lane_f64(tx, j);
                    y += lane_f64(ty, j);
                }
                if (lane == 0) coord[k] = make_double2(x / (double)(2 * d), y / (double)(2 * d));
            }
            if (pull) {
                const double lr = nm == 1 ? lr_1 : lr_2;
                double2 v = coord[k];
                for (int j = 0; j < d; j++) {
                    const double2 c = coord[uniform_i32((int)row[j])];
                    v.x = lr * v.x + (1 - lr) * c.x;
                    v.y = lr * v.y + (1 - lr) * c.y;
                }
                if (lane == 0) coord[k] = v;
            }
            wave_sync();
        }
        double new_sum = sum_domain;
        for (int k0 = 0; k0 < n_new; k0 += 64) {
            const int k = k0 + lane;
            double sv = 0.0;
            if (k < n_new) { const double2 c = coord[n0 + k]; sv = c.x + c.y; }
            const int m = n_new - k0 < 64 ? n_new - k0 : 64;
            for (int j = 0; j < m; j++) new_sum += lane_f64(sv, j);
        }
        diffs = fabs(new_sum - sum_coordinates);
        sum_coordinates = new_sum;
    }
    if (index_error) {
        if (lane == 0 && sweeps_out) sweeps_out[env] = kSmoothIndexError;
        return;
    }
    for (int k = lane; k < n_new; k += 64) vnew[k] = coord[n0 + k];
    if (!which) {
        // front vertices moved with the rest: the ring arrays hold copies of their coordinates
        double2 *rxy = S.ring_xy + (size_t)env * S.cap;
        for (int i = lane; i < n; i += 64) rxy[i] = coord[ringu[i]];
        if (lane == 0) S.scal[env].status = status & ~(kStRm1Bad | kStRp1Bad);
    }
    if (lane == 0) {
        if (sweeps_out) sweeps_out[env] = it;
        if (diff_out) diff_out[env] = diffs;
    }
}

// ------------------------------------------------------------------------------------------ the front smoother
//
// smooth_current_boundary_3, general/mesh.py:939-1028 (the interior=False half of smooth_pave, which move() enters when no
// reference vertex is selectable, rl/boundary_env.py:405-412): every generated vertex ON the front, in ring order and in
// place, by its interior angle -- <= 90: middle_vertex on the bisector for a target angle raised in steps of 5 until the
// new position keeps every surrounding element on its side (is_inside_boundary over clockwise_vertices); (90, 180]:
// side_vertex next to a sharp (< 45) neighbour corner, else find_indention_vertex; (180, 270]: find_indention_vertex;
// beyond: inner_vertex, then find_indention_vertex.  A vertex sees its predecessors' new positions, so the front is
// walked serially by the whole wavefront (wave-uniform control flow, LDS broadcast reads); the lanes carry the two
// atan2-heavy inner loops -- one lane per entry of the surrounding polygon in is_inside_boundary, one lane per front
// vertex / segment in find_indention_vertex's proximity tests.  Graph and coordinates as in k_smooth_final.

struct FrontState {
    double2 *coord;
    const unsigned short *adj;
    const unsigned char *deg;
    const unsigned short *ringu;
    const double *tab;   // the table above
    int n, n0;
    bool exact;    // pow2_glibc reproduces the running libm's pow(x, 2.0) (validated by the host): square like the reference
    int raised;    // 1: the reference raises here (math.sqrt of a negative number, a zero divisor between Python operands);
                   // 2: a construction with a NumPy zero divisor produced nan and the reference ACCEPTED it (see k_smooth_front)
};

__device__ __forceinline__ P2 ldc(const FrontState &f, int v)
{
    const double2 c = f.coord[v];
    return mkp(c.x, c.y);
}

// x ** 2 and Point2D.distance_to as the reference's libm evaluates them (csrc/meshenv_libm.h)
__device__ __forceinline__ double sq(const FrontState &f, double v) { return f.exact ? pow2_nc(v) : v * v; }
__device__ __forceinline__ double distf(const FrontState &f, P2 a, P2 b)
{
    return sqrt_pos(sq(f, a.x - b.x) + sq(f, a.y - b.y));
}

__device__ __forceinline__ double py_sqrt(FrontState &f, double v)
{
    if (v < 0) { f.raised = 1; return 0.0; }   // ValueError (also -inf); math.sqrt(nan) is nan
    return sqrt(v);
}

// a / b as the reference evaluates it: ZeroDivisionError when both operands are Python numbers, NumPy's IEEE quotient
// (inf / nan and a warning) when one of them is an np.float64 -- the coordinates of generated vertices are (B:123 rounds
// an array element), those of domain vertices are not: is_np = "a generated vertex's coordinate went into the divisor".
__device__ __forceinline__ double py_div(FrontState &f, double a, double b, bool is_np)
{
    if (b == 0 && !is_np) { f.raised = 1; return 0.0; }
    return a / b;
}

__device__ __forceinline__ double deg2rad(double a) { return a * (kPi / 180.0); }   // math.radians
__device__ __forceinline__ double rad2deg(double a) { return a * (180.0 / kPi); }   // math.degrees

// the two intersections of the circle |p - (a, b)| = dist with the line A x + B y = W + A a + B b, M:841-858 / 889-904
__device__ __forceinline__ void circle_line(FrontState &f, double a, double b, double A, double B, double W, double dist_,
                                            bool np_A, P2 &v1, P2 &v2)
{
    if (B == 0) {
        const double wa = py_div(f, W, A, np_A);   // W is a Python float (products of math.sqrt / math.cos results)
        const double r = py_sqrt(f, sq(f, dist_) - sq(f, wa));
        v1 = mkp(wa + a, b + r);
        v2 = mkp(wa + a, b - r);
    } else if (A == 0) {
        const double wb = W / B;
        const double r = py_sqrt(f, sq(f, dist_) - sq(f, wb));
        v1 = mkp(a + r, wb + b);
        v2 = mkp(a - r, wb + b);
    } else {
        const double M = -A / B;
        const double N = (W + A * a + B * b) / B;
        const double t = 2 * M * b - 2 * M * N + 2 * a;
        const double m2 = sq(f, M);
        const double disc = fabs(sq(f, t) - 4 * (m2 + 1) * (sq(f, N - b) + sq(f, a) - sq(f, dist_)));
        const double den = 2 * (m2 + 1);
        const double sq = sqrt(disc);
        v1.x = (t + sq) / den;
        v2.x = (t - sq) / den;
        v1.y = M * v1.x + N;
        v2.y = M * v2.x + N;
    }
}

// M:805-832; tan_half = math.tan(math.radians(target_angle / 2)) from the table
__device__ __forceinline__ P2 middle_vertex(const FrontState &f, P2 vertex, P2 left, P2 right, double tan_half)
{
    const P2 m = mkp((left.x + right.x) / 2, (left.y + right.y) / 2);
    const double A = right.x - left.x, B = right.y - left.y;
    const double D = distf(f, left, m) / tan_half;
    P2 v1, v2;
    if (B == 0) {
        v1 = mkp(m.x, m.y + D);
        v2 = mkp(m.x, m.y - D);
    } else if (A == 0) {
        v1 = mkp(m.x + D, m.y);
        v2 = mkp(m.x - D, m.y);
    } else {
        const double M = -A / B;
        const double N = A * m.x / B + m.y;
        const double t = -2 * M * N + 2 * m.x + 2 * M * m.y;
        const double m2 = sq(f, M);
        const double disc = fabs(sq(f, t) - 4 * (m2 + 1) * (sq(f, N - m.y) + sq(f, m.x) - sq(f, D)));
        const double den = 2 * (m2 + 1);
        const double sq = sqrt(disc);
        v1.x = (t + sq) / den;
        v2.x = (t - sq) / den;
        v1.y = M * v1.x + N;
        v2.y = M * v2.x + N;
    }
    return distf(f, v1, vertex) < distf(f, v2, vertex) ? v1 : v2;
}

// M:834-863; cos_a = math.cos(math.radians(angle)) from the table
__device__ __forceinline__ P2 side_vertex(FrontState &f, P2 vertex, P2 next_v, P2 nn_v, double cos_a, double d, bool np_A)
{
    const double W = d * distf(f, next_v, nn_v) * cos_a;
    P2 v1, v2;
    circle_line(f, next_v.x, next_v.y, nn_v.x - next_v.x, nn_v.y - next_v.y, W, d, np_A, v1, v2);
    return distf(f, v1, vertex) < distf(f, v2, vertex) ? v1 : v2;
}

// M:882-909
__device__ __forceinline__ P2 indention_vertex(FrontState &f, P2 vertex, P2 left, P2 right, double cos_a, double d, bool np_A)
{
    const double W = d * distf(f, vertex, left) * cos_a;
    P2 v1, v2;
    circle_line(f, vertex.x, vertex.y, left.x - vertex.x, left.y - vertex.y, W, d, np_A, v1, v2);
    if (f.raised) return vertex;
    return cw(v1, left, right) < cw(v2, left, right) ? v1 : v2;
}

// clockwise_vertices, M:1080-1101: the neighbours of `inner` sorted by clockwise angle (the selection sort as written),
// each followed by the common neighbour it shares with its successor.  out (LDS, 2 x kSmoothMaxDeg): vertex indices.
__device__ __forceinline__ int clockwise_vertices(const FrontState &f, int inner, unsigned short *out)
{
    int vs[kSmoothMaxDeg];
    const int n = uniform_i32((int)f.deg[inner]);
#pragma unroll
    for (int j = 0; j < kSmoothMaxDeg; j++) vs[j] = j < n ? uniform_i32((int)f.adj[(size_t)inner * kSmoothMaxDeg + j]) : 0;
    const P2 pin = ldc(f, inner);
    const int lane = lane_id();
    for (int i = 1; i < n; i++) {
        // one lane per candidate j in [i, n): angle to the previous pick; the largest wins, the first among equals
        int vj = 0, vp = 0;
#pragma unroll
        for (int j = 0; j < kSmoothMaxDeg; j++) {
            vj = j == lane ? vs[j] : vj;
            vp = j == i - 1 ? vs[j] : vp;
        }
        const bool act = lane >= i && lane < n;
        const double a = act ? cw(pin, ldc(f, vj), ldc(f, vp)) : -2.0;
        const double amax = -wave_min_f64(-a);
        const unsigned long long hit = __ballot(act && a == amax);
        const int flag = amax > -1 ? (int)__ffsll((long long)hit) - 1 : i;
        if (flag != i) {
            int t_i = 0, t_f = 0;
#pragma unroll
            for (int j = 0; j < kSmoothMaxDeg; j++) {
                t_i = j == i ? vs[j] : t_i;
                t_f = j == flag ? vs[j] : t_f;
            }
#pragma unroll
            for (int j = 0; j < kSmoothMaxDeg; j++) vs[j] = j == i ? t_f : (j == flag ? t_i : vs[j]);
        }
    }
    int m = 0;
    for (int i = 0; i < n; i++) {
        int cur = 0, prev = 0;
        const int ip = i == 0 ? n - 1 : i - 1;
#pragma unroll
        for (int j = 0; j < kSmoothMaxDeg; j++) {
            cur = j == i ? vs[j] : cur;
            prev = j == ip ? vs[j] : prev;
        }
        int inter = -1;
        const int dc = uniform_i32((int)f.deg[cur]), dp = uniform_i32((int)f.deg[prev]);
        for (int j = 0; j < dc && inter < 0; j++) {
            const int w = uniform_i32((int)f.adj[(size_t)cur * kSmoothMaxDeg + j]);
            if (w == inner) continue;
            bool have = false;
            for (int q = 0; q < dp; q++) have = have || (int)f.adj[(size_t)prev * kSmoothMaxDeg + q] == w;
            if (uniform_i32((int)have)) inter = w;
        }
        if (lane == 0) out[m] = (unsigned short)prev;
        m += 1;
        if (inter >= 0) {
            if (lane == 0) out[m] = (unsigned short)inter;
            m += 1;
        }
    }
    wave_sync();
    return m;
}

// is_inside_boundary, M:1069-1078: one lane per entry of the surrounding polygon
__device__ __forceinline__ bool is_inside_boundary(const FrontState &f, P2 original, P2 moved, const unsigned short *b, int nb,
                                                   int left, int right)
{
    const int lane = lane_id();
    bool bad = false;
    if (lane < nb) {
        const int bi = b[lane], bp = b[lane == 0 ? nb - 1 : lane - 1];
        const bool skip = (left == bi || left == bp) && (right == bi || right == bp);
        if (!skip) bad = (cw(moved, ldc(f, bi), ldc(f, bp)) < kPi) != (cw(original, ldc(f, bi), ldc(f, bp)) < kPi);
    }
    return __ballot(bad) == 0ULL;
}

// M:911-937
__device__ __forceinline__ P2 find_side_vertex(FrontState &f, int v, int _next, int next, int nn, double v_angle,
                                               unsigned short *cb)
{
    const P2 pv = ldc(f, v), pn = ldc(f, next), pnn = ldc(f, nn);
    const double d = (distf(f, pv, ldc(f, _next)) + distf(f, pv, pn) + distf(f, pn, pnn)) / 3;
    const bool np_A = next >= f.n0 || nn >= f.n0;
    double target = 45;
    for (int k = 0; k < 10; k++) {   // target = 45 - 5 k reaches 0 <= v_angle at k = 9 at the latest
        const P2 nv = side_vertex(f, pv, pn, pnn, f.tab[kFtCosSide + k], d, np_A);
        if (f.raised) return pv;
        if (target <= v_angle) return pv;
        const int nb = clockwise_vertices(f, v, cb);
        if (is_inside_boundary(f, pv, nv, cb, nb, _next, next)) return nv;
        target -= 5;
    }
    return pv;
}

// M:1030-1067 with indention_vertex M:882-909; index = ring slot of the vertex, q = quantum of its interior angle
__device__ __forceinline__ P2 find_indention_vertex(FrontState &f, int index, int q, unsigned short *cb)
{
    const int n = f.n, lane = lane_id();
    const int v = f.ringu[index], left = f.ringu[wrapi(index + 1, n)], right = f.ringu[wrapi(index - 1, n)];
    const int l2 = f.ringu[wrapi(index + 2, n)], r2 = f.ringu[wrapi(index - 2, n)];
    const P2 pv = ldc(f, v), pl = ldc(f, left), pr = ldc(f, right);
    const double d = (distf(f, pv, pl) + distf(f, pv, pr)) / 2;
    bool near = false, zero_div = false;
    for (int i = lane; i < n; i += 64) {
        // Boundary2D.get_closet_points(front, vertex, [r2, right, left, l2], d): any other front vertex within d
        const int w = f.ringu[i];
        if (!(w == v || w == r2 || w == right || w == left || w == l2)) near = near || distf(f, pv, ldc(f, w)) <= d;
        // find_closest_segments, M:1103-1112: a front segment not at the vertex whose foot point lies inside it, within d
        const int p1 = f.ringu[i == 0 ? n - 1 : i - 1], p2 = w;
        if (p1 != v && p2 != v) {
            const P2 a = ldc(f, p1), b = ldc(f, p2);
            const double A = b.x - a.x, B = b.y - a.y;
            const double den = sq(f, A) + sq(f, B);
            // C:647: a zero-length segment between two domain vertices raises (Python operands); with a generated end
            // the quotient is NumPy's 0 / 0 = nan, `0 <= s <= 1` is False and the segment is not "inner"
            zero_div = zero_div || (den == 0 && p1 < f.n0 && p2 < f.n0);
            const double s = (A * pv.x + B * pv.y - B * a.y - A * a.x) / den;
            near = near || (0 <= s && s <= 1 && distf(f, pv, mkp(a.x + s * A, a.y + s * B)) <= d);
        }
    }
    // (both loops of the reference run to the end whatever they find: a raising segment is always reached)
    if (__ballot(zero_div) != 0ULL) { f.raised = 1; return pv; }
    if (__ballot(near) == 0ULL) return pv;
    const double cos_a = f.tab[kFtCosInd + q];
    const bool np_A = v >= f.n0 || left >= f.n0;
    for (int times = 4; ; times++) {
        const P2 nv = indention_vertex(f, pv, pl, pr, cos_a, d / times, np_A);
        if (f.raised) return pv;
        if (times >= 10) return pv;
        const int nb = clockwise_vertices(f, v, cb);
        if (is_inside_boundary(f, pv, nv, cb, nb, left, right)) return nv;
    }
}

// One wavefront per env: smooth_current_boundary_3 on the running episode's front.  code_out[env]: 0 done,
// kSmoothSkipped / kSmoothLogOverflow / kSmoothDegree (untouched), kSmoothRaises (the reference raises: the vertices
// moved before that point stay moved, as in the reference), kSmoothNonFinite: a construction divided by a NumPy zero
// (two coincident vertices, one of them generated), the nan position passed is_inside_boundary -- every comparison with
// nan is False, so it passes exactly when the original position reads >= pi against every entry of its surrounding
// polygon -- and the reference assigned it; it then raises in the find_next_state that follows (int(nan), C:1257).  The
// kernel stops at that vertex; the usual outcome of a NumPy zero divisor is the other one: every trial is rejected and the
// vertex stays where it is.
__global__ void __launch_bounds__(64)
k_smooth_front(DevState S, int ring_cap, const uint8_t *__restrict__ mask, int32_t *__restrict__ code_out,
               const double *__restrict__ tab, int libm_exact, double2 *__restrict__ nv_xy, const int32_t *__restrict__ nv_count,
               const int32_t *__restrict__ nv_gid)
{
    extern __shared__ double2 smem[];
    const int env = blockIdx.x, lane = lane_id();
    if (mask && mask[env] == 0) {
        if (lane == 0) code_out[env] = kSmoothSkipped;
        return;
    }
    const int log_cap = S.prm.log_cap;
    const EnvScalars sc = S.scal[env];
    const DevCold cold = load_cold(S);
    const int n = uniform_i32(sc.n), n_elem = uniform_i32(sc.n_elem), n_new = uniform_i32(sc.n_new);
    const int status = uniform_i32(sc.status);
    if ((status & kStLogOverflow) || n_elem > log_cap || n_new > log_cap) {
        if (lane == 0) code_out[env] = kSmoothLogOverflow;
        return;
    }
    const DomConst dc = S.dom[uniform_i32(sc.dom)];
    const int doff = uniform_i32(dc.off), n0 = uniform_i32(dc.n0), nv = n0 + n_new;
    const size_t lbase = ((size_t)env * 2 + ((status >> 4) & 1)) * log_cap;
    const int4 *quads = reinterpret_cast<const int4 *>(cold.log_quads + lbase * 4);
    double2 *vnew = cold.log_vxy + lbase;

    const int V = ring_cap + log_cap;
    double2 *coord = smem;
    unsigned short *adj = (unsigned short *)(coord + V);
    unsigned short *ringu = adj + (size_t)V * kSmoothMaxDeg;
    unsigned short *cb = ringu + ring_cap;                          // clockwise_vertices output, 2 x kSmoothMaxDeg
    unsigned char *deg = (unsigned char *)(cb + 2 * kSmoothMaxDeg);

    for (int i = lane; i < n0; i += 64) coord[i] = cold.dom_xy[doff + i];
    for (int k = lane; k < n_new; k += 64) coord[n0 + k] = vnew[k];
    const int32_t *rid = S.ring_id + (size_t)env * S.cap;
    for (int i = lane; i < n; i += 64) {
        const int g = rid[i];
        ringu[i] = (unsigned short)((g & kNewBit) ? n0 + (g & ~kNewBit) : g);
    }
    const bool too_many = build_segment_lists(quads, n_elem, n0, 0, nv, adj, deg, nullptr);
    if (__ballot(too_many) != 0ULL) {
        if (lane == 0) code_out[env] = kSmoothDegree;
        return;
    }
    wave_sync();
    FrontState f;
    f.coord = coord; f.adj = adj; f.deg = deg; f.ringu = ringu; f.n = n; f.n0 = n0; f.raised = 0; f.tab = tab; f.exact = libm_exact != 0;
    for (int i = 0; i < n && !f.raised; i++) {
        const int v = uniform_i32((int)ringu[i]);
        if (v < n0) continue;   // `in self.original_vertices`
        const int nxt = uniform_i32((int)ringu[i + 1 == n ? 0 : i + 1]), prv = uniform_i32((int)ringu[i == 0 ? n - 1 : i - 1]);
        const P2 pv = ldc(f, v), pn = ldc(f, nxt), pp = ldc(f, prv);
        const double v_rad = cw(pv, pn, pp);
        const double v_angle = rad2deg(v_rad);
        const int q = min(max(uniform_i32(angle_q(v_rad)), 0), kFtQ - 1);
        P2 nw = pv;
        if (v_angle <= 90) {
            double target = v_angle >= 45 ? v_angle : 45;
            const int row = v_angle >= 45 ? min(max(q, kFtMidQ0), kFtMidQ1) - kFtMidQ0 + 1 : 0;
            for (int k = 0; k < kFtMidSteps; k++) {   // target reaches 135 after at most 18 steps of 5
                const P2 cand = middle_vertex(f, pv, pn, pp, tab[kFtTanMid + row * kFtMidSteps + k]);
                if (target >= 135) break;
                const int nb = clockwise_vertices(f, v, cb);
                if (is_inside_boundary(f, pv, cand, cb, nb, nxt, prv)) { nw = cand; break; }
                target += 5;
            }
        } else if (v_angle <= 180) {
            const int nn_r = uniform_i32((int)ringu[wrapi(i - 2, n)]), nn_l = uniform_i32((int)ringu[wrapi(i + 2, n)]);
            const double left_angle = rad2deg(cw(pn, ldc(f, nn_l), pv));    // compute_boundary_angle, C:467-473
            const double right_angle = rad2deg(cw(pp, pv, ldc(f, nn_r)));
            if (right_angle < 45) nw = find_side_vertex(f, v, nxt, prv, nn_r, right_angle, cb);
            else if (left_angle < 45) nw = find_side_vertex(f, v, prv, nxt, nn_l, left_angle, cb);
            else nw = find_indention_vertex(f, i, q, cb);
        } else if (v_angle <= 270) {
            nw = find_indention_vertex(f, i, q, cb);
        } else {
            // inner_vertex(vertex, 45), M:865-880 (A = vertex.x - m.x with a generated vertex: a NumPy divisor)
            const P2 m = mkp((pn.x + pp.x) / 2, (pn.y + pp.y) / 2);
            const double d = distf(f, m, pp) * tab[kFtTan45];
            const double A = pv.x - m.x, B = pv.y - m.y;
            const double qq = py_div(f, sq(f, d), sq(f, A) + sq(f, B), true);
            const double s = py_sqrt(f, qq);
            if (f.raised) break;
            const double ix = m.x + s * A, iy = m.y + s * B;
            if (!(isfinite(ix) && isfinite(iy))) { f.raised = 2; break; }
            if (lane == 0) coord[v] = make_double2(ix, iy);
            wave_sync();
            nw = find_indention_vertex(f, i, q, cb);
        }
        if (f.raised) break;
        if (!(isfinite(nw.x) && isfinite(nw.y))) { f.raised = 2; break; }
        if (lane == 0) coord[v] = make_double2(nw.x, nw.y);
        wave_sync();
    }
    // the front's generated vertices moved: vertex log and ring copies
    double2 *rxy = S.ring_xy + (size_t)env * S.cap;
    for (int i = lane; i < n; i += 64) {
        const int u = ringu[i];
        if (u >= n0) {
            const double2 c = coord[u];
            vnew[u - n0] = c;
            rxy[i] = c;
        }
    }
    // not_valid_points (the move() API) holds the Vertex objects themselves: a listed front vertex that moved is tested
    // (is_vertex_inside_list, M:428-433) at its new position from now on
    if (nv_xy) {
        const int n_nv = uniform_i32(nv_count[env]);
        for (int k = lane; k < n_nv; k += 64) {
            const int g = nv_gid[(size_t)env * ring_cap + k];
            if (g & kNewBit) nv_xy[(size_t)env * ring_cap + k] = coord[n0 + (g & ~kNewBit)];
        }
    }
    if (lane == 0) code_out[env] = f.raised == 2 ? kSmoothNonFinite : (f.raised ? kSmoothRaises : 0);
}

__host__ __device__ __forceinline__ size_t smooth_front_lds_bytes(int ring_cap, int log_cap)
{
    const size_t V = (size_t)ring_cap + log_cap;
    return V * sizeof(double2) + V * kSmoothMaxDeg * 2 + (size_t)ring_cap * 2 + 2 * kSmoothMaxDeg * 2 + V + 64;
}

// What step()'s closing find_next_state (B:250) will select from the rebuilt list, computed at rebuild time.
//
// The reference re-selects the reference vertex from the candidate list at the end of EVERY step(), accepted or not; the
// step kernels do so only after an accepted element, because between two accepted elements nothing the selection reads
// can change -- except through this rebuild (a list ordered by insertion becomes a list ordered by ring position, so its
// head may move).  The reference's next step() still acts on the OLD reference vertex and returns the observation of the
// NEW one; if that action is rejected the ring is the one at hand, so the new selection is known now.  It is parked here
// and committed by k_apply_reselect right after the next step kernel for the envs whose step extracted nothing (an
// extraction re-selects by itself).
struct alignas(32) Reselect {
    int32_t n_elem;      // len(generated_meshes) at the rebuild (-1: nothing pending); the next step rejected <=> unchanged
    int32_t ref;         // ring slot of the new reference vertex (-1: the reference returns None)
    double bl, ct, st;   // base length and action frame of the new point environment
};

// find_reference_candidates(target_angle=0), M:233-261, on the CURRENT front of every selected env: the candidate list is
// rebuilt from scratch (keys from the ring as it is, ties in ring order: stamp = -index, the insertion counter restarts).
// The point environment -- reference vertex, base length, observation -- is left alone, as in the reference, whose next
// step() still acts on the reference vertex chosen before the call; the selection that step will end with is parked in
// pend / pend_obs (above).
// kMode 0: park the selection (above).  kMode 1 / 2: the front moved (smooth_pave(interior=False)), so the point
// environment is recomputed at once -- the find_next_state() / find_next_state(static=True) that move() runs right after
// smooth_pave (B:420) -- and committed: reference vertex, base length, action frame, observation (also to obs_out).
template <int kMode>
__global__ void __launch_bounds__(64)
k_rebuild_candidates(DevState S, int cap, const uint8_t *__restrict__ mask, const int32_t *__restrict__ sweeps,
                     Reselect *__restrict__ pend, float *__restrict__ pend_obs, float *__restrict__ obs_out, const LibmRef lr)
{
    extern __shared__ double2 smem[];
    const int env = blockIdx.x;
    if (mask && mask[env] == 0) return;
    if (sweeps && sweeps[env] < 0) return;   // the smoother refused this env: nothing changed
    Ctx c;
    c.tie = true;   // the smoother has just moved front vertices onto half-quantum angles
    carve_lds(c, smem, cap);
    load_env(c, S, env);
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
    c.counter = 0;
    // the selection of the next step's find_next_state, on a scratch copy of the point environment
    const int ref0 = c.ref, status0 = c.status;
    const double bl0 = c.bl, ct0 = c.ct, st0 = c.st;
    const float obs0 = c.obs;
    BqArgs bq;
    bq.skip = false;
    bq.mode = 0; bq.a = 0; bq.b = 0; bq.ang0 = 0; bq.ang1 = 0; bq.q_ang0 = 0; bq.q_ang2 = 0; bq.half01 = 0; bq.half23 = 0;
    find_next_state(c, S, bq, kMode == 2, nullptr, 0, lr);
    if (kMode == 0) {
        if (c.lane < kObsDim) pend_obs[(size_t)env * kObsDim + c.lane] = c.obs;
        if (c.lane == 0) {
            Reselect r;
            r.n_elem = c.n_elem; r.ref = c.ref; r.bl = c.bl; r.ct = c.ct; r.st = c.st;
            pend[env] = r;
        }
        c.ref = ref0; c.status = status0; c.bl = bl0; c.ct = ct0; c.st = st0; c.obs = obs0;
    } else {
        c.status &= ~(kStRm1Bad | kStRp1Bad);   // the memo of rejected quads is tied to the reference vertex and the ring
        if (c.lane == 0) pend[env].n_elem = -1;
        if (obs_out && c.lane < kObsDim) obs_out[(size_t)env * kObsDim + c.lane] = c.obs;
    }
    c.ring_dirty = true;
    store_env(c, S);
}

// Runs right after the step kernel while a rebuild is pending: an env whose step extracted nothing (n_elem unchanged)
// takes the parked selection -- record, cached observation and the step's observation outputs; the others (extraction:
// re-selected by the step itself; reset: a fresh episode) only drop it.
__global__ void __launch_bounds__(64)
k_apply_reselect(DevState S, Reselect *__restrict__ pend, const float *__restrict__ pend_obs, float *__restrict__ obs_out)
{
    const int env = blockIdx.x, lane = lane_id();
    const Reselect r = pend[env];
    if (uniform_i32(r.n_elem) < 0) return;
    if (lane == 0) pend[env].n_elem = -1;
    EnvScalars *sc = S.scal + env;
    if (uniform_i32(sc->n_elem) != uniform_i32(r.n_elem)) return;
    if (lane < kObsDim) {
        const float o = pend_obs[(size_t)env * kObsDim + lane];
        S.obs_cache[(size_t)env * kObsDim + lane] = o;
        if (obs_out) obs_out[(size_t)env * kObsDim + lane] = o;
        if (S.msg) S.msg[(size_t)env * 21 + lane] = o;
    }
    if (lane == 0) {
        sc->ref = r.ref;
        sc->bl = r.bl; sc->ct = r.ct; sc->st = r.st;
        // the memo of rejected rule -1 / +1 quads is tied to the reference vertex
        sc->status = (sc->status & ~(kStRm1Bad | kStRp1Bad | kStNoReference)) | (r.ref < 0 ? kStNoReference : 0);
    }
}

// ------------------------------------------------------------------------------------------ move() through smooth_pave
//
// rl/boundary_env.py:405-426: when move() finds no selectable reference vertex on a front of more than 4 vertices it runs
// smooth_pave (front + interior smoothing, candidate rebuild), ends the episode if not_valid_points repeats the list of the
// previous smoothing (first entry, last entry, length; last_not_valid_points is set only here and survives reset()),
// empties the list and selects again (static observation); no reference vertex even then ends the episode.  k_move marks
// such envs kMoveNeedsSmoothing; meshenv_move then runs the smoothing kernels under this mask and k_move_finish.

__global__ void k_move_mask(const uint8_t *__restrict__ code, uint8_t *__restrict__ mask, int n)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) mask[e] = code[e] == (uint8_t)kMoveNeedsSmoothing ? 1 : 0;
}

__global__ void k_move_finish(DevState S, int cap, const uint8_t *__restrict__ mask, const int32_t *__restrict__ sweeps,
                              int32_t *__restrict__ nv_count, const int32_t *__restrict__ nv_gid, int32_t *__restrict__ nv_meta,
                              uint8_t *__restrict__ done, uint8_t *__restrict__ complete, uint8_t *__restrict__ code)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.n_envs || mask[e] == 0) return;
    const int sw = sweeps[e];
    if (sw == kSmoothRaises) {   // the reference raises out of move(): the caller resets
        code[e] = (uint8_t)kMoveSmoothRaises;
        done[e] = 1;
        return;
    }
    int32_t *m = nv_meta + (size_t)e * kNvMeta;
    if (sw == kSmoothNonFinite) {
        // the reference accepted a nan position: smooth_pave returns, B:416-422 run, then find_next_state raises (C:1257)
        const int cnt = nv_count[e];
        m[1] = cnt > 0 ? nv_gid[(size_t)e * cap] : 0; m[2] = cnt > 0 ? nv_gid[(size_t)e * cap + cnt - 1] : 0; m[3] = cnt; m[4] = m[0];
        nv_count[e] = 0;
        code[e] = (uint8_t)kMoveSmoothRaises;
        done[e] = 1;
        return;
    }
    if (sw < 0) return;          // graph not rebuildable (log overflow): stays kMoveNeedsSmoothing, done = 1
    const int cnt = nv_count[e], epoch = m[0];
    const int first = cnt > 0 ? nv_gid[(size_t)e * cap] : 0, last = cnt > 0 ? nv_gid[(size_t)e * cap + cnt - 1] : 0;
    bool dn = false;
    if (m[3] > 0 && cnt > 0) {
        // reset() deep-copies the domain (B:69): every vertex of an earlier episode is a different object
        dn = m[4] == epoch && first == m[1] && last == m[2] && cnt == m[3];
    }
    m[1] = first; m[2] = last; m[3] = cnt; m[4] = epoch;   // self.last_not_valid_points = self.not_valid_points
    nv_count[e] = 0;                                        // self.not_valid_points = []
    const bool none = (S.scal[e].status & kStNoReference) != 0;   // the selection k_rebuild_candidates<2> just made
    done[e] = (uint8_t)((dn || none) ? 1 : 0);
    complete[e] = 0;
    code[e] = (uint8_t)(none ? kMoveNone : kMoveOk);
}

}  // namespace meshenv
