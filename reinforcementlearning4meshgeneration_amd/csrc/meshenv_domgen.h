// meshenv_domgen.h -- the domain pipeline of BASELINE.json configs[4] on the device: one wavefront per
// ring generates, orients, densifies and rounds the polygon of seed (seed0 + k) and computes the per-domain constants, so
// that 65 536 ragged rings need no host loop.
//
// What is restated, and from where (host restatement with the same arithmetic: domains.py::random_domain, which the GPU
// test compares with bit for bit):
//   * CPython's random.Random(seed): MT19937 seeded by init_by_array over the 32-bit words of the seed, random() from two
//     outputs (53 bits), uniform(a, b) = a + (b - a) * random(), gauss() = the cached cos / sin pair of one
//     sqrt(-2 log(1 - u)) draw, randint() by rejection on the top bits (Lib/random.py, Modules/_randommodule.c);
//   * generatePolygon, ui/GenerateRandomPolygon.py:5-49: angle steps uniform in 2 pi / n -+ irregularity, normalised to a
//     full turn; radii gauss(aveRadius, spikeyness) clipped to [0, 2 aveRadius]; integer pixel coordinates by int();
//   * the orientation rule of the drawing UI's save (ui/tk-ui.py:84-101, 169-176): negative shoelace sum = clockwise, else
//     reversed;  read_polygon's division by 100 (general/polygon.py:110-117);
//   * domains.py::densify (uniform edge split, even vertex count) and Python's round(x, 4);
//   * Boundary2D.poly_area (general/components.py:477-479) and MeshGeneration.estimate_area_range (general/mesh.py:679-692).
// The only operations that are not IEEE-exact are cos / sin / log of the generator (ocml here, libm on the host: <= 2 ulp),
// and their results are truncated to integer pixels before anything else sees them.
#pragma once

#include "meshenv_geom.h"
#include "meshenv_state.h"

namespace meshenv {

constexpr int kMtN = 624, kMtM = 397;
constexpr int kGenMaxVerts = 64;     // numVerts of config 5 is randint(8, 64)
constexpr double kTwoPi = 6.283185307179586;  // 2 * math.pi == random.TWOPI

struct GenParams {
    unsigned long long seed0;
    double ctr_x, ctr_y, ave_radius, irregularity, spikeyness;  // generatePolygon arguments (250, 250, 100, 0.55, 0.7)
    double edge;                                                // densify target (0.45)
    int fixed_verts;                                            // > 0: numVerts for every ring; 0: randint(8, 64) per stream
    // ---- density mode (meshenv_create_random_density): the reference's own graded subdivision instead of the uniform split
    int density_mode;                    // 0: domains.densify (uniform); 1: Density.calculate_density, ui/tk-ui.py:252-276
    double base_length, density;         // spacing in pixels, per-vertex density (the same for every generated vertex)
    const unsigned long long *seeds;     // [n] explicit seed per ring, or nullptr: seed0 + k
    const double2 *dir_tab;              // (cos, sin)(clockwise_angle) by pixel offset, [(2 R + 1)^2], host libm
    int dir_r;                           // R
    unsigned char *raises;               // [n] nullable: 1 where calculate_density raises ZeroDivisionError
};

constexpr int kDensMaxPts = 2048;    // points of one densified ring (LDS staging of the density mode)

// LDS of one generating wavefront
struct GenScratch {
    unsigned mt[kMtN];           // MT19937 state, then its tempered outputs
    double u[2 * kGenMaxVerts + 8];  // random() draws in consumption order
    double step[kGenMaxVerts];
    double ang[kGenMaxVerts];
    int px[kGenMaxVerts], py[kGenMaxVerts];
    double2 ring[kGenMaxVerts];  // deduplicated polygon / 100, clockwise
    int pieces[kGenMaxVerts], off[kGenMaxVerts + 1];
    int qx[kGenMaxVerts], qy[kGenMaxVerts];   // the same polygon in pixels, generated order (density mode)
};

// ------------------------------------------------------------------------------------------ Density.calculate_density
//
// ui/tk-ui.py:252-276 as domains.calculate_density restates it, followed by the orientation rule of the UI's save
// (:84-101, 169-176) and read_polygon's division by 100 (general/polygon.py:110-117) -- domains.density_domain, the
// reference's whole route from a drawn / generated pixel polygon to a domain ring.  One wavefront per polygon:
//   * the dict keyed by coordinates: a repeated pixel keeps its first position (and its LAST density);
//   * edge i runs from vertex i - 1 to vertex i (edge 0 from the last vertex): A = density(prev) * base_length,
//     B = density(i) * base_length, L = its length, x = round((2 L - A - B) / (A + B)) (Python round: half to even),
//     e = (B - A) / x -- ZeroDivisionError for x == 0, reported per ring -- offsets A (j + 1) + e (j^2 + j) / 2 for j < x,
//     then L; points prev + offset * (cos, sin)(clockwise_angle(prev, i));
//   * the last edge drops its middle point (index len // 2) when the total would be odd;
//   * reversed unless the shoelace sum (builtin sum: left to right from 0) is negative; / 100.
// cos / sin of the edge direction are the one non-IEEE step; pixel polygons have INTEGER offsets, so the host evaluates
// clockwise_angle / math.cos / math.sin with its own libm -- the reference's -- once per offset (dir_tab, meshenv_hip.hip:
// fill_direction_table) or once per edge (edge_dir, explicit polygons): the device result is the reference's bit for bit.
struct DensScratch {
    double px[kGenMaxVerts * 4], py[kGenMaxVerts * 4];   // deduplicated polygon, pixels (explicit polygons: up to 256 vertices)
    double dens[kGenMaxVerts * 4];
    double2 dir[kGenMaxVerts * 4];                     // (cos, sin) of edge i
    double A[kGenMaxVerts * 4], E[kGenMaxVerts * 4], L[kGenMaxVerts * 4];   // L: filled by the caller (edge lengths)
    int X[kGenMaxVerts * 4], cnt[kGenMaxVerts * 4], off[kGenMaxVerts * 4 + 1];
    double2 res[kDensMaxPts];
};
constexpr int kDensMaxVerts = kGenMaxVerts * 4;

// m polygon vertices are in d->px / py / dens (dict-deduplicated, m >= 3), the directions of their edges in d->dir and
// the edge lengths in d->L (edge i = vertex i - 1 -> vertex i).
// Returns the number of ring points (kWrite: written to out_xy), 0 with *status = 1 where the reference raises
// ZeroDivisionError, 0 with *status = 2 when the ring does not fit kDensMaxPts.
template <bool kWrite>
__device__ __forceinline__ int density_ring(DensScratch *d, int m, double base_length, int lane, double2 *out_xy, int *status)
{
    bool zero = false, huge = false;
    for (int i = lane; i < m; i += 64) {
        const int ip = i == 0 ? m - 1 : i - 1;
        const double B = d->dens[i] * base_length, A = d->dens[ip] * base_length;
        const double L = d->L[i];
        const double xr = rint((2 * L - A - B) / (A + B));            // round(): to nearest, ties to even
        // an edge of more points than a whole ring may hold (or a non-finite quotient) never reaches the int conversion:
        // (int)xr is undefined out of range, and a wrapped count would pass the size check below as a negative total
        const bool too_many = !(xr <= (double)kDensMaxPts);
        huge = huge || too_many;
        const int x = too_many ? kDensMaxPts : (xr < -(double)kDensMaxPts ? -kDensMaxPts : (int)xr);
        zero = zero || x == 0;
        d->A[i] = A; d->X[i] = x;
        d->E[i] = x != 0 ? (B - A) / xr : 0.0;
        d->cnt[i] = (x > 0 ? x : 0) + 1;                               // range(x) offsets, then L
    }
    if (__ballot(zero) != 0ULL) {
        *status = 1;
        return 0;
    }
    if (__ballot(huge) != 0ULL) {
        *status = 2;
        return 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int total = 0;
    if (lane == 0) {
        int o = 0;   // (each count <= kDensMaxPts + 1 and m <= kDensMaxVerts: the sum stays far inside int; saturate anyway)
        for (int i = 0; i < m; i++) { d->off[i] = o; o = o > kDensMaxPts ? kDensMaxPts + 2 : o + d->cnt[i]; }
        d->off[m] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    total = d->off[m];
    // M:270-272: on the last edge, (len(res_points) + len(interpolations)) odd -> pop(int(len(interpolations) / 2))
    const int last_len = d->cnt[m - 1];
    const bool pop = (total & 1) != 0;
    const int pop_at = last_len / 2;
    const int n_out = total - (pop ? 1 : 0);
    if (n_out > kDensMaxPts || n_out <= 0) {
        *status = 2;
        return 0;
    }
    if (!kWrite) return n_out;
    for (int e = 0; e < m; e++) {
        const int ep = e == 0 ? m - 1 : e - 1;
        const double A = d->A[e], E = d->E[e], L = d->L[e];
        const int x = d->X[e], c = d->cnt[e], o = d->off[e];
        const double2 cs = d->dir[e];
        const double bx = d->px[ep], by = d->py[ep];
        for (int j = lane; j < c; j += 64) {
            // A * (j + 1) + e * (j ** 2 + j) / 2  -- float * int, float * int, / 2, +
            const double t = j < x ? A * (double)(j + 1) + (E * (double)(j * j + j)) / 2 : L;
            int slot = o + j;
            bool skip = false;
            if (pop && e == m - 1) {
                skip = j == pop_at;
                slot -= j > pop_at ? 1 : 0;
            }
            if (!skip) d->res[slot] = make_double2(bx + t * cs.x, by + t * cs.y);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // check_clockwise: sum([p[i-1].x * p[i].y - p[i-1].y * p[i].x]) < 0, builtin sum from int 0, left to right
    double s = 0.0;
    if (lane == 0) {
        for (int i = 0; i < n_out; i++) {
            const double2 p = d->res[i], q = d->res[i == 0 ? n_out - 1 : i - 1];
            s = s + (q.x * p.y - q.y * p.x);
        }
    }
    const bool reversed = !(uniform_f64(s) < 0.0);
    for (int i = lane; i < n_out; i += 64) {
        const double2 p = d->res[reversed ? n_out - 1 - i : i];
        out_xy[i] = make_double2(p.x / 100, p.y / 100);
    }
    return n_out;
}

// dict keyed by coordinates over n pixel vertices (px, py, dens in `src`): first position, last density; compacted into d.
template <typename T>
__device__ __forceinline__ int density_dedupe(DensScratch *d, const T *sx, const T *sy, const double *sd, double dens_all,
                                              int n, int lane)
{
    int m = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool first = false;
        double dv = dens_all;
        if (i < n) {
            first = true;
            for (int j = 0; j < i; j++) first = first && !(sx[j] == sx[i] && sy[j] == sy[i]);
            if (sd) {
                dv = sd[i];
                for (int j = i + 1; j < n; j++) dv = (sx[j] == sx[i] && sy[j] == sy[i]) ? sd[j] : dv;   // the last density wins
            }
        }
        const unsigned long long mk = __ballot(first);
        if (first) {
            const int r = m + __popcll(mk & ((1ULL << lane) - 1ULL));
            d->px[r] = (double)sx[i]; d->py[r] = (double)sy[i]; d->dens[r] = dv;
        }
        m += __popcll(mk);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return m;
}

// Explicit pixel polygons (meshenv_density_rings): polygon k = vertices [poly_off[k], poly_off[k + 1]) of pxy / dens;
// edge_dir / edge_len: direction (cos, sin) and length of the edge INTO deduplicated vertex r of polygon k at index
// poly_off[k] + r, from the host's libm.  count[k] = ring length, status[k] = 0 ok / 1 raises / 2 too long / 3 fewer than
// 3 distinct vertices.
template <bool kWrite>
__global__ void __launch_bounds__(64)
k_density_rings(int n, const int32_t *poly_off, const double *pxy, const double *dens, const double2 *edge_dir,
                const double *edge_len, double base_length, const int32_t *out_off, double2 *out_xy, int32_t *count,
                unsigned char *status)
{
    __shared__ DensScratch d;
    __shared__ double sx[kDensMaxVerts], sy[kDensMaxVerts];
    const int k = blockIdx.x, lane = threadIdx.x;
    const int o = poly_off[k], nv = poly_off[k + 1] - o;
    for (int i = lane; i < nv; i += 64) { sx[i] = pxy[2 * (o + i)]; sy[i] = pxy[2 * (o + i) + 1]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int m = density_dedupe(&d, sx, sy, dens ? dens + o : nullptr, 1.0, nv, lane);
    int st = 0, c = 0;
    if (m < 3) st = 3;
    else {
        for (int i = lane; i < m; i += 64) { d.dir[i] = edge_dir[o + i]; d.L[i] = edge_len[o + i]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        c = density_ring<kWrite>(&d, m, base_length, lane, kWrite ? out_xy + out_off[k] : nullptr, &st);
    }
    if (lane == 0) {
        if (!kWrite) count[k] = c;
        status[k] = (unsigned char)st;
    }
}

// init_by_array + one twist + tempering: out[0..623] are the stream's first 624 outputs (lane 0 runs the two serial
// recurrences, the tempering is per element)
__device__ __forceinline__ void mt_seed_and_fill(GenScratch *g, unsigned long long seed, int lane)
{
    if (lane == 0) {
        unsigned *mt = g->mt;
        mt[0] = 19650218u;
        for (int i = 1; i < kMtN; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
        const unsigned key[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
        const int klen = key[1] != 0u ? 2 : 1;
        int i = 1, j = 0;
        for (int k = kMtN; k > 0; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (unsigned)j;
            i++; j++;
            if (i >= kMtN) { mt[0] = mt[kMtN - 1]; i = 1; }
            if (j >= klen) j = 0;
        }
        for (int k = kMtN - 1; k > 0; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (unsigned)i;
            i++;
            if (i >= kMtN) { mt[0] = mt[kMtN - 1]; i = 1; }
        }
        mt[0] = 0x80000000u;
        for (int kk = 0; kk < kMtN; kk++) {  // genrand's in-place twist
            const unsigned y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % kMtN] & 0x7fffffffu);
            mt[kk] = mt[(kk + kMtM) % kMtN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kMtN; i += 64) {
        unsigned y = g->mt[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        g->mt[i] = y;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// random(): two consecutive outputs -> 53 bits
__device__ __forceinline__ double mt_random(const GenScratch *g, int pos)
{
    const unsigned a = g->mt[pos] >> 5, b = g->mt[pos + 1] >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double clipd(double x, double lo, double hi) { return lo > hi ? x : (x < lo ? lo : (x > hi ? hi : x)); }

// One ring.  kWrite = false: only the vertex count (the host sizes the domain table from it); true: the ring itself.
// Returns the vertex count, 0 on failure (fail bit set in *err: the stream's first 624 outputs did not suffice).
template <bool kWrite>
__device__ __forceinline__ int gen_ring(GenScratch *g, const GenParams &P, unsigned long long seed, int lane, double2 *out_xy,
                                        int *err, DensScratch *dsc = nullptr, int ring_index = 0)
{
    mt_seed_and_fill(g, seed, lane);
    int pos = 0;
    int nv = P.fixed_verts;
    if (nv <= 0) {  // randint(8, 64) = 8 + _randbelow(57): 6 random bits until the value is below 57
        unsigned r = g->mt[pos++] >> 26;
        while (r >= 57u && pos < kMtN) r = g->mt[pos++] >> 26;
        nv = 8 + (int)r;
    }
    int n_ring = 0;
    {
        const int n_rand = nv + 1 + 2 * ((nv + 1) / 2);
        if (pos + 2 * n_rand > kMtN) {
            if (lane == 0) { if (P.raises) P.raises[ring_index] = 2; else atomicOr(err, 1); }
            return 0;
        }
        // ---- generatePolygon
        const double irr = clipd(P.irregularity, 0.0, 1.0) * 2 * kPi / nv;
        const double spk = clipd(P.spikeyness, 0.0, 1.0) * P.ave_radius;
        const double lower = (2 * kPi / nv) - irr, upper = (2 * kPi / nv) + irr;
        for (int i = lane; i < n_rand; i += 64) g->u[i] = mt_random(g, pos + 2 * i);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < nv) g->step[lane] = lower + (upper - lower) * g->u[lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {  // the two left-to-right sums of the reference
            double total = 0.0;
            for (int i = 0; i < nv; i++) total = total + g->step[i];
            const double k = total / (2 * kPi);
            double angle = 0.0 + (2 * kPi - 0.0) * g->u[nv];
            for (int i = 0; i < nv; i++) {
                g->ang[i] = angle;
                const double s = g->step[i] / k;
                angle = angle + s;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < nv) {
            // gauss(): call 2p draws (u_a, u_b) and returns cos(2 pi u_a) * sqrt(-2 log(1 - u_b)); call 2p + 1 returns the sin half
            const int pr = lane >> 1;
            const double ua = g->u[nv + 1 + 2 * pr], ub = g->u[nv + 2 + 2 * pr];
            const double x2pi = ua * kTwoPi;
            const double g2rad = sqrt(-2.0 * log(1.0 - ub));
            const double z = ((lane & 1) ? sin(x2pi) : cos(x2pi)) * g2rad;
            const double r_i = clipd(P.ave_radius + z * spk, 0.0, 2 * P.ave_radius);
            const double a = g->ang[lane];
            g->px[lane] = (int)(P.ctr_x + r_i * cos(a));
            g->py[lane] = (int)(P.ctr_y + r_i * sin(a));
        }
        pos += 2 * n_rand;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- consecutive duplicates (int() truncation) dropped, closing duplicate dropped; / 100
        bool keep = false;
        if (lane < nv) keep = lane == 0 || g->px[lane] != g->px[lane - 1] || g->py[lane] != g->py[lane - 1];
        unsigned long long m = __ballot(keep);
        int cnt = __popcll(m);
        const int last = 63 - __clzll((long long)m);
        if (cnt > 1 && g->px[last] == g->px[0] && g->py[last] == g->py[0]) {
            m &= ~(1ULL << last);
            keep = keep && lane != last;
            cnt -= 1;
        }
        if (keep) {
            const int r = __popcll(m & ((1ULL << lane) - 1ULL));
            g->ring[r] = make_double2(g->px[lane] / 100.0, g->py[lane] / 100.0);
            g->qx[r] = g->px[lane];
            g->qy[r] = g->py[lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // random_polygon_px() draws another polygon from the same stream when fewer than 5 distinct vertices remain (with
        // numVerts >= 8 that needs four coincident pixels in a row); gauss() carries its cached half over to that draw, which
        // is not restated here: the ring is reported as failed instead, loudly
        if (cnt < 5) {   // (density mode with a flag array: flagged 2 = "not generated here", the caller leaves the seed out)
            if (lane == 0) { if (P.raises) P.raises[ring_index] = 2; else atomicOr(err, 1); }
            return 0;
        }
        n_ring = cnt;
    }
    if (dsc != nullptr) {
        // ---- density mode: calculate_density on the pixel polygon (generated order), then orientation and / 100
        const int m = density_dedupe(dsc, g->qx, g->qy, nullptr, P.density, n_ring, lane);
        int st = 0, c = 0;
        if (m < 3) st = 3;
        else {
            const int R = P.dir_r, W = 2 * R + 1;
            bool out_of_table = false;
            for (int i = lane; i < m; i += 64) {
                const int ip = i == 0 ? m - 1 : i - 1;
                const int dx = (int)(dsc->px[i] - dsc->px[ip]), dy = (int)(dsc->py[i] - dsc->py[ip]);   // clockwise_angle(prev, i)
                const bool in = dx >= -R && dx <= R && dy >= -R && dy <= R;
                out_of_table = out_of_table || !in;
                dsc->dir[i] = in ? P.dir_tab[(size_t)(dx + R) * W + (dy + R)] : make_double2(1.0, 0.0);
                dsc->L[i] = sqrt((double)(dx * dx + dy * dy));   // pixel coordinates are Python ints: math.sqrt of an exact int
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (__ballot(out_of_table) != 0ULL) st = 2;
            else c = density_ring<kWrite>(dsc, m, P.base_length, lane, out_xy, &st);
        }
        if (lane == 0) {
            if (st == 1 && P.raises) P.raises[ring_index] = 1;
            // (with a flag array every failure is reported per ring -- 2 = "not generated here": a ring beyond 2048 points,
            // an edge outside the direction table -- so that a probe over many seeds can leave exactly those seeds out)
            if (st >= 2) { if (P.raises) P.raises[ring_index] = 2; else atomicOr(err, 4); }
            if (st == 1 && !P.raises) atomicOr(err, 8);
        }
        return c;
    }
    // ---- orientation: sum(x[i-1] * y[i] - y[i-1] * x[i]) < 0 is clockwise, else the list is reversed
    bool reversed = false;
    {
        double s = 0.0;
        if (lane == 0) {
            s = 0.0;
            for (int i = 0; i < n_ring; i++) {
                const double2 p = g->ring[i], q = g->ring[i == 0 ? n_ring - 1 : i - 1];
                s = s + (q.x * p.y - q.y * p.x);
            }
        }
        reversed = !(uniform_f64(s) < 0.0);
    }
    auto at = [&](int i) { return g->ring[reversed ? n_ring - 1 - i : i]; };
    // ---- densify: ceil(length / edge) pieces per edge, one more on the last edge when the total is odd
    int pc = 0;
    if (lane < n_ring) {
        const double2 p0 = at(lane), p1 = at(lane + 1 == n_ring ? 0 : lane + 1);
        const double dx = p1.x - p0.x, dy = p1.y - p0.y;
        const double length = sqrt(dx * dx + dy * dy);
        const int c = (int)ceil(length / P.edge);
        pc = c < 1 ? 1 : c;
    }
    int total = pc;
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if ((total & 1) && lane == n_ring - 1) pc += 1;
    total += total & 1;
    if (!kWrite) return total;
    if (lane < n_ring) g->pieces[lane] = pc;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int o = 0;
        for (int i = 0; i < n_ring; i++) { g->off[i] = o; o += g->pieces[i]; }
        g->off[n_ring] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = 0; e < n_ring; e++) {
        const double2 p0 = at(e), p1 = at(e + 1 == n_ring ? 0 : e + 1);
        const int pcs = g->pieces[e], o = g->off[e];
        for (int j = lane; j < pcs; j += 64) {
            const double t = (double)j / (double)pcs;
            const double x = p0.x + (p1.x - p0.x) * t, y = p0.y + (p1.y - p0.y) * t;
            out_xy[o + j] = make_double2(round4_py(x), round4_py(y));
        }
    }
    return total;
}

__device__ __forceinline__ unsigned long long ring_seed(const GenParams &P, int k)
{
    return P.seeds ? P.seeds[k] : P.seed0 + (unsigned long long)k;
}

// pass 1: ring lengths
__global__ void __launch_bounds__(64) k_gen_count(GenParams P, int n, int32_t *count, int *err)
{
    __shared__ GenScratch g;
    const int k = blockIdx.x, lane = threadIdx.x;
    const int c = gen_ring<false>(&g, P, ring_seed(P, k), lane, nullptr, err);
    if (lane == 0) count[k] = c;
}

// pass 2: the rings, written at their offsets of the domain table
__global__ void __launch_bounds__(64) k_gen_rings(GenParams P, int n, const int32_t *offsets, double2 *dom_xy, int *err)
{
    __shared__ GenScratch g;
    const int k = blockIdx.x, lane = threadIdx.x;
    const int c = gen_ring<true>(&g, P, ring_seed(P, k), lane, dom_xy + offsets[k], err);
    if (lane == 0 && c != offsets[k + 1] - offsets[k]) atomicOr(err, 2);
}

// the same two passes in density mode (their own kernels: the 80 KB of density staging stay out of the uniform mode's LDS)
__global__ void __launch_bounds__(64) k_gen_count_density(GenParams P, int n, int32_t *count, int *err)
{
    __shared__ GenScratch g;
    __shared__ DensScratch d;
    const int k = blockIdx.x, lane = threadIdx.x;
    const int c = gen_ring<false>(&g, P, ring_seed(P, k), lane, nullptr, err, &d, k);
    if (lane == 0) count[k] = c;
}

__global__ void __launch_bounds__(64) k_gen_rings_density(GenParams P, int n, const int32_t *offsets, double2 *dom_xy, int *err)
{
    __shared__ GenScratch g;
    __shared__ DensScratch d;
    const int k = blockIdx.x, lane = threadIdx.x;
    const int c = gen_ring<true>(&g, P, ring_seed(P, k), lane, dom_xy + offsets[k], err, &d, k);
    if (lane == 0 && c != offsets[k + 1] - offsets[k]) atomicOr(err, 2);
}

// Per-domain constants on the device (what domains.py::domain_constants computes on the host): original_area =
// 0.5 |x . roll(y, 1) - y . roll(x, 1)| (the host goes through BLAS dot, whose summation order is unspecified: agreement to
// ~1e-15 relative, tested to 1e-12), and estimate_area_range over the ring's edge lengths SORTED ascending and summed in
// that order (rank sort in LDS, one wavefront per ring; dynamic LDS: 16 B per vertex).
__global__ void __launch_bounds__(64) k_dom_consts(int n_domains, const int32_t *offsets, const double2 *dom_xy, DomConst *dom)
{
    extern __shared__ double2 smem[];
    double *len = (double *)smem;   // [n] edge lengths, ring order
    double *srt = len + (offsets[blockIdx.x + 1] - offsets[blockIdx.x]);
    const int d = blockIdx.x, lane = threadIdx.x;
    const int off = offsets[d], n = offsets[d + 1] - off;
    const double2 *xy = dom_xy + off;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double2 p = xy[i], q = xy[i == 0 ? n - 1 : i - 1];
        const double dx = q.x - p.x, dy = q.y - p.y;
        len[i] = sqrt(dx * dx + dy * dy);   // distance(points[i - 1], points[i])
        a += p.x * q.y;                      // x . roll(y, 1)
        b += p.y * q.x;                      // y . roll(x, 1)
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {  // stable rank: sorted() keeps equal lengths in ring order
        const double v = len[i];
        int r = 0;
        for (int j = 0; j < n; j++) {
            const double w = len[j];
            r += (w < v || (w == v && j < i)) ? 1 : 0;
        }
        srt[r] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double total = 0.0;
        for (int i = 0; i < n; i++) total += srt[i];
        const double L = total / n;
        const double max_l = srt[n - 2] < 2 * L ? srt[n - 2] : 2 * L;
        const double min_l = L / sqrt(2.0) < srt[1] ? L / sqrt(2.0) : srt[1];
        const double crit_l = (max_l + 3 * min_l) / 4;
        DomConst dc;
        dc.orig_area = 0.5 * fabs(a - b);
        dc.min_area = min_l * min_l;
        dc.crit_area = crit_l * crit_l;
        dc.bl = 0.0; dc.ct = 1.0; dc.st = 0.0;
        dc.off = off; dc.n0 = n; dc.ref = -1; dc.pad = 0;
        dom[d] = dc;
    }
}

}  // namespace meshenv
