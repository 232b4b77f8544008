// meshenv_quality.h -- per-element quality report of the generated meshes.
//
// The reference scores a finished mesh by exporting it (general/mesh.py:1842-1864) and asking VTK's Verdict for five
// quad measures, then printing range / average / standard deviation per measure
// (Measurement/quality_verdict.py:77-90, 133-148).  VTK is a third-party dependency that is not in the image, so the
// arithmetic follows the reference's own in-repo analogues, Mesh.get_quality(type) (general/components.py:863-933):
//   record[0..7] = min corner angle (deg), max corner angle (deg), 's_jacobian', 'stretch', 'taper', 'robust',
//                  compute_area()[0], 'default'
// (checked in tests/test_gpu_quality.py against the CPU restatement, which is bit-identical to the reference's methods
// on the tests/golden/quality_quads.npz fixture).
//
// One wavefront per environment, one element per lane (episodes have 10-150 elements): the 16-byte quad record and its
// four vertices are gathered straight from the logs (vertex ids < n0 -> the domain table, kNewBit ids -> the env's
// created-vertex log), 64 bytes of measures are written per element, and the per-mesh statistics fall out of one DPP
// reduction -- HBM traffic is the logs once in, the records once out.
#pragma once

#include "meshenv_geom.h"
#include "meshenv_state.h"

namespace meshenv {

constexpr int kQualityDim = 8;

__device__ __forceinline__ double wave_max_f64(double v)
{
#define STEP(CTRL, MASK)                                        \
    {                                                           \
        const double ov = dpp_f64<CTRL, MASK>(-kInf, v);        \
        v = ov > v ? ov : v;                                    \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    return lane_f64(v, 63);
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#define STEP(CTRL, MASK) { v += dpp_f64<CTRL, MASK>(0.0, v); }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    return lane_f64(v, 63);
}

// the eight measures of one quad m[0..3] (Mesh.vertices order)
__device__ __forceinline__ void element_quality(const P2 *m, double *q)
{
    const double kPi = 3.141592653589793;
    double ang[4], amin = kInf, amax = -kInf, err = -kInf;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ang[i] = cw(m[i], m[(i + 1) & 3], m[(i + 3) & 3]);  // components.py:878-881
        amin = ang[i] < amin ? ang[i] : amin;
        amax = ang[i] > amax ? ang[i] : amax;
        const double e = fabs(ang[i] - kPi / 2);  // get_ave_error_angle, components.py:855-861
        err = e > err ? e : err;
    }
    q[0] = amin * (180.0 / kPi);  // math.degrees
    q[1] = amax * (180.0 / kPi);
    // edge lengths e[i] = d(v[i], v[i-1]) and diagonals
    double e[4], emin = kInf, emax = -kInf;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        e[i] = dist(m[i], m[(i + 3) & 3]);
        emin = e[i] < emin ? e[i] : emin;
        emax = e[i] > emax ? e[i] : emax;
    }
    const double d0 = dist(m[0], m[2]), d1 = dist(m[1], m[3]);
    {
        // 's_jacobian', components.py:891-906: p0..p3 = vertices[0], [-1], [-2], [-3]
        const P2 p0 = m[0], p1 = m[3], p2 = m[2], p3 = m[1];
        const double l0x = p1.x - p0.x, l0y = p1.y - p0.y, l1x = p2.x - p1.x, l1y = p2.y - p1.y;
        const double l2x = p3.x - p2.x, l2y = p3.y - p2.y, l3x = p0.x - p3.x, l3y = p0.y - p3.y;
        const double a3 = crossp(l2x, l2y, l3x, l3y), a2 = crossp(l1x, l1y, l2x, l2y);
        const double a1 = crossp(l0x, l0y, l1x, l1y), a0 = crossp(l3x, l3y, l0x, l0y);
        const double n0 = sqrt(l0x * l0x + l0y * l0y), n1 = sqrt(l1x * l1x + l1y * l1y);
        const double n2 = sqrt(l2x * l2x + l2y * l2y), n3 = sqrt(l3x * l3x + l3y * l3y);
        double j = a0 / (n0 * n3), t = a1 / (n0 * n1);
        j = t < j ? t : j;
        t = a2 / (n1 * n2);
        j = t < j ? t : j;
        t = a3 / (n2 * n3);
        j = t < j ? t : j;
        q[2] = j;
        // 'taper', components.py:885-890
        const double x1x = (p1.x - p0.x) + (p2.x - p3.x), x1y = (p1.y - p0.y) + (p2.y - p3.y);
        const double x2x = (p2.x - p1.x) + (p3.x - p0.x), x2y = (p2.y - p1.y) + (p3.y - p0.y);
        const double x12x = (p0.x - p1.x) + (p2.x - p3.x), x12y = (p0.y - p1.y) + (p2.y - p3.y);
        const double len1 = sqrt(x1x * x1x + x1y * x1y), len2 = sqrt(x2x * x2x + x2y * x2y);
        q[4] = sqrt(x12x * x12x + x12y * x12y) / (len2 < len1 ? len2 : len1);
    }
    const double stretch = sqrt(2.0) * emin / (d1 > d0 ? d1 : d0);  // components.py:870-872
    q[3] = stretch;
    q[5] = sqrt(stretch * (amin / amax));  // 'robust', components.py:873-884
    // compute_area, components.py:935-950: corner_1 = angle at v0, corner_3 = angle at v2
    q[6] = 0.5 * e[0] * e[1] * sin(ang[0]) + 0.5 * e[2] * e[3] * sin(ang[2]);
    const double aspect = emin != 0 ? emax / emin : 0.001;  // get_aspect_ratio, components.py:839-844
    q[7] = 1 / (aspect + err);  // 'default', components.py:864-869
}

// which = 0: the running episode of every env; 1: the last archived (finished) episode.
// elem_out [E][log_cap][8] (rows >= count untouched), stats_out [E][8][4] = min, mean, max, variance,
// count_out [E]; each pointer may be NULL.
__global__ void __launch_bounds__(64)
k_element_quality(DevState S, int which, double *__restrict__ elem_out, double *__restrict__ stats_out,
                  int32_t *__restrict__ count_out)
{
    const int env = blockIdx.x, lane = lane_id();
    const int cap = S.prm.log_cap;
    const EnvScalars sc = S.scal[env];
    const DevCold cold = load_cold(S);
    int half = (uniform_i32(sc.status) >> 4) & 1;
    int ne = uniform_i32(sc.n_elem);
    if (which) {
        ne = uniform_i32(cold.last_ep[env].n_elem);
        half ^= 1;
    }
    ne = ne < cap ? ne : cap;
    const int doff = uniform_i32(S.dom[uniform_i32(sc.dom)].off);
    const int4 *quads = reinterpret_cast<const int4 *>(cold.log_quads + ((size_t)env * 2 + half) * cap * 4);
    const double2 *vnew = cold.log_vxy + ((size_t)env * 2 + half) * cap;
    double mn[kQualityDim], mx[kQualityDim], s1[kQualityDim], s2[kQualityDim];
#pragma unroll
    for (int k = 0; k < kQualityDim; k++) { mn[k] = kInf; mx[k] = -kInf; s1[k] = 0; s2[k] = 0; }
    for (int i = lane; i < ne; i += 64) {
        const int4 g = quads[i];
        const int gid[4] = {g.x, g.y, g.z, g.w};
        P2 m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // a created vertex past log_capacity was never stored (MESHENV_ST_LOG_OVERFLOW): NaN record, no read
            const int kn = gid[k] & ~kNewBit;
            const bool created = (gid[k] & kNewBit) != 0;
            const double kNaN = __builtin_nan("");
            const double2 v = created ? (kn < cap ? vnew[kn] : make_double2(kNaN, kNaN)) : cold.dom_xy[doff + gid[k]];
            m[k] = mkp(v.x, v.y);
        }
        double q[kQualityDim];
        element_quality(m, q);
        if (elem_out) {
            double2 *dst = reinterpret_cast<double2 *>(elem_out + ((size_t)env * cap + i) * kQualityDim);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = make_double2(q[2 * k], q[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < kQualityDim; k++) {
            mn[k] = q[k] < mn[k] ? q[k] : mn[k];
            mx[k] = q[k] > mx[k] ? q[k] : mx[k];
            s1[k] += q[k];
            s2[k] += q[k] * q[k];
        }
    }
    if (count_out && lane == 0) count_out[env] = ne;
    if (stats_out) {
        double r = 0;
#pragma unroll
        for (int k = 0; k < kQualityDim; k++) {
            const double a = wave_min_f64(mn[k]), b = wave_sum_f64(s1[k]), c = wave_max_f64(mx[k]), d = wave_sum_f64(s2[k]);
            const double avg = ne ? b / ne : 0.0;
            const double var = ne ? d / ne - avg * avg : 0.0;  // vtkMeshQuality: E[q^2] - E[q]^2
            const int j = lane - 4 * k;
            r = j == 0 ? (ne ? a : 0.0) : (j == 1 ? avg : (j == 2 ? (ne ? c : 0.0) : (j == 3 ? var : r)));
        }
        if (lane < 4 * kQualityDim) stats_out[(size_t)env * 4 * kQualityDim + lane] = r;
    }
}

// MeshGeneration.get_quality(element, index) (general/mesh.py:1728-1747) for arbitrary quads, the indices that are a
// function of the four vertices alone: 0 = Mesh.get_quality() 'default' (C:864-869), 1 = compute_element_quality
// (M:1714-1726) = sqrt(q1 q2) of Mesh.get_quality_3 (C:952-972), 3 = 'stretch' (C:870-872), 4 = 'robust' (C:873-884),
// 5 = 'strong' (C:907-930).  One quad per lane: 64 B in, 8 B out.
__device__ __forceinline__ double quad_quality_index(const P2 *m, int index)
{
    const double kPi = 3.141592653589793;
    double e[4], ang[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        e[i] = dist(m[i], m[(i + 3) & 3]);
        ang[i] = cw(m[i], m[(i + 1) & 3], m[(i + 3) & 3]);
    }
    if (index == 0 || index == 3 || index == 4) {
        double q[kQualityDim];
        element_quality(m, q);
        return index == 0 ? q[7] : (index == 3 ? q[3] : q[5]);
    }
    const double area = 0.5 * e[0] * e[1] * sin(ang[0]) + 0.5 * e[2] * e[3] * sin(ang[2]);  // compute_area, C:935-950
    double q1 = 0.0;  // get_quality_3, C:952-972
    if (area > 0) {
        const double ra = sqrt(area);
        double product = 1.0;
#pragma unroll
        for (int i = 0; i < 4; i++) product *= (ra - e[i] > 0) ? e[i] / ra : 1.0 / (e[i] / ra);
        q1 = pow(product, 0.25);
    }
    double ap = 1.0;
#pragma unroll
    for (int i = 0; i < 4; i++) ap *= 1 - (fabs(ang[i] * (180.0 / kPi) - 90) / 90);
    const double q2 = ap < 0 ? 0.0 : pow(ap, 0.25);
    if (index == 1) return sqrt(q1 * q2);
    double amin = kInf, amax = -kInf;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double a = fabs(ang[i]);
        amin = a < amin ? a : amin;
        amax = a > amax ? a : amax;
    }
    return sqrt(q1 * (amin / amax));  // 'strong'
}

__global__ void __launch_bounds__(64)
k_quad_quality(int n, const double2 *__restrict__ quad_xy, int index, double *__restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    P2 m[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double2 v = quad_xy[(size_t)i * 4 + k];
        m[k] = mkp(v.x, v.y);
    }
    out[i] = quad_quality_index(m, index);
}

}  // namespace meshenv
