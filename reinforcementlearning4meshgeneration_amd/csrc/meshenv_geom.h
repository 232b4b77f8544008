// meshenv_geom.h -- fp64 geometry primitives of the element-extraction hot path, device side (gfx950).
//
// Every function cites the reference lines it follows (general/components.py = "C:", general/mesh.py =
// "M:", rl/boundary_env.py = "B:", general/data.py = "D:").  Build with -ffp-contract=off: the reference
// is CPython float arithmetic, which never fuses a*b+c; the only fused operation is the explicit fma()
// inside round4_py.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace meshenv {

constexpr double kPi = 3.141592653589793;  // math.pi

struct P2 {
    double x, y;
};

__device__ __forceinline__ P2 mkp(double x, double y)
{
    P2 p;
    p.x = x;
    p.y = y;
    return p;
}

// ---------------------------------------------------------------------------------- rounding
// Python's round(float, 4): correctly rounded decimal rounding of the exact binary value, ties to even.
// y = |x|*1e4 (rounded) and e = fma(|x|,1e4,-y) (exact residual) give the exact product as y + e, so the
// comparison of its fractional part against 0.5 is exact; r/1e4 with IEEE division is the double nearest
// to the decimal r*1e-4, which is what CPython's dtoa/strtod round trip returns.
__device__ __forceinline__ double round4_py(double x)
{
    const double ax = fabs(x);
    const double y = ax * 1e4;
    if (!(y < 4503599627370496.0)) return x;  // also NaN/inf
    const double e = fma(ax, 1e4, -y);
    const double f = floor(y);
    const double t = (y - f) - 0.5;
    const double s = t + e;
    double r = f;
    if (s > 0.0) r = f + 1.0;
    else if (s == 0.0 && (((long long)f) & 1LL)) r = f + 1.0;
    return copysign(r / 1e4, x);
}

// round(np.float64, 4): numpy's multiply / rint / divide
__device__ __forceinline__ double round4_np(double x) { return rint(x * 1e4) / 1e4; }

// round(np.float32, 4): the same in float32
__device__ __forceinline__ float round4_npf(float x) { return rintf(x * 1e4f) / 1e4f; }

// ---------------------------------------------------------------------------------- primitives
// Point2D.distance_to, C:17-18.  (The reference's `** 2` is libm pow(x, 2.0), which differs from x*x by
// one ulp for 0.085 % of doubles; the device uses the exactly rounded product -- see DESIGN.md.)
__device__ __forceinline__ double dist(P2 a, P2 b)
{
    const double dx = a.x - b.x, dy = a.y - b.y;
    return sqrt(dx * dx + dy * dy);
}

// numerator / denominator of the clockwise angle: cross(v1,v2), dot(v1,v2) with v1 = p1-s, v2 = p2-s
__device__ __forceinline__ void cw_terms(P2 s, P2 p1, P2 p2, double &c, double &d)
{
    const double v1x = p1.x - s.x, v1y = p1.y - s.y;
    const double v2x = p2.x - s.x, v2y = p2.y - s.y;
    c = v1x * v2y - v1y * v2x;
    d = v1x * v2x + v1y * v2y;
}

__device__ __forceinline__ double cw_from_terms(double c, double d)
{
    const double theta = -atan2(c, d);
    return signbit(theta) ? round4_py(2 * kPi + theta) : round4_py(theta);
}

// Vertex.to_find_clockwise_angle, C:91-100: angle in [0, 2pi] quantised to 1e-4 rad
__device__ __forceinline__ double cw(P2 s, P2 p1, P2 p2)
{
    double c, d;
    cw_terms(s, p1, p2, c, d);
    return cw_from_terms(c, d);
}

// True when round(sin(cw(...)), 4) can only be non-zero.  The rounded angle is within 5e-5 rad of the
// true angle, so if the true angle is more than 1e-3 rad away from every multiple of pi (|tan| > 1e-3)
// the rounded one is at least 9.5e-4 away and |sin| >= 9.4e-4, which rounds to a non-zero 4-decimal
// value.  Lets Segment.straddle skip atan2/sin for every non-degenerate configuration; the exact path is
// taken otherwise, so results are identical to the unfiltered evaluation.
__device__ __forceinline__ bool surely_not_collinear(double c, double d) { return fabs(c) > 1e-3 * fabs(d); }

// cross_product, C:482-483
__device__ __forceinline__ double crossp(double ax, double ay, double bx, double by) { return ax * by - bx * ay; }

// Segment.straddle, C:491-516; self = (p1,p2), another = (q1,q2)
__device__ __noinline__ bool straddle(P2 p1, P2 p2, P2 q1, P2 q2)
{
    double c1, d1, c2, d2;
    cw_terms(p1, q1, p2, c1, d1);
    cw_terms(p1, q2, p2, c2, d2);
    bool collinear = false;
    if (!surely_not_collinear(c1, d1) && !surely_not_collinear(c2, d2)) {
        const double s1 = round4_py(sin(cw_from_terms(c1, d1)));
        const double s2 = round4_py(sin(cw_from_terms(c2, d2)));
        collinear = (s1 == s2) && (s2 == 0.0);
    }
    if (collinear) {
        const double l1 = dist(p1, p2), l2 = dist(q1, q2);
        if (l1 > l2) {
            const P2 m = mkp((p2.x + p1.x) / 2, (p2.y + p1.y) / 2);
            const double a = dist(m, q2), b = dist(m, q1);
            return (b < a ? b : a) <= l1 / 2;
        }
        const P2 m = mkp((q2.x + q1.x) / 2, (q2.y + q1.y) / 2);
        const double a = dist(m, p2), b = dist(m, p1);
        return (b < a ? b : a) <= l2 / 2;
    }
    const double v1x = q1.x - p1.x, v1y = q1.y - p1.y;
    const double v2x = q2.x - p1.x, v2y = q2.y - p1.y;
    const double vmx = p2.x - p1.x, vmy = p2.y - p1.y;
    return crossp(v1x, v1y, vmx, vmy) * crossp(v2x, v2y, vmx, vmy) <= 0.0;
}

// Segment.is_cross, C:518-533
__device__ __forceinline__ bool is_cross(P2 a1, P2 a2, P2 b1, P2 b2)
{
    return straddle(a1, a2, b1, b2) && straddle(b1, b2, a1, a2);
}

// Segment.distance(Vertex), C:670-684; segment = (p1, p2)
__device__ __forceinline__ double seg_point_distance(P2 p1, P2 p2, P2 v)
{
    const double a = p1.x, b = p1.y;
    const double A = p2.x - p1.x, B = p2.y - p1.y;
    const double s = (A * v.x + B * v.y - B * b - A * a) / (A * A + B * B);
    if (0.0 <= s && s <= 1.0) return dist(v, mkp(a + s * A, b + s * B));
    if (s < 0.0) return dist(v, p1);
    return dist(v, p2);
}

// ---------------------------------------------------------------------------------- wave helpers (64 lanes)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bcast_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

// wave-wide argmin of (v, ord): smallest v, ties -> smallest ord.  Lanes without a candidate pass
// v = +inf (and any ord).  Every lane receives the result.
__device__ __forceinline__ void wave_argmin_f64(double &v, int &ord)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = shfl_xor_f64(v, m);
        const int oo = __shfl_xor(ord, m, 64);
        if (ov < v || (ov == v && oo < ord)) {
            v = ov;
            ord = oo;
        }
    }
}

__device__ __forceinline__ void wave_argmin_f32(float &v, int &ord)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(v, m, 64);
        const int oo = __shfl_xor(ord, m, 64);
        if (ov < v || (ov == v && oo < ord)) {
            v = ov;
            ord = oo;
        }
    }
}

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = shfl_xor_f64(v, m);
        v = ov < v ? ov : v;
    }
    return v;
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int ov = __shfl_xor(v, m, 64);
        v = ov < v ? ov : v;
    }
    return v;
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

}  // namespace meshenv
