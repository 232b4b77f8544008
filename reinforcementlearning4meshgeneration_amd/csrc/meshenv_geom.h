// meshenv_geom.h -- fp64 geometry primitives of the element-extraction hot path, device side (gfx950).
//
// Every function cites the reference lines it follows (general/components.py = "C:", general/mesh.py =
// "M:", rl/boundary_env.py = "B:", general/data.py = "D:").  Build with -ffp-contract=off: the reference
// is CPython float arithmetic, which never fuses a*b+c; the only fused operation is the explicit fma()
// inside round4_py.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "meshenv_libm.h"

#ifndef MESHENV_NOINLINE
#define MESHENV_NOINLINE __noinline__
#endif

namespace meshenv {

constexpr double kPi = 3.141592653589793;  // math.pi
constexpr double kInfGeom = __builtin_huge_val();

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
// r / 1e4 for a non-negative integer-valued r.  For r < 2^33 the three-operation FMA sequence below returns the
// correctly rounded quotient -- checked exhaustively against IEEE division for every integer in [0, 2^33)
// (tools/verify_div1e4.c) -- at a third of the latency of the fp64 divide; larger r take the divide.
__device__ MESHENV_NOINLINE double div1e4_slow(double r) { return r / 1e4; }
__device__ __forceinline__ double div1e4(double r)
{
    const double q0 = r * 1e-4;
    const double rem = fma(-q0, 1e4, r);
    double q = fma(rem, 1e-4, q0);
    if (__builtin_expect(!(r < 8589934592.0), 0)) q = div1e4_slow(r);  // never taken for coordinates / angles
    return q;
}

// Python's round(float, 4): correctly rounded decimal rounding of the exact binary value, ties to even.
// y = |x|*1e4 (rounded) and e = fma(|x|,1e4,-y) (exact residual) give the exact product as y + e, so the
// comparison of its fractional part against 0.5 is exact; r/1e4 with IEEE division is the double nearest
// to the decimal r*1e-4, which is what CPython's dtoa/strtod round trip returns.
__device__ __forceinline__ double round4_py(double x)
{
    // written with selects only: the lanes of a wave take different rounding directions
    const double ax = fabs(x);
    const double y = ax * 1e4;
    const double e = fma(ax, 1e4, -y);
    const double f = floor(y);
    const double t = (y - f) - 0.5;
    const double s = t + e;
    const bool up = (s > 0.0) || (s == 0.0 && ((((long long)f) & 1LL) != 0));
    const double r = up ? f + 1.0 : f;
    const double q = copysign(div1e4(r), x);
    return (y < 4503599627370496.0) ? q : x;  // huge / inf / NaN pass through
}

// Python's round(float, 6) (the move() API rounds its local coordinates to 6 places, B:269): the same construction with
// scale 1e6 and the IEEE divide (not a hot path).
__device__ __forceinline__ double round6_py(double x)
{
    const double ax = fabs(x);
    const double y = ax * 1e6;
    const double e = fma(ax, 1e6, -y);
    const double f = floor(y);
    const double t = (y - f) - 0.5;
    const double s = t + e;
    const bool up = (s > 0.0) || (s == 0.0 && ((((long long)f) & 1LL) != 0));
    const double r = up ? f + 1.0 : f;
    const double q = copysign(r / 1e6, x);
    return (y < 4503599627370496.0) ? q : x;
}

// round(np.float64, 4): numpy's multiply / rint / divide
__device__ __forceinline__ double round4_np(double x)
{
    const double r = rint(x * 1e4);
    return copysign(div1e4(fabs(r)), r);
}

// round(np.float32, 4): the same in float32
__device__ __forceinline__ float round4_npf(float x) { return rintf(x * 1e4f) / 1e4f; }

// ---------------------------------------------------------------------------------- transcendentals
// One out-of-line copy of each libm-style routine: the kernels evaluate them for a handful of lanes per
// call site ("job lanes"), so sharing the body keeps the kernel inside the instruction cache.
// atan2, two out-of-line bodies.
//  * atan2_nc: ocml's, a leaf.  Within an ulp of the libm the reference calls, so the 1e-4 quantum of a clockwise angle
//    (every use but two feeds cw_finish) is the reference's unless the angle lies within ~1e-15 of a rounding boundary
//    (k + 0.5) e-4.  The step kernels use this one: their coordinates are all multiples of 1e-4, nothing puts an angle on
//    a boundary, and a chance hit is a 1e-11-per-step event (none in 4.5e9 steps of soak) -- while a test on every angle
//    cost 2-3 % of the throughput at 65 536 envs in each of the three forms tried (tools/ab_all.sh).
//  * atan2_x_nc: the same, then angles inside a band of 1e-11 rad around a boundary are re-evaluated the way glibc does,
//    bit for bit (csrc/meshenv_libm.h, atan2_tie).  Everything on the move() / smoothing path uses this one (Ctx::tie,
//    cw(), cw_exact): the front smoother CONSTRUCTS half-quantum angles -- it places vertices at tan(q e-4 / 2) of a
//    quantised angle and the next observation measures that angle back.
__device__ MESHENV_NOINLINE double atan2_nc(double y, double x) { return atan2(y, x); }
__device__ MESHENV_NOINLINE double atan2_tie_nc(double y, double x, double t) { return atan2_tie(y, x, t); }
__device__ MESHENV_NOINLINE double atan2_x_nc(double y, double x)
{
    double t = atan2(y, x);
#ifndef MESHENV_NO_ATAN_TIE   // (A/B builds only)
    const double theta = -t;
    const double ang = signbit(theta) ? 2 * 3.141592653589793 + theta : theta;
    const double yq = ang * 1e4;
    const double fr = (yq - floor(yq)) - 0.5;
    if (__builtin_expect(fabs(fr) < 1e-7, 0)) t = atan2_tie_nc(y, x, t);
#endif
    return t;
}
__device__ __forceinline__ double atan2_sel(const bool tie, double y, double x) { return tie ? atan2_x_nc(y, x) : atan2_nc(y, x); }
struct SinCos {
    double s, c;
};
__device__ MESHENV_NOINLINE SinCos sincos_nc(double a)
{
    SinCos r;
    r.s = sin(a);
    r.c = cos(a);
    return r;
}
__device__ MESHENV_NOINLINE double sin_nc(double a) { return sin(a); }
// sin / cos of the angles the step kernels form themselves (all in [0, 3 pi]): the small-range form of csrc/meshenv_libm.h,
// a sixth of ocml's instructions
__device__ MESHENV_NOINLINE SinCos sincos_small_nc(double a)
{
#ifdef MESHENV_OCML_SINCOS   // (A/B builds only)
    return sincos_nc(a);
#else
    const SinCosD r = sincos_small(a);
    SinCos o;
    o.s = r.s;
    o.c = r.c;
    return o;
#endif
}

// ---------------------------------------------------------------------------------- primitives
// Point2D.distance_to, C:17-18.  (The reference's `** 2` is libm pow(x, 2.0), which differs from x*x by
// IEEE-correct square root for arguments that are zero or lie in [2^-767, 2^1023): the compiler's own expansion of
// sqrt(double) (v_rsq_f64 seed, two coupled Newton steps, two residual corrections) without its range scaling -- the
// ldexp by 2^+-256 / 2^-+128 and the compare that decide it are 6 of its ~19 instructions, and sums of squared
// coordinate differences are never subnormal-small.  Bit-identical to sqrt() on that range (tests/test_gpu_primitives.py).
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y, h0 = 0.5 * y;
    const double r0 = fma(-h0, s0, 0.5);
    const double h1 = fma(h0, r0, h0), s1 = fma(s0, r0, s0);
    const double d0 = fma(-s1, s1, x);
    const double s2 = fma(d0, h1, s1);
    const double d1 = fma(-s2, s2, x);
    const double r = fma(d1, h1, s2);
    return (x == 0.0 || x == __builtin_huge_val()) ? x : r;  // rsq(0) = inf, rsq(inf) = 0: both would give NaN
}

// one ulp for 0.085 % of doubles; the device uses the exactly rounded product -- see DESIGN.md.)
__device__ __forceinline__ double dist(P2 a, P2 b)
{
    const double dx = a.x - b.x, dy = a.y - b.y;
    return sqrt_pos(dx * dx + dy * dy);
}

// dist(a, b) < t without the square root.  The ring-wide filters (find_same_point, the distance filter of
// check_intersection_with_boundary, the near-vertex scan of compute_boundary_quality) only compare a distance with a
// threshold; sqrt is monotone and correctly rounded, so s = dx^2 + dy^2 < t^2 (1 - 1e-15) implies sqrt(s) < t and
// s > t^2 (1 + 1e-15) implies sqrt(s) >= t (the band covers the rounding of t * t, of the factor and of the root with a
// margin of 3); a sum inside the band -- 2e-15 of its range -- and every non-positive or non-finite threshold takes the
// root.  The same truth value as dist(a, b) < t in every case, 19 instructions shorter per lane.
struct DistThr {
    double t, lo, hi;
};
__device__ __forceinline__ DistThr dist_thr(double t)
{
    DistThr q;
    const double t2 = t * t;
    const bool usable = t > 1e-140 && t < 1e140;
    q.t = t;
    q.lo = usable ? t2 * (1.0 - 1e-15) : -1.0;
    q.hi = usable ? t2 * (1.0 + 1e-15) : kInfGeom;
    return q;
}
__device__ __forceinline__ bool dist_lt(P2 a, P2 b, const DistThr &q)
{
#ifdef MESHENV_NO_FILTERS
    return dist(a, b) < q.t;
#else
    const double dx = a.x - b.x, dy = a.y - b.y;
    const double s = dx * dx + dy * dy;
    bool r = s < q.lo;
    if (__builtin_expect(!r && !(s > q.hi), 0)) r = sqrt_pos(s) < q.t;
    return r;
#endif
}

// numerator / denominator of the clockwise angle: cross(v1,v2), dot(v1,v2) with v1 = p1-s, v2 = p2-s
__device__ __forceinline__ void cw_terms(P2 s, P2 p1, P2 p2, double &c, double &d)
{
    const double v1x = p1.x - s.x, v1y = p1.y - s.y;
    const double v2x = p2.x - s.x, v2y = p2.y - s.y;
    c = v1x * v2y - v1y * v2x;
    d = v1x * v2x + v1y * v2y;
}

// tail of Vertex.to_find_clockwise_angle given t = atan2(c, d): theta = -t, quantised to 1e-4 rad in [0, 2pi]
__device__ __forceinline__ double cw_finish(double t)
{
    const double theta = -t;
    return round4_py(signbit(theta) ? 2 * kPi + theta : theta);
}

// Vertex.to_find_clockwise_angle from its atan2 terms: the reference's quantum in every case / per the kernel's choice
__device__ __forceinline__ double cw_exact(double c, double d) { return cw_finish(atan2_x_nc(c, d)); }
__device__ __forceinline__ double cw_sel(const bool tie, double c, double d) { return cw_finish(atan2_sel(tie, c, d)); }

// cw_finish(atan2(c, d)) without the libm-grade atan2 (309 instructions): the result is quantised to 1e-4 rad, so an
// angle known to 1e-12 decides the quantum everywhere except within a guard band of the rounding boundaries (k + 0.5) e-4,
// where -- and for the degenerate inputs -- need_exact is raised and the caller evaluates the exact form.  ~50
// instructions: |c|, |d| ordered, quotient by reciprocal + two Newton steps, reduction to |t| <= tan(pi/8), degree-7
// polynomial in t^2 (max error 2.9e-13 on that interval, coefficients from a Chebyshev fit), quadrant and sign restored.
// Identical to the exact form whenever need_exact is false (tests/test_gpu_primitives.py: 2e7 random + boundary cases).
__device__ __forceinline__ double cw_fast(double c, double d, bool &need_exact)
{
    const double ax = fabs(d), ay = fabs(c);
    const bool steep = ay > ax;
    const double mx = steep ? ay : ax, mn = steep ? ax : ay;
    double rc = __builtin_amdgcn_rcp(mx);
    rc = fma(fma(-mx, rc, 1.0), rc, rc);
    rc = fma(fma(-mx, rc, 1.0), rc, rc);
    const double r = mn * rc;                       // in [0, 1]
    const bool big = r > 0.41421356237309503;        // tan(pi / 8)
    const double den = r + 1.0;
    double rd = __builtin_amdgcn_rcp(den);
    rd = fma(fma(-den, rd, 1.0), rd, rd);
    rd = fma(fma(-den, rd, 1.0), rd, rd);
    const double t = big ? (r - 1.0) * rd : r;       // atan(r) = pi / 4 + atan((r - 1) / (r + 1))
    const double u = t * t;
    double p = -0.03770173601926599;
    p = fma(p, u, 0.069770051145731);
    p = fma(p, u, -0.08993217256091926);
    p = fma(p, u, 0.11103534914795694);
    p = fma(p, u, -0.14285391259506042);
    p = fma(p, u, 0.1999999318889726);
    p = fma(p, u, -0.33333333278384714);
    p = fma(p, u, 0.9999999999992708);
    double a = t * p + (big ? 0.7853981633974483 : 0.0);
    a = steep ? 1.5707963267948966 - a : a;
    a = d < 0.0 ? kPi - a : a;
    const double at = copysign(a, c);                // ~ atan2(c, d)
    const double theta = -at;
    const double ang = signbit(theta) ? 2 * kPi + theta : theta;
    const double y = ang * 1e4;
    const double f = floor(y);
    const double fr = y - f;
    // guard band 1e-5 quanta = 1e-9 rad around the rounding boundary (the fast angle is good to ~1e-12); zero, NaN and
    // magnitudes outside [1e-290, 1e290] (reciprocal range) take the exact form
    need_exact = !(mx > 1e-290) || !(mx < 1e290) || !(mn >= 0.0) || fabs(fr - 0.5) < 1e-5;
    return div1e4(fr > 0.5 ? f + 1.0 : f);
}

// Vertex.to_find_clockwise_angle, C:91-100
__device__ __forceinline__ double cw(P2 s, P2 p1, P2 p2)
{
    double c, d;
    cw_terms(s, p1, p2, c, d);
    return cw_exact(c, d);
}

// sin_rounds_to_zero(c, d)  ==  (round(sin(cw_angle), 4) == 0)  with cw_angle = the 1e-4-quantised clockwise angle
// whose atan2 arguments are (c, d) -- the collinearity test of Segment.straddle (C:498-500).
//
// The rounded angle a is a multiple of 1e-4 in [0, 2pi]; |sin(a)| < 5e-5 only for a in {0.0, 3.1416, 6.2832}.
//  * |c| > 1e-3 |d|: the true angle is > 9.99e-4 rad away from every multiple of pi, the rounded one > 9.4e-4:
//    never zero.  (Every non-degenerate configuration ends here: two cross products, no transcendental.)
//  * otherwise r = c/d, |r| <= 1e-3 and atan(r) = r to within 3.4e-10, so the angle is known to 3.4e-10 and each of
//    the four sign cases below reduces "a in {0, 3.1416, 6.2832}" to one comparison of |r| against the matching
//    rounding boundary -- done as |c| against boundary * |d|, no division.  Within 1e-8 of a boundary (and for
//    d == 0, which implies c == 0) the reference evaluation (atan2, round, sin, round) is used, so the result is the
//    reference's in every case.
// (out of line: the rare guard-band / degenerate path; keeping it out of straddle() leaves straddle a leaf that
// inlines into the kernels -- as a non-leaf function it saved its return address through a scratch-memory spill,
// a full memory round trip per call)
__device__ MESHENV_NOINLINE bool sin_rounds_to_zero_exact(double c, double d)
{
    return round4_py(sin(cw_exact(c, d))) == 0.0;
}

__device__ __forceinline__ bool sin_rounds_to_zero(double c, double d)
{
#ifdef MESHENV_NO_FILTERS
    return sin_rounds_to_zero_exact(c, d);
#else
    const double ac = fabs(c), ad = fabs(d);
    if (ac > 1e-3 * ad) return false;
    if (d != 0.0) {
        // In all four sign cases the quantity compared with its boundary is |c / d|:
        //   d > 0, c < 0 (or -0): theta = -t >= +0, a = round(theta) == 0.0            <=>  |r| < 5e-5
        //   d > 0, c >= +0:       theta < 0, a = round(2pi + theta) == 6.2832          <=>  |r| < 2pi - 6.28315
        //   d < 0, c >= +0:       t = pi - |r|, a = round(pi + |r|) == 3.1416          <=>  |r| < 3.14165 - pi
        //   d < 0, c < 0:         t = -pi + |r|, a = round(pi - |r|) == 3.1416         <=>  |r| < pi - 3.14155
        // compared without the division: |c| against thr * |d|, guard band 1e-8 * |d| (the products round to ~1e-16
        // relative, far inside the band)
        const bool cneg = signbit(c), dpos = d > 0.0;
        const double thr = dpos ? (cneg ? 5e-5 : 2 * kPi - 6.28315) : (cneg ? kPi - 3.14155 : 3.14165 - kPi);
        const double rhs = thr * ad;
        if (fabs(ac - rhs) > 1e-8 * ad) return ac < rhs;
    }
    return sin_rounds_to_zero_exact(c, d);
#endif
}

// cross_product, C:482-483
__device__ __forceinline__ double crossp(double ax, double ay, double bx, double by) { return ax * by - bx * ay; }

// Segment.straddle, C:491-516; self = (p1,p2), another = (q1,q2)
__device__ __forceinline__ bool straddle(P2 p1, P2 p2, P2 q1, P2 q2)
{
    double c1, d1, c2, d2;
    cw_terms(p1, q1, p2, c1, d1);
    cw_terms(p1, q2, p2, c2, d2);
    // s1 == s2 and s2 == 0 (C:498-500); both roundings are (+-)0 then, and -0.0 == 0.0
    const bool collinear = sin_rounds_to_zero(c1, d1) && sin_rounds_to_zero(c2, d2);
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

// value of lane `l` (wave-uniform l) as a scalar; v_readlane instead of an LDS-crossbar shuffle
__device__ __forceinline__ int lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform_f64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// DPP building block: lanes whose source is outside the row / masked row keep `identity`
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_i32(int identity, int v)
{
    return __builtin_amdgcn_update_dpp(identity, v, kCtrl, kRowMask, 0xf, false);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_f64(double identity, double v)
{
    const int lo = dpp_i32<kCtrl, kRowMask>(__double2loint(identity), __double2loint(v));
    const int hi = dpp_i32<kCtrl, kRowMask>(__double2hiint(identity), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143;

#define MESHENV_DPP_REDUCE(STEP)              \
    STEP(kDppShr1, 0xf)                       \
    STEP(kDppShr2, 0xf)                       \
    STEP(kDppShr4, 0xf)                       \
    STEP(kDppShr8, 0xf)                       \
    STEP(kDppBcast15, 0xa)                    \
    STEP(kDppBcast31, 0xc)

constexpr double kInf = __builtin_huge_val();

// wave-wide argmin of (v, ord): smallest v, ties -> smallest ord.  Lanes without a candidate pass
// v = +inf.  The result is wave-uniform (read from lane 63 after a DPP scan).
__device__ __forceinline__ void wave_argmin_f64(double &v, int &ord)
{
#define STEP(CTRL, MASK)                                                        \
    {                                                                           \
        const double ov = dpp_f64<CTRL, MASK>(kInf, v);                         \
        const int oo = dpp_i32<CTRL, MASK>(0x7fffffff, ord);                    \
        if (ov < v || (ov == v && oo < ord)) { v = ov; ord = oo; }              \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    v = lane_f64(v, 63);
    ord = lane_i32(ord, 63);
}

__device__ __forceinline__ void wave_argmin_f32(float &v, int &ord)
{
#define STEP(CTRL, MASK)                                                                                  \
    {                                                                                                     \
        const float ov = __int_as_float(dpp_i32<CTRL, MASK>(0x7f800000, __float_as_int(v)));              \
        const int oo = dpp_i32<CTRL, MASK>(0x7fffffff, ord);                                              \
        if (ov < v || (ov == v && oo < ord)) { v = ov; ord = oo; }                                        \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    v = lane_f32(v, 63);
    ord = lane_i32(ord, 63);
}

__device__ __forceinline__ double wave_min_f64(double v)
{
#define STEP(CTRL, MASK)                                        \
    {                                                           \
        const double ov = dpp_f64<CTRL, MASK>(kInf, v);         \
        v = ov < v ? ov : v;                                    \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    return lane_f64(v, 63);
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#define STEP(CTRL, MASK)                                                  \
    {                                                                     \
        const int ov = dpp_i32<CTRL, MASK>((int)0x80000000, v);           \
        v = ov > v ? ov : v;                                              \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    return lane_i32(v, 63);
}

typedef unsigned long long u64;

template <int kCtrl, int kRowMask>
__device__ __forceinline__ u64 dpp_u64(u64 identity, u64 v)
{
    const unsigned lo = (unsigned)dpp_i32<kCtrl, kRowMask>((int)(unsigned)identity, (int)(unsigned)v);
    const unsigned hi = (unsigned)dpp_i32<kCtrl, kRowMask>((int)(unsigned)(identity >> 32), (int)(unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 lane_u64(u64 v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// Five independent unsigned 64-bit min-reductions in one DPP scan (the chains interleave, hiding the DPP
// and compare latencies).  Non-negative floats/doubles order like their bit patterns, so (value, order)
// pairs packed as (bits << 32 | order) reduce to the first-in-order minimum in one pass.
__device__ __forceinline__ void wave_min5_u64(u64 &a, u64 &b, u64 &c, u64 &d, u64 &e)
{
#define STEP(CTRL, MASK)                                        \
    {                                                           \
        const u64 oa = dpp_u64<CTRL, MASK>(~0ULL, a);           \
        const u64 ob = dpp_u64<CTRL, MASK>(~0ULL, b);           \
        const u64 oc = dpp_u64<CTRL, MASK>(~0ULL, c);           \
        const u64 od = dpp_u64<CTRL, MASK>(~0ULL, d);           \
        const u64 oe = dpp_u64<CTRL, MASK>(~0ULL, e);           \
        a = oa < a ? oa : a;                                    \
        b = ob < b ? ob : b;                                    \
        c = oc < c ? oc : c;                                    \
        d = od < d ? od : d;                                    \
        e = oe < e ? oe : e;                                    \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    a = lane_u64(a, 63);
    b = lane_u64(b, 63);
    c = lane_u64(c, 63);
    d = lane_u64(d, 63);
    e = lane_u64(e, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#define STEP(CTRL, MASK)                                                        \
    {                                                                           \
        const unsigned ov = (unsigned)dpp_i32<CTRL, MASK>(-1, (int)v);          \
        v = ov < v ? ov : v;                                                    \
    }
    MESHENV_DPP_REDUCE(STEP)
#undef STEP
    return (unsigned)lane_i32((int)v, 63);
}

// sign/zero of round(x, 4) without the final division: returns 1e4 * round(x, 4) as an integer-valued
// double carrying x's sign (tests `== 0`, `< 0` and sign-of-product are unchanged by the exact scaling)
__device__ __forceinline__ double round4_py_scaled(double x)
{
    const double ax = fabs(x);
    const double y = ax * 1e4;
    const double e = fma(ax, 1e4, -y);
    const double f = floor(y);
    const double t = (y - f) - 0.5;
    const double s = t + e;
    const bool up = (s > 0.0) || (s == 0.0 && ((((long long)f) & 1LL) != 0));
    const double r = copysign(up ? f + 1.0 : f, x);
    return (y < 4503599627370496.0) ? r : x;
}
__device__ __forceinline__ double round4_np_scaled(double x) { return rint(x * 1e4); }

}  // namespace meshenv
