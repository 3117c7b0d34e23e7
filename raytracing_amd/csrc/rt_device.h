// rt_device.h -- device-side numerics of the ray propagation path (gfx950).
// One ray per lane; everything here is per-lane scalar code on T = double | float.
// Reference lines are RT_bench.py file:line of neyuru/RayTracing.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_polytab.h"

#ifndef RTMI_CHORD_SERIES
// the step's arclength from the advancement's closed form (chord_length) instead of the positions: 1 for the first-order
// advancement (op1/op2: the chord IS DELTA_S; op1 12.5 -> 11.8 ms), 2 for the second-order one as well (measured slower:
// the vote and eps carried across the lookup cost more than the square root -- headline 14.9 -> 15.5 ms, cfg2 2.20 -> 2.48)
#define RTMI_CHORD_SERIES 1
#endif
#ifndef RTMI_POLY_BATCH
#define RTMI_POLY_BATCH 1     // scalar loads of a lookup: 0 row by row, 1 one spline + n then the other (measured best), 2 all at once (spills SGPRs)
#endif
#ifndef RTMI_FLAT_MAP
#define RTMI_FLAT_MAP 1     // 0 compiles the flat-cell map's tests out of the lookups (A/B of what they cost a field without flat cells)
#endif
#ifndef RTMI_LAT_RELOAD
#define RTMI_LAT_RELOAD 0   // k_advance_lat's kept cell: 0 reloaded with vector loads (vmcnt(0): also waits for the row stores before), 1 through the scalar cache + copies
#endif
#ifndef RTMI_POLY
#define RTMI_POLY 1     // 1: the fast-form step methods look the field up as one polynomial per cell (PolyGather); 0: B-spline sums
#endif

// A wave vote straight from the predicate.  HIP's __ballot(int) converts the bool to an int and compares it with zero again
// (v_cndmask + v_cmp per vote, 12 vector instructions per step of the bench kernel); the builtin takes the i1.
__device__ __forceinline__ unsigned long long rt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

namespace rt {

// The library is compiled with -ffp-contract=off: every fusion below is an explicit fma(), so a value is
// computed with the same roundings wherever the function is inlined (results do not depend on how a run is
// cut into launches, on the ray->lane assignment, or on which kernel variant executes it).  For the same
// reason every data-dependent choice between two formulas below is made PER LANE from that ray's own
// values, never from a wave vote.
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// a*b + c where c is a loop-invariant constant held in a VGPR pair: written as the three-address v_fma_f64.  Left to the
// compiler this becomes v_mov_b64 (copy the constant) + v_fmac_f64 (accumulate into the copy), two instructions for one.
#ifndef RTMI_NO_FMA3
__device__ __forceinline__ double fma_const(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#else
__device__ __forceinline__ double fma_const(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
__device__ __forceinline__ float fma_const(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------- sin/cos
// fp64 sincos: 3-constant Cody-Waite reduction by pi/2 (the first step is exact under fma, so the reduction
// holds ~1e-16 relative accuracy in r for every |x| < 2^30) and the fdlibm minimax kernels on [-pi/4, pi/4];
// <= 1 ulp.  No data-dependent branch; |x| >= 2^30 or non-finite x gives NaN.
__device__ __forceinline__ void sincos_k(double x, double* sp, double* cp) {
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = fma_(-n, 1.57079632679489655800e+00, x);
    r = fma_(-n, 6.12323399573676603587e-17, r);
    r = fma_(-n, -1.49738490485916983294e-33, r);
    r = __builtin_fabs(x) < 1073741824.0 ? r : __builtin_nan("");
    const double z = r * r;
    double ps = fma_(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma_(z, ps, 2.75573137070700676789e-06);
    ps = fma_(z, ps, -1.98412698298579493134e-04);
    ps = fma_(z, ps, 8.33333333332248946124e-03);
    ps = fma_(z, ps, -1.66666666666666324348e-01);
    const double s = fma_(z * r, ps, r);
    double pc = fma_(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma_(z, pc, -2.75573143513906633035e-07);
    pc = fma_(z, pc, 2.48015872894767294178e-05);
    pc = fma_(z, pc, -1.38888888888741095749e-03);
    pc = fma_(z, pc, 4.16666666666666019037e-02);
    const double c = fma_(z * z, pc, fma_(-0.5, z, 1.0));
    const int q = (int)n;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sp = (q & 2) ? -ss : ss;
    *cp = ((q + 1) & 2) ? -cc : cc;
}

// fp32 sincos of an fp64 angle (the angle is one of the fp64 accumulators of a ray): the reduction by pi/2 is done in
// fp64 (one fma, exact enough for any |x| < 2^30) so the angle's low bits are not lost before the trig, the minimax
// kernels on [-pi/4, pi/4] in fp32; ~1 ulp.  The reference has no fp32 path, so this defines it.
__device__ __forceinline__ void sincos_kf(double x, float* sp, float* cp) {
    const float xf = (float)x;
    const float n = __builtin_rintf(xf * 6.3661977237e-01f);
    float r = (float)fma_(-(double)n, 1.57079632679489655800e+00, x);
    r = __builtin_fabsf(xf) < 1073741824.0f ? r : __builtin_nanf("");
    const float z = r * r;
    float ps = fma_(z, 2.7557314297e-06f, -1.9841270114e-04f);
    ps = fma_(z, ps, 8.3333337680e-03f);
    ps = fma_(z, ps, -1.6666667163e-01f);
    const float s = fma_(z * r, ps, r);
    float pc = fma_(z, -2.7557314297e-07f, 2.4801587642e-05f);
    pc = fma_(z, pc, -1.3888889225e-03f);
    pc = fma_(z, pc, 4.1666667908e-02f);
    const float c = fma_(z * z, pc, fma_(-0.5f, z, 1.0f));
    const int q = (int)n;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sp = (q & 2) ? -ss : ss;
    *cp = ((q + 1) & 2) ? -cc : cc;
}

// ---------------------------------------------------------------- math traits
template <typename T> struct M;
template <> struct M<double> {
    // sqrt for arguments in the normal range (squared step lengths ~1e-5, s^2+c^2 ~ 1): the hardware rsq estimate
    // with one coupled Newton step and two residual corrections -- the compiler's own correctly-rounded
    // sequence without its subnormal rescaling and class checks (6 instructions fewer per call).
    static __device__ __forceinline__ double sqrt_(double x) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = 0.5 * y;
        const double r = fma_(-h, g, 0.5);
        g = fma_(g, r, g);
        h = fma_(h, r, h);
        return fma_(fma_(-g, g, x), h, g);      // one residual correction: <= 1 ulp (a second one would round correctly)
    }
    static __device__ __forceinline__ double sqrt_full(double x) { return __dsqrt_rn(x); }
    static __device__ __forceinline__ void sincos_(double x, double* s, double* c) { sincos_k(x, s, c); }
    static __device__ __forceinline__ double atan2_(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double abs_(double x) { return __builtin_fabs(x); }
    static __device__ __forceinline__ double max_(double x, double y) { return __builtin_fmax(x, y); }
    static __device__ __forceinline__ double min_(double x, double y) { return __builtin_fmin(x, y); }
    // x[i] of numpy.linspace: i*step + a with separate roundings (contraction is off), so knots equal the host's bit for bit
    static __device__ __forceinline__ double lin(int i, double h, double a) { return (double)i * h + a; }
    static constexpr double gold_tol = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON), RT_bench.py:66
    static constexpr double small_angle = 0.03125;              // 2^-5: sincos_add's series bound
};
template <> struct M<float> {
    // fp32 is the reduced-precision path (no reference to match bit for bit): hardware sqrt/rcp (1 ulp)
    static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float sqrt_full(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ void sincos_(double x, float* s, float* c) { sincos_kf(x, s, c); }
    static __device__ __forceinline__ float atan2_(float y, float x) { return ::atan2f(y, x); }
    static __device__ __forceinline__ float abs_(float x) { return __builtin_fabsf(x); }
    static __device__ __forceinline__ float max_(float x, float y) { return __builtin_fmaxf(x, y); }
    static __device__ __forceinline__ float min_(float x, float y) { return __builtin_fminf(x, y); }
    static __device__ __forceinline__ float lin(int i, float h, float a) { return (float)i * h + a; }
    static constexpr float gold_tol = 3.4526698300124393e-04f;  // sqrt(FLT_EPSILON): the fp32 analogue
    static constexpr float small_angle = 0.03125f;
};

// sin/cos of (theta + k) given s = sin(theta), c = cos(theta): for |k| < 2^-5 the angle-addition formulas
// with 4-term series of sin k and 1 - cos k (truncation < 3e-18 relative), else a full evaluation.  (2^-5 so that the
// fisheye's RK2 half-step, k1 ~ 0.02 at the calibrated DELTA_S, stays on the series; a fifth term for 2^-4 cost the
// global-gather kernel two more spilled registers and 11 % of its speed.)
template <typename T> __device__ __forceinline__ void sincos_add_small(T s, T c, T k, T* so, T* co) {
    const T z = k * k;
    T ps = fma_const(z, T(-1.0 / 5040.0), T(1.0 / 120.0));
    ps = fma_const(z, ps, T(-1.0 / 6.0));
    const T sk = fma_(z * k, ps, k);                 // sin k
    T pc = fma_const(z, T(-1.0 / 40320.0), T(1.0 / 720.0));
    pc = fma_const(z, pc, T(-1.0 / 24.0));
    pc = fma_(z, pc, T(0.5));
    const T ck1 = z * pc;                            // 1 - cos k
    *so = fma_(c, sk, fma_(-s, ck1, s));
    *co = fma_(-s, sk, fma_(-c, ck1, c));
}
// The same for |k| < 2^-9 (a turn of 0.1 degree per step: every step of the vert_heterogeneous and interface fans, 4e-4 and
// less): one series term fewer each -- the dropped ones are k^6/5040 < 2^-66 of sin k and k^6/720 < 2^-63 of 1: nothing
// in fp64 -- 10 instructions for 13.  Measured (A/B, one session, profiles/r03_ab_small_savings.txt): vert_heterogeneous without
// recording 8.78 -> 8.52 ms, the recording headline unchanged, but fisheye -- whose turns are not tiny, and which now pays
// a second vote -- 7.9 -> 8.7 ms and the few-waves build 2.20 -> 2.24: off.
#ifndef RTMI_TINY_ROT
#define RTMI_TINY_ROT 0
#endif
template <typename T> __device__ __forceinline__ void sincos_add_tiny(T s, T c, T k, T* so, T* co) {
    const T z = k * k;
    const T ps = fma_const(z, T(1.0 / 120.0), T(-1.0 / 6.0));
    const T sk = fma_(z * k, ps, k);                 // sin k
    const T pc = fma_const(z, T(-1.0 / 24.0), T(0.5));
    const T ck1 = z * pc;                            // 1 - cos k
    *so = fma_(c, sk, fma_(-s, ck1, s));
    *co = fma_(-s, sk, fma_(-c, ck1, c));
}
// Which formula a lane uses depends on its own k only (tiny / small / a full evaluation at `target`); the votes select a
// layout without exec-mask bookkeeping for the two common cases that every lane of the wave is tiny, or every lane small.
// (the full evaluation is at base + (add_k ? k : 0), formed only where it is needed)
template <typename T, bool ADD_K> __device__ __forceinline__ void sincos_add_select(double base, T s, T c, T k, T* so, T* co, bool refresh) {
    const T ak = M<T>::abs_(k);
    const bool small = ak < M<T>::small_angle && !refresh;
#if RTMI_TINY_ROT
    const bool tiny = ak < T(0.001953125) && !refresh;
    if (rt_ballot(!tiny) == 0ull) {
        sincos_add_tiny(s, c, k, so, co);
    } else if (rt_ballot(!small || tiny) == 0ull) {
        sincos_add_small(s, c, k, so, co);
    } else if (tiny) {
        sincos_add_tiny(s, c, k, so, co);
    } else if (small) {
        sincos_add_small(s, c, k, so, co);
    } else {
        M<T>::sincos_(ADD_K ? base + (double)k : base, so, co);
    }
#else
    if (rt_ballot(!small) == 0ull) {
        sincos_add_small(s, c, k, so, co);
    } else if (small) {
        sincos_add_small(s, c, k, so, co);
    } else {
        M<T>::sincos_(ADD_K ? base + (double)k : base, so, co);
    }
#endif
}
template <typename T> __device__ __forceinline__ void sincos_add(double theta, T s, T c, T k, T* so, T* co) {
    sincos_add_select<T, true>(theta, s, c, k, so, co, false);
}
// sin/cos of `target` == (angle of (s, c)) + k: by rotation when k is small and no refresh is due, else from scratch
template <typename T> __device__ __forceinline__ void sincos_add(double target, T s, T c, T k, T* so, T* co, bool refresh) {
    sincos_add_select<T, false>(target, s, c, k, so, co, refresh);
}

constexpr double kGoldRatio = 0.6180339887498949;   // (sqrt(5)-1)/2, RT_bench.py:65
constexpr double kHalfPi = 1.5707963267948966;      // DELTA_G, RT_bench.py:64
constexpr int kGoldMaxIter = 96;                    // hard exit for every lane (37 iterations in fp64 from width pi)

// ---------------------------------------------------------------- field in HBM
// Axes are numpy.linspace(a, b, q): x[i] = i*h + a (i < q-1), x[q-1] = b  (genZ, :429).
// zn  : [qy][qx]    bilinear coefficients == n samples (:455)
// g   : [qy][qx][2] bicubic coefficients, interleaved (d/dx spline, d/dy spline) so that one 4x4 window
//       is four contiguous 64-byte (fp64) row segments serving both gradient components (:456-457).
template <typename T> struct FieldDev {
    const T* zn;
    const T* g;
    int qx, qy;
    T ax, hx, bx, inv_hx;
    T ay, hy, by, inv_hy;
    int exact;   // 1: FITPACK's arithmetic with true knots in every cell (see axis_eval)
    // window: 1 when rt::ex::n_gradient may serve a wave whose live lanes share one cell through the scalar cache (lookup_uniform).
    // Set per batch (batch_dev): it pays from two waves per SIMD on -- a lone wave waits longer for its scalar loads than for its
    // vector loads (4 096 rays: -11 %; 65 536: -1 .. -10 %; 131 072: +1 .. +4 %; 262 144: +12 %; 1 M: +5 .. +21 %,
    // profiles/r04_ab_uniform_window.txt)
    int window;
    // poly: [(qy-1)*(qx-1)][kPolyStride] the same three splines as one polynomial per grid cell (rt_polytab.h) -- what the
    // fast-form step methods evaluate (PolyGather below); zn / g serve the reference-order methods and rtmi_field_eval
    const T* poly;
    int ncx;     // cells per grid row, qx - 1
    // flat: 0, or the distance (in elements) from the FLAT-CELL MAP to poly: flatn[cell] = poly[cell - flat] holds the constant
    // index of a cell where the medium is constant (rt::poly_cell_flat), all-ones bits elsewhere.  Such a cell's lookup is
    // (flatn, 0, 0) and never touches its 36 coefficients; 8 (fp64) cells of a grid row share a cache line of the map, where
    // every cell of the table proper is five lines of its own.  0 when no cell of the grid is flat: one scalar test per lookup.
    int flat;
    // rdx / rdy: [q][24] fp64, per cell index of an axis the correctly rounded reciprocals of the seven knot differences
    // FITPACK's fpbspl divides by there, then the cell's knots and the differences themselves (rt_exact.h: axis_exact reads the
    // reciprocals per lane, AxisTab the whole entry for a wave in one cell); nullptr in fp32 fields (the reference-order path is fp64)
    const double *rdx, *rdy;
    // (128 bytes in fp64, and it stays that: the struct heads every step kernel's argument segment, and the register allocation of the
    // kernels at their register cap follows the offsets of what comes after it -- one more member here, 8 bytes, cost interface x op9 12 %:
    // profiles/r05_bisect_iface_op9.txt)
};

// Rare branches of the step loop (a lookup near the grid's rim, re-staging the LDS tile, a lane falling back to a global
// gather) need field members the common path never touches (h, b, the array pointers).  Kept in scalar registers for the
// whole loop they cost ~16 SGPRs, and the step loop is short of exactly those (it spills SGPRs to VGPR lanes and
// rematerialises constants every iteration).  Every kernel that looks the field up has its FieldDev at offset 0 of the
// kernel-argument segment (k_advance / k_trace_refill / k_init: BatchDev::F is the first member; k_field_eval: first
// argument), so a rare branch can re-read it from there; the empty asm keeps the loads inside the branch.
#ifndef RTMI_NO_KERNARG_FIELD
template <typename T> __device__ __forceinline__ FieldDev<T> rare_field(const FieldDev<T>&) {
    typedef const FieldDev<T> __attribute__((address_space(4))) * KP;
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    FieldDev<T> out;
    __builtin_memcpy(&out, p, sizeof(out));      // scalar loads from the constant address space
    return out;                                   // scalar loads, live only inside the branch
}
#else
template <typename T> __device__ __forceinline__ const FieldDev<T>& rare_field(const FieldDev<T>& F) { return F; }
#endif

template <typename T> __device__ __forceinline__ T axis_at(int i, int q, T a, T h, T b) {
    return i >= q - 1 ? b : M<T>::lin(i, h, a);
}

// fpbisp's argument clamp (quirk Q4) and interval search on a linspace axis: j with x[j] <= v < x[j+1],
// j in [0, q-2] (v == x[q-1] gives q-2); also returns t0 = x[j], t1 = x[j+1].  The guess from one multiply is
// off by at most one cell, and only when v sits within rounding of a grid line (rare, per-lane fix-up).
template <typename T> __device__ __forceinline__ int locate(T& v, int q, T a, T h, T b, T inv_h, T& t0, T& t1) {
    v = v < a ? a : v;
    v = v > b ? b : v;
    int j = (int)((v - a) * inv_h);
    j = j < 0 ? 0 : (j > q - 2 ? q - 2 : j);
    t0 = M<T>::lin(j, h, a);
    t1 = axis_at(j + 1, q, a, h, b);
    if (t0 > v) {
        if (j > 0) { j -= 1; t1 = t0; t0 = M<T>::lin(j, h, a); }
    } else if (t1 <= v && j < q - 2) {
        j += 1; t0 = t1; t1 = axis_at(j + 1, q, a, h, b);
    }
    return j;
}

// cubic interpolating knot vector of FITPACK regrid(s=0): t[l] = x[0] (l<=3), x[l-2] (4<=l<=q-1), x[q-1] (l>=q)
template <typename T> __device__ __forceinline__ T knot3(int l, int q, T a, T h, T b) {
    return l <= 3 ? a : (l >= q ? b : M<T>::lin(l - 2, h, a));
}

// 1/d for a knot difference d that equals r0^-1 up to grid rounding (d = k*h*(1 + O(1e-13))): one Newton
// step from r0 is accurate to < 1 ulp and costs two fma instead of an IEEE division.
template <typename T> __device__ __forceinline__ T rcp_near(T d, T r0) { return fma_(r0, fma_(-d, r0, T(1)), r0); }

// 1/d to < 1 ulp from the hardware estimate (v_rcp, ~23 bits) and two Newton steps; used where no good
// starting value is known (the not-a-knot end intervals).
__device__ __forceinline__ double rcp_full(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma_(r, fma_(-d, r, 1.0), r);
    return fma_(r, fma_(-d, r, 1.0), r);
}
__device__ __forceinline__ float rcp_full(float d) { return __builtin_amdgcn_rcpf(d); }

// General form of one axis of n_gradient's basis (any cell, FITPACK's arithmetic with true knots): cell j (from
// locate, with t0 = x[j], t1 = x[j+1]) -> the two linear weights (fpbspl k=1 on [t0, t1]) and the four cubic
// weights (fpbspl k=3 on the interval l = clamp(j+2, 3, q-1)); reciprocals of knot differences by rcp_full.
template <typename T>
__device__ __forceinline__ int axis_basis(T v, int j, T t0, T t1, int q, T a, T h, T b, T wl[2], T w[4]) {
    {   // linear: f = 1/(t1 - t0); h0 = f*(t1 - v), h1 = f*(v - t0)
        const T f = rcp_full(t1 - t0);
        wl[0] = f * (t1 - v);
        wl[1] = f * (v - t0);
    }
    int l = j + 2;
    l = l < 3 ? 3 : (l > q - 1 ? q - 1 : l);
    const T tm2 = knot3(l - 2, q, a, h, b), tm1 = knot3(l - 1, q, a, h, b), k0 = knot3(l, q, a, h, b);
    const T k1 = knot3(l + 1, q, a, h, b), t2 = knot3(l + 2, q, a, h, b), t3 = knot3(l + 3, q, a, h, b);
    const T r10 = rcp_full(k1 - k0);
    const T r1m1 = rcp_full(k1 - tm1), r20 = rcp_full(t2 - k0);
    const T r1m2 = rcp_full(k1 - tm2), r2m1 = rcp_full(t2 - tm1), r30 = rcp_full(t3 - k0);
    const T a1 = k1 - v, a2 = t2 - v, a3 = t3 - v;      // t[li] - x
    const T b0 = v - k0, bm1 = v - tm1, bm2 = v - tm2;  // x - t[lj]
    // fpbspl recurrence, j = 1, 2, 3, with f = hh[i] / (t[li] - t[lj]) taken as hh[i] * reciprocal
    const T h0 = r10 * a1, h1 = r10 * b0;
    T f = h0 * r1m1;
    const T g0 = f * a1;
    T g1 = f * bm1;
    f = h1 * r20;
    g1 = fma_(f, a2, g1);
    const T g2 = f * b0;
    f = g0 * r1m2;
    w[0] = f * a1;
    w[1] = f * bm2;
    f = g1 * r2m1;
    w[1] = fma_(f, a2, w[1]);
    w[2] = f * bm1;
    f = g2 * r30;
    w[2] = fma_(f, a3, w[2]);
    w[3] = f * b0;
    return l;
}

// One axis of n_gradient: clamp (Q4), locate the cell, evaluate both bases.  Returns j (cell) and l (cubic
// interval; the 4x4 window starts at column l-3).
//
// Fast path (per lane; unless F.exact): a cell at least five cells inside the grid has the six cubic knots
// x[j-2..j+3] = consecutive linspace points, i.e. equally spaced up to the rounding of i*h + a (measured:
// <= 3e-13 relative to h on the reference's grids).  There the position in the cell is u = (v - a)/h - j straight
// from the quotient that locates the cell (no knot is formed at all), the linear weights are (1 - u, u) and the cubic
// weights are the uniform B-spline polynomials of u -- 16 flops instead of 61.  u differs from FITPACK's
// (v - x[j])/(x[j+1] - x[j]) by the rounding of the quotient (<= 6e-14 of a cell for grids up to 1000 cells) plus the
// knots' own departure from equal spacing; a lookup that lands within that distance of a grid line may take the
// neighbouring cell with u = 1 - eps instead of eps, which the splines' continuity makes immaterial.  Weights differ
// from fpbspl on the true knots by <= 1.5e-13 absolute (tools/basis_error.py).  F.exact forces the general form
// everywhere; rtmi_field_eval (the n_gradient call surface) always uses it.
// (the unclamped quotient is used: a point that far inside the grid is not affected by the clamp of quirk Q4)
template <typename T>
__device__ __forceinline__ void axis_fast(T ur, T jf, int& j, int& l, T wl[2], T w[4]) {
    const T u = ur - jf, om = T(1) - u;
    j = (int)jf;
    l = j + 2;
    wl[0] = om; wl[1] = u;
    const T u2 = u * u, om2 = om * om;
    w[0] = om2 * om * T(1.0 / 6.0);
    w[1] = fma_(u2, fma_(u, T(0.5), T(-1)), T(2.0 / 3.0));
    w[2] = fma_(om2, fma_(om, T(0.5), T(-1)), T(2.0 / 3.0));
    w[3] = u2 * u * T(1.0 / 6.0);
}
__device__ __forceinline__ double floor_(double x) { return __builtin_floor(x); }
__device__ __forceinline__ float floor_(float x) { return __builtin_floorf(x); }

// general form of one axis (any cell): clamp, interval search on the true knots, fpbspl
template <typename T>
__device__ __forceinline__ void axis_general(T v, int q, T a, T h, T b, T ih, int& j, int& l, T wl[2], T w[4]) {
    T t0, t1;
    j = locate(v, q, a, h, b, ih, t0, t1);      // clamps v (quirk Q4)
    l = axis_basis(v, j, t0, t1, q, a, h, b, wl, w);
}

// ---------------------------------------------------------------- n_gradient = locate + gather + combine
// The cell of one lookup: indices and the six basis weights per axis.
template <typename T> struct Cell {
    int jx, jy, lx, ly;          // bilinear cell (jx, jy); the cubic 4x4 window starts at (lx-3, ly-3)
    T lwx[2], lwy[2], wx[4], wy[4];
};

// The rare layout of field_locate: some lane of the wave is near the grid's rim (or F.exact is set).  (Keeping it out
// of line was tried: the Cell then lives in scratch memory and the step loop runs at half speed.)
template <typename T>
__device__ __forceinline__ void field_locate_mixed(const FieldDev<T>& F, T x, T y, bool fx, bool fy, Cell<T>& c) {
    const T urx = (x - F.ax) * F.inv_hx, ury = (y - F.ay) * F.inv_hy;
    if (fx) axis_fast(urx, floor_(urx), c.jx, c.lx, c.lwx, c.wx);
    else axis_general(x, F.qx, F.ax, F.hx, F.bx, F.inv_hx, c.jx, c.lx, c.lwx, c.wx);
    if (fy) axis_fast(ury, floor_(ury), c.jy, c.ly, c.lwy, c.wy);
    else axis_general(y, F.qy, F.ay, F.hy, F.by, F.inv_hy, c.jy, c.ly, c.lwy, c.wy);
}

// Both axes.  A lane takes the fast form on an axis when the UNCLAMPED quotient (v - a)/h lies in [5, q - 7): the point
// is then at least five cells inside the grid (no clamp can apply) and cell j = floor(quotient) has equally spaced
// knots around it.  Which FORMULA a lane uses depends on its own values only; the wave vote below merely picks a code
// layout without exec-mask bookkeeping for the overwhelmingly common case that every lane is interior on both axes.
template <typename T> __device__ __forceinline__ void field_locate(const FieldDev<T>& F, T x, T y, Cell<T>& c) {
    const T urx = (x - F.ax) * F.inv_hx, ury = (y - F.ay) * F.inv_hy;
    const bool fx = !F.exact && urx >= T(5) && urx < (T)(F.qx - 7);     // false for NaN
    const bool fy = !F.exact && ury >= T(5) && ury < (T)(F.qy - 7);
    if (rt_ballot(!(fx && fy)) == 0ull) {
        axis_fast(urx, floor_(urx), c.jx, c.lx, c.lwx, c.wx);
        axis_fast(ury, floor_(ury), c.jy, c.ly, c.lwy, c.wy);
    } else {
        field_locate_mixed(rare_field(F), x, y, fx, fy, c);
    }
}

// one (d/dx-spline, d/dy-spline) coefficient pair: a single 16-byte (fp64) access in HBM and in LDS
template <typename T> using Pair = T __attribute__((ext_vector_type(2)));
#define RT_LDS __attribute__((address_space(3)))

// 36 coefficients straight from HBM/L2: 2x2 of zn and the 4x4 window of interleaved pairs.
template <typename T>
__device__ __forceinline__ void gather_global(const FieldDev<T>& F, const Cell<T>& c, T z[4], Pair<T> g[4][4]) {
    // grids hold < 2^31 values and < 2^24 per axis (field_alloc), so the element indices are 24-bit products in 32 bits
    // (64-bit index arithmetic on purpose: 24-bit multiplies measured 8 % slower here -- two more spilled registers)
    const T* zp = F.zn + (size_t)c.jy * F.qx + c.jx;
    z[0] = zp[0]; z[1] = zp[1]; z[2] = zp[F.qx]; z[3] = zp[F.qx + 1];
    const Pair<T>* gp = reinterpret_cast<const Pair<T>*>(F.g) + ((size_t)(c.ly - 3) * F.qx + (c.lx - 3));
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int q = 0; q < 4; q++) g[r][q] = gp[(size_t)r * F.qx + q];
    }
}

template <typename T> __device__ __forceinline__ void gather_none(T z[4], Pair<T> g[4][4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        z[r] = T(1);
#pragma unroll
        for (int q = 0; q < 4; q++) g[r][q] = Pair<T>{T(0), T(0)};
    }
}

// bilinear n: the two x-interpolations, then the y-interpolation (6 instructions; fpbisp's four triple products take 8)
template <typename T> __device__ __forceinline__ T bilinear(const Cell<T>& c, T z0, T z1, T z2, T z3) {
    const T r0 = fma_(z1, c.lwx[1], z0 * c.lwx[0]), r1 = fma_(z3, c.lwx[1], z2 * c.lwx[0]);
    return fma_(r1, c.lwy[1], r0 * c.lwy[0]);
}
// the bicubic gradient: one basis for both components (same knots); row sums, then the column sum.
template <typename T>
__device__ __forceinline__ void field_combine(const Cell<T>& c, const T z[4], const Pair<T> g[4][4], T& n, T& gx, T& gy) {
    n = bilinear(c, z[0], z[1], z[2], z[3]);
    T sx = 0, sy = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        T rx = g[r][0].x * c.wx[0], ry = g[r][0].y * c.wx[0];
#pragma unroll
        for (int q = 1; q < 4; q++) {
            rx = fma_(g[r][q].x, c.wx[q], rx);
            ry = fma_(g[r][q].y, c.wx[q], ry);
        }
        sx = r == 0 ? rx * c.wy[0] : fma_(rx, c.wy[r], sx);
        sy = r == 0 ? ry * c.wy[0] : fma_(ry, c.wy[r], sy);
    }
    gx = sx;
    gy = sy;
}

// gather_global + field_combine with the window consumed row by row (PHASES of 4/PHASES rows each): the same sums in the
// same order as field_combine, hence the same bits, with 1/PHASES of the window in registers at a time.  PHASES = 4 is
// the register-frugal form for the rare per-lane fallback of the tile kernel.
template <typename T, int PHASES>
__device__ __forceinline__ void lookup_global_rows(const FieldDev<T>& F, const Cell<T>& c, T& n, T& gx, T& gy) {
    const T* zp = F.zn + (size_t)c.jy * F.qx + c.jx;
    const T z0 = zp[0], z1 = zp[1], z2 = zp[F.qx], z3 = zp[F.qx + 1];
    const Pair<T>* gp = reinterpret_cast<const Pair<T>*>(F.g) + ((size_t)(c.ly - 3) * F.qx + (c.lx - 3));
    constexpr int ROWS = 4 / PHASES;
    T sx = 0, sy = 0;
#pragma unroll
    for (int ph = 0; ph < PHASES; ph++) {
        Pair<T> a[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
#pragma unroll
            for (int q = 0; q < 4; q++) a[r][q] = gp[(size_t)(ph * ROWS + r) * F.qx + q];
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            T rx = a[r][0].x * c.wx[0], ry = a[r][0].y * c.wx[0];
#pragma unroll
            for (int q = 1; q < 4; q++) {
                rx = fma_(a[r][q].x, c.wx[q], rx);
                ry = fma_(a[r][q].y, c.wx[q], ry);
            }
            const int rr = ph * ROWS + r;
            sx = rr == 0 ? rx * c.wy[0] : fma_(rx, c.wy[rr], sx);
            sy = rr == 0 ? ry * c.wy[0] : fma_(ry, c.wy[rr], sy);
        }
        if (PHASES > 1) asm volatile("" : "+v"(sx), "+v"(sy) : : "memory");   // the next rows' loads stay behind these sums
    }
    n = bilinear(c, z0, z1, z2, z3);
    gx = sx; gy = sy;
}

#ifndef RTMI_TILE_PHASES
#define RTMI_TILE_PHASES 4     // the 4x4 window is read from the LDS tile and summed in this many groups of rows
#endif
#ifndef RTMI_GLOBAL_PHASES
#define RTMI_GLOBAL_PHASES 2
#endif
// Gather policy 1: every lookup reads its 36 coefficients from global memory (L1/L2-resident in practice).
// FLATMAP (rt_exact.h, the reference-order step where the medium is constant): false in the builds for fields whose flat-cell map is
// empty -- the flat path is not even compiled in there (as run-time tests it cost the vert_heterogeneous fan in reference order a third).
template <typename T, bool FLATMAP = true> struct GlobalGather {
    // gflat: what FITPACK's gradient evaluates to at most in a FLAT cell of the map (2^-72 of the grid's largest gradient-spline
    // coefficient; rt_exact.h, the reference-order step where the medium is constant); set by the kernels from BatchDev::gflat
    double gflat = 0.0;
    static constexpr bool kUniformWindow = true;     // rt::ex::n_gradient: a wave in one cell reads the window through the scalar cache
    __device__ __forceinline__ void fetch(const FieldDev<T>& F, const Cell<T>& c, bool active, T z[4], Pair<T> g[4][4]) {
        // an idle lane reads the grid's first window instead of its stale cell: all idle lanes then share one
        // cache line, without a branch around the loads
        Cell<T> cc = c;
        cc.jx = active ? c.jx : 0; cc.jy = active ? c.jy : 0; cc.lx = active ? c.lx : 3; cc.ly = active ? c.ly : 3;
        gather_global(F, cc, z, g);
    }
    __device__ __forceinline__ void lookup(const FieldDev<T>& F, const Cell<T>& c, bool active, T& n, T& gx, T& gy);
};

// Wave-wide min / max of an int, returned wave-uniform.  DPP row shifts and row broadcasts (an inclusive scan within each
// row of 16, then rows 0/2 into 1/3 and the lower half into the upper) leave the result in lane 63: six v_min/v_max with
// a DPP operand and one v_readlane, instead of six ds_bpermute round trips through the LDS crossbar (__shfl_xor).
#ifndef RTMI_SHFL_REDUCE
template <bool MIN> __device__ __forceinline__ int wave_reduce_i(int v) {
    constexpr int ident = MIN ? 0x7fffffff : (int)0x80000000;
    auto op = [](int a, int b) { return MIN ? (b < a ? b : a) : (b > a ? b : a); };
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) { return wave_reduce_i<true>(v); }
__device__ __forceinline__ int wave_max_i(int v) { return wave_reduce_i<false>(v); }
#else
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
#endif

// Gather policy 2: a wave-private LDS tile of the field.  The rays of a wave travel together (they leave one
// origin with neighbouring angles), so their 4x4 windows overlap almost completely and move ~0.15 cell per step:
// the wave keeps the TILE x TILE block of coefficients around them in LDS, serves lookups with ds_read (a few
// LDS cycles, broadcast for equal addresses) instead of 18 texture-path loads, and re-stages only when a live
// lane's window leaves the tile (every ~50 steps).  A lane whose window is outside (incoherent wave, grid edge)
// reads global memory for that lookup -- the coefficient VALUES are the same bits either way, so results do not
// depend on the policy.  No block barrier: the tile is private to one wave and LDS executes a wave's DS
// instructions in order; fetch() must be reached in wave-uniform control flow (it votes and shuffles).
template <typename T, int PHASES = RTMI_TILE_PHASES> struct LdsGather {
    static constexpr bool kUniformWindow = false;
    static constexpr int TILE = 16;                 // coefficient rows/cols held
    static constexpr int GPITCH = TILE + 1;         // pairs per g row (one pad pair against bank aliasing)
    static constexpr int ZPITCH = TILE + 2;         // elements per zn row
    static constexpr int ELEMS = 2 * TILE * GPITCH + TILE * ZPITCH;   // in units of T
    RT_LDS Pair<T>* gt;   // this wave's region: g tile ...
    RT_LDS T* zt;         // ... then zn tile
    int ox, oy;      // tile origin in coefficient indices (wave-uniform)
    int valid;       // tile holds data
    int cooldown;    // steps to wait before trying to stage again after the wave's windows did not fit

    __device__ __forceinline__ void init(T* wave_lds) {
        gt = (RT_LDS Pair<T>*)wave_lds;
        zt = (RT_LDS T*)(wave_lds + 2 * TILE * GPITCH);
        ox = oy = 0; valid = 0; cooldown = 0;
    }

    __device__ __forceinline__ void stage(const FieldDev<T>& F) {
        // lane l copies tile column l % 16 of rows l / 16, +4, +8, +12: one source offset, then a constant stride
        const int lane = (int)__lane_id(), row0 = lane >> 4, col = lane & 15;
        const size_t src0 = (size_t)(oy + row0) * F.qx + (size_t)(ox + col), stride = (size_t)4 * F.qx;
        const Pair<T>* gp = reinterpret_cast<const Pair<T>*>(F.g) + src0;
        const T* zp = F.zn + src0;
        RT_LDS Pair<T>* gl = gt + (row0 * GPITCH + col);
        RT_LDS T* zl = zt + (row0 * ZPITCH + col);
        static_assert(TILE == 16, "stage() copies 4 rows per pass with 64 lanes");
        __builtin_amdgcn_wave_barrier();
        Pair<T> gv[4];
        T zv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { gv[q] = gp[q * stride]; zv[q] = zp[q * stride]; }
#pragma unroll
        for (int q = 0; q < 4; q++) { gl[q * 4 * GPITCH] = gv[q]; zl[q * 4 * ZPITCH] = zv[q]; }
        __builtin_amdgcn_wave_barrier();
    }

    __device__ __forceinline__ void read_tile(int cx, int cy, T z[4], Pair<T> g[4][4]) const {
        const RT_LDS Pair<T>* gw = gt + cy * GPITCH + cx;
        const RT_LDS T* zw = zt + (cy + 1) * ZPITCH + (cx + 1);
        z[0] = zw[0]; z[1] = zw[1]; z[2] = zw[ZPITCH]; z[3] = zw[ZPITCH + 1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int q = 0; q < 4; q++) g[r][q] = gw[r * GPITCH + q];
        }
    }

    // Where this lane's window sits in the tile (cx, cy) and whether it does; re-stages the tile when a live lane's
    // window has left it.  Must be reached in wave-uniform control flow (it votes and shuffles).
    __device__ __forceinline__ bool place(const FieldDev<T>& F_, const Cell<T>& c, bool active, int& cx, int& cy) {
        cx = c.lx - 3 - ox; cy = c.ly - 3 - oy;
        // the bilinear 2x2 sits at window offset (+1,+1) when the cell is not at a not-a-knot end
        const bool regular = c.jx == c.lx - 2 && c.jy == c.ly - 2;
        // an idle lane counts as served (it reads tile corner (0,0); nobody uses what it computes)
        bool fits = !active || (valid && regular && (unsigned)cx <= (unsigned)(TILE - 4) && (unsigned)cy <= (unsigned)(TILE - 4));
        if (rt_ballot(!fits) != 0ull) {
            const FieldDev<T> F = rare_field(F_);
            if (cooldown == 0 && F.qx >= TILE && F.qy >= TILE) {
                const int mnx = wave_min_i(active && regular ? c.lx - 3 : 0x7fffffff);
                const int mxx = wave_max_i(active && regular ? c.lx : -0x7fffffff);
                const int mny = wave_min_i(active && regular ? c.ly - 3 : 0x7fffffff);
                const int mxy = wave_max_i(active && regular ? c.ly : -0x7fffffff);
                if (mxx >= mnx && mxx - mnx < TILE && mxy - mny < TILE) {
                    int nx = mnx - (TILE - (mxx - mnx + 1)) / 2, ny = mny - (TILE - (mxy - mny + 1)) / 2;
                    nx = nx < 0 ? 0 : (nx > F.qx - TILE ? F.qx - TILE : nx);
                    ny = ny < 0 ? 0 : (ny > F.qy - TILE ? F.qy - TILE : ny);
                    ox = nx; oy = ny; valid = 1;
                    stage(F);
                    cx = c.lx - 3 - ox; cy = c.ly - 3 - oy;
                    fits = !active || (regular && (unsigned)cx <= (unsigned)(TILE - 4) && (unsigned)cy <= (unsigned)(TILE - 4));
                } else {
                    cooldown = 32;   // incoherent wave: lanes outside read global memory; look again later
                }
            } else if (cooldown > 0) {
                --cooldown;
            }
        }
        cx = (fits && active) ? cx : 0;
        cy = (fits && active) ? cy : 0;
        return fits;
    }

    __device__ __forceinline__ void fetch(const FieldDev<T>& F, const Cell<T>& c, bool active, T z[4], Pair<T> g[4][4]) {
        int cx, cy;
        const bool fits = place(F, c, active, cx, cy);
        if (rt_ballot(!fits) == 0ull) {
            read_tile(cx, cy, z, g);              // the common case, wave-uniform: straight-line LDS reads
        } else if (fits) {
            read_tile(cx, cy, z, g);
        } else {
            gather_global(rare_field(F), c, z, g);   // this lane's window is outside the tile (or at a grid end)
            // Retire these loads here.  Otherwise the compiler, which shares the ds_read block between this
            // mixed case and the all-lanes-fit case, guards the LDS reads with vmcnt waits -- and vmcnt also
            // counts the trajectory stores of the previous step, so every step would wait for HBM writes.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
        }
    }

    // fetch + field_combine.  With every lane in the tile the window is read and summed in RTMI_TILE_PHASES groups of rows
    // (same sums in the same order as field_combine, so the same bits).  Row by row (RTMI_TILE_PHASES 4): 4 LDS reads in
    // flight instead of 16 and 128 instead of 182 VGPRs, i.e. four waves per SIMD without spilling for op2/op6.
    __device__ __forceinline__ void lookup(const FieldDev<T>& F, const Cell<T>& c, bool active, T& n, T& gx, T& gy) {
        int cx, cy;
        const bool fits = place(F, c, active, cx, cy);
        if (rt_ballot(!fits) == 0ull || fits) {   // every lane in the tile (the common, wave-uniform case), or this one is
            if constexpr (PHASES > 1) {
            const RT_LDS Pair<T>* gw = gt + cy * GPITCH + cx;
            const RT_LDS T* zw = zt + (cy + 1) * ZPITCH + (cx + 1);
            const T z0 = zw[0], z1 = zw[1], z2 = zw[ZPITCH], z3 = zw[ZPITCH + 1];
            constexpr int ROWS = 4 / PHASES;
            T sx = 0, sy = 0;
#pragma unroll
            for (int ph = 0; ph < PHASES; ph++) {
                Pair<T> a[ROWS][4];
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) a[r][q] = gw[(ph * ROWS + r) * GPITCH + q];
                }
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    T rx = a[r][0].x * c.wx[0], ry = a[r][0].y * c.wx[0];
#pragma unroll
                    for (int q = 1; q < 4; q++) {
                        rx = fma_(a[r][q].x, c.wx[q], rx);
                        ry = fma_(a[r][q].y, c.wx[q], ry);
                    }
                    const int rr = ph * ROWS + r;
                    sx = rr == 0 ? rx * c.wy[0] : fma_(rx, c.wy[rr], sx);
                    sy = rr == 0 ? ry * c.wy[0] : fma_(ry, c.wy[rr], sy);
                }
                asm volatile("" : "+v"(sx), "+v"(sy) : : "memory");   // the next rows' reads stay behind these sums
            }
            n = bilinear(c, z0, z1, z2, z3);
            gx = sx; gy = sy;
            } else {   // PHASES == 1: the whole window in flight at once (the latency build: registers to spare, nothing to hide behind)
            T z[4];
            Pair<T> g[4][4];
            read_tile(cx, cy, z, g);
            field_combine(c, z, g, n, gx, gy);
            }
        } else {
            // this lane's window is outside the tile (or at a grid end): row by row from global memory, so that this rare
            // branch does not set the kernel's register count (36 coefficients + 18 addresses in flight did)
            lookup_global_rows<T, 4>(rare_field(F), c, n, gx, gy);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), see fetch()
        }
    }
};

// n_gradient(vector, grd, z) (:141-156): bilinear n, bicubic dn/dx and dn/dy at (x, y) -> (n, [gx, gy]).
// `active` tells the gather policy whether this lane's lookup matters (idle lanes still execute it).
template <typename G> struct IsPoly { static constexpr bool value = false; };
template <typename T, int MODE, bool FLAT> struct PolyGather;
template <typename T, int MODE, bool FLAT> struct IsPoly<PolyGather<T, MODE, FLAT>> { static constexpr bool value = true; };
template <typename G> struct HasFlatMap { static constexpr bool value = false; };
template <typename T, int MODE, bool FLAT> struct HasFlatMap<PolyGather<T, MODE, FLAT>> { static constexpr bool value = RTMI_FLAT_MAP && FLAT; };
template <typename G> struct ReportsSteep { static constexpr bool value = false; };
template <typename T, int MODE, bool FLAT> struct ReportsSteep<PolyGather<T, MODE, FLAT>> { static constexpr bool value = RTMI_FLAT_MAP && FLAT && sizeof(T) == 8; };
template <typename T, bool FLAT> struct PolyLaneKept;
template <typename T, bool FLAT> struct IsPoly<PolyLaneKept<T, FLAT>> { static constexpr bool value = true; };
template <typename T, bool FLAT> struct ReportsSteep<PolyLaneKept<T, FLAT>> { static constexpr bool value = RTMI_FLAT_MAP && FLAT && sizeof(T) == 8; };
// ... and gathers the step loops call prefetch() on after every step
template <typename G> struct HasPrefetch { static constexpr bool value = false; };
template <typename T, bool FLAT> struct HasPrefetch<PolyLaneKept<T, FLAT>> { static constexpr bool value = true; };
// ... and the steepness of the cell (flat-cell map, FlatBits): 0 for every lookup that is not a PolyGather with the map compiled in
template <typename T, typename G>
__device__ __forceinline__ void n_gradient(const FieldDev<T>& F, G& gather, bool active, T x, T y, T& n, T& gx, T& gy, float& lam) {
    if constexpr (IsPoly<G>::value) {
        gather.lookup_xy(F, active, x, y, n, gx, gy, lam);
    } else {
        lam = 0.f;
        Cell<T> c;
        field_locate(F, x, y, c);
        gather.lookup(F, c, active, n, gx, gy);
    }
}
template <typename T, typename G>
__device__ __forceinline__ void n_gradient(const FieldDev<T>& F, G& gather, bool active, T x, T y, T& n, T& gx, T& gy) {
    if constexpr (IsPoly<G>::value) {
        gather.lookup_xy(F, active, x, y, n, gx, gy);
    } else {
        Cell<T> c;
        field_locate(F, x, y, c);
        gather.lookup(F, c, active, n, gx, gy);
    }
}
template <typename T, bool FLATMAP>
__device__ __forceinline__ void GlobalGather<T, FLATMAP>::lookup(const FieldDev<T>& F, const Cell<T>& c, bool active, T& n, T& gx, T& gy) {
    // The window in two halves of two rows: 9 loads in flight instead of 18, and the kernel fits three waves per SIMD
    // without spilling (143 instead of 168 VGPRs + 35 spilled).  A/B in one session: 15.0 vs 18.1 ms without recording,
    // 24.0 vs 30.2 ms with the full record (four phases / four waves per SIMD: 15.2 / 24.1 ms).
    Cell<T> cc = c;   // an idle lane reads the grid's first window instead of its stale cell (one shared cache line)
    cc.jx = active ? c.jx : 0; cc.jy = active ? c.jy : 0; cc.lx = active ? c.lx : 3; cc.ly = active ? c.ly : 3;
    lookup_global_rows<T, RTMI_GLOBAL_PHASES>(F, cc, n, gx, gy);
}

// ---------------------------------------------------------------- the field as one polynomial per cell (rt_polytab.h)
// Where a lookup lands: the cell and the position in it.  u = (x - a)*inv_h - j with the product taken EXACTLY (fma): the
// table's polynomials are written in precisely this coordinate (poly_axis_build inverts the same map), so u carries one
// rounding and the lookup agrees with FITPACK's evaluation to < 1e-15 of the field's scale in every cell of the grid, the
// not-a-knot end cells included (tests/test_polytab_host.py) -- there is no rim case.  What is left of the old one is
// FITPACK's argument clamp (quirk Q4) for a point outside the grid: per lane, behind a wave vote that only picks the layout.
template <typename T> struct PolyCell { int cell; T u, v; };
template <typename T> __device__ __forceinline__ void poly_axis_clamped(T x, T a, T b, T inv_h, int ncell, T& xa, T& jf, int& j) {
    x = x < a ? a : x;
    x = x > b ? b : x;
    xa = x - a;
    jf = floor_(xa * inv_h);
    jf = jf < T(0) ? T(0) : (jf > (T)(ncell - 1) ? (T)(ncell - 1) : jf);
    j = (int)jf;
}
// live: the wave's vote on `active` (the caller has it anyway).  An idle lane (a terminated ray's stale state drifts out of the
// grid) never addresses the table with its cell, so only LIVE lanes outside the grid send the wave into the clamp layout.
// (The votes are combined as lane masks in scalar registers: booleans passed around by reference come back as byte values in
// vector registers and cost a dozen vector instructions per step.)
template <typename T>
__device__ __forceinline__ void poly_locate(const FieldDev<T>& F, T x, T y, unsigned long long live, PolyCell<T>& c) {
    T xa = x - F.ax, ya = y - F.ay;
    T jfx = floor_(xa * F.inv_hx), jfy = floor_(ya * F.inv_hy);
    int jx = (int)jfx, jy = (int)jfy;                // NaN converts to 0: "inside", and the NaN flows through u
    const unsigned long long outx = rt_ballot((unsigned)jx >= (unsigned)F.ncx), outy = rt_ballot((unsigned)jy >= (unsigned)(F.qy - 1));
    if (((outx | outy) & live) != 0ull) {
        const FieldDev<T> G = rare_field(F);
        if ((unsigned)jx >= (unsigned)G.ncx) poly_axis_clamped(x, G.ax, G.bx, G.inv_hx, G.ncx, xa, jfx, jx);
        if ((unsigned)jy >= (unsigned)(G.qy - 1)) poly_axis_clamped(y, G.ay, G.by, G.inv_hy, G.qy - 1, ya, jfy, jy);
    }
    c.u = fma_(xa, F.inv_hx, -jfx);
    c.v = fma_(ya, F.inv_hy, -jfy);
    c.cell = jy * F.ncx + jx;                     // < 2^31 cells (field_alloc)
}

// Horner in u for the four rows, then in v.  ROW(k) yields row k's four coefficients; with a wave-uniform cell they sit in
// scalar registers and every fma below reads one of them as its scalar operand.
template <typename T> using Quad = T __attribute__((ext_vector_type(4)));
// a*u + c with a and c in SCALAR registers (a wave-uniform cell's coefficients), u per lane.  A vector instruction reads at
// most one scalar operand on gfx9, so c is copied to a vector register first; left to the compiler every scalar addend of
// the Horner chain is copied (it selects the two-address v_fmac, 26 copies per lookup) -- hence the explicit forms.
__device__ __forceinline__ double fma_sus(double a, double u, double c) {
    double t, d;
    asm("v_mov_b64 %0, %1" : "=v"(t) : "s"(c));
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(u), "v"(t));
    return d;
}
__device__ __forceinline__ float fma_sus(float a, float u, float c) {
    float t, d;
    asm("v_mov_b32 %0, %1" : "=v"(t) : "s"(c));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(u), "v"(t));
    return d;
}
// a*u + c with only the addend c in scalar registers: the three-address form, no copy
__device__ __forceinline__ double fma_vus(double a, double u, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(u), "s"(c));
    return d;
}
__device__ __forceinline__ float fma_vus(float a, float u, float c) {
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(u), "s"(c));
    return d;
}
// Horner in u for one row.  SC 1: the coefficients are scalar-register values; SC 2: they are vector registers that must
// survive the lookup (the kept cell of PolyGather's CACHED mode): three-address fma, or the compiler copies each addend for
// its two-address v_fmac.  Same operations in every form: same bits.
template <typename T, int SC> __device__ __forceinline__ T poly_row(Quad<T> a, T u) {
    if constexpr (SC == 1) return fma_vus(fma_vus(fma_sus(a.w, u, a.z), u, a.y), u, a.x);
    else if constexpr (SC == 2) return fma_const(fma_const(fma_const(a.w, u, a.z), u, a.y), u, a.x);
    else return fma_(fma_(fma_(a.w, u, a.z), u, a.y), u, a.x);
}
template <typename T, int SC, typename ROW> __device__ __forceinline__ T poly_bicubic(ROW row, int base, T u, T v) {
    const T r3 = poly_row<T, SC>(row(base + 3), u), r2 = poly_row<T, SC>(row(base + 2), u);
    const T r1 = poly_row<T, SC>(row(base + 1), u), r0 = poly_row<T, SC>(row(base), u);
    return fma_(fma_(fma_(r3, v, r2), v, r1), v, r0);
}
template <typename T, int SC> __device__ __forceinline__ T poly_bilinear(Quad<T> b, T u, T v) {
    if constexpr (SC == 1) return fma_(fma_sus(b.w, u, b.z), v, fma_sus(b.y, u, b.x));
    else if constexpr (SC == 2) return fma_(fma_const(b.w, u, b.z), v, fma_const(b.y, u, b.x));
    else return fma_(fma_(b.w, u, b.z), v, fma_(b.y, u, b.x));
}

// Gather policy 3 (the fast-form step methods): the cell's polynomial.
//   SCALAR: the wave's live lanes are asked for their cell; the lanes in the first live lane's cell evaluate with that cell's
//   36 coefficients read through the scalar cache into SGPRs (5 s_load per lookup, no vector register, no LDS), the rest go
//   round once more, and whoever is still left after two rounds -- an incoherent wave: shuffled or random rays -- reads its
//   own cell with vector loads.  A fan's wave sits in ONE cell on 98 % of its steps (64 neighbouring rays of 1 M span 4 % of
//   a cell) and in two on the rest.
//   !SCALAR: every lane reads its own cell with vector loads (two rows in flight): the per-ray-DELTA_S sweep, field_path 1.
//   CACHED (the build for at most two waves per SIMD, k_advance_lat): nothing hides a scalar load's latency there (six waits
//   of ~130 clocks per step measured: cfg2 3.3 ms against 2.3 with the LDS tile), and registers are plentiful, so the wave
//   KEEPS its cell's 36 coefficients in vector registers (every lane the same values) and reloads them only when the first
//   live lane's cell changes -- every seventh step on the vert fan; a lookup in the kept cell is 33 fma and no memory access.
// The arithmetic is poly_bicubic / poly_bilinear on the same numbers either way: the result does not depend on the policy,
// on the wave mates or on which round served the lane.
constexpr int kPolyLane = 0, kPolyScalar = 1, kPolyCached = 2;
// The flat-cell map (FieldDev::flat).  An entry is the cell's constant index, or -- for an ordinary cell -- a NaN: in fp32 fields
// all-ones bits; in fp64 fields the high word all ones and the low word the cell's STEEPNESS as float bits (0 for most cells).
// Steepness (k_polytab; rtmi.hip "critical rays"): lambda = sqrt(|Hessian of n| / n) over the cell, the rate (per unit length) at
// which a ray running along the iso-lines of a transition that sharp drifts away from its neighbours, kept when it is at least
// FieldDev-wide lambda_0 (so: only in cells of a SHARP transition -- the interface scenario's sigmoid; no cell of the fisheye or
// vert_heterogeneous grids).  The fused step kernels add it up over the steps a ray hovers in such cells (Ray::hov) to find the
// few rays per million whose trajectory amplifies rounding differences past the 1e-9 the path is held to; those are re-traced in
// the reference's operation order (rtmi.hip, retrace).
template <typename T> struct FlatBits;
template <> struct FlatBits<double> { typedef unsigned long long type; };
template <> struct FlatBits<float> { typedef unsigned type; };
template <typename T> __device__ __forceinline__ bool flat_entry(typename FlatBits<T>::type b) {
    if constexpr (sizeof(T) == 8) return (unsigned)(b >> 32) != 0xffffffffu;
    else return b != ~(typename FlatBits<T>::type)0;
}
template <typename T> __device__ __forceinline__ float steep_of(typename FlatBits<T>::type b) {
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(float, (unsigned)b);     // only read when !flat_entry(b)
    else return 0.f;
}
__host__ __device__ inline unsigned long long steep_entry_bits(float lam) {
    return 0xffffffff00000000ull | (unsigned long long)__builtin_bit_cast(unsigned, lam);
}
// per lane (vector load): the lanes of an incoherent wave, the one-lane-per-point lookups.  lam: the cell's steepness (0: none)
template <typename T> __device__ __forceinline__ bool flat_lane(const FieldDev<T>& F, int cell, T& c, float& lam) {
    typedef typename FlatBits<T>::type B;
    const B b = reinterpret_cast<const B*>(F.poly)[(long)cell - (long)F.flat];
    c = __builtin_bit_cast(T, b);
    const bool fl = flat_entry<T>(b);
    lam = fl ? 0.f : steep_of<T>(b);
    return fl;
}
// for a wave-uniform cell, through the scalar cache
template <typename T> __device__ __forceinline__ bool flat_uniform(const FieldDev<T>& F, int cu, T& c, float& lam) {
    typedef typename FlatBits<T>::type B;
    typedef const B __attribute__((address_space(4)))* SP;
    SP q = (SP)(F.poly) + ((long)cu - (long)F.flat);
    asm volatile("" : "+s"(q));
    const B b = *q;
    c = __builtin_bit_cast(T, b);
    const bool fl = flat_entry<T>(b);
    lam = fl ? 0.f : steep_of<T>(b);
    return fl;
}
// FLAT false: built for fields WITHOUT flat cells (the host knows: rtmi_field::flat_cells) -- the map's tests are not even
// compiled in.  Left in as run-time tests on a scalar they cost the fisheye fan 4.5 % and the vert fan 1.4 % (register allocation
// of kernels that sit at their budget, not the two scalar instructions: profiles/r04_ab_flat_tests_cost.txt), so the two hot
// kernels (k_advance and k_advance_sliced on the wave-shared path) exist in both forms and the host picks.
template <typename T, int MODE, bool FLAT = true> struct PolyGather {
    static constexpr bool SCALAR = MODE == kPolyScalar;
    static constexpr bool CACHED = MODE == kPolyCached;
    static constexpr int NA = CACHED ? 9 : 1;     // rows held
    static constexpr bool kPoly = true;
    typedef const Quad<T> __attribute__((address_space(4)))* ScalarRows;
    static __device__ __forceinline__ void eval_scalar(ScalarRows p, T u, T v, T& n, T& gx, T& gy) {
#if RTMI_POLY_BATCH == 0
        auto row = [&](int k) -> Quad<T> { return p[k]; };
        gx = poly_bicubic<T, 1>(row, 0, u, v);
        gy = poly_bicubic<T, 1>(row, 4, u, v);
        n = poly_bilinear<T, 1>(p[8], u, v);
#else
        // scalar loads return out of order, so every wait is for all of them: ask for as many rows at once as the scalar
        // registers hold (RTMI_POLY_BATCH 1: one spline + n, then the other; 2: everything) instead of row by row
        Quad<T> a[9];
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = p[k];
        a[8] = p[8];
#if RTMI_POLY_BATCH == 2
#pragma unroll
        for (int k = 4; k < 8; k++) a[k] = p[k];
#endif
        auto row = [&](int k) -> Quad<T> { return a[k]; };
        gx = poly_bicubic<T, 1>(row, 0, u, v);
        n = poly_bilinear<T, 1>(a[8], u, v);
#if RTMI_POLY_BATCH == 1
        asm volatile("" : "+v"(gx), "+v"(n) : : "memory");
#pragma unroll
        for (int k = 4; k < 8; k++) a[k] = p[k];
#endif
        gy = poly_bicubic<T, 1>(row, 4, u, v);
#endif
    }
    // one lane, its own cell: the flat-cell rule first (a flat cell's coefficients are never read), else the polynomial
    static __device__ __forceinline__ void eval_lane(const FieldDev<T>& F, int cell, T u, T v, T& n, T& gx, T& gy, float& lam) {
        if (RTMI_FLAT_MAP && FLAT && F.flat) {
            T cf;
            if (flat_lane(F, cell, cf, lam)) { n = cf; gx = T(0); gy = T(0); return; }
        }
        eval_lane_poly(F, cell, u, v, n, gx, gy);
    }
    static __device__ __forceinline__ void eval_lane_poly(const FieldDev<T>& F, int cell, T u, T v, T& n, T& gx, T& gy) {
        const Quad<T>* p = reinterpret_cast<const Quad<T>*>(F.poly + (size_t)cell * kPolyStride);
        if constexpr (CACHED) {
            // this build has the registers: all nine rows in flight, one memory latency (the cells of a wave that straddles a
            // grid line were all used a step ago: L1 hits)
            Quad<T> a[9];
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = p[k];
            auto row = [&](int k) -> Quad<T> { return a[k]; };
            gx = poly_bicubic<T, 0>(row, 0, u, v);
            gy = poly_bicubic<T, 0>(row, 4, u, v);
            n = poly_bilinear<T, 0>(a[8], u, v);
            return;
        }
        T g[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            // two rows in flight (16 VGPRs in fp64), the next two behind their sums: this path must not set the register count
            Quad<T> a3 = p[4 * s + 3], a2 = p[4 * s + 2];
            T r3 = poly_row<T, 0>(a3, u), r2 = poly_row<T, 0>(a2, u);
            asm volatile("" : "+v"(r3), "+v"(r2) : : "memory");
            Quad<T> a1 = p[4 * s + 1], a0 = p[4 * s];
            T r1 = poly_row<T, 0>(a1, u), r0 = poly_row<T, 0>(a0, u);
            g[s] = fma_(fma_(fma_(r3, v, r2), v, r1), v, r0);
            asm volatile("" : "+v"(g[s]) : : "memory");
        }
        gx = g[0]; gy = g[1];
        n = poly_bilinear<T, 0>(p[8], u, v);
    }
    // CACHED: the kept cell and its nine rows (wave-uniform values in vector registers); lanes outside it read their own cell
    // (218 VGPRs, two waves per SIMD).  Tried and measured slower, A/B in one session each: TWO kept cells (a wave that is
    // crossing a grid line has lanes on both sides for a few steps; the cell ahead into the slot not served from last) --
    // 304 VGPRs and twice the control flow: cfg2 2.59 vs 2.21 ms, a one-cell-wide fan 2.00 vs 1.86; reloading through the
    // scalar cache with 72 copies into the vector registers (no vmcnt wait behind the trajectory stores): 2.54 vs 2.25 ms.
    int tagA;
    float lamA;           // CACHED: the kept cell's steepness
    // critical rays (hover_update): the hover sum beyond which a fused fp64 op1/2/6/8 step ends its ray for the re-trace, in units of
    // steepness (kHoverLimit / DELTA_S); +inf where nothing is handed over.  Set by the kernels from BatchDev::hov_limit.
    float hov_limit = __builtin_inff();
    Quad<T> rowsA[NA];
    // Does this lookup report steepness?  Only where the map's tests are compiled in, and only fp64 (the fp32 map has no room
    // for it and fp32 batches have no reference to be re-traced against).
    static constexpr bool kSteep = RTMI_FLAT_MAP && FLAT && sizeof(T) == 8;
    __device__ __forceinline__ void init() { tagA = -1; lamA = 0.f; }
    template <int N> static __device__ __forceinline__ void eval_rows(const Quad<T> (&rows)[N], T u, T v, T& n, T& gx, T& gy) {
        auto row = [&](int k) -> Quad<T> { return rows[N == 9 ? k : 0]; };
        gx = poly_bicubic<T, 2>(row, 0, u, v);
        gy = poly_bicubic<T, 2>(row, 4, u, v);
        n = poly_bilinear<T, 2>(rows[N == 9 ? 8 : 0], u, v);
    }
    template <int N> static __device__ __forceinline__ void load_rows(Quad<T> (&rows)[N], const FieldDev<T>& F, int cu, float& lam) {
        lam = 0.f;
        if (RTMI_FLAT_MAP && FLAT && F.flat) {       // a flat cell is kept as the polynomial (b0, 0, ...): eval_rows then gives (b0, 0, 0) exactly
            T cf;
            if (flat_uniform(F, cu, cf, lam)) {
#pragma unroll
                for (int k = 0; k < N; k++) rows[k] = Quad<T>{T(0), T(0), T(0), T(0)};
                rows[N == 9 ? 8 : 0].x = cf;
                return;
            }
        }
#if RTMI_LAT_RELOAD == 1
        // through the scalar cache, three rows at a time, copied into the kept cell's vector registers: scalar loads count in
        // lgkmcnt, so this reload does not wait for the trajectory stores of the steps before (vmcnt is one in-order counter)
        {
            ScalarRows p = (ScalarRows)(F.poly + (size_t)cu * kPolyStride);
            asm volatile("" : "+s"(p));
#pragma unroll
            for (int g = 0; g < N; g += 3) {
                Quad<T> a0 = p[g], a1 = p[g + 1 < N ? g + 1 : g], a2 = p[g + 2 < N ? g + 2 : g];
                rows[g] = a0;
                if (g + 1 < N) rows[g + 1] = a1;
                if (g + 2 < N) rows[g + 2] = a2;
                asm volatile("" : "+v"(rows[g]) : : "memory");
            }
            return;
        }
#endif
        // every lane loads the same 288 bytes (one address per instruction: a broadcast in the texture path)
        typedef const Quad<T> __attribute__((address_space(1)))* GlobalRows;
        GlobalRows p = (GlobalRows)(F.poly + (size_t)cu * kPolyStride);
        asm volatile("" : "+v"(p));          // a per-lane address on purpose: vector loads into vector registers
#pragma unroll
        for (int k = 0; k < N; k++) rows[k] = p[k];
        // retire the loads here, in the rare branch: left pending they make every later step's first use of a row wait for
        // vmcnt(0), which also counts the trajectory stores of the step before
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    // lam: the steepness of the cell the lane's point lies in (0 in almost every cell, see FlatBits above), for Ray::hov
    __device__ __forceinline__ void lookup_xy(const FieldDev<T>& F, bool active, T x, T y, T& n, T& gx, T& gy, float& lam) {
        PolyCell<T> c;
        const unsigned long long live = rt_ballot(active);
        poly_locate(F, x, y, live, c);
        lam = 0.f;
        if constexpr (CACHED || SCALAR) {
            // no live lane (the step loops leave before this can happen): nothing addresses the table with an idle lane's cell
            if (live == 0ull) { n = T(1); gx = T(0); gy = T(0); return; }
        }
        if constexpr (CACHED) {
            const int cu = __builtin_amdgcn_readlane(c.cell, __builtin_ctzll(live));
            if (cu != tagA) { load_rows(rowsA, F, cu, lamA); tagA = cu; }
            // every lane evaluates the first live lane's cell in straight-line code; lanes of another cell are redone
            eval_rows(rowsA, c.u, c.v, n, gx, gy);
            lam = lamA;
            if (active && c.cell != cu) eval_lane(F, c.cell, c.u, c.v, n, gx, gy, lam);
        } else if constexpr (SCALAR) {
            // the first live lane's cell; when every live lane is in it (98 % of a fan's wave-steps) all lanes evaluate its
            // polynomial in straight-line code -- an idle lane too, at its own (u, v) in [0, 1)^2: finite, and nobody reads it
            int cu = __builtin_amdgcn_readlane(c.cell, __builtin_ctzll(live));
            // the entry's address is formed from the scalar and pinned to scalar registers BEFORE any comparison with the
            // per-lane cell: inside "cell == cu" the compiler would otherwise substitute the lane's value and load per lane
            ScalarRows p = (ScalarRows)(F.poly + (size_t)cu * kPolyStride);
            asm volatile("" : "+s"(p));
            if ((rt_ballot(c.cell != cu) & live) == 0ull) {
                if (RTMI_FLAT_MAP && FLAT && F.flat) {
                    T cf;
                    if (flat_uniform(F, cu, cf, lam)) { n = cf; gx = T(0); gy = T(0); return; }     // scalar branch: the map entry is wave-uniform
                }
                eval_scalar(p, c.u, c.v, n, gx, gy);
                return;
            }
            // a wave in several cells: two rounds of "the first waiting lane's cell", then per-lane loads for whoever is left
            n = T(1); gx = T(0); gy = T(0);
            bool todo = active;
#pragma nounroll
            for (int round = 0; round < 2; ++round) {
                if (rt_ballot(todo) == 0ull) break;
                if (todo) {
                    cu = __builtin_amdgcn_readfirstlane(c.cell);
                    p = (ScalarRows)(F.poly + (size_t)cu * kPolyStride);
                    asm volatile("" : "+s"(p));
                    if (c.cell == cu) {
                        T cf;
                        float lu = 0.f;
                        if (RTMI_FLAT_MAP && FLAT && F.flat && flat_uniform(F, cu, cf, lu)) { n = cf; gx = T(0); gy = T(0); }
                        else eval_scalar(p, c.u, c.v, n, gx, gy);
                        lam = lu;
                        todo = false;
                    }
                }
            }
            if (todo) eval_lane(F, c.cell, c.u, c.v, n, gx, gy, lam);
        } else {
            n = T(1); gx = T(0); gy = T(0);          // what an idle lane steps on with (finite; nobody reads its state)
            if (active) eval_lane(F, c.cell, c.u, c.v, n, gx, gy, lam);
        }
    }
    __device__ __forceinline__ void lookup_xy(const FieldDev<T>& F, bool active, T x, T y, T& n, T& gx, T& gy) {
        float lam;
        lookup_xy(F, active, x, y, n, gx, gy, lam);
    }
};

// One lane, one ray, ITS OWN kept cell (k_retrace_tail: the fused tails of re-traced critical rays -- a lone wave whose 64 rays are
// anywhere).  A lone wave's step is mostly the wait for its lookup: 1.4 us per step with per-lane loads (map entry, then the rows:
// two round trips) where the arithmetic takes 0.5.  Here every lane keeps the nine rows and the map entry of the cell its ray is
// in (a ray stays ~10 steps in a cell), all ten loads of a new cell go out together, and the step loop starts them a step AHEAD
// (prefetch: where the ray will be after the next step if it goes on as it goes now), so they travel while the step's arithmetic
// runs.  A wrong guess costs a reload at the point of use, never a wrong value: the tag is the cell the registers hold.
// The polynomial is evaluated on vector registers exactly as PolyGather's kept cell is (the same Horner chain, the same bits);
// a flat cell's entry overrides it with (constant, 0, 0) like flat_lane does.
template <typename T, bool FLAT = true> struct PolyLaneKept {
    static constexpr bool kPoly = true;
    typedef typename FlatBits<T>::type B;
    int tag;
    B ent;
    Quad<T> rows[9];
    float hov_limit = __builtin_inff();
    __device__ __forceinline__ void init() {
        tag = -1; ent = ~(B)0;
#pragma unroll
        for (int k = 0; k < 9; k++) rows[k] = Quad<T>{T(0), T(0), T(0), T(0)};
    }
    __device__ __forceinline__ void load(const FieldDev<T>& F, int cell) {
        tag = cell;
        if (RTMI_FLAT_MAP && FLAT) ent = F.flat ? reinterpret_cast<const B*>(F.poly)[(long)cell - (long)F.flat] : ~(B)0;
        const Quad<T>* p = reinterpret_cast<const Quad<T>*>(F.poly + (size_t)cell * kPolyStride);
#pragma unroll
        for (int k = 0; k < 9; k++) rows[k] = p[k];
    }
    __device__ __forceinline__ void lookup_xy(const FieldDev<T>& F, bool active, T x, T y, T& n, T& gx, T& gy, float& lam) {
        PolyCell<T> c;
        poly_locate(F, x, y, rt_ballot(active), c);
        if (active && c.cell != tag) load(F, c.cell);
        auto row = [&](int k) -> Quad<T> { return rows[k]; };
        gx = poly_bicubic<T, 0>(row, 0, c.u, c.v);
        gy = poly_bicubic<T, 0>(row, 4, c.u, c.v);
        n = poly_bilinear<T, 0>(rows[8], c.u, c.v);
        lam = 0.f;
        if (RTMI_FLAT_MAP && FLAT) {
            const bool fl = flat_entry<T>(ent);
            lam = fl ? 0.f : steep_of<T>(ent);
            if (fl) { n = __builtin_bit_cast(T, ent); gx = T(0); gy = T(0); }
        }
        if (!active) { n = T(1); gx = T(0); gy = T(0); lam = 0.f; }      // what an idle lane steps on with (finite; nobody reads its state)
    }
    __device__ __forceinline__ void lookup_xy(const FieldDev<T>& F, bool active, T x, T y, T& n, T& gx, T& gy) {
        float lam;
        lookup_xy(F, active, x, y, n, gx, gy, lam);
    }
    __device__ __forceinline__ void prefetch(const FieldDev<T>& F, bool active, T xp, T yp) {
        const T jfx = floor_((xp - F.ax) * F.inv_hx), jfy = floor_((yp - F.ay) * F.inv_hy);
        if (active && jfx >= T(0) && jfx < (T)F.ncx && jfy >= T(0) && jfy < (T)(F.qy - 1)) {
            const int cell = (int)jfy * F.ncx + (int)jfx;
            if (cell != tag) load(F, cell);
        }
    }
};

// ---------------------------------------------------------------- per-ray state
// The six quantities a ray ACCUMULATES over thousands of steps -- position, angle, the two arclengths, traveltime --
// are fp64 in both precisions (in registers and in HBM).  With T = float the field, its lookup and the whole step
// arithmetic are fp32 and each step's increment is formed in fp32, then added to the fp64 accumulator: adding 2.6e-3
// steps onto coordinates of order 5 in fp32 would round every step to 2.4e-7.  With T = double nothing changes.
using Acc = double;
template <typename T> struct Ray {
    Acc x, y, th;            // position, angle (accumulators)
    T n, gx, gy;             // index, gradient at the current point
    T ux, uy, coef, nray;    // derived: unit tangent, anisotropy(theta,gamma), coef*n
    T rn;                    // derived: 1/n (rcp_full, < 1 ulp), shared by the three divisions by n of a step
    Acc dsim, dreal, tt;     // simulated / expected arclength, traveltime (accumulators)
    T mx, my;                // momenta of the current row (output only)
    T hx0, hy0, hx1, hy1;    // op7: the two positions before (x,y), oldest first (VECTOR_LIST, Q11)
    float hov;               // fused fp64 op1/2/6/8 on a field with steep cells: sum of the steepness of the cells in which the ray
                             // ran along the iso-lines (hover_update); times DELTA_S it grows with the factor by which the ray's
                             // trajectory amplifies a rounding difference -- past kHoverLimit the ray is re-traced in reference order
};
template <typename T> constexpr bool kMixed = !__is_same(T, double);   // fp32 arithmetic on fp64 accumulators

template <typename T> struct Consts {
    T step, step2h;          // DELTA_S and pow(DELTA_S, 2)/2 (host-computed from numpy's step**2, :330)
    T step2;                 // pow(DELTA_S, 2) itself (= 2*step2h exactly), for the reference-order arithmetic of rt_exact.h
    T gamma, g2m1;           // trazar's gamma, gamma**2-1 (:230)
    T gamma_s, g2m1_s;       // module-global gamma of op10/op11 (Q12)
    T box[4];
    // op10/op11 (rt_exact.h, golden_filtered phase T): suprema over all angles of the k-th derivatives (k = 1..4) of the unit
    // momentum curve (cos t, gamma_s^2 sin t)/a(t) -- [0..3], the larger of the two components -- and of a(t) -- [4..7]
    T gold_sup[8];
    // (nothing is added here or to FieldDev lightly: both sit in front of the batch's members in every step kernel's argument segment,
    // and the register allocation of the kernels at their register cap follows those offsets -- 8 bytes more cost interface x op9 5-12 %)
};

// anisotropy(theta, gamma) (:118-119) from sin/cos.  ISO (gamma == 1): sqrt(s^2 + c^2), which is 1 +- ulp
// and is kept (quirk Q5), the product gamma*s is exact.
// ISO (gamma == 1): the reference's sqrt(sin^2 + cos^2) is 1 +- 1 ulp (quirk Q5) -- rounding noise of ITS sin/cos; with
// this path's own sin/cos (within 1 ulp of libm's, not equal) the noise would be a different one, so the factor is taken
// as exactly 1 here (momenta, n_ray and traveltime move by <= 1 ulp; the reference-order methods of rt_exact.h keep it).
template <typename T, bool ISO> __device__ __forceinline__ T aniso(T s, T c, T gamma) {
    if (ISO) return T(1);
    const T gs = gamma * s;
    return M<T>::sqrt_(fma_(gs, gs, c * c));
}
// moment() (:217-230) given coef = anisotropy(theta, gamma).  ISO: gamma**2-1 == 0 makes the bracket exactly 1.
template <typename T, bool ISO> __device__ __forceinline__ T moment(T n, T coef, T g2m1, T o0, T o1) {
    if (ISO) return n * o0;
    return n * coef * o0 * (T(1) + o1 * g2m1 / (coef * coef));
}
template <typename T> __device__ __forceinline__ T impulse(T a, T b, T step) { return step * (a + b) * T(0.5); }

// ---- advancement (:300-365)
template <typename T> __device__ __forceinline__ void adv_first(const Ray<T>& r, T step, Acc& fx, Acc& fy) {
    if constexpr (kMixed<T>) {
        fx = r.x + (Acc)(r.ux * step);
        fy = r.y + (Acc)(r.uy * step);
    } else {
        fx = fma_(r.ux, step, r.x);
        fy = fma_(r.uy, step, r.y);
    }
}
// eps: the step's chord is DELTA_S sqrt(1 + eps) -- the displacement is DELTA_S u + (DELTA_S^2 / 2n) v with v = grad n - (grad n . u) u
// perpendicular to the unit tangent u, so eps = (DELTA_S / 2n)^2 |v|^2 (see chord_length)
template <typename T> __device__ __forceinline__ void adv_second(const Ray<T>& r, const Consts<T>& k, Acc& fx, Acc& fy, T& eps) {
    const T d = fma_(r.gy, r.uy, r.gx * r.ux);  // np.dot on 2 elements rounds exactly like this
    const T s = k.step2h * r.rn;                // step**2 / (2 n)
    const T vx = fma_(-d, r.ux, r.gx), vy = fma_(-d, r.uy, r.gy);
    if constexpr (kMixed<T>) {
        fx = r.x + (Acc)fma_(vx, s, r.ux * k.step);
        fy = r.y + (Acc)fma_(vy, s, r.uy * k.step);
    } else {
        fx = fma_(vx, s, fma_(r.ux, k.step, r.x));
        fy = fma_(vy, s, fma_(r.uy, k.step, r.y));
    }
#if RTMI_CHORD_SERIES > 1
    const T q = (k.step * T(0.5)) * r.rn;       // the constant product folds on the host side of the loop
    eps = q * q * fma_(vy, vy, vx * vx);
#else
    eps = T(-1);
#endif
}
// The arclength of one step, np.linalg.norm(old - new) in the reference (:785).  The reference takes it from the ROUNDED
// positions (a 2.6e-3 difference of coordinates of order 5: 2e-13 relative noise per step); from the advancement itself it is
// DELTA_S sqrt(1 + eps) exactly, and for eps < 2^-26 (everywhere but within a few cells of the interface scenario's jump)
// DELTA_S (1 + eps/2) to 2^-55: one fma instead of two subtractions, a square root by rsq + 6 and its transcendental.  eps < 0:
// no closed form (curvature advancement): from the positions.  Per-lane choice; the vote selects the layout.
template <typename T> __device__ __forceinline__ T chord_length(const Consts<T>& k, const Ray<T>& r, Acc fx, Acc fy, T eps) {
#if RTMI_CHORD_SERIES
    const bool series = eps >= T(0) && eps < T(1.4901161193847656e-08);
    if (rt_ballot(!series) == 0ull) return fma_(k.step * T(0.5), eps, k.step);
    if (series) return fma_(k.step * T(0.5), eps, k.step);
    if (eps >= T(0)) return k.step * M<T>::sqrt_(T(1) + eps);
#endif
    const T dx = (T)(r.x - fx), dy = (T)(r.y - fy);
    return M<T>::sqrt_(fma_(dy, dy, dx * dx));  // np.linalg.norm on 2 elements
}
// returns the reference's flag: true == curvature NOT negligible (quirk Q14)
template <typename T> __device__ __forceinline__ bool adv_curv(const Ray<T>& r, const Consts<T>& k, Acc& fx, Acc& fy) {
    const T d = fma_(r.gy, r.uy, r.gx * r.ux);
    const T vx = fma_(-d, r.ux, r.gx), vy = fma_(-d, r.uy, r.gy);
    const T curv = M<T>::sqrt_full(fma_(vy, vy, vx * vx)) * r.rn;   // may be exactly 0: full-range sqrt
    if (curv < T(1.4901161193847656e-08)) {  // GOLD_TOL (:355), same constant in both precisions
        adv_first(r, k.step, fx, fy);
        return false;
    }
    const T dc = curv * k.step;
    const bool neg = r.gx * r.uy - r.gy * r.ux > T(0);  // np.cross (:360), two rounded products
    T s2, c2;
    M<T>::sincos_(neg ? r.th - (Acc)dc : r.th + (Acc)dc, &s2, &c2);
    // (:361) [sin th - sin(th-dc), cos(th-dc) - cos th]/curv ; (:363) [sin(th+dc) - sin th, -cos(th+dc) + cos th]/curv
    const T rc = T(1) / curv;
    if constexpr (kMixed<T>) {
        fx = r.x + (Acc)((neg ? (r.uy - s2) : (s2 - r.uy)) * rc);
        fy = r.y + (Acc)((neg ? (c2 - r.ux) : (-c2 + r.ux)) * rc);
    } else {
        fx = fma_(neg ? (r.uy - s2) : (s2 - r.uy), rc, r.x);
        fy = fma_(neg ? (c2 - r.ux) : (-c2 + r.ux), rc, r.y);
    }
    return true;
}

// ---- angle determination (:370-407)
// fn_rcp = 1/fn; the intermediate angle theta+k1 is never stored, so its sin/cos come from sincos_add.
template <typename T> __device__ __forceinline__ Acc ang_rk2(const Ray<T>& r, T step, T fn_rcp, T fgx, T fgy) {
    const T k1 = step * fma_(r.ux, r.gy, -(r.uy * r.gx)) * r.rn;
    T s2, c2;
    sincos_add(r.th, r.uy, r.ux, k1, &s2, &c2);
    const T k2 = step * fma_(c2, fgy, -(s2 * fgx)) * fn_rcp;
    return r.th + (Acc)((k1 + k2) * T(0.5));
}
// atan2(ys, yc) + theta, folded into (-pi, pi], for a vector given by its components ACROSS (ys) and ALONG (yc) the ray's unit
// tangent (cos theta, sin theta): the direction a step turns to is always next to the one it comes from, so the new angle is
// theta + atan(ys / yc) with |ys / yc| << 1 -- a reciprocal and a five-term series (|t| < 2^-5: the next term is t^10/11 <
// 2^-53) instead of a full atan2 of the untouched components (~65 instructions in ocml) -- and it is better conditioned too:
// the products with n that the two components share cancel before anything is rounded.  A lane whose turn is larger, or
// whose vector points backwards, takes atan2 itself; the vote only selects the layout.  RTMI_ATAN_NEAR 0: always atan2.
#ifndef RTMI_ATAN_NEAR
#define RTMI_ATAN_NEAR 1
#endif
template <typename T> __device__ __forceinline__ Acc angle_near(Acc theta, T ux, T uy, T ys, T yc) {
    const T t = ys * rcp_full(yc);
    const bool near = yc > T(0) && M<T>::abs_(t) < T(0.03125) && __builtin_fabs(theta) <= Acc(3.141592653589793);   // (a launch angle beyond +-pi: atan2 folds it)
    auto series = [&]() -> Acc {
        const T z = t * t;
        T p = fma_const(z, T(1.0 / 9.0), T(-1.0 / 7.0));
        p = fma_const(z, p, T(1.0 / 5.0));
        p = fma_const(z, p, T(-1.0 / 3.0));
        const Acc th = theta + (Acc)fma_(t * z, p, t);
        // atan2's range: theta itself is in (-pi, pi], the turn is small
        return th > Acc(3.141592653589793) ? th - Acc(6.283185307179586) : (th <= Acc(-3.141592653589793) ? th + Acc(6.283185307179586) : th);
    };
    auto full = [&]() -> Acc {      // the vector in the fixed frame again: (yc, ys) turned by theta
        return (Acc)M<T>::atan2_(fma_(yc, uy, ys * ux), fma_(yc, ux, -(ys * uy)));
    };
    if (rt_ballot(!near) == 0ull) return series();
    return near ? series() : full();
}
template <typename T> __device__ __forceinline__ Acc ang_cost(const Ray<T>& r, T step, T fgx, T fgy) {
#if RTMI_ATAN_NEAR
    // P = n u + I (momentum + impulse): across the tangent I x u... = I_y u_x - I_x u_y, along it n + I . u (|u| = 1)
    const T ix = impulse(r.gx, fgx, step), iy = impulse(r.gy, fgy, step);
    return angle_near<T>(r.th, r.ux, r.uy, fma_(iy, r.ux, -(ix * r.uy)), r.n + fma_(ix, r.ux, iy * r.uy));
#else
    return (Acc)M<T>::atan2_(fma_(r.n, r.uy, impulse(r.gy, fgy, step)), fma_(r.n, r.ux, impulse(r.gx, fgx, step)));
#endif
}

// golden() (:175-199) on a cost functor; recomputes both cost values every iteration like the
// reference (Q13) so the comparison sequence, and with it the returned midpoint, is the same.
template <typename T, typename F> __device__ __forceinline__ T golden(F cost, T a, T b) {
    const T GR = T(kGoldRatio);
    // bracket arithmetic stays unfused: the iteration count and the returned midpoint then follow the
    // reference's roundings exactly
    T c = b - (b - a) * GR, d = a + (b - a) * GR;
    for (int it = 0; it < kGoldMaxIter && M<T>::abs_(c - d) > M<T>::gold_tol; ++it) {
        if (cost(c) < cost(d)) b = d; else a = c;
        c = b - (b - a) * GR;
        d = a + (b - a) * GR;
    }
    return (b + a) * T(0.5);
}

template <typename T> __device__ __forceinline__ T ang_golden_iso(const Ray<T>& r, T step, T fn, T fgx, T fgy) {
    const T px = r.n * r.ux, py = r.n * r.uy;
    const T ix = impulse(r.gx, fgx, step), iy = impulse(r.gy, fgy, step);
    auto cost = [=](T t) {  // (:595, :697)
        T s, c;
        M<T>::sincos_(t, &s, &c);
        const T ex = fma_(fn, c, -px) - ix, ey = fma_(fn, s, -py) - iy;
        return fma_(ey, ey, ex * ex);
    };
    return golden<T>(cost, (T)r.th - T(kHalfPi), (T)r.th + T(kHalfPi));
}
template <typename T>
__device__ __forceinline__ T ang_golden_aniso(const Ray<T>& r, const Consts<T>& k, T fn, T fgx, T fgy) {
    // (:725-728 / :758-761); the step functions read the module-global gamma (Q12)
    const T c0 = aniso<T, false>(r.uy, r.ux, k.gamma_s);
    const T mix = moment<T, false>(r.n, c0, k.g2m1_s, r.ux, -(r.uy * r.uy));
    const T miy = moment<T, false>(r.n, c0, k.g2m1_s, r.uy, r.ux * r.ux);
    const T cgx = r.coef * r.gx, cgy = r.coef * r.gy;
    const T gam = k.gamma_s, g2 = k.g2m1_s, hstep = k.step * T(0.5);   // step*x*0.5 == (0.5*step)*x bit for bit
    auto cost = [=](T t) {  // 74 evaluations per step: every instruction here costs 74
        T s, c;
        M<T>::sincos_(t, &s, &c);
        const T a = aniso<T, false>(s, c, gam);
        // n*coef and (gamma**2-1)/coef**2, shared by both moments; the quotient by rcp_full (< 1 ulp)
        const T q = fn * a, w = g2 * rcp_full(a * a);
        const T ex = q * c * fma_(-(s * s), w, T(1)) - mix - (cgx + a * fgx) * hstep;
        const T ey = q * s * fma_(c * c, w, T(1)) - miy - (cgy + a * fgy) * hstep;
        return fma_(ey, ey, ex * ex);
    };
    return golden<T>(cost, (T)r.th - T(kHalfPi), (T)r.th + T(kHalfPi));
}

// ---- opN (:469-764), split around the field lookup: advancement, then angle determination
template <typename T, int METHOD>
__device__ __forceinline__ bool op_advance(const Consts<T>& k, const Ray<T>& r, Acc& fx, Acc& fy, T& eps) {
    if constexpr (METHOD == 1 || METHOD == 2) { adv_first(r, k.step, fx, fy); eps = RTMI_CHORD_SERIES ? T(0) : T(-1); return true; }   // |u| = 1: the chord is DELTA_S
    else if constexpr (METHOD == 3 || METHOD == 4 || METHOD == 5 || METHOD == 10) { eps = T(-1); return adv_curv(r, k, fx, fy); }
    else { adv_second(r, k, fx, fy, eps); return true; }
}
template <typename T, int METHOD>
__device__ __forceinline__ Acc op_angle(const Consts<T>& k, const Ray<T>& r, bool flag, Acc fx, Acc fy, T fn, T fgx, T fgy, T frn) {
    if constexpr (METHOD == 1 || METHOD == 8) return ang_cost(r, k.step, fgx, fgy);
    else if constexpr (METHOD == 2 || METHOD == 6) return ang_rk2(r, k.step, frn, fgx, fgy);
    else if constexpr (METHOD == 3) return flag ? ang_rk2(r, k.step, frn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 4) return flag ? ang_cost(r, k.step, fgx, fgy) : r.th;
    else if constexpr (METHOD == 5) return flag ? (Acc)ang_golden_iso(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 9) return (Acc)ang_golden_iso(r, k.step, fn, fgx, fgy);
    else if constexpr (METHOD == 10) return flag ? (Acc)ang_golden_aniso(r, k, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 11) return (Acc)ang_golden_aniso(r, k, fn, fgx, fgy);
    else {  // 7: finite_diff (:370-372) over [P0, P1, P2, P3] = [h0, h1, (x,y), f]; position differences in fp64
        const T vx = (T)(Acc(11) * fx - Acc(18) * r.x + Acc(9) * (Acc)r.hx1 - Acc(2) * (Acc)r.hx0);
        const T vy = (T)(Acc(11) * fy - Acc(18) * r.y + Acc(9) * (Acc)r.hy1 - Acc(2) * (Acc)r.hy0);
#if RTMI_ATAN_NEAR
        return angle_near<T>(r.th, r.ux, r.uy, fma_(vy, r.ux, -(vx * r.uy)), fma_(vx, r.ux, vy * r.uy));
#else
        return (Acc)M<T>::atan2_(vy, vx);
#endif
    }
}

// store_update_results (:783-790) + the row bookkeeping of the loop body (:871-875)
template <typename T, bool ISO, bool ROT = false>
__device__ __forceinline__ void store_update(const Consts<T>& k, Ray<T>& r, Acc fx, Acc fy, Acc fth, T fn, T fgx, T fgy,
                                             T frn, T eps, bool refresh = false) {
    const T dist = chord_length(k, r, fx, fy, eps);
    r.dsim += (Acc)dist;
    r.dreal += (Acc)k.step;  // quirk Q16: accumulated, not i*step
    T s, c;
    if constexpr (ROT) sincos_add<T>(fth, r.uy, r.ux, (T)(fth - r.th), &s, &c, refresh);
    else M<T>::sincos_(fth, &s, &c);
    const T coef = aniso<T, ISO>(s, c, k.gamma);
    r.mx = moment<T, ISO>(fn, coef, k.g2m1, c, -(s * s));
    r.my = moment<T, ISO>(fn, coef, k.g2m1, s, c * c);
    r.hx0 = r.hx1; r.hy0 = r.hy1; r.hx1 = (T)r.x; r.hy1 = (T)r.y;
    r.x = fx; r.y = fy; r.th = fth; r.n = fn; r.gx = fgx; r.gy = fgy; r.rn = frn;
    r.ux = c; r.uy = s; r.coef = coef;
    const T nray = ISO ? fn : coef * fn;                    // (:873)
    r.tt = r.tt + (Acc)(dist * (r.nray + nray) * T(0.5));   // (:874) quirk Q6
    r.nray = nray;
}

// derived quantities from the stored state (used when a launch (re)loads a ray from HBM)
template <typename T, bool ISO, bool HAVE_UNIT = false> __device__ __forceinline__ void derive(const Consts<T>& k, Ray<T>& r) {
    if constexpr (!HAVE_UNIT) M<T>::sincos_(r.th, &r.uy, &r.ux);      // HAVE_UNIT: (ux, uy) came with the state (RotatesUnit)
    r.coef = aniso<T, ISO>(r.uy, r.ux, k.gamma);
    r.nray = ISO ? r.n : r.coef * r.n;
    r.rn = rcp_full(r.n);
    r.mx = moment<T, ISO>(r.n, r.coef, k.g2m1, r.ux, -(r.uy * r.uy));
    r.my = moment<T, ISO>(r.n, r.coef, k.g2m1, r.uy, r.ux * r.ux);
}

template <typename T> __device__ __forceinline__ bool outside(const Consts<T>& k, const Ray<T>& r) {  // (:878)
    return r.x > (Acc)k.box[1] || r.x < (Acc)k.box[0] || r.y > (Acc)k.box[3] || r.y < (Acc)k.box[2];
}

}  // namespace rt
#include "rt_exact.h"
namespace rt {

// Methods that run in the reference's own operation order (rt_exact.h): curvature advancement and/or golden-section
// angle search, fp64 only (the reference has no fp32).
// The kernels' METHOD template value is the step method 1..11, plus kRefOrder for op1/2/6/7/8 when the batch asks for the
// reference's operation order throughout (rtmi_params.reference_order, fp64): those five then take rt_exact.h's path too --
// op2/op6 become the oracle's (= the reference's) bits, op1/7/8 differ from it by their atan2 alone.
constexpr int kRefOrder = 16;
// ... plus kFastField (with kRefOrder; op7 with RTMI_ORDER_FAST_FIELD): the reference-order step on the FAST field lookup (the
// cell's polynomial, flat-cell map, scalar cache) instead of FITPACK's sums.  op7 needs its POSITIONS to round like the
// reference's (its angle differentiates them: rtmi.hip, ref_order); n and grad n enter a position only through the second-order
// term (DELTA_S^2 / 2n)(grad n - ...) ~ 1e-6, where the lookup's 1e-15 relative difference from FITPACK's moves a rounding on
// few steps.  Everything else of the step -- the advancement's operation order, numpy's arctan2, glibc's sin / cos -- is
// rt::ex's, bit for bit.  <= 8e-11 from the reference on 4 096- and 65 536-ray fans of every scenario at 2.3 times the speed of
// reference order throughout; NOT the default, because a ray that grazes a sharp interface at its critical angle amplifies
// the few rounding differences past 1e-9 (one of 16 384 sampled rays of the 1 M-ray interface fan: 2.6e-9).
constexpr int kFastField = 32;
constexpr int base_method(int m) { return m & 15; }
template <typename T, int METHOD> struct IsExact { static constexpr bool value = false; };
template <int METHOD> struct IsExact<double, METHOD> {
    static constexpr bool value = (METHOD & kRefOrder) != 0 || METHOD == 3 || METHOD == 4 || METHOD == 5 || METHOD == 9 || METHOD == 10 || METHOD == 11;
};
inline bool is_exact_method(int method) { return method == 3 || method == 4 || method == 5 || method >= 9; }

// op2/op6 in fp64 turn by less than 2^-5 rad per step almost everywhere, so the unit vector (cos, sin) of the new angle is
// the old one rotated by the angle's increment (a 7th-order series, 14 fp64 instructions) instead of a from-scratch
// sincos with range reduction and quadrant selection (37); every kUnitRefresh-th row of a ray (by its own row index, so
// the result does not depend on lanes, waves or launch boundaries) and every larger turn recompute it from the angle.
// The unit vector is then part of a ray's state: it is stored with it (BatchDev::unit) and reloaded, never re-derived,
// so a trajectory is the same whether it runs in one launch or row by row.  Measured against the from-scratch build over
// 65 536-ray fans: 7e-14 (vert_heterogeneous), 1.4e-12 (fisheye), 4e-11 (interface, where a ray's position error is
// amplified ~1e5 times by the sharp index change) in the final state -- the size of either build's distance from the
// reference.  1024 rather than a shorter period: with lane refill the lanes of a wave are at rows of their own, and a
// wave takes the slow two-formula layout whenever any of its 64 lanes refreshes (6 % of its steps at 1024, 22 % at 256).
constexpr int kUnitRefresh = 1024;
#ifndef RTMI_UNIT_REFRESH_F32
// fp32 batches keep the from-scratch sincos: rotating in fp32 with a refresh every 16 rows gains 6 % on cfg4 but moves end
// points from 3.9e-7 to 1.5e-6 of the fp64 path (every 64 rows: 6e-6) -- measured on 65 536-ray fans, not adopted.
#define RTMI_UNIT_REFRESH_F32 0     // > 0: fp32 op2/op6 rotate the unit vector too, recomputing it every this many rows
#endif
template <typename T, int METHOD> struct RotatesUnit {
    static constexpr bool value = (RTMI_UNIT_REFRESH_F32 > 0) && (METHOD == 2 || METHOD == 6);
    static constexpr int refresh = RTMI_UNIT_REFRESH_F32 > 0 ? RTMI_UNIT_REFRESH_F32 : 1;
};
// op1 and op8 as well (RTMI_ROT_18): their new angle is the old one plus a small turn too (angle_near), so their unit tangent
// is rotated like op2/op6's.  op7 is not: its two extra state arrays' slots are the history's.
#ifndef RTMI_ROT_18
#define RTMI_ROT_18 1
#endif
constexpr bool rotating_method(int m) { return m == 2 || m == 6 || (RTMI_ROT_18 && RTMI_ATAN_NEAR && (m == 1 || m == 8)); }
template <int METHOD> struct RotatesUnit<double, METHOD> {
    static constexpr bool value = rotating_method(METHOD);
    static constexpr int refresh = kUnitRefresh;
};
inline bool rotates_unit(int method, bool f64) {
    return f64 ? rotating_method(method) : (RTMI_UNIT_REFRESH_F32 > 0 && (method == 2 || method == 6));
}

// One iteration of trazar's loop for row index i (the row being produced); returns "still inside the box".
// For op7 rows 1 and 2 are the bootstrap steps (:833-864): first- and second-order backward differences and
// no boundary test.  Every lane of a wave calls this together (the gather policy may vote); `active` marks
// the lanes whose ray is really stepping -- an idle lane just evolves a stale, finite state nobody reads.
// Critical rays.  A ray that runs ALONG a sharp transition of the medium amplifies any rounding difference: at the
// interface scenario's critical angle a million times (the reference's own rows move 1e-6 for a 1e-12 change of the launch
// angle), and there a fused step -- ~1e-16 per step from the reference's roundings -- ends up past 1e-9.  Which rays those are
// cannot be told from the launch conditions, but it shows on the way.  A displacement q across the ray obeys q'' = (n_vv / n) q
// along it (n_vv: the second derivative of n across the ray), so it grows by at most exp(integral of sqrt(|n_vv| / n) ds); in a
// wall the Hessian is n'' g g' (g the unit gradient), n_vv = n'' (v . g)^2, and with the cell's steepness lambda =
// sqrt(|Hessian n|_inf / n) (FlatBits; kept for the few cells with lambda >= lambda_0) the exponent is the sum over the ray's steps
// of lambda |v . g| DELTA_S.  hov adds lambda sqrt(1 - (g . u)^2) over the steps that end in a steep cell, with the gradient and
// the unit tangent at the new point: a ray that crosses a wall squarely adds next to nothing, one that reflects a
// few units, one that runs along it hundreds of steps' worth.  Five fp64 and four fp32 instructions, executed only when some lane
// of the wave is in a steep cell; per lane from the ray's own values: independent of wave mates, schedule and partition.
// Calibration (tools/hover_measures.py: the oracle's own trajectories around the split of the 1 M-ray fan, against the movement
// of their rows under a 1e-12 change of the launch angle, for the interface scenario's wall as it is and tilted by 3 and 11
// degrees): correlation of the sum with ln(amplification) 0.98-0.99 (0.88-0.92 for round 5's first form, which counted lambda
// over the steps heading within 0.02 rad of the iso-lines: on a tilted wall the spline's gradient wobbles by more than that and
// critical rays went uncounted); the smallest sum among the rays with amplification > 1e4 is 14.65 / 15.22 / 17.24, and a limit
// there flags exactly those (352 / 352 / 384 of the million); no other ray of the fan comes near (reflecting rays at grazing
// incidence reach 6-8).  The fused forms' rows leave 1e-9 from amplification 2.5e5 on (their distance from the reference's
// roundings is 4e-15 of a launch angle): 12 -- amplification 1e3 on the straight wall, 2 016 rays of the million -- keeps a
// factor 250 in hand.  What the sum does not see: sensitivity that is not
// exponential growth along a wall (a bent wall's focusing: tools/tilted_interface_probe.py's arcs) -- reference_order = 1 is the
// answer there.
constexpr float kHoverLimit = 12.0f;     // hov * DELTA_S beyond which a ray is re-traced in reference order
// Returns false for a lane whose sum has passed the limit: the step then reports the ray as ended, with hov = +inf as
// the mark -- the step loops have no test of their own for this on their hot path; they look at the mark where they store an ended ray.
// (The limit travels with the gather object, PolyGather::hov_limit: see Consts for why not with the constants.)
// hover_weight: |v . g| from d = g . u (not normalised) and g2 = |g|^2; the same expression wherever a step is weighed.
template <typename T> __device__ __forceinline__ float hover_weight(T d, T g2) {
    const float q = (float)(d * d), g = (float)g2;
    const float w = 1.f - q * __builtin_amdgcn_rcpf(g);
    return g > 0.f && w > 0.f ? __builtin_amdgcn_sqrtf(w) : 0.f;
}
template <typename T> __device__ __forceinline__ bool hover_update(float hov_limit, Ray<T>& r, bool active, float lam) {
    if (rt_ballot(lam != 0.f) == 0ull) return true;
    // at the END of the step, from the ray's own new state (gradient and unit tangent at the new point): nothing else of the step is
    // live there -- weighed in the middle of the step, with the tangent the step started with, the sum's few temporaries sent
    // op1's and op8's kernels, which sit at their 128 registers, into spilling 56 / 42 of them in the step loop (27 vs 22 ms)
    const T d = fma_(r.gy, r.uy, r.gx * r.ux), g2 = fma_(r.gy, r.gy, r.gx * r.gx);
    if (active && lam != 0.f) {
        r.hov = fmaf(lam, hover_weight(d, g2), r.hov);
        if (r.hov > hov_limit) { r.hov = INFINITY; return false; }
    }
    return true;
}

template <typename T, int METHOD, bool ISO, typename G>
__device__ __forceinline__ bool ray_step(const FieldDev<T>& F, const Consts<T>& k, G& gather, bool active, Ray<T>& r, int i) {
    if constexpr (IsExact<T, METHOD>::value) {
        return ex::ray_step<base_method(METHOD)>(F, k, gather, active, r, i);
    } else {
    Acc fx, fy, fth;
    T fn, fgx, fgy;
    T eps;
    const bool flag = op_advance<T, METHOD>(k, r, fx, fy, eps);
    constexpr bool kHover = ReportsSteep<G>::value && RotatesUnit<T, METHOD>::value;     // fp64 op1/2/6/8 with the flat-cell map compiled in
    float lam = 0.f;
    if constexpr (kHover) n_gradient(F, gather, active, (T)fx, (T)fy, fn, fgx, fgy, lam);
    else n_gradient(F, gather, active, (T)fx, (T)fy, fn, fgx, fgy);
    const T frn = rcp_full(fn);
    // (A "constant-medium step" -- where the gradient is exactly zero at both ends of the step the angle determination and the rotation
    // of the unit tangent return what they were given, term by term, and can be skipped: 55 of the step's 135 vector instructions on
    // three quarters of the interface fan's steps -- was built, proven bit-identical, and measured: 23.5 vs 23.1 ms.  The interface
    // kernel waits for the flat-cell map's entry of each new position, not for arithmetic.  Not kept.)
    bool boot = false;
    if (METHOD == 7 && i <= 2) {
        T vx, vy;
        if (i == 1) { vx = (T)(fx - r.x); vy = (T)(fy - r.y); }                                // (:843)
        else { vx = (T)(Acc(3) * fx - Acc(4) * r.x + (Acc)r.hx1); vy = (T)(Acc(3) * fy - Acc(4) * r.y + (Acc)r.hy1); }   // (:856)
        fth = (Acc)M<T>::atan2_(vy, vx);
        boot = true;  // no boundary test in the bootstrap
    } else {
        fth = op_angle<T, METHOD>(k, r, flag, fx, fy, fn, fgx, fgy, frn);
    }
    if constexpr (RotatesUnit<T, METHOD>::value)
        store_update<T, ISO, true>(k, r, fx, fy, fth, fn, fgx, fgy, frn, eps, (i & (RotatesUnit<T, METHOD>::refresh - 1)) == 0);
    else
        store_update<T, ISO>(k, r, fx, fy, fth, fn, fgx, fgy, frn, eps);
    bool calm = true;
    if constexpr (kHover) calm = hover_update(gather.hov_limit, r, active, lam);
    return calm && (boot || !outside(k, r));
    }
}

}  // namespace rt
