// rt_exact.h -- the step arithmetic of the methods whose results hinge on last bits, in the reference's own
// operation order (fp64 only).  Included by rt_device.h.
//
// For op3/4/5/9/10/11 (curvature advancement and/or golden-section angle search) a 1-ulp difference in an
// intermediate value can move a trajectory by 1e-8 .. 1e-6, so those methods do not use the
// fused / reciprocal / uniform-knot forms of rt_device.h.  Everything here is written operation by operation like
// the reference's scalar numpy code as restated in oracle/rt_oracle.c (same association, no contraction, IEEE
// division and square root, libm-identical sin/cos from rt_libm.h), so that a ray's state is the SAME BITS as the
// oracle's after every step (atan2, used by op4 only, is ocml's and within 1 ulp).  Reference lines are RT_bench.py.
//
// The golden-section search keeps the reference's comparison sequence without paying for 74 such cost evaluations
// per step: see golden_filtered below.
#pragma once
#include "rt_libm.h"
#include "rt_golden_rot.h"

namespace rt {
namespace ex {

__device__ const double kSinCosTab[4 * RT_SINCOS_TAB_ENTRIES] = {RT_SINCOS_TAB_VALUES};

// Real (not inlined) functions: each is ~150 instructions with four argument ranges, called a handful of times per
// step; inlining them at every call site multiplied the kernels' code size and register pressure for nothing.
// Beyond the range rt_libm.h reproduces (|x| >= 1.05e8, never reached by the path) the value is sincos_k's.
__device__ __attribute__((noinline)) double sin_(double x) {
    if (gl::in_range(x)) return gl::sin(kSinCosTab, x);
    double s, c;
    sincos_k(x, &s, &c);
    return s;
}
__device__ __attribute__((noinline)) double cos_(double x) {
    if (gl::in_range(x)) return gl::cos(kSinCosTab, x);
    double s, c;
    sincos_k(x, &s, &c);
    return c;
}
__device__ __forceinline__ double sqrt_(double x) { return __dsqrt_rn(x); }            // correctly rounded
__device__ __forceinline__ double sq(double x) { return x * x; }                        // oracle SQ()
__device__ __forceinline__ double dot2(double a0, double a1, double b0, double b1) { return __builtin_fma(a1, b1, a0 * b0); }
__device__ __forceinline__ double norm2(double a0, double a1) { return sqrt_(dot2(a0, a1, a0, a1)); }
__device__ __forceinline__ double aniso(double s, double c, double gamma) { return sqrt_(sq(gamma * s) + sq(c)); }   // :118-119
// moment() (:217-230) with coef = anisotropy(theta, gamma) passed in
__device__ __forceinline__ double moment(double n, double coef, double g2m1, double o0, double o1) {
    return n * coef * o0 * (1.0 + o1 * g2m1 / sq(coef));
}
__device__ __forceinline__ double impulse(double a, double b, double step) { return step * (a + b) / 2.0; }          // :214

// ---------------------------------------------------------------- n_gradient in FITPACK's arithmetic
// One axis: fpbisp's clamp + interval search (through rt::locate, exact on the true knots) and fpbspl for k = 1
// and k = 3 with IEEE divisions and separate multiply/add, exactly as oracle/rt_oracle.c fpbspl().
__device__ __forceinline__ void axis_exact(double v, int q, double a, double h, double b, double ih, int& j, int& l,
                                           double wl[2], double w[4]) {
    double t0, t1;
    j = locate(v, q, a, h, b, ih, t0, t1);      // v is clamped in place (quirk Q4)
    {   // k = 1 on [t0, t1]
        const double f = 1.0 / (t1 - t0);
        wl[0] = 0.0 + f * (t1 - v);
        wl[1] = f * (v - t0);
    }
    l = j + 2;
    l = l < 3 ? 3 : (l > q - 1 ? q - 1 : l);
    const double tm2 = knot3(l - 2, q, a, h, b), tm1 = knot3(l - 1, q, a, h, b), k0 = knot3(l, q, a, h, b);
    const double k1 = knot3(l + 1, q, a, h, b), k2 = knot3(l + 2, q, a, h, b), k3 = knot3(l + 3, q, a, h, b);
    // j = 1
    double f = 1.0 / (k1 - k0);
    double h0 = 0.0 + f * (k1 - v), h1 = f * (v - k0), h2, h3;
    // j = 2
    double hh0 = h0, hh1 = h1, hh2;
    f = hh0 / (k1 - tm1);
    h0 = 0.0 + f * (k1 - v);
    h1 = f * (v - tm1);
    f = hh1 / (k2 - k0);
    h1 = h1 + f * (k2 - v);
    h2 = f * (v - k0);
    // j = 3
    hh0 = h0; hh1 = h1; hh2 = h2;
    f = hh0 / (k1 - tm2);
    h0 = 0.0 + f * (k1 - v);
    h1 = f * (v - tm2);
    f = hh1 / (k2 - tm1);
    h1 = h1 + f * (k2 - v);
    h2 = f * (v - tm1);
    f = hh2 / (k3 - k0);
    h2 = h2 + f * (k3 - v);
    h3 = f * (v - k0);
    w[0] = h0; w[1] = h1; w[2] = h2; w[3] = h3;
}

__device__ __forceinline__ void field_locate(const FieldDev<double>& F, double x, double y, Cell<double>& c) {
    axis_exact(x, F.qx, F.ax, F.hx, F.bx, F.inv_hx, c.jx, c.lx, c.lwx, c.wx);
    axis_exact(y, F.qy, F.ay, F.hy, F.by, F.inv_hy, c.jy, c.ly, c.lwy, c.wy);
}

// fpbisp's double sum: sp += c * wy[i] * wx[j], y index outer (splines are built as (y, x), :455)
__device__ __forceinline__ void field_combine(const Cell<double>& c, const double z[4], const Pair<double> g[4][4],
                                              double& n, double& gx, double& gy) {
    double sp = 0.0;
    sp += z[0] * c.lwy[0] * c.lwx[0];
    sp += z[1] * c.lwy[0] * c.lwx[1];
    sp += z[2] * c.lwy[1] * c.lwx[0];
    sp += z[3] * c.lwy[1] * c.lwx[1];
    n = sp;
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            sx += g[r][q].x * c.wy[r] * c.wx[q];
            sy += g[r][q].y * c.wy[r] * c.wx[q];
        }
    }
    gx = sx;
    gy = sy;
}

template <typename G>
__device__ __forceinline__ void n_gradient(const FieldDev<double>& F, G& gather, bool active, double x, double y,
                                           double& n, double& gx, double& gy) {
    Cell<double> c;
    ex::field_locate(F, x, y, c);
    double z[4];
    Pair<double> g[4][4];
    gather.fetch(F, c, active, z, g);
    ex::field_combine(c, z, g, n, gx, gy);
}

// ---------------------------------------------------------------- advancement (:300-365)
__device__ __forceinline__ void adv_first(const Ray<double>& r, double step, double& fx, double& fy) {
    fx = r.x + r.ux * step;
    fy = r.y + r.uy * step;
}
__device__ __forceinline__ void adv_second(const Ray<double>& r, const Consts<double>& k, double& fx, double& fy) {   // :330
    const double d = dot2(r.gx, r.gy, r.ux, r.uy);
    fx = (r.x + r.ux * k.step) + (r.gx - d * r.ux) * k.step2 / (2.0 * r.n);
    fy = (r.y + r.uy * k.step) + (r.gy - d * r.uy) * k.step2 / (2.0 * r.n);
}
// returns the reference's flag: true == curvature NOT negligible (quirk Q14)
__device__ __forceinline__ bool adv_curv(const Ray<double>& r, const Consts<double>& k, double& fx, double& fy) {
    const double d = dot2(r.gx, r.gy, r.ux, r.uy);
    const double curv = norm2(r.gx - d * r.ux, r.gy - d * r.uy) / r.n;
    if (curv < 1.4901161193847656e-08) {   // GOLD_TOL (:355)
        adv_first(r, k.step, fx, fy);
        return false;
    }
    const double dc = curv * k.step;
    if (r.gx * r.uy - r.gy * r.ux > 0) {   // np.cross (:360)
        const double t = r.th - dc;
        fx = r.x + (r.uy - sin_(t)) / curv;             // r.uy == sin(theta), r.ux == cos(theta): same function, same bits
        fy = r.y + (cos_(t) - r.ux) / curv;
    } else {
        const double t = r.th + dc;
        fx = r.x + (sin_(t) - r.uy) / curv;
        fy = r.y + (-cos_(t) + r.ux) / curv;
    }
    return true;
}

// ---------------------------------------------------------------- angle determination (:370-407)
__device__ __forceinline__ double ang_rk2(const Ray<double>& r, double step, double fn, double fgx, double fgy) {
    const double k1 = step * (r.ux * r.gy - r.uy * r.gx) / r.n;
    const double t = r.th + k1;
    const double k2 = step * (cos_(t) * fgy - sin_(t) * fgx) / fn;
    return r.th + (k1 + k2) / 2.0;
}
__device__ __forceinline__ double ang_cost(const Ray<double>& r, double step, double fgx, double fgy) {
    return ::atan2(r.n * r.uy + impulse(r.gy, fgy, step), r.n * r.ux + impulse(r.gx, fgx, step));
}

// ---------------------------------------------------------------- golden() (:175-199), filtered
// The reference evaluates func(c) and func(d) afresh in each of its 37 iterations and keeps [a, d] when
// func(c) < func(d), else [c, b]; the returned midpoint depends only on that sequence of outcomes.  Here every
// comparison is first attempted with FAST cost values that carry a bound on their distance from what the reference's
// arithmetic gives at the current points; when the two fast values are further apart than the bounds allow, the
// outcome is decided.  Otherwise (per lane, rare: the costs must agree to ~1e-14 of their terms) both costs are
// recomputed at the current points in the reference's arithmetic (`exact`) and compared like the reference does.
// Either way the outcome is the reference's, so a, b, c, d -- advanced with the reference's own unfused bracket
// arithmetic -- are its bits, and so is the returned midpoint.
//
// One new point per iteration: of the two points of the new bracket one is the previous iteration's better point
// ("survivor"; equal up to a few ulps of t to the point the reference would evaluate again), the other is the
// survivor moved by h_j = pi*GR^(j+4) towards the kept side.
//   Phase A (the first kGoldRotIters iterations -- with the default, all 37 of a search that starts from width pi):
//     sin/cos of the new point by rotating the survivor's sin/cos through h_j (table rt_golden_rot.h), no sincos
//     evaluation at all.  The tracked angle drifts from the actual point by rounding (bounded by DPHI below), which
//     only loosens the bound; measured on the BASELINE configurations the exact re-evaluation stays rare even in the
//     last iterations (cfg5: 8.4e9 ray-steps/s with 24 rotation iterations, 9.5e9 with all of them).
//   Phase B (whatever remains): sincos_k at the actual point, tight bound, the survivor's value reused with its actual
//     distance |t - t_eval| priced in.
//
// FastSC(s, c, f, g): cost f and g = |e_x| + |e_y| of the residual vector from sin/cos of the evaluation angle.
// Bound of one stored value: (g + l)*(K1 + l) + f*K2, l = LIP*|t - t_eval|, plus K3 once, where
//   K1  = 2 x (absolute error bound of one residual component: reference arithmetic + fast arithmetic),
//   LIP = 2 x (bound on |d e/dt|), K2 = relative error of squaring and adding, K3 = second-order terms.
struct GoldBounds { double K1, LIP, K2, K3; };

#ifndef RTMI_GOLD_ROT_ITERS
#define RTMI_GOLD_ROT_ITERS 48   // <= RT_GOLD_ROT_ENTRIES; from a bracket of width pi the search ends after 37 iterations
#endif
constexpr int kGoldRotIters = RTMI_GOLD_ROT_ITERS;
__device__ const double kGoldRot[2 * RT_GOLD_ROT_ENTRIES] = {RT_GOLD_ROT_VALUES};
constexpr double kU = 1.1102230246251565e-16;   // 2^-53

// fast_a: the phase-A evaluation (may be cheaper and looser than fast_sc by at most EA in one residual component)
template <typename FastA, typename FastSC, typename Exact>
__device__ __forceinline__ double golden_filtered(FastA fast_a, double EA, FastSC fast_sc, Exact exact, const GoldBounds& B,
                                                  double th, double s0, double c0) {
    const double GR = kGoldRatio, tol = M<double>::gold_tol;
    double a = th - kHalfPi, b = th + kHalfPi;
    double c = b - (b - a) * GR, d = a + (b - a) * GR;
    int it = 0;
    {   // ---- phase A.  X: survivor, Y: the newer point; x_is_c: X is the lower point c
        // tracked angle vs actual point: rotation arithmetic (<= 6u each, table constants included) and the bracket
        // arithmetic's roundings (<= ulp(t) per update, t up to |theta| + pi/2)
        const double dphi = kU * (8.0 + kGoldRotIters * (6.0 + 2.0 * (__builtin_fabs(th) + 2.0)));
        const double lA = B.LIP * dphi;
        const double K1A = B.K1 + lA + 2.0 * EA, K3A = 4.0 * K1A * K1A;
        const double sk = RT_GOLD_KAPPA_SIN, ck = RT_GOLD_KAPPA_COS;
        double sX = fma_(-c0, sk, s0 * ck), cX = fma_(s0, sk, c0 * ck);   // theta0 - kappa
        double sY = fma_(c0, sk, s0 * ck), cY = fma_(-s0, sk, c0 * ck);   // theta0 + kappa
        double fX, gX, fY, gY;
        fast_a(sX, cX, fX, gX);
        fast_a(sY, cY, fY, gY);
        bool x_is_c = true;
        for (; it < kGoldRotIters && __builtin_fabs(c - d) > tol; ++it) {
            const double bound = fma_(gX + gY, K1A, fma_(fX + fY, B.K2, K3A));
            bool xless = fX < fY;
            bool lt = xless == x_is_c;
            if (!(__builtin_fabs(fX - fY) > bound)) {        // also taken when anything is NaN
                lt = exact(c) < exact(d);
                xless = lt == x_is_c;
            }
            if (lt) b = d; else a = c;
            c = b - (b - a) * GR;
            d = a + (b - a) * GR;
            if (!xless) { fX = fY; gX = gY; sX = sY; cX = cY; }
            const double ch = kGoldRot[2 * it + 1];
            const double sh = lt ? -kGoldRot[2 * it] : kGoldRot[2 * it];
            sY = fma_(cX, sh, sX * ch);
            cY = fma_(-sX, sh, cX * ch);
            fast_a(sY, cY, fY, gY);
            x_is_c = !lt;
        }
    }
    // ---- phase B (only if the search is still running)
    double fc = 0, gc = 0, fd = 0, gd = 0, tc = c, td = d;
    if (it < kGoldMaxIter && __builtin_fabs(c - d) > tol) {
        double s, cs;
        sincos_k(c, &s, &cs); fast_sc(s, cs, fc, gc);
        sincos_k(d, &s, &cs); fast_sc(s, cs, fd, gd);
    }
    for (; it < kGoldMaxIter && __builtin_fabs(c - d) > tol; ++it) {
        const double lc = B.LIP * __builtin_fabs(c - tc), ld = B.LIP * __builtin_fabs(d - td);
        const double bound = fma_(gc + lc, B.K1 + lc, fc * B.K2) + fma_(gd + ld, B.K1 + ld, fd * B.K2) + B.K3;
        bool lt = fc < fd;
        if (!(__builtin_fabs(fc - fd) > bound)) lt = exact(c) < exact(d);
        double s, cs;
        if (lt) {   // keep [a, d]: the new d is the old c (up to rounding), the new c is fresh
            b = d;
            c = b - (b - a) * GR;
            d = a + (b - a) * GR;
            fd = fc; gd = gc; td = tc;
            sincos_k(c, &s, &cs); fast_sc(s, cs, fc, gc); tc = c;
        } else {    // keep [c, b]
            a = c;
            c = b - (b - a) * GR;
            d = a + (b - a) * GR;
            fc = fd; gc = gd; tc = td;
            sincos_k(d, &s, &cs); fast_sc(s, cs, fd, gd); td = d;
        }
    }
    return (b + a) / 2.0;
}

// The two cost functions in the reference's arithmetic -- the rare path of golden_filtered, kept out of line.
__device__ __attribute__((noinline)) double exact_cost_iso(double t, double fn, double px, double py, double ix, double iy) {
    return sq(fn * cos_(t) - px - ix) + sq(fn * sin_(t) - py - iy);                                   // (:595, :697)
}
__device__ __attribute__((noinline)) double exact_cost_aniso(double t, double fn, double gam, double g2, double mix, double miy,
                                                            double cgx, double cgy, double fgx, double fgy, double step) {
    const double s = sin_(t), c = cos_(t);                                                            // (:728, :761)
    const double a = aniso(s, c, gam);
    const double mx = moment(fn, a, g2, c, -sq(s));
    const double my = moment(fn, a, g2, s, sq(c));
    return sq(mx - mix - impulse(cgx, a * fgx, step)) + sq(my - miy - impulse(cgy, a * fgy, step));
}

// isotropic cost (:595, :697): (n' cos t - n u_x - I_x)^2 + (n' sin t - n u_y - I_y)^2
__device__ __forceinline__ double ang_golden_iso(const Ray<double>& r, double step, double fn, double fgx, double fgy) {
    const double px = r.n * r.ux, py = r.n * r.uy;
    const double ix = impulse(r.gx, fgx, step), iy = impulse(r.gy, fgy, step);
    auto exact = [=](double t) { return exact_cost_iso(t, fn, px, py, ix, iy); };
    auto fast_sc = [=](double s, double c, double& f, double& g) {
        const double e0 = fma_(fn, c, -px) - ix, e1 = fma_(fn, s, -py) - iy;
        f = fma_(e1, e1, e0 * e0);
        g = __builtin_fabs(e0) + __builtin_fabs(e1);
    };
    // one residual component: |n' cos t| carries the sincos error (libm <= 0.55 ulp, sincos_k taken as <= 2 ulp) and a
    // product rounding, the two subtractions round at most u * (|n'| + |p| + |I|) each -- in both arithmetics
    const double afn = __builtin_fabs(fn);
    const double mag = afn + __builtin_fmax(__builtin_fabs(px), __builtin_fabs(py)) + __builtin_fmax(__builtin_fabs(ix), __builtin_fabs(iy));
    const double e1 = 12.0 * kU * mag;
    GoldBounds B;
    B.K1 = 2.0 * e1; B.LIP = 2.0 * afn; B.K2 = 8.0 * kU; B.K3 = 16.0 * e1 * e1;
    return golden_filtered(fast_sc, 0.0, fast_sc, exact, B, r.th, r.uy, r.ux);
}

// anisotropic cost (:725-728 / :758-761); the step functions read the module-global gamma (quirk Q12)
__device__ __forceinline__ double ang_golden_aniso(const Ray<double>& r, const Consts<double>& k, double fn, double fgx,
                                                   double fgy) {
    const double gam = k.gamma_s, g2 = k.g2m1_s, step = k.step;
    const double c0 = aniso(r.uy, r.ux, gam);
    const double mix = moment(r.n, c0, g2, r.ux, -sq(r.uy));
    const double miy = moment(r.n, c0, g2, r.uy, sq(r.ux));
    const double cgx = r.coef * r.gx, cgy = r.coef * r.gy;
    auto exact = [=](double t) { return exact_cost_aniso(t, fn, gam, g2, mix, miy, cgx, cgy, fgx, fgy, step); };
    // Fast form: with a^2 = gamma^2 s^2 + c^2 the reference's brackets are 1 - s^2 (gamma^2-1)/a^2 = 1/a^2 and
    // 1 + c^2 (gamma^2-1)/a^2 = gamma^2/a^2, so m_x = n' c / a and m_y = n' gamma^2 s / a: no cancellation, one rsq.
    const double hstep = step * 0.5, gam2 = gam * gam, fng2 = fn * gam2;
    auto fast_n = [=](double s, double c, double& f, double& g, const int newton) {
        const double gs = gam * s;
        const double q2 = fma_(gs, gs, c * c);
        double ra = __builtin_amdgcn_rsq(q2);                       // 1/a: hardware estimate + Newton steps
        ra = fma_(ra * 0.5, fma_(-q2 * ra, ra, 1.0), ra);
        if (newton > 1) ra = fma_(ra * 0.5, fma_(-q2 * ra, ra, 1.0), ra);
        const double a = q2 * ra;
        const double e0 = fn * c * ra - mix - fma_(a, fgx, cgx) * hstep;
        const double e1 = fng2 * s * ra - miy - fma_(a, fgy, cgy) * hstep;
        f = fma_(e1, e1, e0 * e0);
        g = __builtin_fabs(e0) + __builtin_fabs(e1);
    };
    auto fast_sc = [=](double s, double c, double& f, double& g) { fast_n(s, c, f, g, 2); };   // < 2 ulp in 1/a
    // phase A: one Newton step on the >= 20-bit hardware estimate leaves 1/a within 2^-38 (3.7e-12) relative
    auto fast_a = [=](double s, double c, double& f, double& g) { fast_n(s, c, f, g, 1); };
    // Error of one residual component (u = 2^-53, relative errors unless stated).  Reference arithmetic: s, c <= 1.1u
    // (libm); a <= 4.1u; n'*a*o0 <= 7.2u; w = o1*(gamma^2-1)/a^2 <= 14.4u with |w| <= wmax = |gamma^2-1| / min(1, gamma^2);
    // the bracket F = 1 + w cancels, |dF| <= 14.4u*wmax + u*|F|, |F| <= 1 + wmax; so |dm| <= |n'|*gmax*(14.4u*wmax +
    // 9.2u*(1 + wmax)).  Fast arithmetic (s, c taken as <= 2 ulp = 4u; q2 <= 11u; 1/a <= 7.5u): |dm| <= 14.5u*|m| with
    // |m| <= |n'|*max(1, gamma^2)/amin.  The impulse and the two subtractions add, in both arithmetics together,
    // 4u*(|m| + |m_i| + |I|) + 22u*step*(|coef g| + gmax*|g'|).
    const double afn = __builtin_fabs(fn), g2a = __builtin_fabs(g2);
    const double amin = __builtin_fmin(1.0, gam), gmax = __builtin_fmax(1.0, gam), g2max = __builtin_fmax(1.0, gam2);
    const double wmax = g2a / __builtin_fmin(1.0, gam2);
    const double mom = afn * (gmax * (15.0 * wmax + 10.0 * (1.0 + wmax)) + 16.0 * g2max / amin);
    const double gsum = __builtin_fabs(fgx) + __builtin_fabs(fgy);
    const double oth = 4.0 * (__builtin_fabs(mix) + __builtin_fabs(miy) + afn * gmax * (1.0 + wmax)) +
                       22.0 * step * (__builtin_fabs(cgx) + __builtin_fabs(cgy) + gmax * gsum);
    const double e1 = kU * (mom + oth);
    GoldBounds B;
    B.K1 = 2.0 * e1;
    // |a'| <= |gamma^2-1| / (2 amin): |dm/dt| <= n' max(1, gamma^2) (1/amin + |gamma^2-1| / (2 amin^3)); the impulse
    // varies with a(t) only
    B.LIP = 3.0 * (afn * g2max * (1.0 / amin + g2a / (2.0 * amin * amin * amin)) + step * g2a / amin * gsum);
    B.K2 = 8.0 * kU; B.K3 = 16.0 * e1 * e1;
    const double EA = 4.0e-12 * (afn * g2max / amin + step * gmax * gsum);     // |m| * 3.7e-12, and the impulse's a(t)
    return golden_filtered(fast_a, EA, fast_sc, exact, B, r.th, r.uy, r.ux);
}

// ---------------------------------------------------------------- opN around the field lookup
template <int METHOD>
__device__ __forceinline__ bool op_advance(const Consts<double>& k, const Ray<double>& r, double& fx, double& fy) {
    if constexpr (METHOD == 3 || METHOD == 4 || METHOD == 5 || METHOD == 10) return ex::adv_curv(r, k, fx, fy);
    else { ex::adv_second(r, k, fx, fy); return true; }
}
template <int METHOD>
__device__ __forceinline__ double op_angle(const Consts<double>& k, const Ray<double>& r, bool flag, double fn, double fgx,
                                           double fgy) {
    if constexpr (METHOD == 3) return flag ? ex::ang_rk2(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 4) return flag ? ex::ang_cost(r, k.step, fgx, fgy) : r.th;
    else if constexpr (METHOD == 5) return flag ? ex::ang_golden_iso(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 9) return ex::ang_golden_iso(r, k.step, fn, fgx, fgy);
    else if constexpr (METHOD == 10) return flag ? ex::ang_golden_aniso(r, k, fn, fgx, fgy) : r.th;
    else return ex::ang_golden_aniso(r, k, fn, fgx, fgy);   // 11
}

// store_update_results (:783-790) + the row bookkeeping of the loop body (:871-875)
__device__ __forceinline__ void store_update(const Consts<double>& k, Ray<double>& r, double fx, double fy, double fth,
                                             double fn, double fgx, double fgy) {
    const double dist = norm2(r.x - fx, r.y - fy);
    r.dsim += dist;
    r.dreal += k.step;
    const double c = cos_(fth), s = sin_(fth);
    const double coef = aniso(s, c, k.gamma);
    r.mx = moment(fn, coef, k.g2m1, c, -sq(s));
    r.my = moment(fn, coef, k.g2m1, s, sq(c));
    r.hx0 = r.hx1; r.hy0 = r.hy1; r.hx1 = r.x; r.hy1 = r.y;
    r.x = fx; r.y = fy; r.th = fth; r.n = fn; r.gx = fgx; r.gy = fgy;
    r.ux = c; r.uy = s; r.coef = coef;
    const double nray = coef * fn;                              // (:873)
    r.tt = r.tt + dist * (r.nray + nray) / 2.0;                 // (:874) quirk Q6
    r.nray = nray;
}

// derived quantities from the stored state: the same functions of the same bits as when they were first formed
__device__ __forceinline__ void derive(const Consts<double>& k, Ray<double>& r) {
    r.ux = cos_(r.th); r.uy = sin_(r.th);
    r.coef = aniso(r.uy, r.ux, k.gamma);
    r.nray = r.coef * r.n;
    r.rn = 0;
    r.mx = moment(r.n, r.coef, k.g2m1, r.ux, -sq(r.uy));
    r.my = moment(r.n, r.coef, k.g2m1, r.uy, sq(r.ux));
}

template <int METHOD, typename G>
__device__ __forceinline__ bool ray_step(const FieldDev<double>& F, const Consts<double>& k, G& gather, bool active,
                                         Ray<double>& r) {
    double fx, fy, fn, fgx, fgy;
    const bool flag = ex::op_advance<METHOD>(k, r, fx, fy);
    ex::n_gradient(F, gather, active, fx, fy, fn, fgx, fgy);
    const double fth = ex::op_angle<METHOD>(k, r, flag, fn, fgx, fgy);
    ex::store_update(k, r, fx, fy, fth, fn, fgx, fgy);
    return !outside(k, r);
}

}  // namespace ex
}  // namespace rt
