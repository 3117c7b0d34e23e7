// rt_device.h -- device-side numerics of the ray propagation path (gfx950).
// One ray per lane; everything here is per-lane scalar code on T = double | float.
// Reference lines are RT_bench.py file:line of neyuru/RayTracing.
#pragma once
#include <hip/hip_runtime.h>

namespace rt {

// The library is compiled with -ffp-contract=off: every fusion below is an explicit fma(), so a value is
// computed with the same roundings wherever the function is inlined (results do not depend on how a run is
// cut into launches or which kernel variant executes it).
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------- math traits
template <typename T> struct M;
template <> struct M<double> {
    static __device__ __forceinline__ double sqrt_(double x) { return __dsqrt_rn(x); }
    static __device__ __forceinline__ void sincos_(double x, double* s, double* c) { ::sincos(x, s, c); }
    static __device__ __forceinline__ double atan2_(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double abs_(double x) { return ::fabs(x); }
    static __device__ __forceinline__ double exp_(double x) { return ::exp(x); }
    // x[i] of numpy.linspace: i*step + a with separate roundings (contraction is off), so knots equal the host's bit for bit
    static __device__ __forceinline__ double lin(int i, double h, double a) { return (double)i * h + a; }
    static constexpr double gold_tol = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON), RT_bench.py:66
};
template <> struct M<float> {
    static __device__ __forceinline__ float sqrt_(float x) { return __fsqrt_rn(x); }
    static __device__ __forceinline__ void sincos_(float x, float* s, float* c) { ::sincosf(x, s, c); }
    static __device__ __forceinline__ float atan2_(float y, float x) { return ::atan2f(y, x); }
    static __device__ __forceinline__ float abs_(float x) { return ::fabsf(x); }
    static __device__ __forceinline__ float exp_(float x) { return ::expf(x); }
    static __device__ __forceinline__ float lin(int i, float h, float a) { return (float)i * h + a; }
    static constexpr float gold_tol = 3.4526698300124393e-04f;  // sqrt(FLT_EPSILON): the fp32 analogue
};

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
};

template <typename T> __device__ __forceinline__ T axis_at(int i, int q, T a, T h, T b) {
    return i >= q - 1 ? b : M<T>::lin(i, h, a);
}

// fpbisp's argument clamp (quirk Q4) and interval search, on a linspace axis: returns j with
// x[j] <= v < x[j+1], j in [0, q-2] (v == x[q-1] gives q-2).
template <typename T> __device__ __forceinline__ int locate(T& v, int q, T a, T h, T b, T inv_h) {
    v = v < a ? a : v;
    v = v > b ? b : v;
    int j = (int)((v - a) * inv_h);
    j = j < 0 ? 0 : (j > q - 2 ? q - 2 : j);
    if (axis_at(j, q, a, h, b) > v) {
        j = j > 0 ? j - 1 : 0;
    } else if (j < q - 2 && axis_at(j + 1, q, a, h, b) <= v) {
        j = j + 1;
    }
    return j;
}

// cubic interpolating knot vector of FITPACK regrid(s=0): t[l] = x[0] (l<=3), x[l-2] (4<=l<=q-1), x[q-1] (l>=q)
template <typename T> __device__ __forceinline__ T knot3(int l, int q, T a, T h, T b) {
    return l <= 3 ? a : (l >= q ? b : M<T>::lin(l - 2, h, a));
}

// FITPACK fpbspl, k = 3: the four non-zero cubic B-splines on [t[l], t[l+1]) at v.
template <typename T> __device__ __forceinline__ void bspl3(T v, int l, int q, T a, T h, T b, T w[4]) {
    const T tm2 = knot3(l - 2, q, a, h, b), tm1 = knot3(l - 1, q, a, h, b), t0 = knot3(l, q, a, h, b);
    const T t1 = knot3(l + 1, q, a, h, b), t2 = knot3(l + 2, q, a, h, b), t3 = knot3(l + 3, q, a, h, b);
    // j = 1
    T f = T(1) / (t1 - t0);
    T h0 = f * (t1 - v), h1 = f * (v - t0);
    // j = 2
    f = h0 / (t1 - tm1);
    T g0 = f * (t1 - v), g1 = f * (v - tm1);
    f = h1 / (t2 - t0);
    g1 = fma_(f, t2 - v, g1);
    T g2 = f * (v - t0);
    // j = 3
    f = g0 / (t1 - tm2);
    w[0] = f * (t1 - v);
    w[1] = f * (v - tm2);
    f = g1 / (t2 - tm1);
    w[1] = fma_(f, t2 - v, w[1]);
    w[2] = f * (v - tm1);
    f = g2 / (t3 - t0);
    w[2] = fma_(f, t3 - v, w[2]);
    w[3] = f * (v - t0);
}

// n_gradient(vector, grd, z) (:141-156): bilinear n, bicubic dn/dx and dn/dy at (x, y).
// 36 coefficients are gathered per call: 2x2 of zn and a 4x4 window of interleaved pairs.
template <typename T>
__device__ __forceinline__ void n_gradient(const FieldDev<T>& F, T x, T y, T& n, T& gx, T& gy) {
    const int jx = locate(x, F.qx, F.ax, F.hx, F.bx, F.inv_hx);
    const int jy = locate(y, F.qy, F.ay, F.hy, F.by, F.inv_hy);
    // ---- gather (issued first; the basis evaluation below needs no memory)
    int lx = jx + 2; lx = lx < 3 ? 3 : (lx > F.qx - 1 ? F.qx - 1 : lx);
    int ly = jy + 2; ly = ly < 3 ? 3 : (ly > F.qy - 1 ? F.qy - 1 : ly);
    const T* zp = F.zn + (size_t)jy * F.qx + jx;
    const T z00 = zp[0], z01 = zp[1], z10 = zp[F.qx], z11 = zp[F.qx + 1];
    const T* gp = F.g + ((size_t)(ly - 3) * F.qx + (lx - 3)) * 2;
    T c[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) c[r][q] = gp[(size_t)r * F.qx * 2 + q];
    }
    // ---- bilinear n: fpbspl k=1 on knots x[jx], x[jx+1]
    {
        const T xa = axis_at(jx, F.qx, F.ax, F.hx, F.bx), xb = axis_at(jx + 1, F.qx, F.ax, F.hx, F.bx);
        const T ya = axis_at(jy, F.qy, F.ay, F.hy, F.by), yb = axis_at(jy + 1, F.qy, F.ay, F.hy, F.by);
        const T fx = T(1) / (xb - xa), fy = T(1) / (yb - ya);
        const T wx0 = fx * (xb - x), wx1 = fx * (x - xa);
        const T wy0 = fy * (yb - y), wy1 = fy * (y - ya);
        n = fma_(z11 * wy1, wx1, fma_(z10 * wy1, wx0, fma_(z01 * wy0, wx1, (z00 * wy0) * wx0)));
    }
    // ---- bicubic gradient: shared basis for both components (same knots)
    T wx[4], wy[4];
    bspl3(x, lx, F.qx, F.ax, F.hx, F.bx, wx);
    bspl3(y, ly, F.qy, F.ay, F.hy, F.by, wy);
    T sx = 0, sy = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        T rx = 0, ry = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            rx = fma_(c[r][2 * q], wx[q], rx);
            ry = fma_(c[r][2 * q + 1], wx[q], ry);
        }
        sx = fma_(rx, wy[r], sx);
        sy = fma_(ry, wy[r], sy);
    }
    gx = sx;
    gy = sy;
}

// ---------------------------------------------------------------- per-ray state
template <typename T> struct Ray {
    T x, y, th, n, gx, gy;   // position, angle, index, gradient at the current point
    T ux, uy, coef, nray;    // derived: unit tangent, anisotropy(theta,gamma), coef*n
    T dsim, dreal, tt;       // simulated / expected arclength, traveltime
    T mx, my;                // momenta of the current row (output only)
    T hx0, hy0, hx1, hy1;    // op7: the two positions before (x,y), oldest first (VECTOR_LIST, Q11)
};

template <typename T> struct Consts {
    T step, step2;           // DELTA_S and pow(DELTA_S, 2) (host-computed, :330)
    T gamma, g2m1;           // trazar's gamma, gamma**2-1 (:230)
    T gamma_s, g2m1_s;       // module-global gamma of op10/op11 (Q12)
    T box[4];
};

// anisotropy(theta, gamma) (:118-119) from sin/cos
template <typename T> __device__ __forceinline__ T aniso(T s, T c, T gamma) {
    const T gs = gamma * s;
    return M<T>::sqrt_(fma_(gs, gs, c * c));
}
// moment() (:217-230) given coef = anisotropy(theta, gamma)
template <typename T> __device__ __forceinline__ T moment(T n, T coef, T g2m1, T o0, T o1) {
    return n * coef * o0 * (T(1) + o1 * g2m1 / (coef * coef));
}
template <typename T> __device__ __forceinline__ T impulse(T a, T b, T step) { return step * (a + b) / T(2); }

// ---- advancement (:300-365)
template <typename T> __device__ __forceinline__ void adv_first(const Ray<T>& r, T step, T& fx, T& fy) {
    fx = fma_(r.ux, step, r.x);
    fy = fma_(r.uy, step, r.y);
}
template <typename T> __device__ __forceinline__ void adv_second(const Ray<T>& r, const Consts<T>& k, T& fx, T& fy) {
    const T d = fma_(r.gy, r.uy, r.gx * r.ux);  // np.dot on 2 elements rounds exactly like this
    const T s = k.step2 / (T(2) * r.n);
    fx = fma_(fma_(-d, r.ux, r.gx), s, fma_(r.ux, k.step, r.x));
    fy = fma_(fma_(-d, r.uy, r.gy), s, fma_(r.uy, k.step, r.y));
}
// returns the reference's flag: true == curvature NOT negligible (quirk Q14)
template <typename T> __device__ __forceinline__ bool adv_curv(const Ray<T>& r, const Consts<T>& k, T& fx, T& fy) {
    const T d = fma_(r.gy, r.uy, r.gx * r.ux);
    const T vx = fma_(-d, r.ux, r.gx), vy = fma_(-d, r.uy, r.gy);
    const T curv = M<T>::sqrt_(fma_(vy, vy, vx * vx)) / r.n;
    if (curv < T(1.4901161193847656e-08)) {  // GOLD_TOL (:355), same constant in both precisions
        adv_first(r, k.step, fx, fy);
        return false;
    }
    const T dc = curv * k.step;
    const T sgn = (r.gx * r.uy - r.gy * r.ux > T(0)) ? T(-1) : T(1);  // np.cross (:360), two rounded products
    T s2, c2;
    M<T>::sincos_(sgn < T(0) ? r.th - dc : r.th + dc, &s2, &c2);
    // (:361) [sin th - sin(th-dc), cos(th-dc) - cos th]/curv ; (:363) [sin(th+dc) - sin th, -cos(th+dc) + cos th]/curv
    fx = r.x + (sgn < T(0) ? (r.uy - s2) : (s2 - r.uy)) / curv;
    fy = r.y + (sgn < T(0) ? (c2 - r.ux) : (-c2 + r.ux)) / curv;
    return true;
}

// ---- angle determination (:370-407)
template <typename T> __device__ __forceinline__ T ang_rk2(const Ray<T>& r, T step, T fn, T fgx, T fgy) {
    const T k1 = step * fma_(r.ux, r.gy, -(r.uy * r.gx)) / r.n;
    T s2, c2;
    M<T>::sincos_(r.th + k1, &s2, &c2);
    const T k2 = step * fma_(c2, fgy, -(s2 * fgx)) / fn;
    return r.th + (k1 + k2) / T(2);
}
template <typename T> __device__ __forceinline__ T ang_cost(const Ray<T>& r, T step, T fgx, T fgy) {
    return M<T>::atan2_(fma_(r.n, r.uy, impulse(r.gy, fgy, step)), fma_(r.n, r.ux, impulse(r.gx, fgx, step)));
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
    return (b + a) / T(2);
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
    return golden<T>(cost, r.th - T(kHalfPi), r.th + T(kHalfPi));
}
template <typename T>
__device__ __forceinline__ T ang_golden_aniso(const Ray<T>& r, const Consts<T>& k, T fn, T fgx, T fgy) {
    // (:725-728 / :758-761); the step functions read the module-global gamma (Q12)
    const T c0 = aniso(r.uy, r.ux, k.gamma_s);
    const T mix = moment(r.n, c0, k.g2m1_s, r.ux, -(r.uy * r.uy));
    const T miy = moment(r.n, c0, k.g2m1_s, r.uy, r.ux * r.ux);
    const T cgx = r.coef * r.gx, cgy = r.coef * r.gy;
    const T gam = k.gamma_s, g2 = k.g2m1_s, step = k.step;
    auto cost = [=](T t) {
        T s, c;
        M<T>::sincos_(t, &s, &c);
        const T a = aniso(s, c, gam);
        const T ex = moment(fn, a, g2, c, -(s * s)) - mix - impulse(cgx, a * fgx, step);
        const T ey = moment(fn, a, g2, s, c * c) - miy - impulse(cgy, a * fgy, step);
        return fma_(ey, ey, ex * ex);
    };
    return golden<T>(cost, r.th - T(kHalfPi), r.th + T(kHalfPi));
}

// ---- opN (:469-764): final position / angle / n / gradient of one DELTA_S step
template <typename T, int METHOD>
__device__ __forceinline__ void op_step(const FieldDev<T>& F, const Consts<T>& k, const Ray<T>& r, T& fx, T& fy,
                                        T& fth, T& fn, T& fgx, T& fgy) {
    bool flag = true;
    if constexpr (METHOD == 1 || METHOD == 2) adv_first(r, k.step, fx, fy);
    else if constexpr (METHOD == 3 || METHOD == 4 || METHOD == 5 || METHOD == 10) flag = adv_curv(r, k, fx, fy);
    else adv_second(r, k, fx, fy);
    n_gradient(F, fx, fy, fn, fgx, fgy);
    if constexpr (METHOD == 1 || METHOD == 8) fth = ang_cost(r, k.step, fgx, fgy);
    else if constexpr (METHOD == 2 || METHOD == 6) fth = ang_rk2(r, k.step, fn, fgx, fgy);
    else if constexpr (METHOD == 3) fth = flag ? ang_rk2(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 4) fth = flag ? ang_cost(r, k.step, fgx, fgy) : r.th;
    else if constexpr (METHOD == 5) fth = flag ? ang_golden_iso(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 9) fth = ang_golden_iso(r, k.step, fn, fgx, fgy);
    else if constexpr (METHOD == 10) fth = flag ? ang_golden_aniso(r, k, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 11) fth = ang_golden_aniso(r, k, fn, fgx, fgy);
    else {  // 7: finite_diff (:370-372) over [P0, P1, P2, P3] = [h0, h1, (x,y), f]
        const T vx = T(11) * fx - T(18) * r.x + T(9) * r.hx1 - T(2) * r.hx0;
        const T vy = T(11) * fy - T(18) * r.y + T(9) * r.hy1 - T(2) * r.hy0;
        fth = M<T>::atan2_(vy, vx);
    }
}

// store_update_results (:783-790) + the row bookkeeping of the loop body (:871-875)
template <typename T>
__device__ __forceinline__ void store_update(const Consts<T>& k, Ray<T>& r, T fx, T fy, T fth, T fn, T fgx, T fgy) {
    const T dx = r.x - fx, dy = r.y - fy;
    const T dist = M<T>::sqrt_(fma_(dy, dy, dx * dx));  // np.linalg.norm on 2 elements
    r.dsim += dist;
    r.dreal += k.step;  // quirk Q16: accumulated, not i*step
    T s, c;
    M<T>::sincos_(fth, &s, &c);
    const T coef = aniso(s, c, k.gamma);
    r.mx = moment(fn, coef, k.g2m1, c, -(s * s));
    r.my = moment(fn, coef, k.g2m1, s, c * c);
    r.hx0 = r.hx1; r.hy0 = r.hy1; r.hx1 = r.x; r.hy1 = r.y;
    r.x = fx; r.y = fy; r.th = fth; r.n = fn; r.gx = fgx; r.gy = fgy;
    r.ux = c; r.uy = s; r.coef = coef;
    const T nray = coef * fn;                           // (:873)
    r.tt = r.tt + dist * (r.nray + nray) / T(2);        // (:874) quirk Q6
    r.nray = nray;
}

// derived quantities from the stored state (used when a launch (re)loads a ray from HBM)
template <typename T> __device__ __forceinline__ void derive(const Consts<T>& k, Ray<T>& r) {
    M<T>::sincos_(r.th, &r.uy, &r.ux);
    r.coef = aniso(r.uy, r.ux, k.gamma);
    r.nray = r.coef * r.n;
    r.mx = moment(r.n, r.coef, k.g2m1, r.ux, -(r.uy * r.uy));
    r.my = moment(r.n, r.coef, k.g2m1, r.uy, r.ux * r.ux);
}

template <typename T> __device__ __forceinline__ bool outside(const Consts<T>& k, const Ray<T>& r) {  // (:878)
    return r.x > k.box[1] || r.x < k.box[0] || r.y > k.box[3] || r.y < k.box[2];
}

// One iteration of trazar's loop for row index i (the row being produced).  For op7 rows 1 and 2 are
// the bootstrap steps (:833-864): first- and second-order backward differences and no boundary test.
template <typename T, int METHOD>
__device__ __forceinline__ bool ray_step(const FieldDev<T>& F, const Consts<T>& k, Ray<T>& r, int i) {
    T fx, fy, fth, fn, fgx, fgy;
    if (METHOD == 7 && i <= 2) {
        adv_second(r, k, fx, fy);
        n_gradient(F, fx, fy, fn, fgx, fgy);
        T vx, vy;
        if (i == 1) { vx = fx - r.x; vy = fy - r.y; }                                          // (:843)
        else { vx = T(3) * fx - T(4) * r.x + r.hx1; vy = T(3) * fy - T(4) * r.y + r.hy1; }     // (:856)
        fth = M<T>::atan2_(vy, vx);
        store_update(k, r, fx, fy, fth, fn, fgx, fgy);
        return true;  // still alive (no boundary test in the bootstrap)
    }
    op_step<T, METHOD>(F, k, r, fx, fy, fth, fn, fgx, fgy);
    store_update(k, r, fx, fy, fth, fn, fgx, fgy);
    return !outside(k, r);
}

}  // namespace rt
