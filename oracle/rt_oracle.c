/*
 * rt_oracle.c -- CPU restatement of neyuru/RayTracing's per-ray shooting-method
 * propagation path (RT_bench.py: trazar -> opN -> n_gradient).
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (raytracing_amd/ -> librtmi.so -> HIP kernels) never calls it.
 *
 * Parity status: PINNED.  Every function below is checked against golden
 * vectors captured from the unmodified reference (oracle/gen_golden.py, run in
 * the build container where /root/reference exists) -- tests/test_oracle_golden.py.
 *
 * Arithmetic notes (why some lines look odd).  The reference is scalar numpy
 * on 2-element arrays.  To stay within an ulp of it this restatement keeps the
 * reference's evaluation order and reproduces three library behaviours that
 * were measured against numpy 2.2.6 / scipy 1.15.3 in the build container:
 *   - np.dot / np.linalg.norm on 2-vectors round as fma(a1,b1, a0*b0)
 *     (OpenBLAS ddot tail loop) -> DOT2 below;
 *   - `x**2` on a numpy float64 scalar is libm pow(x, 2.0), not x*x -> see SQ() and libm_square();
 *   - np.sin/np.cos/np.sqrt equal glibc's sin()/cos()/sqrt(); np.arctan2 and np.exp do not
 *     (<= 1 ulp apart), so atan2-based methods match to ~1e-15, not bit for bit.
 *     glibc's sincos() is NOT bit-identical to its sin() and cos() (0.14 % of arguments differ, FMA
 *     build), and gcc merges sin(x), cos(x) pairs into sincos(x): the Makefile therefore passes
 *     -fno-builtin-sin -fno-builtin-cos so that every call below is the separate libm function the
 *     reference calls (tools/check_libm_sincos.cpp counts the difference).
 * Build with -ffp-contract=off so the compiler adds no fusions of its own.
 *
 * Third-party arithmetic restated here (absent from /root/reference):
 *   scipy FITPACK (scipy 1.15.3 in the build container; README of the reference
 *   lists 1.12.0; no lock file pins it): regrid with s=0 (interpolating
 *   not-a-knot knots, fpregr.f/fpgrre.f), bispev/fpbisp/fpbspl evaluation;
 *   numpy: linspace, meshgrid, gradient(edge_order=2).
 *   Reference call sites: RT_bench.py:429-432 (grid), 450 (gradient),
 *   455-457 (fits), 153-155 (evaluation).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_API __attribute__((visibility("default")))

/* ---- constants: RT_bench.py:59-66 ---------------------------------------- */
static const double THCK_PARAM = 0.005;                 /* :59 */
static const double GOLD_TOL = 1.4901161193847656e-08;  /* sqrt(eps) = 2^-26, :66 */
#define DELTA_G (M_PI / 2)                               /* :64 */
static double gold_ratio(void) { return (sqrt(5.0) - 1.0) / 2.0; } /* :65 */

/* numpy's scalar x**2 is libm pow(x, 2.0).  On glibc 2.35 that differs from the rounded product x*x by one ulp for
 * ~0.09 % of arguments.  gcc folds pow(x, 2.0) into x*x, so by default SQ() IS the rounded product: a known deviation
 * from the reference, kept because the device path reproduces x*x bit for bit and not glibc's pow.  Built with
 * -DRTO_SQ_POW (oracle/librt_oracle_pow.so, `make pow`) SQ() is the reference-faithful libm pow through a volatile
 * exponent; tests/test_oracle_golden.py runs the trajectory fixtures with both and records what the deviation costs
 * (none of the reference's fixtures moves by more than 3e-17 because of it).  The one place where the reference's pow is
 * honoured in both builds is DELTA_S**2 (:330), computed once per call through libm_square(). */
#ifdef RTO_SQ_POW
static double SQ(double x) { volatile double two = 2.0; return pow(x, two); }
#else
static inline double SQ(double x) { return x * x; }
#endif
static double libm_square(double x) { volatile double two = 2.0; return pow(x, two); }
static inline double DOT2(double a0, double a1, double b0, double b1) {
    return fma(a1, b1, a0 * b0);                                    /* np.dot, 2 elements */
}
static inline double NORM2(double a0, double a1) { return sqrt(DOT2(a0, a1, a0, a1)); }

/* np.exp on float64 ARRAYS as numpy 2.2.6 evaluates it on an AVX512_SKX machine (the build container; RT_bench.py:107
 * calls it on the meshgrid): numpy dispatches to Intel SVML's __svml_exp8_ha (numpy/_core/src/umath/svml, BSD-3), not
 * to libm -- the two differ in the last bit for 4.5 % of arguments.  Restated from the published routine's main path:
 * N = floor(x*log2(e)*16)/16 (an fma rounded toward zero onto a 2^-4 grid), r = x - N*ln2 (hi/lo), a degree-6
 * polynomial in three interleaved pairs, 2^(j/16) from a 16-entry table with a correction term, scaled by 2^floor(N).
 * tools/check_np_exp.py compares it with np.exp on 2.6e7 arguments: 0 mismatches.  |x| >= 707.7 takes SVML's scalar
 * fallback, restated here as libm exp: there the interface field is sqrt(2) or 1 to the last bit whatever the
 * exponential's last bits are (1 + e rounds to e or to 1). */
static double np_exp(double x) {
    static const double T16[16] = {0x1.0000000000000p+0, 0x1.0b5586cf9890fp+0, 0x1.172b83c7d517bp+0, 0x1.2387a6e756238p+0,
        0x1.306fe0a31b715p+0, 0x1.3dea64c123422p+0, 0x1.4bfdad5362a27p+0, 0x1.5ab07dd485429p+0, 0x1.6a09e667f3bcdp+0,
        0x1.7a11473eb0187p+0, 0x1.8ace5422aa0dbp+0, 0x1.9c49182a3f090p+0, 0x1.ae89f995ad3adp+0, 0x1.c199bdd85529cp+0,
        0x1.d5818dcfba487p+0, 0x1.ea4afa2a490dap+0};                                  /* 2^(j/16), correctly rounded */
    static const double TL16[16] = {0x0.0p+0, 0x1.79aa65d837b6dp-54, -0x1.01b15eaa59348p-55, 0x1.68efde3a8a894p-54,
        0x1.34d754db0abb6p-55, 0x1.59f48a72a4c6dp-55, 0x1.690cebb7aafb0p-56, 0x1.063e1e21c5409p-54, -0x1.3b3efbf5e2228p-54,
        -0x1.b32dcb94da51dp-56, 0x1.db72fc1f0eab4p-55, 0x1.1affc2b91ce27p-56, 0x1.c1a7792cb3387p-55, 0x1.36eae30af0cb3p-56,
        0x1.4a385a63d07a7p-56, -0x1.ff7128fd391f0p-55};                                /* (2^(j/16) - T16[j]) / T16[j] */
    const double L2E = 0x1.71547652b82fep+0, LN2H = 0x1.62e42fefa39efp-1, LN2L = 0x1.abc9e3b39803fp-56;
    const double A = 0x1.7411836940c04p-10, B = 0x1.1101cbbc265c0p-7, C = 0x1.55557242d68fep-5, D = 0x1.5555553939732p-3,
                 E = 0x1.000000000d008p-1, F = 0x1.fffffffffff70p-1;
    if (!(fabs(x) < 0x1.61da04cbafe44p+9)) return exp(x);
    /* fma(x, L2E, 1.5*2^48 + 1023) rounded toward zero, minus the shifter: floor of the EXACT product on the 1/16 grid */
    double p = x * L2E, e = fma(x, L2E, -p);
    double f16 = floor(p * 16.0);
    if (f16 == p * 16.0 && e < 0) f16 -= 1.0;
    double N = f16 / 16.0;
    int j = (int)((long long)f16 & 15);
    double r = fma(-N, LN2H, x);
    r = fma(-N, LN2L, r);
    double r2 = r * r;
    double P1 = fma(A, r, B), P2 = fma(C, r, D), P3 = fma(E, r, F);
    double q = fma(r2, P1, P2);
    q = fma(r2, q, P3);
    double t = fma(q, r, TL16[j]);
    t = fma(T16[j], t, T16[j]);
    return ldexp(t, (int)floor(N));
}

/* np.arctan2 on float64 as numpy 2.2.6 evaluates it on an AVX512_SKX machine: Intel SVML's __svml_atan28_ha (differs from
 * libm's atan2 in the last bit for 7 % of arguments).  Main path restated: the ratio |y|/|x| against 0.4375, 0.6875, 1.1875,
 * 2.4375 picks a base point c in {0, 0.5, 1, 1.5, inf}; q = (|y| - c|x|) / (|x| + c|y|) (inf: -|x|/|y|) by a reciprocal -- the
 * VRCP14PD instruction's 14-bit estimate and two Newton steps -- as a hi + lo pair; an odd polynomial of degree 21 in two
 * interleaved chains of q^4; atan(c) hi/lo, pi hi/lo and the signs put back.  VRCP14PD itself is a table
 * (raytracing_amd/csrc/rt_rcp14_table.h, captured from the instruction by tools/gen_rcp14_table.c).  tools/check_np_atan2.py:
 * 0 mismatches against np.arctan2 on 1.6e7 argument pairs.  Operands with exponents below 2^-1020 or above 2^993 (zeros and
 * infinities among them) take SVML's scalar fall-back, restated as libm atan2. */
#include "../raytracing_amd/csrc/rt_rcp14_table.h"
static uint16_t g_rcp14[65536];
static int g_rcp14_ready = 0;
static void rcp14_init(void) {
    static const uint64_t words[RT_RCP14_WORDS] = {RT_RCP14_DELTAS};
    uint16_t v = RT_RCP14_T0;
    for (int k = 0; k < 65536; k++) { v = (uint16_t)(v - ((words[k >> 5] >> (2 * (k & 31))) & 3)); g_rcp14[k] = v; }
    g_rcp14_ready = 1;
}
static double vrcp14pd(double x) {
    uint64_t xb, rb;
    memcpy(&xb, &x, 8);
    const uint64_t e = (xb >> 52) & 0x7ff, m = xb & 0xfffffffffffffull;
    rb = m == 0 ? (0x7fe - e) << 52 : ((0x7fd - e) << 52) | ((uint64_t)g_rcp14[m >> 36] << 36);
    double r;
    memcpy(&r, &rb, 8);
    return r;
}
static double np_arctan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    uint64_t b;
    memcpy(&b, &ax, 8); const int32_t ix = (int32_t)((uint32_t)(b >> 32) - 0x80300000u);
    memcpy(&b, &ay, 8); const int32_t iy = (int32_t)((uint32_t)(b >> 32) - 0x80300000u);
    if (ix >= (int32_t)0xfdd00000u || iy >= (int32_t)0xfdd00000u) return atan2(y, x);
    const int k5 = 0.4375 * ax < ay, k1 = 0.6875 * ax < ay, k2 = 1.1875 * ax < ay, k3 = 2.4375 * ax < ay;
    const double c = k2 ? (k3 ? 1.0 : 1.5) : (k1 ? 1.0 : 0.5);
    const double ahi = k2 ? (k3 ? 0x1.921fb54442d18p+0 : 0x1.f730bd281f69bp-1) : (k1 ? 0x1.921fb54442d18p-1 : 0x1.dac670561bb4fp-2);
    const double alo = k2 ? (k3 ? 0x1.1a62633145c07p-54 : 0x1.007887af0cbbdp-56) : (k1 ? 0x1.1a62633145c07p-55 : 0x1.a2b7f222f65e2p-56);
    double den = k3 ? 0.0 : ax, num = k3 ? 0.0 : ay;
    if (k5) { den = fma(c, ay, den); num = fma(-c, ax, num); }
    double r = vrcp14pd(den);
    double e = fma(-r, den, 1.0);
    r = fma(e, r, r);
    e = fma(-r, den, 1.0);
    r = fma(e, r, r);
    const double q = num * r, q2 = q * q;
    const double res = fma(-q, den, num);
    const double q4 = q2 * q2;
    double ql = res * r;
    if (k5) ql = ql + alo;
    double A = fma(0x1.be4fbe6733718p-7, q4, 0x1.6ad5558fe19c9p-5), B = fma(-0x1.04cd71f92185ep-5, q4, -0x1.a9e755ca13d23p-5);
    A = fma(q4, A, 0x1.e12f1edf7c393p-5);  B = fma(q4, B, -0x1.1108d326c68edp-4);
    A = fma(q4, A, 0x1.3b132b731e73ap-4);  B = fma(q4, B, -0x1.745d119677a4fp-4);
    A = fma(q4, A, 0x1.c71c719f99f96p-4);  B = fma(q4, B, -0x1.2492492441a21p-3);
    A = fma(q4, A, 0x1.9999999998f43p-3);  B = fma(q4, B, -0x1.5555555555552p-2);
    double t = fma(q2, A, B) * q2;
    const int xneg = x < 0.0;
    if (xneg) ql = ql - 0x1.1a64000000000p-53;                  /* pi's low part, with the sign the final flip undoes */
    t = fma(q, t, ql);
    double s = q + t;
    if (k5) s = s + ahi;
    if (xneg) s = -s + 0x1.921fb54442d18p+1;
    return copysign(s, y);                                       /* s >= 0 here; SVML ORs y's sign bit in */
}
RTO_API void rto_np_arctan2_many(const double *y, const double *x, double *o, long n) {
    if (!g_rcp14_ready) rcp14_init();
    for (long i = 0; i < n; i++) o[i] = np_arctan2(y[i], x[i]);
}
RTO_API void rto_np_exp_many(const double *x, double *y, long n) { for (long i = 0; i < n; i++) y[i] = np_exp(x[i]); }

/* ---- scenario fields: RT_bench.py:106-116 -------------------------------- */
enum { SC_INTERFACE = 1, SC_FISHEYE = 2, SC_VERT = 3, SC_ANISO = 4 };

static double scenario_n(int sc, double a, double b) {
    switch (sc) {
    case SC_INTERFACE: /* :107; np.exp on the meshgrid array -> numpy's vector loop, see np_exp() */
        return sqrt(2.0) - (sqrt(2.0) - 1.0) / (1.0 + np_exp(-b / THCK_PARAM));
    case SC_FISHEYE:   /* :111, np.power(a,2) on arrays is an exact square */
        return 1.0 / (1.0 + a * a + b * b);
    default: {         /* :115-116, scenario 4 reuses it (:1579) */
        double v = 18.0 + 2.0 * b;
        return 1.0 / v;
    }
    }
}

/* anisotropy(theta, gamma): RT_bench.py:118-119 */
static inline double anisotropy_sc(double s, double c, double gamma) {
    return sqrt(SQ(gamma * s) + SQ(c));
}

/* ---- field ---------------------------------------------------------------- */
typedef struct {
    int qx, qy;
    double *x, *y;       /* grid axes (genZ :429) */
    double *Z;           /* [qy][qx] samples == bilinear coefficients (:455) */
    double *cdy, *cdx;   /* bicubic coefficients of GradX (d/dy) and GradY (d/dx) (:456-457) */
    double *tx3, *ty3;   /* cubic knot vectors, len qx+4 / qy+4 */
    double *tx1, *ty1;   /* linear knot vectors, len qx+2 / qy+2 */
} rto_field;

/* numpy.linspace(a, b, n): y = arange(n)*step + a, y[-1] = b */
static void np_linspace(double a, double b, int n, double *out) {
    double step = (b - a) / (double)(n - 1);
    for (int i = 0; i < n; i++) out[i] = (double)i * step + a;
    out[n - 1] = b;
}

/* numpy.gradient(f, dx, edge_order=2) along one axis of a [n0][n1] array.
 * axis 0: stride n1, axis 1: stride 1. */
static void np_gradient_axis(const double *f, int n0, int n1, int axis, double dx, double *out) {
    int n = axis == 0 ? n0 : n1, m = axis == 0 ? n1 : n0;
    long sa = axis == 0 ? n1 : 1, sm = axis == 0 ? 1 : n1;
    double a0 = -1.5 / dx, b0 = 2.0 / dx, c0 = -0.5 / dx;
    double a1 = 0.5 / dx, b1 = -2.0 / dx, c1 = 1.5 / dx;
    for (int j = 0; j < m; j++) {
        const double *p = f + j * sm;
        double *o = out + j * sm;
        for (int i = 1; i < n - 1; i++) o[i * sa] = (p[(i + 1) * sa] - p[(i - 1) * sa]) / (2.0 * dx);
        o[0] = a0 * p[0] + b0 * p[sa] + c0 * p[2 * sa];
        o[(n - 1) * sa] = a1 * p[(n - 3) * sa] + b1 * p[(n - 2) * sa] + c1 * p[(n - 1) * sa];
    }
}

/* FITPACK fpbspl: the k+1 non-zero B-splines of degree k at x, t[l] <= x < t[l+1]
 * (l is the 0-based index of the interval's left knot). */
static void fpbspl(const double *t, int k, double x, int l, double *h) {
    double hh[6];
    h[0] = 1.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 0; i < j; i++) {
            int li = l + 1 + i, lj = li - j;
            double f = hh[i] / (t[li] - t[lj]);
            h[i] = h[i] + f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}

/* interval search of fpbisp: clamp, then walk from l=k while x >= t[l+1]. n = knot count. */
static int fp_interval(const double *t, int n, int k, double *x) {
    double tb = t[k], te = t[n - k - 1];
    if (*x < tb) *x = tb;
    if (*x > te) *x = te;
    int l = k;
    while (*x >= t[l + 1] && l != n - k - 2) l++;
    return l;
}

/* interpolating knots of regrid(s=0): fpregr.f -- k odd: t = [x0 x (k+1), x[(k+1)/2 .. m-1-(k+1)/2], x[m-1] x (k+1)] */
static void interp_knots(const double *x, int m, int k, double *t) {
    int n = m + k + 1, k3 = k / 2;
    for (int i = 0; i <= k; i++) { t[i] = x[0]; t[n - 1 - i] = x[m - 1]; }
    for (int i = k + 1, j = k3 + 1; i < n - k - 1; i++, j++) t[i] = x[j];
}

/* Solve the collocation system  sum_j c_j B_j(x_i) = d_i  (cubic, not-a-knot knots)
 * for `nrhs` right-hand sides laid out with stride (rs between i, cs between rhs).
 * The matrix has bandwidth 2/2; Gaussian elimination without pivoting (totally
 * positive matrix).  Equals FITPACK's Givens-QR result to rounding (1.3e-15
 * measured in SURVEY.md 8.a4 and pinned by tests/golden/field_*.npz). */
static void collocation_solve(const double *x, int m, const double *t, double *d, long rs, long cs, int nrhs) {
    /* band storage A[i][0..4] = columns i-2..i+2 */
    double (*A)[5] = calloc((size_t)m, sizeof *A);
    for (int i = 0; i < m; i++) {
        double xv = x[i], h[4];
        int l = fp_interval(t, m + 4, 3, &xv);
        fpbspl(t, 3, xv, l, h);
        for (int q = 0; q < 4; q++) {
            int j = l - 3 + q, off = j - i + 2;
            if (off >= 0 && off < 5) A[i][off] = h[q];
            /* entries outside the band are exactly zero (B_{l}(t_l)=0) */
        }
    }
    /* LU in place, no pivoting; L multipliers stored in A[i][0..1] */
    for (int i = 0; i < m; i++) {
        for (int r = i + 1; r <= i + 2 && r < m; r++) {
            int off = i - r + 2; /* column i in row r */
            double mlt = A[r][off] / A[i][2];
            A[r][off] = mlt;
            for (int q = 1; q <= 2; q++) {
                if (off + q < 5) A[r][off + q] -= mlt * A[i][2 + q];
            }
        }
    }
    for (int r0 = 0; r0 < nrhs; r0++) {
        double *b = d + r0 * cs;
        for (int i = 0; i < m; i++) { /* forward */
            double v = b[i * rs];
            if (i >= 1) v -= A[i][1] * b[(i - 1) * rs];
            if (i >= 2) v -= A[i][0] * b[(i - 2) * rs];
            b[i * rs] = v;
        }
        for (int i = m - 1; i >= 0; i--) { /* back */
            double v = b[i * rs];
            if (i + 1 < m) v -= A[i][3] * b[(i + 1) * rs];
            if (i + 2 < m) v -= A[i][4] * b[(i + 2) * rs];
            b[i * rs] = v / A[i][2];
        }
    }
    free(A);
}

/* FITPACK regrid with s = 0 (fpregr.f -> one call of fpgrre.f with p = -1): the interpolating spline's coefficients as the
 * least-squares solution of (spy) c (spx)' = z by Givens rotations WITHOUT square-root-free tricks -- each data row of
 * the observation matrix is rotated into the band triangle (fpgivs/fprota), the same rotations go over the right-hand
 * sides, first along FITPACK's x (the rows of our [qy][qx] arrays, :456 passes (y, x, Z)), then along its y; two
 * back substitutions (fpback) finish.  Restated operation by operation; the Fortran is compiled without FMA in scipy's
 * wheels, this file with -ffp-contract=off.  Result: the coefficient arrays are scipy's get_coeffs() bit for bit
 * (tests/test_oracle_golden.py: field_*.npz blocks compared with array_equal).  The banded LU above solves the same system
 * and lands 1.3e-15 away; it is kept as rto_set_field_solver(1) to measure exactly that. */
typedef struct { int m; int *nr; double (*cs)[4][2]; double (*a)[4]; } fp_axis;   /* rotations + band triangle of one axis */
static void fpgivs(double piv, double *ww, double *c, double *s) {
    double store = fabs(piv), dd;
    if (store >= *ww) { double q = *ww / piv; dd = store * sqrt(1.0 + q * q); }
    else { double q = piv / *ww; dd = *ww * sqrt(1.0 + q * q); }
    *c = *ww / dd; *s = piv / dd; *ww = dd;
}
static inline void fprota(double c, double s, double *a, double *b) {
    double s1 = *a, s2 = *b;
    *b = c * s2 + s * s1;
    *a = c * s1 - s * s2;
}
static fp_axis fp_axis_build(const double *x, int m, const double *t) {
    fp_axis A; A.m = m;
    A.nr = malloc(m * sizeof(int)); A.cs = calloc((size_t)m, sizeof *A.cs); A.a = calloc((size_t)m, sizeof *A.a);
    int l = 3, number = 0;                        /* fpgrre: l = kx1 (1-based), the interval of x(it) */
    for (int it = 0; it < m; it++) {
        double h[5];
        while (!(x[it] < t[l + 1] || l == m - 1)) { l++; number++; }
        fpbspl(t, 3, x[it], l, h);
        A.nr[it] = number;
        int irot = number - 1;
        for (int i = 0; i < 4; i++) {
            irot++;
            double piv = h[i];
            A.cs[it][i][0] = A.cs[it][i][1] = 0.0;           /* (0, 0): no rotation (piv == 0) */
            if (piv == 0.0) continue;
            double c, s;
            fpgivs(piv, &A.a[irot][0], &c, &s);
            A.cs[it][i][0] = c; A.cs[it][i][1] = s;
            for (int j = i + 1, i2 = 1; j < 4; j++, i2++) fprota(c, s, &h[j], &A.a[irot][i2]);
        }
    }
    return A;
}
static void fp_axis_free(fp_axis *A) { free(A->nr); free(A->cs); free(A->a); }
/* fpback with bandwidth 4 on n values of stride es */
static void fpback4(const double (*a)[4], double *z, int n, long es) {
    z[(n - 1) * es] = z[(n - 1) * es] / a[n - 1][0];
    for (int i = n - 2, j = 2; i >= 0; i--, j++) {
        double store = z[i * es];
        int i1 = j <= 3 ? j - 1 : 3;
        for (int l = 1; l <= i1; l++) store = store - z[(i + l) * es] * a[i][l];
        z[i * es] = store / a[i][0];
    }
}
static void regrid_interp(const double *xs, int qx, const double *tx, const double *ys, int qy, const double *ty, double *d) {
    fp_axis AY = fp_axis_build(ys, qy, ty), AX = fp_axis_build(xs, qx, tx);   /* FITPACK's x is our y (rows) */
    size_t nz = (size_t)qx * qy;
    double *q = calloc(nz, sizeof(double)), *right = malloc((qx > qy ? qx : qy) * sizeof(double));
    for (int it = 0; it < qy; it++) {               /* rows of z into the triangle of FITPACK-x; q = g [qy][qx] */
        memcpy(right, d + (size_t)it * qx, qx * sizeof(double));
        for (int i = 0; i < 4; i++) {
            double c = AY.cs[it][i][0], s = AY.cs[it][i][1];
            if (c == 0.0 && s == 0.0) continue;
            double *qr = q + (size_t)(AY.nr[it] + i) * qx;
            for (int j = 0; j < qx; j++) fprota(c, s, &right[j], &qr[j]);
        }
    }
    memset(d, 0, nz * sizeof(double));               /* c = 0; columns of g into the triangle of FITPACK-y */
    for (int it = 0; it < qx; it++) {
        for (int j = 0; j < qy; j++) right[j] = q[(size_t)j * qx + it];
        for (int i = 0; i < 4; i++) {
            double c = AX.cs[it][i][0], s = AX.cs[it][i][1];
            if (c == 0.0 && s == 0.0) continue;
            int col = AX.nr[it] + i;
            for (int j = 0; j < qy; j++) fprota(c, s, &right[j], &d[(size_t)j * qx + col]);
        }
    }
    for (int i = 0; i < qy; i++) fpback4((const double (*)[4])AX.a, d + (size_t)i * qx, qx, 1);      /* (ry) c1 = h */
    for (int j = 0; j < qx; j++) fpback4((const double (*)[4])AY.a, d + j, qy, qx);                    /* c (rx)' = c1 */
    free(q); free(right); fp_axis_free(&AY); fp_axis_free(&AX);
}
static int g_field_solver = 0;    /* 0: FITPACK's Givens QR (the reference's bits); 1: banded LU (round 1-2's solver) */
RTO_API void rto_set_field_solver(int lu) { g_field_solver = lu; }

RTO_API void rto_field_free(rto_field *f) {
    if (!f) return;
    free(f->x); free(f->y); free(f->Z); free(f->cdy); free(f->cdx);
    free(f->tx3); free(f->ty3); free(f->tx1); free(f->ty1); free(f);
}

/* interpolacion(): RT_bench.py:435-464 (Hessian splines :459-462 are never read -> skipped) */
RTO_API rto_field *rto_field_from_samples(const double *x, int qx, const double *y, int qy,
                                          const double *Z, double delta) {
    if (!g_rcp14_ready) rcp14_init();           /* np_arctan2's table, before any (threaded) trace */
    rto_field *f = calloc(1, sizeof *f);
    size_t nz = (size_t)qx * qy;
    f->qx = qx; f->qy = qy;
    f->x = malloc(qx * sizeof(double)); memcpy(f->x, x, qx * sizeof(double));
    f->y = malloc(qy * sizeof(double)); memcpy(f->y, y, qy * sizeof(double));
    f->Z = malloc(nz * sizeof(double)); memcpy(f->Z, Z, nz * sizeof(double));
    f->cdy = malloc(nz * sizeof(double)); f->cdx = malloc(nz * sizeof(double));
    /* :450  GradX = d/d(axis0) = d/dy, GradY = d/d(axis1) = d/dx, both scaled by DELTA (quirk Q1) */
    np_gradient_axis(Z, qy, qx, 0, delta, f->cdy);
    np_gradient_axis(Z, qy, qx, 1, delta, f->cdx);
    f->tx3 = malloc((qx + 4) * sizeof(double)); f->ty3 = malloc((qy + 4) * sizeof(double));
    f->tx1 = malloc((qx + 2) * sizeof(double)); f->ty1 = malloc((qy + 2) * sizeof(double));
    interp_knots(x, qx, 3, f->tx3); interp_knots(y, qy, 3, f->ty3);
    interp_knots(x, qx, 1, f->tx1); interp_knots(y, qy, 1, f->ty1);
    /* :456-457 separable interpolation: along axis 1 (x) for every row, then axis 0 (y) */
    if (g_field_solver == 0) {
        regrid_interp(x, qx, f->tx3, y, qy, f->ty3, f->cdy);
        regrid_interp(x, qx, f->tx3, y, qy, f->ty3, f->cdx);
    } else {
        collocation_solve(x, qx, f->tx3, f->cdy, 1, qx, qy);
        collocation_solve(y, qy, f->ty3, f->cdy, qx, 1, qx);
        collocation_solve(x, qx, f->tx3, f->cdx, 1, qx, qy);
        collocation_solve(y, qy, f->ty3, f->cdx, qx, 1, qx);
    }
    return f;
}

/* genZ(): RT_bench.py:412-433, then interpolacion() */
RTO_API rto_field *rto_field_build(int scenario, double xi, double xs, double yi, double ys, double delta) {
    int qx = (int)((xs - xi + 6) / delta + 1);  /* :426 */
    int qy = (int)((ys - yi + 6) / delta + 1);  /* :427 */
    double *x = malloc(qx * sizeof(double)), *y = malloc(qy * sizeof(double));
    np_linspace(xi - 3, xs + 3, qx, x);         /* :429 */
    np_linspace(yi - 3, ys + 3, qy, y);
    double *Z = malloc((size_t)qx * qy * sizeof(double));
    for (int i = 0; i < qy; i++)
        for (int j = 0; j < qx; j++) Z[(size_t)i * qx + j] = scenario_n(scenario, x[j], y[i]); /* :430-432 */
    rto_field *f = rto_field_from_samples(x, qx, y, qy, Z, delta);
    free(x); free(y); free(Z);
    return f;
}

RTO_API void rto_field_dims(const rto_field *f, int *qx, int *qy) { *qx = f->qx; *qy = f->qy; }
RTO_API void rto_field_get(const rto_field *f, double *x, double *y, double *Z, double *cdy, double *cdx) {
    size_t nz = (size_t)f->qx * f->qy;
    if (x) memcpy(x, f->x, f->qx * sizeof(double));
    if (y) memcpy(y, f->y, f->qy * sizeof(double));
    if (Z) memcpy(Z, f->Z, nz * sizeof(double));
    if (cdy) memcpy(cdy, f->cdy, nz * sizeof(double));
    if (cdx) memcpy(cdx, f->cdx, nz * sizeof(double));
}

/* fpbisp single-point evaluation; first spline argument is y (splines are built (y, x, .), :455) */
static double bisp_eval(const double *ty, int ny, const double *tx, int nx, const double *c, int k,
                        double yv, double xv) {
    double wy[4], wx[4];
    int ly = fp_interval(ty, ny, k, &yv); fpbspl(ty, k, yv, ly, wy);
    int lx = fp_interval(tx, nx, k, &xv); fpbspl(tx, k, xv, lx, wx);
    int ncx = nx - k - 1;
    double sp = 0.0;
    for (int i1 = 0; i1 <= k; i1++)
        for (int j1 = 0; j1 <= k; j1++)
            sp += c[(size_t)(ly - k + i1) * ncx + (lx - k + j1)] * wy[i1] * wx[j1];
    return sp;
}

/* n_gradient(vector, grd, z): RT_bench.py:141-156 -> (n, [dn/dx, dn/dy]) (quirks Q2-Q4) */
RTO_API void rto_n_gradient(const rto_field *f, double x, double y, double *n, double *gx, double *gy) {
    *n = bisp_eval(f->ty1, f->qy + 2, f->tx1, f->qx + 2, f->Z, 1, y, x);     /* :153 */
    *gx = bisp_eval(f->ty3, f->qy + 4, f->tx3, f->qx + 4, f->cdx, 3, y, x);  /* :154 grd[1] */
    *gy = bisp_eval(f->ty3, f->qy + 4, f->tx3, f->qx + 4, f->cdy, 3, y, x);  /* :155 grd[0] */
}

RTO_API void rto_n_gradient_many(const rto_field *f, int npts, const double *x, const double *y,
                                 double *n, double *gx, double *gy) {
    for (int i = 0; i < npts; i++) rto_n_gradient(f, x[i], y[i], n + i, gx + i, gy + i);
}

/* ---- one-step numerics ---------------------------------------------------- */
typedef struct {
    double x, y, theta, n, gx, gy, coef, ux, uy;
    double mx, my;               /* momenta (output only) */
    double dist_sim, dist_real, T, nray;
    double hx[3], hy[3];         /* op7: previous positions, oldest first (VECTOR_LIST, Q11) */
} rto_state;

typedef struct {
    const rto_field *f;
    int method;                  /* 1..11 */
    double gamma;                /* trazar's local gamma (:793) */
    double gamma_step;           /* module-global gamma read by op10/op11 (Q12) */
    double step, step2;          /* step, pow(step,2) */
} rto_ctx;

/* moment(): :217-230 ; o0/o1 as built in moments() :245 */
static double moment_f(double n, double s, double c, double gamma, double o0, double o1) {
    double coef = anisotropy_sc(s, c, gamma);
    double g2m1 = gamma * gamma - 1.0; /* python int arithmetic, exact */
    return n * coef * o0 * (1.0 + o1 * g2m1 / SQ(coef));
}
static void moments_f(double n, double s, double c, double ux, double uy, double gamma, double *mx, double *my) {
    *mx = moment_f(n, s, c, gamma, ux, -SQ(uy));
    *my = moment_f(n, s, c, gamma, uy, SQ(ux));
}

static inline double impulse_t(double a, double b, double step) { return step * (a + b) / 2.0; } /* :214 */

/* golden(): :175-199 (Q13: both cost values recomputed every iteration) */
typedef double (*cost_fn)(double, const void *);
static double golden(cost_fn fn, const void *env, double a, double b) {
    const double GR = gold_ratio();
    double c = b - (b - a) * GR, d = a + (b - a) * GR;
    while (fabs(c - d) > GOLD_TOL) {
        if (fn(c, env) < fn(d, env)) b = d; else a = c;
        c = b - (b - a) * GR;
        d = a + (b - a) * GR;
    }
    return (b + a) / 2.0;
}

typedef struct { double fn, n, ux, uy, ix, iy; } iso_env;
static double cost_iso(double t, const void *e_) { /* :595 / :697 */
    const iso_env *e = e_;
    return SQ(e->fn * cos(t) - e->n * e->ux - e->ix) + SQ(e->fn * sin(t) - e->n * e->uy - e->iy);
}
typedef struct { double fn, gamma, mix, miy, cgx, cgy, fgx, fgy, step; } aniso_env;
static double cost_aniso(double t, const void *e_) { /* :728 / :761 */
    const aniso_env *e = e_;
    double s = sin(t), c = cos(t);
    double a = anisotropy_sc(s, c, e->gamma);
    double mx = moment_f(e->fn, s, c, e->gamma, c, -SQ(s));
    double my = moment_f(e->fn, s, c, e->gamma, s, SQ(c));
    return SQ(mx - e->mix - impulse_t(e->cgx, a * e->fgx, e->step)) +
           SQ(my - e->miy - impulse_t(e->cgy, a * e->fgy, e->step));
}

/* advancement. returns flag for curvature_t (Q14: true == curvature NOT negligible) */
static void adv_first(const rto_state *s, double step, double *fx, double *fy) { /* :312 */
    *fx = s->x + s->ux * step; *fy = s->y + s->uy * step;
}
static void adv_second(const rto_state *s, const rto_ctx *c, double *fx, double *fy) { /* :330 */
    double d = DOT2(s->gx, s->gy, s->ux, s->uy);
    *fx = (s->x + s->ux * c->step) + (s->gx - d * s->ux) * c->step2 / (2.0 * s->n);
    *fy = (s->y + s->uy * c->step) + (s->gy - d * s->uy) * c->step2 / (2.0 * s->n);
}
static int adv_curv(const rto_state *s, const rto_ctx *c, double *fx, double *fy) { /* :335-365 */
    double d = DOT2(s->gx, s->gy, s->ux, s->uy);
    double curv = NORM2(s->gx - d * s->ux, s->gy - d * s->uy) / s->n;
    if (curv < GOLD_TOL) { adv_first(s, c->step, fx, fy); return 0; }
    double dc = curv * c->step, th = s->theta;
    if (s->gx * s->uy - s->gy * s->ux > 0) { /* np.cross 2-D, :360 */
        *fx = s->x + (sin(th) - sin(th - dc)) / curv;
        *fy = s->y + (cos(th - dc) - cos(th)) / curv;
    } else {
        *fx = s->x + (sin(th + dc) - sin(th)) / curv;
        *fy = s->y + (-cos(th + dc) + cos(th)) / curv;
    }
    return 1;
}
/* angle determination */
static double ang_rk2(const rto_state *s, double step, double fn, double fgx, double fgy) { /* :389-391 */
    double k1 = step * (cos(s->theta) * s->gy - sin(s->theta) * s->gx) / s->n;
    double k2 = step * (cos(s->theta + k1) * fgy - sin(s->theta + k1) * fgx) / fn;
    return s->theta + (k1 + k2) / 2.0;
}
static double ang_cost(const rto_state *s, double step, double fgx, double fgy) { /* :407 */
    return np_arctan2(s->n * sin(s->theta) + impulse_t(s->gy, fgy, step),
                      s->n * cos(s->theta) + impulse_t(s->gx, fgx, step));
}
static double ang_golden_iso(const rto_state *s, double step, double fn, double fgx, double fgy) {
    iso_env e = { fn, s->n, s->ux, s->uy, impulse_t(s->gx, fgx, step), impulse_t(s->gy, fgy, step) };
    return golden(cost_iso, &e, s->theta - DELTA_G, s->theta + DELTA_G);
}
static double ang_golden_aniso(const rto_state *s, const rto_ctx *c, double fn, double fgx, double fgy) {
    double sn = sin(s->theta), cs = cos(s->theta), g = c->gamma_step;
    aniso_env e;
    e.fn = fn; e.gamma = g; e.step = c->step;
    e.mix = moment_f(s->n, sn, cs, g, s->ux, -SQ(s->uy)); /* :725 / :758 */
    e.miy = moment_f(s->n, sn, cs, g, s->uy, SQ(s->ux));
    e.cgx = s->coef * s->gx; e.cgy = s->coef * s->gy; e.fgx = fgx; e.fgy = fgy;
    return golden(cost_aniso, &e, s->theta - DELTA_G, s->theta + DELTA_G);
}

/* opN: RT_bench.py:469-764.  out: final position/angle/n/grad */
static void op_step(const rto_ctx *c, const rto_state *s, double *fx, double *fy, double *fth,
                    double *fn, double *fgx, double *fgy) {
    int m = c->method, flag = 1;
    switch (m) {
    case 1: case 2: adv_first(s, c->step, fx, fy); break;
    case 3: case 4: case 5: case 10: flag = adv_curv(s, c, fx, fy); break;
    default: adv_second(s, c, fx, fy); break; /* 6,7,8,9,11 */
    }
    rto_n_gradient(c->f, *fx, *fy, fn, fgx, fgy);
    switch (m) {
    case 1: case 8: *fth = ang_cost(s, c->step, *fgx, *fgy); break;
    case 2: case 6: *fth = ang_rk2(s, c->step, *fn, *fgx, *fgy); break;
    case 3: *fth = flag ? ang_rk2(s, c->step, *fn, *fgx, *fgy) : s->theta; break;
    case 4: *fth = flag ? ang_cost(s, c->step, *fgx, *fgy) : s->theta; break;
    case 5: *fth = flag ? ang_golden_iso(s, c->step, *fn, *fgx, *fgy) : s->theta; break;
    case 9: *fth = ang_golden_iso(s, c->step, *fn, *fgx, *fgy); break;
    case 10: *fth = flag ? ang_golden_aniso(s, c, *fn, *fgx, *fgy) : s->theta; break;
    case 11: *fth = ang_golden_aniso(s, c, *fn, *fgx, *fgy); break;
    case 7: { /* :646-648 finite_diff :370-372 */
        double vx = 11 * *fx - 18 * s->hx[2] + 9 * s->hx[1] - 2 * s->hx[0];
        double vy = 11 * *fy - 18 * s->hy[2] + 9 * s->hy[1] - 2 * s->hy[0];
        *fth = np_arctan2(vy, vx);
        break;
    }
    }
}

/* store_update_results (:783-790) + row bookkeeping (:871-875) */
static void store_update(const rto_ctx *c, rto_state *s, double fx, double fy, double fth, double fn,
                         double fgx, double fgy) {
    double dist = NORM2(s->x - fx, s->y - fy);
    s->dist_sim += dist;
    s->dist_real += c->step;                                   /* Q16 */
    double cs = cos(fth), sn = sin(fth);
    double coef_f = anisotropy_sc(sn, cs, c->gamma);
    moments_f(fn, sn, cs, cs, sn, c->gamma, &s->mx, &s->my);
    /* op7 history shifts (VECTOR_LIST.pop(0) after append) */
    s->hx[0] = s->hx[1]; s->hx[1] = s->hx[2]; s->hx[2] = fx;
    s->hy[0] = s->hy[1]; s->hy[1] = s->hy[2]; s->hy[2] = fy;
    s->x = fx; s->y = fy; s->theta = fth; s->n = fn; s->gx = fgx; s->gy = fgy;
    s->coef = coef_f; s->ux = cs; s->uy = sn;
    double nray = coef_f * fn;                                 /* :873 */
    s->T = s->T + dist * (s->nray + nray) / 2.0;               /* :874, Q6 */
    s->nray = nray;
}

static void state_init(const rto_ctx *c, rto_state *s, double x0, double y0, double th0) { /* :809-826 */
    memset(s, 0, sizeof *s);
    s->x = x0; s->y = y0; s->theta = th0;
    s->ux = cos(th0); s->uy = sin(th0);
    rto_n_gradient(c->f, x0, y0, &s->n, &s->gx, &s->gy);
    s->coef = anisotropy_sc(s->uy, s->ux, c->gamma);
    moments_f(s->n, s->uy, s->ux, s->ux, s->uy, c->gamma, &s->mx, &s->my);
    s->nray = s->coef * s->n;
    s->hx[2] = x0; s->hy[2] = y0;
}

/* Single step from an explicit state; used for the per-method fixtures.
 * st[] = x,y,theta,n,gx,gy,coef ; hist[] = 3 previous positions (x0,y0,x1,y1,x2,y2, oldest first,
 * the newest equals (x,y)); out[] = fx,fy,ftheta,fn,fgx,fgy */
RTO_API void rto_single_step(const rto_field *f, int method, double gamma, double step,
                             const double *st, const double *hist, double *out) {
    rto_ctx c = { f, method, gamma, gamma, step, libm_square(step) };
    rto_state s; memset(&s, 0, sizeof s);
    s.x = st[0]; s.y = st[1]; s.theta = st[2]; s.n = st[3]; s.gx = st[4]; s.gy = st[5]; s.coef = st[6];
    s.ux = cos(s.theta); s.uy = sin(s.theta);
    if (hist) for (int i = 0; i < 3; i++) { s.hx[i] = hist[2 * i]; s.hy[i] = hist[2 * i + 1]; }
    op_step(&c, &s, out + 0, out + 1, out + 2, out + 3, out + 4, out + 5);
}

/* ---- trazar: RT_bench.py:766-948 ------------------------------------------ */
typedef struct {
    int method;
    double gamma, gamma_step;
    double step;
    int max_size;            /* rows incl. row 0 (:797/:799) */
    double box[4];           /* limx_i, limx_s, limy_i, limy_s */
    int record_stride;       /* 0: no trajectory; s>=1: rows i with i % s == 0 plus nothing else */
    long rec_rows;           /* rows allocated in s_ray/n_ray (recorded row r holds step r*stride) */
    int nthreads;            /* OpenMP threads over rays (1 = the reference's sequential loop) */
} rto_params;

static inline void write_row(const rto_params *p, double *s_ray, double *n_ray, long R, long k, long i,
                             const rto_state *s) {
    if (!p->record_stride || i % p->record_stride) return;
    long r = i / p->record_stride;
    if (r >= p->rec_rows) return;
    double *row = s_ray + (size_t)r * 6 * R;
    row[0 * R + k] = s->x; row[1 * R + k] = s->y; row[2 * R + k] = s->mx; row[3 * R + k] = s->my;
    row[4 * R + k] = s->T; row[5 * R + k] = s->theta;
    if (n_ray) n_ray[(size_t)r * R + k] = s->nray;
}

/* final[] is [9][R]: x,y,theta,n,gx,gy,mx,my,T of the last written row; d_ray is [3][R] (:888-890) */
RTO_API long rto_trazar(const rto_field *f, const rto_params *p, int R, const double *x0, const double *y0,
                        const double *th0, double *s_ray, double *n_ray, double *d_ray, double *final) {
    long total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total) num_threads(p->nthreads > 0 ? p->nthreads : 1)
#endif
    for (int k = 0; k < R; k++) {
        rto_ctx c = { f, p->method, p->gamma, p->gamma_step, p->step, libm_square(p->step) };
        rto_state s;
        state_init(&c, &s, x0[k], y0[k], th0[k]);
        write_row(p, s_ray, n_ray, R, k, 0, &s);
        long i = 0, loop_iter = 1;
        double fx, fy, fth, fn, fgx, fgy;
        if (p->method == 7) { /* :833-864 bootstrap, no boundary test */
            loop_iter = 3;
            for (i = 1; i <= 2 && i < p->max_size; i++) {
                adv_second(&s, &c, &fx, &fy);
                rto_n_gradient(f, fx, fy, &fn, &fgx, &fgy);
                double vx, vy;
                if (i == 1) { vx = fx - s.hx[2]; vy = fy - s.hy[2]; }               /* :843 */
                else { vx = 3 * fx - 4 * s.hx[2] + s.hx[1]; vy = 3 * fy - 4 * s.hy[2] + s.hy[1]; } /* :856 */
                fth = np_arctan2(vy, vx);
                store_update(&c, &s, fx, fy, fth, fn, fgx, fgy);
                write_row(p, s_ray, n_ray, R, k, i, &s);
            }
            i = 2;
        }
        for (long it = loop_iter; it < p->max_size; it++) { /* :866 */
            i = it;
            op_step(&c, &s, &fx, &fy, &fth, &fn, &fgx, &fgy);
            store_update(&c, &s, fx, fy, fth, fn, fgx, fgy);
            write_row(p, s_ray, n_ray, R, k, i, &s);
            if (s.x > p->box[1] || s.x < p->box[0] || s.y > p->box[3] || s.y < p->box[2]) break; /* :878 */
        }
        d_ray[0 * (size_t)R + k] = s.dist_real;
        d_ray[1 * (size_t)R + k] = s.dist_sim;
        d_ray[2 * (size_t)R + k] = (double)i;
        if (final) {
            double v[9] = { s.x, s.y, s.theta, s.n, s.gx, s.gy, s.mx, s.my, s.T };
            for (int q = 0; q < 9; q++) final[(size_t)q * R + k] = v[q];
        }
        total += i;
    }
    return total;
}

RTO_API int rto_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
