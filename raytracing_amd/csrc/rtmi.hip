// rtmi.hip -- librtmi.so: HIP kernels + C ABI (include/rtmi.h) for the ray propagation hot path.
// gfx950 only.  Reference lines are RT_bench.py file:line of neyuru/RayTracing.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <exception>
#include <mutex>
#include <numeric>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_internal.h"
#include "rt_device.h"

#define RTMI_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(RTMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ \
                                          ":" + std::to_string(__LINE__) + ")");                        \
    } while (0)
#define ARG_TRY(cond, msg)                              \
    do {                                                \
        if (!(cond)) return fail(RTMI_ERR_ARG, (msg));  \
    } while (0)

// Suprema over t of |d^k/dt^k| (k = 1..4) of the unit momentum curve's components c/a, gamma^2 s/a and of a(t) =
// sqrt(gamma^2 s^2 + c^2), by truncated Taylor arithmetic (order 4) at equally spaced angles, with 10 % on top for what lies
// between the samples.  Bounds for the remainder of the golden-section search's third-order expansion (rt_exact.h, phase T);
// gamma is a property of the batch, so once per batch.
// The curve's features are 1/G wide, G = max(gamma, 1/gamma) (the ellipse's sharp ends), so the sample count grows with G:
// >= 430 samples across a feature -- what gamma = 3, the reference's anisotropy (RT_bench.py:286), has at 8192 and what the
// op10/op11 bit-identity tests validate (gamma 0.05 .. 50, tests/test_gpu_exact.py); neighbouring samples then differ by a
// fraction of the 10 %.  Beyond G = 64 nothing is claimed: the suprema are returned infinite, phase T's `ok` is false and
// every comparison of the search runs the reference's own arithmetic (correct, slower).
static void gold_sup_derivatives(double gamma, double out[8]) {
    const double G = std::max(gamma, 1.0 / gamma);
    if (!(G <= 64.0)) {
        for (int i = 0; i < 8; i++) out[i] = INFINITY;
        return;
    }
    constexpr int K = 4;
    const int N = 8192 * std::max(1, (int)std::ceil(G / 3.0));
    typedef double Jet[K + 1];
    auto mul = [](const Jet a, const Jet b, Jet o) {
        for (int i = 0; i <= K; i++) { double v = 0; for (int j = 0; j <= i; j++) v += a[j] * b[i - j]; o[i] = v; }
    };
    const double fact[K + 1] = {1, 1, 2, 6, 24};
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (int q = 0; q < N; q++) {
        const double t = 2.0 * M_PI * q / N;
        Jet s, c, ss, cc, a2, A, a, mx, my, t1, t2;
        for (int k = 0; k <= K; k++) { s[k] = std::sin(t + k * M_PI / 2) / fact[k]; c[k] = std::cos(t + k * M_PI / 2) / fact[k]; }
        mul(s, s, ss); mul(c, c, cc);
        for (int k = 0; k <= K; k++) a2[k] = gamma * gamma * ss[k] + cc[k];
        for (int k = 0; k <= K; k++) A[k] = 0;
        A[0] = 1.0 / std::sqrt(a2[0]);
        for (int it = 0; it < 5; it++) {          // A <- A (1.5 - 0.5 a2 A^2): one more exact series coefficient per pass
            mul(A, A, t1); mul(a2, t1, t2);
            for (int k = 0; k <= K; k++) t2[k] = -0.5 * t2[k];
            t2[0] += 1.5;
            mul(A, t2, t1);
            for (int k = 0; k <= K; k++) A[k] = t1[k];
        }
        mul(a2, A, a); mul(c, A, mx); mul(s, A, my);
        for (int k = 1; k <= K; k++) {
            const double m = std::max(std::fabs(mx[k]), gamma * gamma * std::fabs(my[k])) * fact[k];
            out[k - 1] = std::max(out[k - 1], m);
            out[3 + k] = std::max(out[3 + k], std::fabs(a[k]) * fact[k]);
        }
    }
    for (int i = 0; i < 8; i++) out[i] *= 1.1;
}

// numpy's scalar x**2 calls libm pow(x, 2.0), which is not always the rounded product x*x (it differs by one ulp for
// ~0.1 % of arguments on glibc 2.35); the exponent is volatile so that no compiler folds the call into a multiply.
// Steep cells (k_polytab): lambda * (the grid's shorter side) >= kSteepRate.  40: the interface scenario's sigmoid (lambda up
// to 36 on a 12 x 28 grid) has a band of them 0.07 wide; the fisheye (lambda <= 2, 9 x 9) and vert_heterogeneous (0.2) have none.
constexpr double kSteepRate = 40.0;
static double libm_square(double x) {
    volatile double two = 2.0;
    return std::pow(x, two);
}

// Does this batch step op1/2/6/7/8 in the reference's own operation order (rt_exact.h)?  fp64 only.  RTMI_ORDER_REFERENCE: all
// five.  RTMI_ORDER_DEFAULT: op7 alone -- it differentiates POSITIONS (11 P3 - 18 P2 + 9 P1 - 2 P0 over 6 DELTA_S), so the
// last bits of the positions enter every new angle at 2e-12 and a fused position update (an ulp or two from the reference's)
// random-walks away from it: 8.4e-9 on recorded rows of the interface scenario, past the 1e-9 the path is held to; only the
// reference's own roundings reproduce its noise.  RTMI_ORDER_FUSED keeps op7 in the fused form (3.3 times faster, fp32 always).
static bool ref_order(const rtmi_params& p) {
    return p.dtype == RTMI_F64 && (p.reference_order == RTMI_ORDER_REFERENCE ||
                                   (p.method == 7 && (p.reference_order == RTMI_ORDER_DEFAULT || p.reference_order == RTMI_ORDER_FAST_FIELD)));
}
// RTMI_ORDER_FAST_FIELD: op7 takes that step on the FAST field lookup (rt::kFastField, rt_device.h): 2.3 times faster, and within
// 1e-9 of the reference everywhere but on rays that graze a sharp interface at its critical angle (one of 16 384 sampled rays of
// the 1 M-ray interface fan: 2.6e-9; profiles/r04_op7_offenders_interface_1m.txt) -- which is why it is not the default.
static bool fast_field_order(const rtmi_params& p) { return p.dtype == RTMI_F64 && p.method == 7 && p.reference_order == RTMI_ORDER_FAST_FIELD; }

// ------------------------------------------------------------------ handles
struct rtmi_field {
    int device = 0;
    int dtype = RTMI_F64;
    int qx = 0, qy = 0;
    double ax = 0, hx = 0, bx = 0, ay = 0, hy = 0, by = 0;
    // fp64 build products (kept for rtmi_field_read and as the source of the packed arrays)
    double *dZ = nullptr, *dCdy = nullptr, *dCdx = nullptr;
    // packed, dtype-typed arrays the trace kernels gather from
    void *zn = nullptr, *g = nullptr;
    void* poly = nullptr;        // [(qy-1)*(qx-1)][rt::kPolyStride] of dtype: one polynomial per cell (rt_polytab.h)
    void* poly_base = nullptr;   // the allocation: the flat-cell map ([flat_pad] of dtype, rt::FieldDev::flat), then the table
    long flat_pad = 0;           // elements from the map's start to the table's
    long flat_cells = 0;         // cells the map marks flat
    double gmax = 0;             // the largest gradient-spline coefficient of the grid in magnitude (k_absmax)
    long steep_cells = 0;        // fp64 fields: cells whose map entry carries a steepness (k_polytab); with neither kind the kernels never look at the map
    double* rdiv = nullptr;      // [qx][24] then [qy][24]: reciprocals of the knot differences fpbspl divides by, knots, differences (rt_exact.h, AxisTab)
    hipStream_t stream = nullptr;
};

// A field's device memory is only valid on the device it was built on; callers that switch devices
// (rtmi_set_device, torch.cuda.set_device) get RTMI_ERR_ARG instead of a cross-device access.
static hipError_t check_device_impl(const rtmi_field* f, const char* who, int* rc) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != f->device) {
        *rc = fail(RTMI_ERR_ARG, std::string(who) + ": the field lives on device " + std::to_string(f->device) +
                                     " but the current device is " + std::to_string(dev));
    }
    return hipSuccess;
}
#define DEVICE_TRY(f, who)                                  \
    do {                                                    \
        int rc_ = RTMI_OK;                                  \
        HIP_TRY(check_device_impl((f), (who), &rc_));       \
        if (rc_) return rc_;                                \
    } while (0)

template <typename T> static rt::FieldDev<T> field_dev(const rtmi_field* f, int exact) {
    rt::FieldDev<T> F;
    F.exact = exact;
    F.zn = (const T*)f->zn;
    F.g = (const T*)f->g;
    F.qx = f->qx; F.qy = f->qy;
    F.ax = (T)f->ax; F.hx = (T)f->hx; F.bx = (T)f->bx;
    F.inv_hx = (T)(1.0 / f->hx);
    F.ay = (T)f->ay; F.hy = (T)f->hy; F.by = (T)f->by;
    F.inv_hy = (T)(1.0 / f->hy);
    F.poly = (const T*)f->poly;
    F.ncx = f->qx - 1;
    F.flat = (f->flat_cells > 0 || f->steep_cells > 0) ? (int)f->flat_pad : 0;
    F.rdx = f->rdiv;
    F.rdy = f->rdiv ? f->rdiv + (size_t)f->qx * rt::ex::kAxisTab : nullptr;
    F.window = 0;
    return F;
}

// ================================================================== field build kernels (fp64)
// np.exp on a float64 array as the reference's numpy evaluates it (RT_bench.py:107 calls it on the meshgrid): on the
// AVX512 machines numpy's wheels dispatch to Intel SVML's __svml_exp8_ha (numpy/_core/src/umath/svml, BSD-3), which is not
// libm's exp in the last bit for 4.5 % of arguments.  The routine's main path restated (oracle/rt_oracle.c np_exp has the
// same text; tools/check_np_exp.py: 0 mismatches against np.exp on 2.6e7 arguments): N = floor(x*log2(e)*16)/16 -- an fma
// rounded toward zero onto a 2^-4 grid -- r = x - N*ln2 in two pieces, a degree-6 polynomial in three interleaved pairs,
// 2^(j/16) from a 16-entry table with its correction term, scaled by 2^floor(N).  |x| >= 707.7 (SVML's scalar fall-back):
// ocml's exp -- there 1 + e rounds to e or to 1 and the interface's n is sqrt(2) or 1 whatever the last bits of e.
__device__ static double np_exp(double x) {
    static const double T16[16] = {0x1.0000000000000p+0, 0x1.0b5586cf9890fp+0, 0x1.172b83c7d517bp+0, 0x1.2387a6e756238p+0,
        0x1.306fe0a31b715p+0, 0x1.3dea64c123422p+0, 0x1.4bfdad5362a27p+0, 0x1.5ab07dd485429p+0, 0x1.6a09e667f3bcdp+0,
        0x1.7a11473eb0187p+0, 0x1.8ace5422aa0dbp+0, 0x1.9c49182a3f090p+0, 0x1.ae89f995ad3adp+0, 0x1.c199bdd85529cp+0,
        0x1.d5818dcfba487p+0, 0x1.ea4afa2a490dap+0};
    static const double TL16[16] = {0x0.0p+0, 0x1.79aa65d837b6dp-54, -0x1.01b15eaa59348p-55, 0x1.68efde3a8a894p-54,
        0x1.34d754db0abb6p-55, 0x1.59f48a72a4c6dp-55, 0x1.690cebb7aafb0p-56, 0x1.063e1e21c5409p-54, -0x1.3b3efbf5e2228p-54,
        -0x1.b32dcb94da51dp-56, 0x1.db72fc1f0eab4p-55, 0x1.1affc2b91ce27p-56, 0x1.c1a7792cb3387p-55, 0x1.36eae30af0cb3p-56,
        0x1.4a385a63d07a7p-56, -0x1.ff7128fd391f0p-55};
    const double L2E = 0x1.71547652b82fep+0, LN2H = 0x1.62e42fefa39efp-1, LN2L = 0x1.abc9e3b39803fp-56;
    const double A = 0x1.7411836940c04p-10, B = 0x1.1101cbbc265c0p-7, C = 0x1.55557242d68fep-5, D = 0x1.5555553939732p-3,
                 E = 0x1.000000000d008p-1, F = 0x1.fffffffffff70p-1;
    if (!(fabs(x) < 0x1.61da04cbafe44p+9)) return exp(x);
    // floor of the EXACT product x*L2E on the 1/16 grid (== the toward-zero fma onto the shifter 1.5*2^48 + 1023)
    const double p = x * L2E, e = __builtin_fma(x, L2E, -p);
    double f16 = floor(p * 16.0);
    if (f16 == p * 16.0 && e < 0) f16 -= 1.0;
    const double N = f16 * 0.0625;
    const int j = (int)((long long)f16 & 15);
    double r = __builtin_fma(-N, LN2H, x);
    r = __builtin_fma(-N, LN2L, r);
    const double r2 = r * r;
    const double P1 = __builtin_fma(A, r, B), P2 = __builtin_fma(C, r, D), P3 = __builtin_fma(E, r, F);
    double q = __builtin_fma(r2, P1, P2);
    q = __builtin_fma(r2, q, P3);
    double t = __builtin_fma(q, r, TL16[j]);
    t = __builtin_fma(T16[j], t, T16[j]);
    return ldexp(t, (int)floor(N));
}

__device__ static double scenario_n(int sc, double a, double b) {
    if (sc == RTMI_INTERFACE)  // :107 (exp overflows to inf for y < -3.55, result sqrt(2): same as numpy)
        return __dsqrt_rn(2.0) - (__dsqrt_rn(2.0) - 1.0) / (1.0 + np_exp(-b / 0.005));
    if (sc == RTMI_FISHEYE)    // :111
        return 1.0 / (1.0 + a * a + b * b);
    return 1.0 / (18.0 + 2.0 * b);  // :115-116
}

__global__ void k_sample(int sc, double* Z, int qx, int qy, double ax, double hx, double bx, double ay, double hy,
                         double by) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= qx || i >= qy) return;
    const double x = rt::axis_at<double>(j, qx, ax, hx, bx), y = rt::axis_at<double>(i, qy, ay, hy, by);
    Z[(size_t)i * qx + j] = scenario_n(sc, x, y);  // meshgrid X[i,j]=x[j], Y[i,j]=y[i] (:430-432)
}

// np.gradient(Z, delta, edge_order=2) along one axis (:450); numpy's evaluation order (contraction is off).
__global__ void k_gradient(const double* Z, double* out, int qx, int qy, int axis, double dx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= qx || i >= qy) return;
    const int n = axis == 0 ? qy : qx, p = axis == 0 ? i : j;
    const long s = axis == 0 ? qx : 1;
    const double* c = Z + (size_t)i * qx + j;
    double r;
    if (p == 0) {
        r = (-1.5 / dx) * c[0] + (2.0 / dx) * c[s] + (-0.5 / dx) * c[2 * s];
    } else if (p == n - 1) {
        r = (0.5 / dx) * c[-2 * s] + (-2.0 / dx) * c[-s] + (1.5 / dx) * c[0];
    } else {
        r = (c[s] - c[-s]) / (2.0 * dx);
    }
    out[(size_t)i * qx + j] = r;
}

// FITPACK regrid with s = 0 (fpregr.f -> fpgrre.f, p = -1), the fit behind RectBivariateSpline (:456-457): the
// interpolating spline's coefficients as the least-squares solution of (spy) c (spx)' = z by Givens rotations.  Each data
// row of an axis' observation matrix is rotated into a band triangle (fpgivs / fprota); the rotations depend on the axis
// alone, so the host works them out once per axis (fp_axis_build) and the device applies them to all right-hand sides at
// once -- first along FITPACK's x (the rows of our [qy][qx] arrays: the reference passes (y, x, Z)), then along its y --
// followed by the two back substitutions (fpback).  Same operations in the same order as the Fortran (scipy's wheels carry
// no FMA; this library is compiled with -ffp-contract=off), so the coefficient arrays are scipy's get_coeffs() bit for bit
// -- where rounds 1-2's banded LU of the same system landed 1.3e-15 away, enough to move interface x op3/4/5 by 2e-7.
//
// k_givens: lane = one right-hand side (stride ls), it = data rows of the axis (stride is).  Data row it touches triangle
// rows nr[it] .. nr[it]+3; nr never decreases and a triangle row is final once nr has passed it, so the four live rows
// are a register window: no read-modify-write of memory at all (out starts as the zero matrix of the Fortran).
__global__ void k_givens(const double* in, double* out, int m, int nlines, long is, long ls, const int* nr, const double* cs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines) return;
    const double* src = in + (size_t)t * ls;
    double* dst = out + (size_t)t * ls;
    double w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    int base = 0;
    for (int it = 0; it < m; it++) {
        for (const int number = nr[it]; base < number; ++base) {
            dst[(size_t)base * is] = w0;
            w0 = w1; w1 = w2; w2 = w3; w3 = 0;
        }
        double right = src[(size_t)it * is];
        const double* r = cs + (size_t)it * 8;
#define RT_ROTA_(W, I)                                                                  \
        if (!(r[2 * I] == 0.0 && r[2 * I + 1] == 0.0)) {   /* (0, 0): piv == 0, no rotation */ \
            const double c = r[2 * I], sn = r[2 * I + 1], s1 = right, s2 = W;               \
            W = c * s2 + sn * s1;        /* fprota: b = cos*b + sin*a */                    \
            right = c * s1 - sn * s2;    /*         a = cos*a - sin*b */                    \
        }
        RT_ROTA_(w0, 0) RT_ROTA_(w1, 1) RT_ROTA_(w2, 2) RT_ROTA_(w3, 3)
#undef RT_ROTA_
    }
    if (base < m) dst[(size_t)base * is] = w0;
    if (base + 1 < m) dst[(size_t)(base + 1) * is] = w1;
    if (base + 2 < m) dst[(size_t)(base + 2) * is] = w2;
    if (base + 3 < m) dst[(size_t)(base + 3) * is] = w3;
}
// fpback with bandwidth 4: a[n][4] is the band triangle; one line (stride ls) per lane, elements es apart
__global__ void k_fpback(double* d, int n, int nlines, long es, long ls, const double* a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlines) return;
    double* z = d + (size_t)t * ls;
    double c1 = z[(size_t)(n - 1) * es] / a[(size_t)(n - 1) * 4], c2 = 0, c3 = 0;   // c[i+1], c[i+2], c[i+3]
    z[(size_t)(n - 1) * es] = c1;
    for (int i = n - 2, j = 2; i >= 0; i--, j++) {
        const double* ai = a + (size_t)i * 4;
        double store = z[(size_t)i * es];
        store = store - c1 * ai[1];
        if (j > 2) store = store - c2 * ai[2];
        if (j > 3) store = store - c3 * ai[3];
        const double v = store / ai[0];
        z[(size_t)i * es] = v;
        c3 = c2; c2 = c1; c1 = v;
    }
}

template <typename T> __global__ void k_pack(const double* Z, const double* cdy, const double* cdx, T* zn, T* g, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zn[i] = (T)Z[i];
    g[2 * i] = (T)cdx[i];      // d/dx spline = grd[1] (:154)
    g[2 * i + 1] = (T)cdy[i];  // d/dy spline = grd[0] (:155)
}

// The per-cell polynomial table (rt_polytab.h): one thread per cell, fp64 conversion, stored in the field's dtype; and the
// flat-cell map in front of it (rt::poly_cell_flat; thr from the grid's largest gradient-spline coefficient, k_absmax).
// A cell's STEEPNESS lambda = sqrt(max |Hessian of n| / min n) over its four corners, from the cell's own polynomials: the
// gradient splines' first derivatives (d/du, d/dv of both, scaled to x and y) are the Hessian the reference's field has there;
// its infinity norm bounds |w' H w| for every unit w.  A ray that runs along the iso-lines of a transition drifts away from
// its neighbours like exp(lambda s) on the side where n curves upwards: lambda = 36 per unit length in the interface scenario's
// sigmoid, 2 at most in the fisheye, 0.2 in vert_heterogeneous.  Kept (as float bits in the map entry's low word, fp64 fields
// only) when lambda >= lam0 = kSteepRate / (the grid's shorter side): a transition sharp against the size of the scene.
template <typename T>
__global__ void k_polytab(const double* Z, const double* cdx, const double* cdy, int qx, int qy, const double* Cx, const double* Lx,
                          const double* Cy, const double* Ly, T* out, T* flatn, const unsigned long long* gmax_bits,
                          unsigned long long* nflat, double inv_hx, double inv_hy, double lam0) {
    const int jx = blockIdx.x * blockDim.x + threadIdx.x, jy = blockIdx.y;
    bool flat = false, steep = false;
    if (jx < qx - 1 && jy < qy - 1) {
        double c[36];
        rt::poly_cell_convert(Z, cdx, cdy, qx, qy, jx, jy, Cx, Lx, Cy, Ly, c);
        const size_t cell = (size_t)jy * (qx - 1) + jx;
        T* o = out + cell * rt::kPolyStride;
        for (int i = 0; i < 36; i++) o[i] = (T)c[i];
        for (int i = 36; i < rt::kPolyStride; i++) o[i] = T(0);
        flat = rt::poly_cell_flat(c, __builtin_bit_cast(double, *gmax_bits) * rt::kPolyFlatRel);
        typedef typename rt::FlatBits<T>::type B;
        B entry = flat ? __builtin_bit_cast(B, (T)c[32]) : ~(B)0;
        if constexpr (sizeof(T) == 8) {
            if (!flat) {
                double hmax = 0.0, nmin = INFINITY;
                for (int corner = 0; corner < 4; corner++) {
                    const double u = corner & 1, v = corner >> 1;
                    double H[2][2];     // [spline s: 0 = dn/dx, 1 = dn/dy][0: d/dx, 1: d/dy]
                    for (int sp = 0; sp < 2; sp++) {
                        const double* A = c + 16 * sp;      // A[4k + p]: u^p v^k
                        double du = 0.0, dv = 0.0, vk = 1.0;
                        for (int k = 0; k < 4; k++) {
                            du += vk * (A[4 * k + 1] + u * (2.0 * A[4 * k + 2] + u * 3.0 * A[4 * k + 3]));
                            vk *= v;
                        }
                        double up = 1.0;
                        for (int pq = 0; pq < 4; pq++) {
                            dv += up * (A[4 + pq] + v * (2.0 * A[8 + pq] + v * 3.0 * A[12 + pq]));
                            up *= u;
                        }
                        H[sp][0] = du * inv_hx; H[sp][1] = dv * inv_hy;
                    }
                    hmax = fmax(hmax, fmax(fabs(H[0][0]) + fabs(H[0][1]), fabs(H[1][0]) + fabs(H[1][1])));
                    nmin = fmin(nmin, c[32] + u * c[33] + v * c[34] + u * v * c[35]);
                }
                const double lam = nmin > 0.0 ? sqrt(hmax / nmin) : 0.0;
                steep = lam >= lam0 && lam < 3.0e38;
                entry = rt::steep_entry_bits(steep ? (float)lam : 0.f);
            }
        }
        reinterpret_cast<B*>(flatn)[cell] = entry;
    }
    const unsigned long long votes = rt_ballot(flat), svotes = rt_ballot(steep);                 // one atomic per wave
    if ((threadIdx.x & 63) == 0 && votes) atomicAdd(nflat, (unsigned long long)__popcll(votes));
    if ((threadIdx.x & 63) == 0 && svotes) atomicAdd(nflat + 1, (unsigned long long)__popcll(svotes));
}
// max |v| over two arrays as the bit pattern of a non-negative double (ordered like the integers)
__global__ void k_absmax(const double* a, const double* b, size_t n, unsigned long long* out) {
    unsigned long long m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, fabs(a[i])), v = __builtin_bit_cast(unsigned long long, fabs(b[i]));
        m = u > m ? u : m;
        m = v > m ? v : m;                    // (a NaN coefficient orders above everything: no cell is flat then)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(m, o, 64); m = w > m ? w : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// FAST: the fast-form step methods' lookup (the cell's polynomial, one lane per point); else FITPACK's arithmetic on the
// B-spline window (the field is passed with exact = 1)
template <typename T, bool FAST>
__global__ void k_field_eval(rt::FieldDev<T> F, long npts, const double* x, const double* y, double* n, double* gx,
                             double* gy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    T a, b, c;
    if constexpr (FAST) {
        rt::PolyGather<T, rt::kPolyLane> gg;
        rt::n_gradient(F, gg, true, (T)x[i], (T)y[i], a, b, c);
    } else {
        rt::GlobalGather<T> gg;
        rt::n_gradient(F, gg, true, (T)x[i], (T)y[i], a, b, c);
    }
    n[i] = a; gx[i] = b; gy[i] = c;
}

// ------------------------------------------------------------------ host side of the field build
namespace {
// FITPACK fpbspl (k=3) on the host, for the collocation matrix only.
void host_bspl3(const std::vector<double>& t, double x, int l, double h[4]) {
    double hh[4];
    h[0] = 1.0;
    for (int j = 1; j <= 3; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 0; i < j; i++) {
            const int li = l + 1 + i, lj = li - j;
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] = h[i] + f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}
// One axis of fpgrre: the Givens rotations that take the axis' observation matrix (one cubic B-spline row per data point,
// interpolating not-a-knot knots t = [x0 x4, x[2..m-3], x[m-1] x4]) to its band triangle.  Out: nr[it] = first triangle row
// data row it touches, cs[it][i] = (cos, sin) of its i-th rotation ((0, 0): none, the pivot was zero), a[m][4] the triangle.
struct FpAxis { std::vector<int> nr; std::vector<double> cs, a; };
FpAxis fp_axis_build(const std::vector<double>& x) {
    const int m = (int)x.size();
    std::vector<double> t(m + 4);
    for (int i = 0; i <= 3; i++) { t[i] = x[0]; t[m + 3 - i] = x[m - 1]; }
    for (int i = 4, j = 2; i < m; i++, j++) t[i] = x[j];
    FpAxis A;
    A.nr.assign(m, 0); A.cs.assign((size_t)m * 8, 0.0); A.a.assign((size_t)m * 4, 0.0);
    int l = 3, number = 0;
    for (int it = 0; it < m; it++) {
        while (!(x[it] < t[l + 1] || l == m - 1)) { l++; number++; }
        double h[4];
        host_bspl3(t, x[it], l, h);
        A.nr[it] = number;
        for (int i = 0, irot = number; i < 4; i++, irot++) {
            const double piv = h[i];
            if (piv == 0.0) continue;
            double& ww = A.a[(size_t)irot * 4];
            const double store = std::fabs(piv);                 // fpgivs
            double dd;
            if (store >= ww) { const double q = ww / piv; dd = store * std::sqrt(1.0 + q * q); }
            else { const double q = piv / ww; dd = ww * std::sqrt(1.0 + q * q); }
            const double c = ww / dd, sn = piv / dd;
            ww = dd;
            A.cs[(size_t)it * 8 + 2 * i] = c; A.cs[(size_t)it * 8 + 2 * i + 1] = sn;
            for (int j = i + 1, i2 = 1; j < 4; j++, i2++) {      // fprota on the rest of the row
                const double s1 = h[j], s2 = A.a[(size_t)irot * 4 + i2];
                A.a[(size_t)irot * 4 + i2] = c * s2 + sn * s1;
                h[j] = c * s1 - sn * s2;
            }
        }
    }
    return A;
}
// Per cell index j of an axis, rt::ex::kAxisTab = 24 doubles (rt::ex::AxisTab): the correctly rounded reciprocals of the seven knot
// differences rt::ex::axis_exact divides by (same knots by the same operations as the device forms them: x is numpy.linspace as
// linspace() below restates it; then an IEEE division) -- 1/(x[j+1]-x[j]), 1/(k1-k0), 1/(k1-tm1), 1/(k2-k0), 1/(k1-tm2),
// 1/(k2-tm1), 1/(k3-k0), 0 -- then x[j], x[j+1], the six knots tm2 .. k3 and the seven differences (and a 0) in the same order.
std::vector<double> fp_axis_tab_build(const std::vector<double>& x) {
    constexpr int S = rt::ex::kAxisTab;
    const int m = (int)x.size();
    std::vector<double> out((size_t)m * S, 0.0);
    auto knot = [&](int l) { return l <= 3 ? x[0] : (l >= m ? x[m - 1] : x[l - 2]); };   // rt::knot3
    for (int j = 0; j < m - 1; j++) {
        int l = j + 2;
        l = l < 3 ? 3 : (l > m - 1 ? m - 1 : l);
        const double tm2 = knot(l - 2), tm1 = knot(l - 1), k0 = knot(l), k1 = knot(l + 1), k2 = knot(l + 2), k3 = knot(l + 3);
        double* o = &out[(size_t)j * S];
        const double d[7] = {x[j + 1] - x[j], k1 - k0, k1 - tm1, k2 - k0, k1 - tm2, k2 - tm1, k3 - k0};
        for (int i = 0; i < 7; i++) { o[i] = 1.0 / d[i]; o[16 + i] = d[i]; }
        o[8] = x[j]; o[9] = x[j + 1];
        o[10] = tm2; o[11] = tm1; o[12] = k0; o[13] = k1; o[14] = k2; o[15] = k3;
    }
    for (int i = 0; i < S; i++) out[(size_t)(m - 1) * S + i] = out[(size_t)(m - 2) * S + i];
    return out;
}
std::vector<double> linspace(double a, double b, int n) {
    std::vector<double> v(n);
    const double step = (b - a) / (double)(n - 1);
    for (int i = 0; i < n; i++) v[i] = (double)i * step + a;
    v[n - 1] = b;
    return v;
}
}  // namespace

static int field_finish_impl(rtmi_field* f, double delta) {
    const int qx = f->qx, qy = f->qy;
    const size_t nz = (size_t)qx * qy;
    hipStream_t st = f->stream;
    HIP_TRY(hipMalloc(&f->dCdy, nz * sizeof(double)));
    HIP_TRY(hipMalloc(&f->dCdx, nz * sizeof(double)));
    dim3 blk(256), grd((qx + 255) / 256, qy);
    hipLaunchKernelGGL(k_gradient, grd, blk, 0, st, f->dZ, f->dCdy, qx, qy, 0, delta);  // GradX = d/dy (Q2)
    hipLaunchKernelGGL(k_gradient, grd, blk, 0, st, f->dZ, f->dCdx, qx, qy, 1, delta);  // GradY = d/dx
    HIP_TRY(hipGetLastError());
    // RectBivariateSpline(y, x, Grad) (:456-457) = FITPACK regrid, s = 0: Givens QR along y (FITPACK's x), then along x
    const FpAxis AX = fp_axis_build(linspace(f->ax, f->bx, qx));
    const FpAxis AY = fp_axis_build(linspace(f->ay, f->by, qy));
    double* dlux = nullptr;     // rotations + triangles of both axes, then the work matrix g
    double* dluy = nullptr;
    double* dpoly = nullptr;    // per-axis tables of the cell polynomials
    auto solve_and_pack = [&]() -> int {   // dlux/dluy are released below whatever this returns
        const size_t nax = (size_t)qx * 12, nay = (size_t)qy * 12;                 // cs [m][8] + a [m][4]
        HIP_TRY(hipMalloc(&dlux, (nax + nay) * sizeof(double) + (size_t)(qx + qy) * sizeof(int)));
        HIP_TRY(hipMalloc(&dluy, nz * sizeof(double)));
        double *csx = dlux, *ax4 = dlux + (size_t)qx * 8, *csy = dlux + nax, *ay4 = csy + (size_t)qy * 8;
        int *nrx = (int*)(dlux + nax + nay), *nry = nrx + qx;
        HIP_TRY(hipMemcpyAsync(csx, AX.cs.data(), (size_t)qx * 8 * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ax4, AX.a.data(), (size_t)qx * 4 * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(csy, AY.cs.data(), (size_t)qy * 8 * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ay4, AY.a.data(), (size_t)qy * 4 * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(nrx, AX.nr.data(), (size_t)qx * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(nry, AY.nr.data(), (size_t)qy * sizeof(int), hipMemcpyHostToDevice, st));
        double* g = dluy;
        for (double* c : {f->dCdy, f->dCdx}) {
            // rows of z into FITPACK-x's triangle (one lane per column), columns of g into FITPACK-y's (one lane per row)
            hipLaunchKernelGGL(k_givens, dim3((qx + 63) / 64), dim3(64), 0, st, c, g, qy, qx, (long)qx, 1L, nry, csy);
            hipLaunchKernelGGL(k_givens, dim3((qy + 63) / 64), dim3(64), 0, st, g, c, qx, qy, 1L, (long)qx, nrx, csx);
            // (ry) c1 = h along x for every row, then c (rx)' = c1 along y for every column
            hipLaunchKernelGGL(k_fpback, dim3((qy + 63) / 64), dim3(64), 0, st, c, qx, qy, 1L, (long)qx, ax4);
            hipLaunchKernelGGL(k_fpback, dim3((qx + 63) / 64), dim3(64), 0, st, c, qy, qx, (long)qx, 1L, ay4);
        }
        HIP_TRY(hipGetLastError());
        const size_t esz = f->dtype == RTMI_F64 ? 8 : 4;
        HIP_TRY(hipMalloc(&f->zn, nz * esz));
        HIP_TRY(hipMalloc(&f->g, 2 * nz * esz));
        if (f->dtype == RTMI_F64)
            hipLaunchKernelGGL(k_pack<double>, dim3((nz + 255) / 256), dim3(256), 0, st, f->dZ, f->dCdy, f->dCdx,
                               (double*)f->zn, (double*)f->g, nz);
        else
            hipLaunchKernelGGL(k_pack<float>, dim3((nz + 255) / 256), dim3(256), 0, st, f->dZ, f->dCdy, f->dCdx,
                               (float*)f->zn, (float*)f->g, nz);
        HIP_TRY(hipGetLastError());
        {   // reciprocals of the knot differences for the reference-order lookup (rt_exact.h)
            const std::vector<double> RX = fp_axis_tab_build(linspace(f->ax, f->bx, qx)), RY = fp_axis_tab_build(linspace(f->ay, f->by, qy));
            HIP_TRY(hipMalloc(&f->rdiv, (RX.size() + RY.size()) * sizeof(double)));
            HIP_TRY(hipMemcpyAsync(f->rdiv, RX.data(), RX.size() * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(f->rdiv + RX.size(), RY.data(), RY.size() * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));  // the host vectors go out of scope
        }
        // one polynomial per cell for the fast-form lookups: per-axis basis tables on the host (long double), cells on the device
        {
            const double ihx = (double)(f->dtype == RTMI_F64 ? 1.0 / f->hx : (double)(float)(1.0 / f->hx));
            const double ihy = (double)(f->dtype == RTMI_F64 ? 1.0 / f->hy : (double)(float)(1.0 / f->hy));
            const bool f64 = f->dtype == RTMI_F64;
            const rt::PolyAxis PX = rt::poly_axis_build(linspace(f->ax, f->bx, qx), f64 ? f->ax : (double)(float)f->ax, ihx);
            const rt::PolyAxis PY = rt::poly_axis_build(linspace(f->ay, f->by, qy), f64 ? f->ay : (double)(float)f->ay, ihy);
            const size_t ncell = (size_t)(qx - 1) * (qy - 1);
            const size_t nax2 = (size_t)(qx - 1) * 20, nay2 = (size_t)(qy - 1) * 20;
            HIP_TRY(hipMalloc(&dpoly, (nax2 + nay2) * sizeof(double)));
            double *dCx = dpoly, *dLx = dpoly + (size_t)(qx - 1) * 16, *dCy = dpoly + nax2, *dLy = dCy + (size_t)(qy - 1) * 16;
            HIP_TRY(hipMemcpyAsync(dCx, PX.C.data(), PX.C.size() * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(dLx, PX.L.data(), PX.L.size() * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(dCy, PY.C.data(), PY.C.size() * sizeof(double), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(dLy, PY.L.data(), PY.L.size() * sizeof(double), hipMemcpyHostToDevice, st));
            // one allocation: the flat-cell map (one entry per cell, padded to 32 entries), then the table
            f->flat_pad = (long)((ncell + 31) / 32 * 32);
            HIP_TRY(hipMalloc(&f->poly_base, ((size_t)f->flat_pad + ncell * rt::kPolyStride) * esz));
            f->poly = (char*)f->poly_base + (size_t)f->flat_pad * esz;
            unsigned long long* dcnt = nullptr;      // [0] bits of max |gradient-spline coefficient|, [1] flat cells, [2] steep cells
            HIP_TRY(hipMalloc(&dcnt, 3 * sizeof(unsigned long long)));
            unsigned long long hcnt[3] = {0, 0, 0};
            // steep: lambda >= kSteepRate / the grid's shorter side (RTMI_STEEP_RATE overrides, for calibration runs)
            const double steep_rate = getenv("RTMI_STEEP_RATE") ? atof(getenv("RTMI_STEEP_RATE")) : kSteepRate;
            const double lam0 = steep_rate / std::fmin(f->bx - f->ax, f->by - f->ay);
            hipError_t e = hipMemsetAsync(dcnt, 0, sizeof(hcnt), st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, f->dCdx, f->dCdy, nz, dcnt);
                const dim3 pg((qx - 1 + 63) / 64, qy - 1), pb(64);
                if (f->dtype == RTMI_F64)
                    hipLaunchKernelGGL(k_polytab<double>, pg, pb, 0, st, f->dZ, f->dCdx, f->dCdy, qx, qy, dCx, dLx, dCy, dLy, (double*)f->poly,
                                       (double*)f->poly_base, dcnt, dcnt + 1, ihx, ihy, lam0);
                else
                    hipLaunchKernelGGL(k_polytab<float>, pg, pb, 0, st, f->dZ, f->dCdx, f->dCdy, qx, qy, dCx, dLx, dCy, dLy, (float*)f->poly,
                                       (float*)f->poly_base, dcnt, dcnt + 1, ihx, ihy, lam0);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(hcnt, dcnt, sizeof(hcnt), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host tables go out of scope
            (void)hipFree(dcnt);
            HIP_TRY(e);
            f->flat_cells = getenv("RTMI_NO_FLAT") ? 0 : (long)hcnt[1];     // RTMI_NO_FLAT=1: A/B without the map
            f->steep_cells = getenv("RTMI_NO_FLAT") ? 0 : (long)hcnt[2];
            memcpy(&f->gmax, &hcnt[0], sizeof(double));
            if (getenv("RTMI_DEBUG")) fprintf(stderr, "rtmi: field %d x %d: %ld of %zu cells flat, %ld steep (lambda >= %.3g)\n", qx, qy, (long)hcnt[1], ncell, (long)hcnt[2], lam0);
        }
        HIP_TRY(hipStreamSynchronize(st));  // host vectors / LU buffers go out of scope
        return RTMI_OK;
    };
    const int rc = solve_and_pack();
    if (rc) (void)hipStreamSynchronize(st);   // nothing may still read the LU buffers
    (void)hipFree(dlux);
    (void)hipFree(dluy);
    (void)hipFree(dpoly);
    return rc;
}

static int field_finish(rtmi_field* f, double delta) {
    try {
        return field_finish_impl(f, delta);
    } catch (const std::exception& e) {   // host vectors of the collocation factorisation
        return fail(RTMI_ERR_ALLOC, std::string("field build: ") + e.what());
    }
}

RTMI_EXPORT int rtmi_abi_version(void) { return RTMI_ABI_VERSION; }
RTMI_EXPORT const char* rtmi_last_error(void) { return g_err.c_str(); }
RTMI_EXPORT int rtmi_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return RTMI_OK;
}
RTMI_EXPORT int rtmi_device_count(int* count) {
    ARG_TRY(count, "rtmi_device_count: null");
    HIP_TRY(hipGetDeviceCount(count));
    return RTMI_OK;
}

RTMI_EXPORT void rtmi_field_destroy(rtmi_field* f) {
    if (!f) return;
    (void)hipFree(f->dZ); (void)hipFree(f->dCdy); (void)hipFree(f->dCdx); (void)hipFree(f->zn); (void)hipFree(f->g);
    (void)hipFree(f->poly_base);
    (void)hipFree(f->rdiv);
    delete f;
}

static int field_alloc(int dtype, int qx, int qy, void* stream, rtmi_field** out) {
    ARG_TRY(out, "field: out is null");
    ARG_TRY(dtype == RTMI_F64 || dtype == RTMI_F32, "field: dtype must be RTMI_F64 or RTMI_F32");
    ARG_TRY(qx >= 8 && qy >= 8, "field: grid must be at least 8x8 (cubic not-a-knot fit)");
    ARG_TRY((size_t)qx * qy < (1ull << 31) && qx < (1 << 24) && qy < (1 << 24), "field: grid too large");
    rtmi_field* f = new (std::nothrow) rtmi_field();
    if (!f) return fail(RTMI_ERR_ALLOC, "field: host allocation failed");
    f->dtype = dtype; f->qx = qx; f->qy = qy; f->stream = (hipStream_t)stream;
    *out = f;
    HIP_TRY(hipGetDevice(&f->device));
    HIP_TRY(hipMalloc(&f->dZ, (size_t)qx * qy * sizeof(double)));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_field_build(int scenario, double xi, double xs, double yi, double ys, double delta, int dtype,
                                 void* stream, rtmi_field** out) {
    ARG_TRY(scenario >= RTMI_INTERFACE && scenario <= RTMI_ANISOTROPY, "rtmi_field_build: scenario must be 1..4");
    ARG_TRY(delta > 0 && xs > xi && ys > yi, "rtmi_field_build: need delta > 0 and xs > xi, ys > yi");
    const int qx = (int)((xs - xi + 6) / delta + 1);  // :426
    const int qy = (int)((ys - yi + 6) / delta + 1);  // :427
    rtmi_field* f = nullptr;
    int rc = field_alloc(dtype, qx, qy, stream, &f);
    if (rc) { rtmi_field_destroy(f); return rc; }
    f->ax = xi - 3; f->bx = xs + 3; f->hx = (f->bx - f->ax) / (double)(qx - 1);  // :429 linspace
    f->ay = yi - 3; f->by = ys + 3; f->hy = (f->by - f->ay) / (double)(qy - 1);
    hipLaunchKernelGGL(k_sample, dim3((qx + 255) / 256, qy), dim3(256), 0, f->stream, scenario, f->dZ, qx, qy, f->ax,
                       f->hx, f->bx, f->ay, f->hy, f->by);
    rc = field_finish(f, delta);
    if (rc) { rtmi_field_destroy(f); return rc; }
    *out = f;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_field_from_samples(const double* x, int qx, const double* y, int qy, const double* Z, double delta,
                                        int dtype, void* stream, rtmi_field** out) {
    ARG_TRY(x && y && Z, "rtmi_field_from_samples: null input");
    ARG_TRY(delta > 0, "rtmi_field_from_samples: delta must be > 0");
    ARG_TRY(qx >= 8 && qy >= 8, "rtmi_field_from_samples: grid must be at least 8x8");
    try {
        const std::vector<double> lx = linspace(x[0], x[qx - 1], qx), ly = linspace(y[0], y[qy - 1], qy);
        if (memcmp(lx.data(), x, qx * sizeof(double)) || memcmp(ly.data(), y, qy * sizeof(double)))
            return fail(RTMI_ERR_UNSUPPORTED, "rtmi_field_from_samples: axes must be numpy.linspace grids (genZ, RT_bench.py:429)");
    } catch (const std::exception& e) {
        return fail(RTMI_ERR_ALLOC, std::string("rtmi_field_from_samples: ") + e.what());
    }
    rtmi_field* f = nullptr;
    int rc = field_alloc(dtype, qx, qy, stream, &f);
    if (rc) { rtmi_field_destroy(f); return rc; }
    f->ax = x[0]; f->bx = x[qx - 1]; f->hx = (f->bx - f->ax) / (double)(qx - 1);
    f->ay = y[0]; f->by = y[qy - 1]; f->hy = (f->by - f->ay) / (double)(qy - 1);
    hipError_t e = hipMemcpyAsync(f->dZ, Z, (size_t)qx * qy * sizeof(double), hipMemcpyHostToDevice, f->stream);
    if (e != hipSuccess) { rtmi_field_destroy(f); return fail(RTMI_ERR_HIP, hipGetErrorString(e)); }
    rc = field_finish(f, delta);
    if (rc) { rtmi_field_destroy(f); return rc; }
    *out = f;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_field_dims(const rtmi_field* f, int* qx, int* qy) {
    ARG_TRY(f && qx && qy, "rtmi_field_dims: null");
    *qx = f->qx; *qy = f->qy;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_field_read(const rtmi_field* f, double* x, double* y, double* Z, double* cdy, double* cdx) {
    ARG_TRY(f, "rtmi_field_read: null field");
    DEVICE_TRY(f, "rtmi_field_read");
    const size_t nz = (size_t)f->qx * f->qy * sizeof(double);
    HIP_TRY(hipStreamSynchronize(f->stream));
    try {
        if (x) { auto v = linspace(f->ax, f->bx, f->qx); memcpy(x, v.data(), v.size() * sizeof(double)); }
        if (y) { auto v = linspace(f->ay, f->by, f->qy); memcpy(y, v.data(), v.size() * sizeof(double)); }
    } catch (const std::exception& e) {
        return fail(RTMI_ERR_ALLOC, std::string("rtmi_field_read: ") + e.what());
    }
    if (Z) HIP_TRY(hipMemcpy(Z, f->dZ, nz, hipMemcpyDeviceToHost));
    if (cdy) HIP_TRY(hipMemcpy(cdy, f->dCdy, nz, hipMemcpyDeviceToHost));
    if (cdx) HIP_TRY(hipMemcpy(cdx, f->dCdx, nz, hipMemcpyDeviceToHost));
    return RTMI_OK;
}

static int field_eval_impl(const rtmi_field* f, int64_t npts, const double* x, const double* y, double* n, double* gx,
                           double* gy, bool fast, const char* who) {
    if (!(f && x && y && n && gx && gy)) return fail(RTMI_ERR_ARG, std::string(who) + ": null");
    if (npts < 0) return fail(RTMI_ERR_ARG, std::string(who) + ": npts < 0");
    DEVICE_TRY(f, who);
    if (npts == 0) return RTMI_OK;
    double* d = nullptr;
    const size_t nb = (size_t)npts * sizeof(double);
    HIP_TRY(hipMalloc(&d, 5 * nb));
    hipStream_t st = f->stream;
    int rc = RTMI_OK;
    do {
        if (hipMemcpyAsync(d, x, nb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + npts, y, nb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = RTMI_ERR_HIP; break; }
        const dim3 g((unsigned)((npts + 255) / 256)), b(256);
        double *dn = d + 2 * npts, *dgx = d + 3 * npts, *dgy = d + 4 * npts;
        if (f->dtype == RTMI_F64) {
            if (fast) hipLaunchKernelGGL((k_field_eval<double, true>), g, b, 0, st, field_dev<double>(f, 0), (long)npts, d, d + npts, dn, dgx, dgy);
            else hipLaunchKernelGGL((k_field_eval<double, false>), g, b, 0, st, field_dev<double>(f, 1), (long)npts, d, d + npts, dn, dgx, dgy);
        } else {
            if (fast) hipLaunchKernelGGL((k_field_eval<float, true>), g, b, 0, st, field_dev<float>(f, 0), (long)npts, d, d + npts, dn, dgx, dgy);
            else hipLaunchKernelGGL((k_field_eval<float, false>), g, b, 0, st, field_dev<float>(f, 1), (long)npts, d, d + npts, dn, dgx, dgy);
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = RTMI_ERR_HIP; break; }
        if (hipMemcpy(n, dn, nb, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(gx, dgx, nb, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(gy, dgy, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = RTMI_ERR_HIP;
    } while (0);
    (void)hipFree(d);
    if (rc) return fail(rc, std::string(who) + ": HIP failure");
    return RTMI_OK;
}
RTMI_EXPORT int rtmi_field_eval(const rtmi_field* f, int64_t npts, const double* x, const double* y, double* n,
                                double* gx, double* gy) {
    return field_eval_impl(f, npts, x, y, n, gx, gy, false, "rtmi_field_eval");
}
RTMI_EXPORT int rtmi_debug_field_lookup(const rtmi_field* f, int64_t npts, const double* x, const double* y, double* n,
                                        double* gx, double* gy) {
    return field_eval_impl(f, npts, x, y, n, gx, gy, true, "rtmi_debug_field_lookup");
}

// ================================================================== ray batch
template <typename T> struct BatchDev {
    rt::FieldDev<T> F;
    rt::Consts<T> K;
    long R;
    int max_size;
    int stride;       // record stride (0 = none)
    long rec_rows;
    // SoA state in ONE slab: six fp64 accumulator arrays (x, y, theta, dist_sim, dist_real, T; fp64 in both precisions,
    // see rt::Ray) followed by dtype arrays n, gx, gy and op7's four history arrays, each [R].  The per-array pointers
    // are formed where they are used so that the step loop keeps a few scalar registers live instead of twenty.
    double* st;
    int has_hist;     // op7 only: hx0, hy0, hx1, hy1 follow n, gx, gy
    int exact;        // fp64 op3/4/5/9/10/11: derived values and lookups in the reference's operation order (rt_exact.h)
    int iso;          // the step kernels run their ISO build (gamma == 1, method < 10, not exact): coef taken as exactly 1
    int rot;          // fp64 op1/2/6/8 (rt::RotatesUnit): the unit vector (cos, sin) is state, kept in unit(0), unit(1)
    __device__ __forceinline__ double* acc(int q) const { return st + (size_t)q * R; }                 // 0..5: x y th dsim dreal tt
    __device__ __forceinline__ T* aux(int q) const { return reinterpret_cast<T*>(st + (size_t)6 * R) + (size_t)q * R; }   // 0..2: n gx gy; 3..6: history
    __device__ __forceinline__ T* unit(int q) const { return aux(3 + q); }     // rot batches have no history arrays
    int* istep;
    unsigned char* alive;
    T *s_ray, *n_ray;
    unsigned long long* counters;  // [0] ray-steps, [1] live rays (filled by k_stats on demand), [2] refill queue head
    const double *x0, *y0, *th0;   // launch conditions (device, fp64), in slot order
    const int* perm;               // sort_rays: slot k holds caller's ray perm[k] (nullptr: identity)
    const T *vstep, *vstep2h;      // per-ray DELTA_S and DELTA_S**2/2 (rtmi_batch_set_per_ray; nullptr: uniform)
    const int* vmax;               // per-ray max_size (same)
    // Critical rays (rt::hover_update, "retrace" below): hov[R] is the per-ray hover sum (state; nullptr: this batch flags nothing),
    // hov_limit = rt::kHoverLimit (the sum times DELTA_S beyond which a ray is handed over), rq the hand-over queue in device
    // memory -- [0] rays pushed so far (also those beyond rq_cap: they stay), [1..] entries (row where the ray stopped << 32 | ray
    // slot) -- and rq_host one flag per entry in pinned host memory, where the host counts them while the kernel runs (push_critical).
    float* hov;
    float hov_limit;
    unsigned rq_cap;
    unsigned long long* rq;
    unsigned* rq_host;
    int prio;                      // > 0: the kernel's waves raise their issue priority (the re-trace batch, beside the main kernel)
    double gflat;                  // rt::GlobalGather::gflat (the reference-order step where the medium is constant, rt_exact.h)
    unsigned blk_rot;              // k_advance, builds with the hover sum: the bundle the first hardware block takes (Retrace::rot)
};

// A fused ray whose hover sum passed the limit: queue it for the re-trace in reference order and stop it.  False when the
// queue is full (the ray then carries on in the fused form and is counted: rtmi_stats.retrace_overflow).
// The entry goes to the coherence point (device scope) and is acknowledged before the count is published to the host, which
// launches the consumer only after it has read that count: the consumer kernel starts with the entry visible.
// The host learns of an entry through a flag of ITS OWN in pinned host memory (rq_host[slot], stored once the entry is
// acknowledged) and counts the leading run of flags: no store depends on another's order.  (A count published by plain
// stores of slot + 1 landed out of order now and then -- the host sat on a stale count until the main kernel was done, +5 ms
// per pass -- and could run ahead of a slower lane's entry; a count published under a device lock was right but its loop and
// loads cost the kernels that sit at 128 registers spills in the step loop.)
template <typename T> __device__ __forceinline__ bool push_critical(const BatchDev<T>& a, long k, int row) {
    const unsigned slot = atomicAdd(reinterpret_cast<unsigned*>(a.rq), 1u);       // the low word of the 64-bit count
    if (slot >= a.rq_cap) return false;
    __hip_atomic_store(a.rq + 1 + slot, ((unsigned long long)(unsigned)row << 32) | (unsigned long long)(unsigned)k, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __hip_atomic_store(a.rq_host + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
}

template <typename T> __device__ __forceinline__ void write_row(const BatchDev<T>& a, long row, long k, const rt::Ray<T>& r) {
    T* p = a.s_ray + (size_t)row * 6 * a.R + k;
    p[0] = (T)r.x; p[a.R] = (T)r.y; p[2 * a.R] = r.mx; p[3 * a.R] = r.my; p[4 * a.R] = (T)r.tt; p[5 * a.R] = (T)r.th;
    if (a.n_ray) a.n_ray[(size_t)row * a.R + k] = r.nray;
}

// derived values / lookups outside the step loop, where the method is a run-time property of the batch
template <typename T> __device__ __forceinline__ void derive_rt(const BatchDev<T>& a, rt::Ray<T>& r) {
    if (a.iso) rt::derive<T, true>(a.K, r); else rt::derive<T, false>(a.K, r);
}
template <> __device__ __forceinline__ void derive_rt<double>(const BatchDev<double>& a, rt::Ray<double>& r) {
    if (a.exact) rt::ex::derive(a.K, r);
    else if (a.iso) rt::derive<double, true>(a.K, r);
    else rt::derive<double, false>(a.K, r);
}
#if RTMI_POLY
template <typename T> using InitGather = rt::PolyGather<T, rt::kPolyLane>;     // the fast forms' lookup, one lane per ray
#else
template <typename T> using InitGather = rt::GlobalGather<T>;
#endif
template <typename T> __device__ __forceinline__ void n_gradient_rt(const BatchDev<T>& a, T x, T y, T& n, T& gx, T& gy) {
    InitGather<T> gg;
    rt::n_gradient(a.F, gg, true, x, y, n, gx, gy);
}
template <> __device__ __forceinline__ void n_gradient_rt<double>(const BatchDev<double>& a, double x, double y, double& n, double& gx, double& gy) {
    if (a.exact) { rt::GlobalGather<double> gg; rt::ex::n_gradient(a.F, gg, true, x, y, n, gx, gy); }
    else { InitGather<double> gg; rt::n_gradient(a.F, gg, true, x, y, n, gx, gy); }
}

// initial conditions (:809-826) of the ray in slot k
template <typename T> __device__ __forceinline__ void init_ray(const BatchDev<T>& a, long k) {
    rt::Ray<T> r;
    r.x = a.x0[k]; r.y = a.y0[k]; r.th = a.th0[k];
    n_gradient_rt(a, (T)r.x, (T)r.y, r.n, r.gx, r.gy);
    derive_rt(a, r);
    r.dsim = 0; r.dreal = 0; r.tt = 0;
    a.acc(0)[k] = r.x; a.acc(1)[k] = r.y; a.acc(2)[k] = r.th; a.aux(0)[k] = r.n; a.aux(1)[k] = r.gx; a.aux(2)[k] = r.gy;
    a.acc(3)[k] = 0; a.acc(4)[k] = 0; a.acc(5)[k] = 0;
    if (a.has_hist) { a.aux(3)[k] = 0; a.aux(4)[k] = 0; a.aux(5)[k] = 0; a.aux(6)[k] = 0; }
    if (a.rot) { a.unit(0)[k] = r.ux; a.unit(1)[k] = r.uy; }
    if (a.hov) a.hov[k] = 0.f;
    a.istep[k] = 0;
    a.alive[k] = max_size_of(a, k) > 1;
    if (a.stride && a.rec_rows > 0) write_row(a, 0, k, r);
}
// one lane per ray
template <typename T> __global__ void k_init(BatchDev<T> a) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    init_ray(a, k);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Batch totals on demand: counters[0] = sum of istep (ray-steps taken = sum of d_ray[2]), counters[1] = live rays.
// Kept out of the advance kernels on purpose: two same-address atomics per wave cost more than a whole
// one-step launch (16 384 waves x ~12 ns each), and both totals are functions of state already in HBM.
__global__ void k_stats(const int* istep, const unsigned char* alive, long R, unsigned long long* counters) {
    unsigned long long steps = 0, live = 0;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < R; k += (long)gridDim.x * blockDim.x) {
        steps += (unsigned long long)istep[k];
        live += alive[k];
    }
    steps = wave_sum(steps);
    live = wave_sum(live);
    if ((threadIdx.x & 63) == 0) {
        if (steps) atomicAdd(&counters[0], steps);
        if (live) atomicAdd(&counters[1], live);
    }
}

// State accesses.  COH: the time-sliced kernel hands a bundle's state from one block to another, possibly on another XCD
// whose L2 is not coherent with this one's; its state loads and stores are device-scope (relaxed) atomics, which go to the
// coherence point, so that no whole-cache write-back / invalidate is needed around a slice (the rows need none: only the
// host reads them).  Plain accesses otherwise.
template <bool COH, typename V> __device__ __forceinline__ V ld_state(const V* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, typename V> __device__ __forceinline__ void st_state(V* p, V v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// HOV: this build keeps the hover sum (critical rays; only the fp64 op1/2/6/8 builds with the flat-cell map compiled in)
template <typename T, int METHOD, bool ISO, bool COH = false, bool HOV = false>
__device__ __forceinline__ void load_ray(const BatchDev<T>& a, long k, rt::Ray<T>& r, int& i) {
    r.x = ld_state<COH>(a.acc(0) + k); r.y = ld_state<COH>(a.acc(1) + k); r.th = ld_state<COH>(a.acc(2) + k);
    r.n = ld_state<COH>(a.aux(0) + k); r.gx = ld_state<COH>(a.aux(1) + k); r.gy = ld_state<COH>(a.aux(2) + k);
    r.dsim = ld_state<COH>(a.acc(3) + k); r.dreal = ld_state<COH>(a.acc(4) + k); r.tt = ld_state<COH>(a.acc(5) + k);
    if (rt::base_method(METHOD) == 7) {
        r.hx0 = ld_state<COH>(a.aux(3) + k); r.hy0 = ld_state<COH>(a.aux(4) + k);
        r.hx1 = ld_state<COH>(a.aux(5) + k); r.hy1 = ld_state<COH>(a.aux(6) + k);
    } else { r.hx0 = r.hy0 = r.hx1 = r.hy1 = 0; }
    if constexpr (rt::IsExact<T, METHOD>::value) rt::ex::derive(a.K, r);
    else if constexpr (rt::RotatesUnit<T, METHOD>::value) {
        r.ux = ld_state<COH>(a.unit(0) + k); r.uy = ld_state<COH>(a.unit(1) + k);
        rt::derive<T, ISO, true>(a.K, r);
    } else rt::derive<T, ISO>(a.K, r);
    if constexpr (HOV) r.hov = a.hov ? ld_state<COH>(a.hov + k) : 0.f;
    else r.hov = 0.f;
    i = ld_state<COH>(a.istep + k);
}
template <typename T, int METHOD, bool COH = false, bool HOV = false>
__device__ __forceinline__ void store_ray(const BatchDev<T>& a, long k, const rt::Ray<T>& r, int i, bool alive) {
    st_state<COH>(a.acc(0) + k, r.x); st_state<COH>(a.acc(1) + k, r.y); st_state<COH>(a.acc(2) + k, r.th);
    st_state<COH>(a.aux(0) + k, r.n); st_state<COH>(a.aux(1) + k, r.gx); st_state<COH>(a.aux(2) + k, r.gy);
    st_state<COH>(a.acc(3) + k, r.dsim); st_state<COH>(a.acc(4) + k, r.dreal); st_state<COH>(a.acc(5) + k, r.tt);
    if (rt::base_method(METHOD) == 7) {
        st_state<COH>(a.aux(3) + k, r.hx0); st_state<COH>(a.aux(4) + k, r.hy0);
        st_state<COH>(a.aux(5) + k, r.hx1); st_state<COH>(a.aux(6) + k, r.hy1);
    }
    if constexpr (rt::RotatesUnit<T, METHOD>::value) {
        st_state<COH>(a.unit(0) + k, r.ux); st_state<COH>(a.unit(1) + k, r.uy);
        if constexpr (HOV) { if (a.hov) st_state<COH>(a.hov + k, r.hov); }
    }
    st_state<COH>(a.istep + k, i);
    st_state<COH>(a.alive + k, (unsigned char)alive);
}

// A finite, in-grid dummy state for lanes that hold no ray (they still execute every step, see rt::ray_step).
template <typename T> __device__ __forceinline__ void idle_ray(const BatchDev<T>& a, rt::Ray<T>& r) {
    r.x = a.F.ax + a.F.hx * T(8); r.y = a.F.ay + a.F.hy * T(8); r.th = 0; r.n = 1; r.gx = 0; r.gy = 0;
    r.ux = 1; r.uy = 0; r.coef = 1; r.nray = 1; r.rn = 1;
    r.dsim = r.dreal = r.tt = r.mx = r.my = 0;
    r.hx0 = r.hx1 = r.x; r.hy0 = r.hy1 = r.y;
    r.hov = 0.f;
}

// Gather policy of a step kernel.  Reference-order methods (rt_exact.h: FITPACK's sums on the B-spline window): the LDS tile
// (LDS) or global gathers.  Fast-form methods and every fp32 batch: the cell's polynomial (rt::PolyGather), through the scalar
// cache for a coherent wave (LDS) or with per-lane loads.  RTMI_POLY 0 builds the fast forms on the B-spline window as well.
template <typename T, int METHOD> constexpr bool uses_poly() { return RTMI_POLY && (!rt::IsExact<T, METHOD>::value || (METHOD & rt::kFastField) != 0); }
#ifndef RTMI_LAT_LANEKEPT
#define RTMI_LAT_LANEKEPT 1
#endif
template <typename T, int METHOD, bool LDS> constexpr bool uses_tile() { return LDS && !uses_poly<T, METHOD>(); }
// NOFLAT: the field has no flat cell (rt::PolyGather's FLAT false: the flat-cell map's tests compiled out).
template <typename T, int METHOD, bool LDS, int PH = RTMI_TILE_PHASES, bool NOFLAT = false, bool POLY = uses_poly<T, METHOD>()> struct GatherOf { using type = rt::GlobalGather<T, !NOFLAT>; };
template <typename T, int METHOD, int PH, bool NOFLAT> struct GatherOf<T, METHOD, true, PH, NOFLAT, false> { using type = rt::LdsGather<T, PH>; };
template <typename T, int METHOD, bool LDS, int PH, bool NOFLAT> struct GatherOf<T, METHOD, LDS, PH, NOFLAT, true> {
#if RTMI_LAT_LANEKEPT
    // PH 1: k_advance_lat -- every lane its own kept cell, the next cell's loads a step ahead (rt::PolyLaneKept)
    using type = std::conditional_t<LDS && PH == 1, rt::PolyLaneKept<T, !NOFLAT>, rt::PolyGather<T, !LDS ? rt::kPolyLane : rt::kPolyScalar, !NOFLAT>>;
#else
    using type = rt::PolyGather<T, !LDS ? rt::kPolyLane : PH == 1 ? rt::kPolyCached : rt::kPolyScalar, !NOFLAT>;   // PH 1: k_advance_lat
#endif
};
template <typename T, bool LDS, bool FM> __device__ __forceinline__ void gather_init(rt::GlobalGather<T, FM>&, T*) {}
// the per-lane gather of a build with the flat path compiled in (reference-order op1/2/6/8 on a field with flat cells) carries gflat
template <typename T, typename G> __device__ __forceinline__ void gather_flat_bound(G&, const BatchDev<T>&) {}
template <> __device__ __forceinline__ void gather_flat_bound<double, rt::GlobalGather<double, true>>(rt::GlobalGather<double, true>& g, const BatchDev<double>& a) { g.gflat = a.gflat; }
template <typename T, bool LDS, int MODE, bool FLAT> __device__ __forceinline__ void gather_init(rt::PolyGather<T, MODE, FLAT>& g, T*) { g.init(); }
template <typename T, bool LDS, bool FLAT> __device__ __forceinline__ void gather_init(rt::PolyLaneKept<T, FLAT>& g, T*) { g.init(); }
// LDS of a step kernel in units of T: the reference-order methods' tile; the polynomial lookup needs none.  (An L2 prefetch of
// the cells ahead -- global_load_lds into a per-wave sink whenever the wave's cell changes -- was measured: interface 23.6 ->
// 23.0 ms, but fisheye, a new cell every step, 8.3 -> 9.1, vert_heterogeneous 8.8 -> 9.0, fp32 48.7 -> 49.6: not kept.)
template <typename T, int METHOD, bool LDS, int PH = RTMI_TILE_PHASES> constexpr int kernel_lds_elems() {
    return uses_tile<T, METHOD, LDS>() ? 4 * rt::LdsGather<T, PH>::ELEMS : 2;
}
template <typename T, bool LDS, int PH> __device__ __forceinline__ void gather_init(rt::LdsGather<T, PH>& g, T* lds) {
    g.init(lds + (threadIdx.x >> 6) * rt::LdsGather<T, PH>::ELEMS);
}

// A ray's state is stored once, when it terminates: the batch members that needs (state slab, istep, alive) are re-read
// from the kernel-argument segment there instead of occupying scalar registers for the whole loop (see rt::rare_field).
#ifndef RTMI_NO_KERNARG_FIELD
template <typename T> __device__ __forceinline__ BatchDev<T> rare_batch(const BatchDev<T>&) {
    static_assert(__builtin_offsetof(BatchDev<T>, F) == 0, "rt::rare_field reads the FieldDev at offset 0 of the kernel arguments");
    typedef const BatchDev<T> __attribute__((address_space(4))) * KP;
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    BatchDev<T> out;
    __builtin_memcpy(&out, p, sizeof(out));      // scalar loads from the constant address space
    return out;
}
#else
template <typename T> __device__ __forceinline__ const BatchDev<T>& rare_batch(const BatchDev<T>& a) { return a; }
#endif

// ---- trajectory rows through a wave-uniform buffer descriptor
// In k_advance every live lane of a wave is at the same row (they start together and step together), so a row's
// address splits into a wave-uniform part -- row base + block offset, kept in SGPRs as a buffer descriptor -- a
// per-quantity scalar offset (q*R elements) and a per-lane constant (lane*sizeof(T)).  That removes the per-lane
// 64-bit address arithmetic of write_row (seven v_mad_u64_u32 / v_lshl_add_u64 chains per step) and frees their
// registers.  RTMI_ROW_STORE_AUX sets the stores' cache policy (0 plain, 2 nt, 16 sc1, 18 sc1 nt).
#ifndef RTMI_TILE_WAVES
#define RTMI_TILE_WAVES 4      // waves per SIMD the LDS-tile variant of k_advance is built for (the light methods)
#endif
// op2 and op6 (one field lookup per step, no golden section, no curvature terms) fit one more wave per SIMD than the rest;
// with the cell-polynomial lookup op1/7/8 do too (they spill 60-90 bytes per lane at 128 VGPRs and are still 8-10 % faster at
// four waves than at three: op1 8.97 -> 8.03 ms)
constexpr bool light_method(int m) { return m == 2 || m == 6 || (RTMI_POLY && (m == 1 || m == 7 || m == 8)); }
// op4 and the golden-section methods carry the most state: their tile builds keep two waves per SIMD
constexpr bool heavy_method(int m) { return m == 4 || m == 5 || m >= 9; }
#ifndef RTMI_GLOBAL_WAVES
#define RTMI_GLOBAL_WAVES 3    // waves per SIMD the fp64 global-gather builds are compiled for
#endif
#ifndef RTMI_GOLD_WAVES
#define RTMI_GOLD_WAVES 3      // waves per SIMD the golden-section builds (op5/9/10/11, global gather) are compiled for
#endif
#ifndef RTMI_SLICED_SLEEP
#define RTMI_SLICED_SLEEP 32   // s_sleep argument (x64 clocks) between two polls of a bundle's slice counter
#endif
#ifndef RTMI_F32_SLICED_WAVES
#define RTMI_F32_SLICED_WAVES 5   // the fp32 k_advance fits five waves per SIMD by itself (94 VGPRs); the sliced build has to be told
#endif
#ifndef RTMI_F32_WAVES
#define RTMI_F32_WAVES 4       // waves per SIMD the fp32 builds of k_advance are compiled for
#endif
#ifndef RTMI_STEP_PAIRS
#define RTMI_STEP_PAIRS 2      // 1: op2/op6 fp64, 2: every fast-form build -- two steps per loop iteration (see advance_loop)
#endif
#ifndef RTMI_ROW_STORE_AUX
#define RTMI_ROW_STORE_AUX 2   // nt: rows are written once and never read by the kernel (measured 3-5 % over plain stores)
#endif
typedef unsigned rt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rt_u32x2, v), rs, voff, soff, RTMI_ROW_STORE_AUX);
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, RTMI_ROW_STORE_AUX);
}
// A pointer every lane of the wave holds the same value of, pinned to scalar registers.  The compiler keeps the row
// pointers of advance_loop in VGPRs (their loop-carried updates get moved to the VALU) and would otherwise wrap every
// buffer store in a readfirstlane/compare waterfall loop: 6 VALU + 5 SALU instructions per store for one iteration.
template <typename T>
__device__ __forceinline__ T* wave_uniform_ptr(T* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long)hi << 32) | lo);
}
// rowp / nrowp: this block's slice of the current row of s_ray / n_ray (wave-uniform pointers the loop advances)
template <typename T>
__device__ __forceinline__ void write_row_uniform(const BatchDev<T>& a, T* rowp, T* nrowp, int voff, const rt::Ray<T>& r) {
    int qR = (int)(a.R * (long)sizeof(T));                              // byte distance between quantities (< 2^31 / 6: pick_advance)
#ifndef RTMI_NO_QR_LAUNDER
    // the multiples of qR are formed here, per row, with one scalar instruction each: hoisted out of the step loop they
    // do not fit the scalar registers and come back through v_readlane + 5 wait states apiece
    asm volatile("" : "+s"(qR));
#endif
    rowp = wave_uniform_ptr(rowp);
    nrowp = wave_uniform_ptr(nrowp);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rowp, 0, 0x7fffffff, 0x00020000);
    row_store(rs, voff, 0, (T)r.x);
    row_store(rs, voff, qR, (T)r.y);
    row_store(rs, voff, 2 * qR, r.mx);
    row_store(rs, voff, 3 * qR, r.my);
    row_store(rs, voff, 4 * qR, (T)r.tt);
    row_store(rs, voff, 5 * qR, (T)r.th);
    if (nrowp) {
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(nrowp, 0, 0x7fffffff, 0x00020000);
        row_store(rn, voff, 0, r.nray);
    }
}

// The step loop of k_advance.  UROW: rows are recorded and every live lane of the wave is at the same row, so the
// row counter lives in scalar registers and rows go out through write_row_uniform; otherwise per-lane bookkeeping
// (also when nothing is recorded).
template <typename T, int METHOD, bool ISO, typename G, bool UROW, bool COH = false>
__device__ __forceinline__ void advance_loop(const BatchDev<T>& a, const rt::Consts<T>& K, G& gather, rt::Ray<T>& r, long k, int& i,
                                             bool& alive, int max_size, int nsteps, long blk) {
    const bool RECORD = a.stride != 0;
    // the record stride as an opaque scalar-register value: left as a kernel argument the compiler re-reads it from the
    // argument segment for every recorded row (s_load + s_waitcnt lgkmcnt(0): a scalar-cache round trip per row)
    int stride = a.stride;
    asm volatile("" : "+s"(stride));
    int until = 0;
    long row = 0;
    // UROW: scalar row bookkeeping common to the live lanes: steps until the next recorded row, rows that may still be
    // written, and this block's slice of the current row (advanced by one row of 6*R / R values per recorded row)
    int rows_left = 0;
    T *rowp = nullptr, *nrowp = nullptr;
    const int voff = (int)(threadIdx.x * sizeof(T));
    if (UROW && RECORD) {
        const int i0 = __builtin_amdgcn_readfirstlane(rt::wave_max_i(alive ? i : 0));
        const int row0 = i0 / stride;
        until = stride - (i0 % stride);
        rows_left = (int)(a.rec_rows - 1 - row0 < 0x7fffffffL ? a.rec_rows - 1 - row0 : 0x7fffffffL);
        rowp = a.s_ray + (size_t)row0 * 6 * a.R + blk;
        nrowp = a.n_ray ? a.n_ray + (size_t)row0 * a.R + blk : nullptr;
    } else if (RECORD && alive) {
        until = stride - (i % stride);   // steps until the next recorded row
        row = i / stride;
    }
    // critical rays: the step itself ends a ray whose hover sum passed PolyGather::hov_limit (rt::hover_update: `inside` comes back false and
    // the sum +inf); which of the two it was is looked at where a ray's state is stored, off the loop's hot path
    constexpr bool HOV = rt::ReportsSteep<G>::value && rt::RotatesUnit<T, METHOD>::value;
    // one DELTA_S step of every lane; false once no lane of the wave is live
    auto one_step = [&]() -> bool {
        if (rt_ballot(alive) == 0ull) return false;
        const bool active = alive;
        ++i;
        const bool inside = rt::ray_step<T, METHOD, ISO>(a.F, K, gather, active, r, i);
        if (UROW && RECORD) {
            if (--until == 0) {
                until = stride;
                rowp += 6 * a.R;
                if (nrowp) nrowp += a.R;
                if (--rows_left >= 0 && active) write_row_uniform(a, rowp, nrowp, voff, r);
            }
        }
        if (active) {
            if (!UROW && RECORD) {
                if (--until == 0) {
                    until = stride;
                    ++row;
                    if (row < a.rec_rows) write_row(a, row, k, r);
                }
            }
            alive = inside && (i + 1 < max_size);
            if (!alive) {
                if constexpr (rt::IsExact<T, METHOD>::value) rt::ex::finish_state<rt::base_method(METHOD)>(a.F, gather, r);   // the gradient at the end point, if the flat path left it out
                const BatchDev<T> ab = rare_batch(a);
                if constexpr (HOV) {
                    if (r.hov == INFINITY) {      // a critical ray (a few hundred of a million on the interface fan, none elsewhere)
                        store_ray<T, METHOD, COH, true>(ab, k, r, i, false);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (!push_critical(ab, k, i)) {          // queue full: the ray carries on in the fused form (counted)
                            r.hov = -INFINITY;
                            alive = !rt::outside(K, r) && (i + 1 < max_size);
                            if (alive) st_state<COH>(ab.alive + k, (unsigned char)1);
                        }
                        return true;
                    }
                }
                store_ray<T, METHOD, COH, HOV>(ab, k, r, i, false);
            }
        }
        // (k_advance_lat) where the next step will look the field up if the ray goes on as it goes now: a lane about to enter
        // another cell starts that cell's loads here.  (Issued BEFORE the step's row stores instead -- the memory counter is in
        // order -- the vert configurations lose 3-10 %: cfg2 full 2.47 -> 2.71 ms, strong8 full 3.24 -> 3.33; GPU call 49.)
        if constexpr (rt::HasPrefetch<G>::value) gather.prefetch(a.F, alive, (T)r.x + r.ux * K.step, (T)r.y + r.uy * K.step);
        return true;
    };
    if constexpr (RTMI_STEP_PAIRS == 2 ? !rt::IsExact<T, METHOD>::value : (RTMI_STEP_PAIRS && light_method(METHOD) && sizeof(T) == 8)) {
        // Two steps per loop iteration: the second step's results land in the registers the first step's inputs left, so
        // the ~15 register copies that rotate the new state into the loop-carried registers every step disappear.
        for (int it = 0; it < nsteps; it += 2) {
            if (!one_step()) break;
            if (it + 1 >= nsteps) break;
            if (!one_step()) break;
        }
    } else {
        for (int it = 0; it < nsteps; ++it)
            if (!one_step()) break;
    }
}

template <typename T, int METHOD, bool ISO, bool LDS, bool VAR, bool COH = false, int PH = RTMI_TILE_PHASES, bool NOFLAT = false>
__device__ __forceinline__ bool advance_bundle(const BatchDev<T>& a, T* lds, long blk, int nsteps);

// Which 256-ray bundle a hardware block takes.  Blocks are dealt to the eight XCDs round-robin (block h runs on XCD h % 8,
// as the h / 8-th block there), each XCD with an L2 of its own; neighbouring bundles of a fan walk through the same cells of
// the field, so RTMI_XCD_GROUP consecutive bundles go to ONE XCD (its L2 then serves the second to G-th from the first's
// misses) while the groups still interleave over the XCDs (a contiguous eighth of the fan per XCD is badly balanced:
// DESIGN.md 5.1).  Identity for the blocks past the last whole round of 8 x G.
#ifndef RTMI_XCD_GROUP
#define RTMI_XCD_GROUP 8     // measured (A/B, one session): interface 24.6 -> 23.6 ms at 8, 24.0 at 4; vert, fisheye, fp32 unchanged
#endif
__device__ __forceinline__ unsigned xcd_grouped_block(unsigned h, unsigned nblocks) {
#if RTMI_XCD_GROUP > 1
    constexpr unsigned G = RTMI_XCD_GROUP, ROUND = 8u * G;
    if (h < nblocks / ROUND * ROUND) {
        const unsigned xcd = h & 7u, idx = h >> 3;
        return ((idx / G) * 8u + xcd) * G + idx % G;
    }
#endif
    return h;
}

// The loop at :866-879: one lane per ray, state in registers for up to nsteps DELTA_S steps.
// ISO (gamma == 1) drops the anisotropic factor's dead arithmetic; LDS selects the wave-private field tile
// (rt::LdsGather) over per-lookup global gathers.  Results are bit-identical across all four variants.
// Register budgets (RTMI_*_WAVES): the fp64 tile variant fits four waves per SIMD (128 VGPRs, the window read and summed
// row by row), the fp64 global-gather variants three (137-168 VGPRs, gathers in two halves); none of them spills.
// Every lane runs every iteration until no lane of its wave is active; a ray's state is stored the moment it
// terminates (or when the launch's step budget ends), so idle lanes never write.
template <typename T, int METHOD, bool ISO, bool LDS, bool VAR, bool NOFLAT = false>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? RTMI_F32_WAVES : LDS ? (light_method(METHOD) ? RTMI_TILE_WAVES : heavy_method(METHOD) ? 2 : 3) : ((METHOD == 5 || METHOD >= 9) ? RTMI_GOLD_WAVES : RTMI_GLOBAL_WAVES))
void k_advance(BatchDev<T> a, int nsteps) {
    __shared__ __attribute__((aligned(16))) T lds[kernel_lds_elems<T, METHOD, LDS>()];
    if (a.prio > 0) __builtin_amdgcn_s_setprio(3);     // the re-trace of a few hundred critical rays beside the main kernel's waves
    unsigned bundle = xcd_grouped_block(blockIdx.x, gridDim.x);
    if constexpr (!NOFLAT && sizeof(T) == 8 && rt::rotating_method(METHOD)) {
        // a re-run batch that handed critical rays over last time starts with THEIR bundles (Retrace::rot): the re-trace is one long
        // dependent chain per ray and ends the pass the later the later it starts
        bundle += a.blk_rot;
        bundle = bundle >= gridDim.x ? bundle - gridDim.x : bundle;
    }
    advance_bundle<T, METHOD, ISO, LDS, VAR, false, RTMI_TILE_PHASES, NOFLAT>(a, lds, (long)bundle * blockDim.x, nsteps);
}
// The kernel built for FEW waves: a batch of <= 2 waves per SIMD (cfg2's 65 536 rays: one) has nothing to hide a step's
// dependent chain behind -- 1 800 cycles per step at one wave per SIMD against 545 of issue -- so this build spends registers
// on latency (launch bound: two waves per SIMD): every LANE keeps its own cell's 36 polynomial coefficients in vector registers
// and the step loop starts the next cell's loads a step ahead of their use (rt::PolyLaneKept, round 5; 165 VGPRs); a lookup in
// the kept cell touches no memory.  Rounds 3-4 kept the WAVE's cell (rt::PolyGather, CACHED: 218 VGPRs; lanes in other cells
// loaded theirs at the point of use): A/B in one session, profiles/r05_ab_lat_lanekept.txt -- cfg2 full / none 2.66 -> 2.49 /
// 2.26 -> 2.11 ms, 32 768 rays full 2.93 -> 2.46, the fisheye shard of 8 4.78 -> 4.06, strong8 full 3.30 -> 3.21; strong8
// without the record loses 5 % (2.27 -> 2.39).  Same arithmetic as every other build: same bits.
// NOFLAT: for a field whose flat-cell map is empty (vert_heterogeneous, fisheye): neither the map's tests nor the hover sum are compiled in
template <typename T, int METHOD, bool ISO, bool NOFLAT = false>
__global__ __launch_bounds__(256, 2) void k_advance_lat(BatchDev<T> a, int nsteps) {
    __shared__ __attribute__((aligned(16))) T lds[uses_tile<T, METHOD, true>() ? 4 * rt::LdsGather<T, 1>::ELEMS : 2];
    advance_bundle<T, METHOD, ISO, true, false, false, 1, NOFLAT>(a, lds, (long)blockIdx.x * blockDim.x, nsteps);
}
template <typename T, int METHOD, bool ISO, bool LDS, bool VAR, bool COH, int PH, bool NOFLAT>
__device__ __forceinline__ bool advance_bundle(const BatchDev<T>& a, T* lds, long blk, int nsteps) {
    typename GatherOf<T, METHOD, LDS, PH, NOFLAT>::type gather;
    gather_init<T, LDS>(gather, lds);
    if constexpr (rt::IsExact<T, METHOD>::value && rt::ex::flat_shortcut<rt::base_method(METHOD)>()) gather_flat_bound<T>(gather, a);
    if constexpr (rt::IsExact<T, METHOD>::value) {
        if constexpr (rt::ex::inline_sincos(rt::base_method(METHOD)) || (METHOD & rt::kFastField) != 0) rt::ex::stage_sincos_tab();   // glibc's table into LDS (rt_exact.h)
    }
    const long k = blk + threadIdx.x;
    rt::Ray<T> r;
    int i = 0;
    bool alive = k < a.R && ld_state<COH>(a.alive + (k < a.R ? k : 0));
    // VAR: every ray carries its own DELTA_S and max_size (the calibration sweep as one candidate x ray batch)
    rt::Consts<T> K = a.K;
    int max_size = a.max_size;
    if (VAR && a.vstep && k < a.R) { K.step = a.vstep[k]; K.step2h = a.vstep2h[k]; K.step2 = K.step2h * T(2); max_size = a.vmax[k]; }
    constexpr bool HOV = rt::ReportsSteep<decltype(gather)>::value && rt::RotatesUnit<T, METHOD>::value;
    if constexpr (HOV) gather.hov_limit = a.hov_limit / (float)K.step;       // critical rays: the hover sum's limit in units of steepness (+inf: nothing is handed over)
    if (alive) load_ray<T, METHOD, ISO, COH, HOV>(a, k, r, i);
    else idle_ray(a, r);
    // Rows are recorded through the wave-uniform descriptor path (UROW) by every build except the VAR one: the host
    // launches a non-VAR build only while every live ray of the batch is at the same row (always, unless
    // rtmi_batch_set_state gave rays rows of their own) and a row's six quantities lie within 31-bit byte offsets of each
    // other (pick_advance); the VAR build keeps the per-lane row bookkeeping.
    // Drain the state loads here: a load still pending at the loop header stays "pending" in the compiler's wait-count
    // model around the back edge, and every first use in the loop then waits for vmcnt(0), i.e. for the previous
    // step's row stores to be acknowledged.
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), expcnt and lgkmcnt untouched
    advance_loop<T, METHOD, ISO, decltype(gather), !VAR, COH>(a, K, gather, r, k, i, alive, max_size, nsteps, blk);
    if (alive) {
        if constexpr (rt::IsExact<T, METHOD>::value) rt::ex::finish_state<rt::base_method(METHOD)>(a.F, gather, r);
        store_ray<T, METHOD, COH, HOV>(a, k, r, i, true);
    }
    return alive;
}

// Time-sliced bundles on persistent blocks (launch_mode 2).  A launch of k_advance gives every 256-ray bundle a block
// slot for its whole life; when the long bundles of a fan are a little more than a whole multiple of the slots (cfg3:
// 2 060 bundles of 3 039 rows for 1 024 slots) the last round runs on a nearly empty chip.  Here as many blocks as fit
// the device serve one FIFO of bundles: entry i < NB is bundle i (implicit), later entries are bundles pushed back by the
// block that ran a slice of them and found rays still alive.  A slice is `slice` DELTA_S steps (the first two of a bundle
// are 4 and 2 slices long: balance is decided at the end of a fan's life and every slice costs an inter-slice latency)
// from the state the previous slice stored -- the arithmetic of rtmi_step repeated, so the results are the bits of every
// other mode.  A bundle is in the queue at most once, so no two blocks ever hold it; its state goes through device-scope
// accesses (ld_state / st_state) and a block drains its own stores (vmcnt(0) + barrier) before it pushes the bundle --
// no whole-L2 write-back or invalidate (the agent-scope release/acquire fences of a first version cost 31.6 vs 18.4 ms
// with the full record).  ctl: [0..1] head, [2..3] pushed, [4..5] finished (64-bit counters), [6] stalled, [8..] entries
// (64-bit: slices done so far << 32 | bundle + 1; 0 = not written yet).
// Exit, reached by every block: its entry index is past everything that was or will be pushed (finished == NB + pushed,
// read in that order: then nothing is in flight that could push), or past the queue's capacity, or a wait ran out.
// (An earlier version handed out tickets for every (bundle, pass) pair: blocks racing through the tickets of dead
// bundles cost 0.5 ms on a balanced fan.)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) {
    return uniform_u64(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// 100 MHz constant counter (s_memrealtime): the wait bound below is wall-clock, not a poll count
__device__ __forceinline__ unsigned long long realtime_ticks() { return __builtin_amdgcn_s_memrealtime(); }
// Compiler-only fence: the hand-over below orders its memory operations for the HARDWARE with device-scope atomics to the
// coherence point, s_waitcnt vmcnt(0) and the block barrier; this keeps the optimizer from moving state accesses across the
// entry load / entry store (s_waitcnt is not a memory operation to it, and the barrier's fence is workgroup scope).
__device__ __forceinline__ void compiler_fence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }
template <typename T, int METHOD, bool ISO, bool LDS, bool NOFLAT = false>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? RTMI_F32_SLICED_WAVES : LDS ? (light_method(METHOD) ? RTMI_TILE_WAVES : heavy_method(METHOD) ? 2 : 3) : ((METHOD == 5 || METHOD >= 9) ? RTMI_GOLD_WAVES : RTMI_GLOBAL_WAVES))
void k_advance_sliced(BatchDev<T> a, int slice, unsigned long long capacity, unsigned long long* ctl, unsigned long long timeout_ticks) {
    __shared__ __attribute__((aligned(16))) T lds[kernel_lds_elems<T, METHOD, LDS>()];
    __shared__ unsigned long long s_entry;
    const unsigned long long NB = (unsigned long long)((a.R + 255) / 256);
    unsigned long long* head = ctl;
    unsigned long long* pushed = ctl + 1;
    unsigned long long* finished = ctl + 2;
    unsigned long long* stalled = ctl + 3;
    unsigned long long* entries = ctl + 4;                   // [capacity]: entry NB + j
    const unsigned long long kStop = ~0ull;
    // The block's bookkeeping is done by wave 0 with all of its lanes (every lane the same address and value; an atomic
    // add is 1 from lane 0 and 0 from the others): a branch on the wave index is scalar, whereas `if (threadIdx.x == 0)`
    // in front of a barrier had the compiler split this loop into a lane-0 loop around an other-lanes loop, with the
    // barriers in the inner one -- lane 0 never came back for a second entry.
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
    const unsigned long long one = (threadIdx.x & 63) == 0 ? 1ull : 0ull;
    bool running = true;
    while (running) {
        if (wave0) {
            const unsigned long long i = uniform_u64(atomicAdd(head, one));          // lane 0's return value is the index
            unsigned long long e = kStop;
            if (ld_u64(stalled) == 0ull) {
                if (i < NB) {
                    e = i + 1ull;                                                     // slices done: 0
                } else if (i - NB < capacity) {
                    e = ld_u64(entries + (i - NB));
                    // Wait for the entry.  Blocks past the queue's tail sit here until the last live bundles finish, which
                    // may legitimately take as long as the longest slice of the slowest method; the wait is abandoned only
                    // after timeout_ticks of wall-clock time in which NOTHING moved (no slice finished, nothing pushed) --
                    // the host sizes that from the longest possible slice (sliced_timeout_ticks) -- and the launch then
                    // reports RTMI_ERR_STATE instead of hanging.
                    unsigned long long f0 = ~0ull, p0 = ~0ull, t0 = realtime_ticks();
                    while (e == 0ull) {
                        const unsigned long long f = ld_u64(finished);
                        const unsigned long long p = ld_u64(pushed + (f >> 63));     // address depends on f: read after it
                        if (f == NB + p && i >= NB + p) { e = kStop; break; }         // nothing in flight, nothing left
                        const unsigned long long now = realtime_ticks();
                        if (f != f0 || p != p0) { f0 = f; p0 = p; t0 = now; }         // progress somewhere: start over
                        else if (now - t0 > timeout_ticks) break;
                        __builtin_amdgcn_s_sleep(RTMI_SLICED_SLEEP);
                        e = ld_u64(entries + (i - NB));
                    }
                    if (e == 0ull) { __hip_atomic_store(stalled, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); e = kStop; }
                } else {
                    __hip_atomic_store(stalled, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // queue capacity (never expected)
                }
            }
            compiler_fence();                                // no state load of the bundle moves above the entry load
            s_entry = e;
        }
        __syncthreads();
        // block-uniform values read from LDS are pinned to scalar registers: the loop and its barriers must be uniform
        // control flow for the compiler too
        const unsigned long long e = uniform_u64(s_entry);
        __syncthreads();                                     // s_entry is rewritten next round
        if (e == kStop) {
            running = false;
        } else {
            const long bundle = (long)((e & 0xffffffffull) - 1ull);
            const unsigned k = (unsigned)(e >> 32);           // slices this bundle has had
            const int nsteps = k == 0u ? 4 * slice : k == 1u ? 2 * slice : slice;
            compiler_fence();
            const bool alive = advance_bundle<T, METHOD, ISO, LDS, false, true, RTMI_TILE_PHASES, NOFLAT>(a, lds, bundle * 256, nsteps);
            compiler_fence();                                // no state store of the bundle moves below the entry store
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this lane's state stores are acknowledged
            const int any = __builtin_amdgcn_readfirstlane(__syncthreads_or(alive));
            if (wave0) {
                if (any) {
                    const unsigned long long slot = uniform_u64(atomicAdd(pushed, one));
                    compiler_fence();
                    if (slot < capacity)
                        __hip_atomic_store(entries + slot, ((unsigned long long)(k + 1u) << 32) | (unsigned long long)(bundle + 1),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_s_waitcnt(0x0F70);      // the entry before this slice counts as finished
                }
                atomicAdd(finished, one);
            }
        }
    }
}

// Persistent waves with lane refill (launch_mode 1): every wave draws rays from a device-side queue
// (counters[2] = next unclaimed ray).  Whenever at least `refill_min` of its lanes hold a terminated ray the
// wave compacts: a 64-bit ballot marks the dead lanes, one lane claims that many rays with a single atomic,
// and each dead lane takes ray base + (number of dead lanes below it) (mbcnt prefix).  A ray's arithmetic
// never depends on its lane or wave mates, so results are bit-identical to k_advance.
// Exit: the queue is exhausted and no lane is live -- reached by every wave because each ray takes at most
// max_size steps and the queue only advances.
#ifndef RTMI_REFILL_WAVES
#define RTMI_REFILL_WAVES 3
#endif
template <typename T, int METHOD, bool ISO, bool LDS>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 4 : light_method(METHOD) ? RTMI_REFILL_WAVES : 2) void k_trace_refill(BatchDev<T> a, int refill_min, int chunk) {
    __shared__ __attribute__((aligned(16))) T lds[kernel_lds_elems<T, METHOD, LDS>()];
    typename GatherOf<T, METHOD, LDS>::type gather;
    gather_init<T, LDS>(gather, lds);
    if constexpr (rt::IsExact<T, METHOD>::value && rt::ex::flat_shortcut<rt::base_method(METHOD)>()) gather_flat_bound<T>(gather, a);
    if constexpr (rt::IsExact<T, METHOD>::value) {
        if constexpr (rt::ex::inline_sincos(rt::base_method(METHOD)) || (METHOD & rt::kFastField) != 0) rt::ex::stage_sincos_tab();
    }
    const bool RECORD = a.stride != 0;
    constexpr bool RHOV = rt::ReportsSteep<typename GatherOf<T, METHOD, LDS>::type>::value && rt::RotatesUnit<T, METHOD>::value;
    rt::Consts<T> K = a.K;
    if constexpr (RHOV) gather.hov_limit = a.hov_limit / (float)K.step;
    const unsigned lane = threadIdx.x & 63;
    rt::Ray<T> r;
    idle_ray(a, r);
    long k = 0, row = 0;
    int i = 0, until = 0;
    bool alive = false;
    bool exhausted = false;  // wave-uniform
    for (;;) {
        unsigned long long live_mask = rt_ballot(alive);
        const int n_dead = 64 - __popcll(live_mask);
        if (!exhausted && (n_dead >= refill_min || live_mask == 0)) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&a.counters[2], (unsigned long long)n_dead);
            base = __shfl(base, 0, 64);
            const unsigned long long dead_mask = ~live_mask;
            const unsigned slot = __builtin_amdgcn_mbcnt_hi((unsigned)(dead_mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)dead_mask, 0u));
            if (!alive) {
                const long kk = (long)(base + slot);
                if (kk < a.R && a.alive[kk]) {
                    k = kk;
                    load_ray<T, METHOD, ISO, false, RHOV>(a, k, r, i);
                    until = RECORD ? a.stride - (i % a.stride) : 0;
                    row = RECORD ? i / a.stride : 0;
                    alive = true;
                }
            }
            exhausted = base + (unsigned long long)n_dead >= (unsigned long long)a.R;
            live_mask = rt_ballot(alive);
            __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): no state load pending into the step loop (see k_advance)
        }
        if (live_mask == 0) {
            if (exhausted) break;
            continue;  // the claimed rays were all finished already; claim again (the queue advanced)
        }
        for (int it = 0; it < chunk; ++it) {
            if (rt_ballot(alive) == 0ull) break;
            const bool active = alive;
            ++i;
            const bool inside = rt::ray_step<T, METHOD, ISO>(a.F, K, gather, active, r, i);
            if (active) {
                if (RECORD) {
                    if (--until == 0) {
                        until = a.stride;
                        ++row;
                        if (row < a.rec_rows) write_row(a, row, k, r);
                    }
                }
                alive = inside && (i + 1 < a.max_size);
                if constexpr (RHOV) {
                    if (!alive && r.hov == INFINITY) {          // a critical ray: see advance_loop
                        store_ray<T, METHOD, false, true>(a, k, r, i, false);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (!push_critical(a, k, i)) {
                            r.hov = -INFINITY;
                            alive = !rt::outside(K, r) && (i + 1 < a.max_size);
                            if (alive) a.alive[k] = 1;
                        }
                        continue;
                    }
                }
                if (!alive) {
                    if constexpr (rt::IsExact<T, METHOD>::value) rt::ex::finish_state<rt::base_method(METHOD)>(a.F, gather, r);
                    store_ray<T, METHOD, false, RHOV>(a, k, r, i, false);
                }
            }
        }
    }
}

template <typename T> __device__ __forceinline__ int max_size_of(const BatchDev<T>& a, long k) { return a.vmax ? a.vmax[k] : a.max_size; }

// index of slot k's ray in the caller's order
template <typename T> __device__ __forceinline__ long out_index(const BatchDev<T>& a, long k) { return a.perm ? (long)a.perm[k] : k; }

template <typename T> __global__ void k_pack_d_ray(BatchDev<T> a, double* out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    const long o = out_index(a, k);
    out[o] = a.acc(4)[k]; out[a.R + o] = a.acc(3)[k]; out[2 * a.R + o] = (double)a.istep[k];  // :888-890
}
template <typename T> __global__ void k_pack_final(BatchDev<T> a, double* out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    rt::Ray<T> r;
    r.x = a.acc(0)[k]; r.y = a.acc(1)[k]; r.th = a.acc(2)[k]; r.n = a.aux(0)[k]; r.gx = a.aux(1)[k]; r.gy = a.aux(2)[k]; r.tt = a.acc(5)[k];
    if (a.rot) {   // the carried unit tangent is what the last recorded row's momenta were formed from
        r.ux = a.unit(0)[k]; r.uy = a.unit(1)[k];
        if (a.iso) rt::derive<T, true, true>(a.K, r); else rt::derive<T, false, true>(a.K, r);
    } else derive_rt(a, r);
    const double v[9] = {(double)r.x, (double)r.y, (double)r.th, (double)r.n, (double)r.gx, (double)r.gy,
                         (double)r.mx, (double)r.my, (double)r.tt};
    const long o = out_index(a, k);
#pragma unroll
    for (int q = 0; q < 9; q++) out[(size_t)q * a.R + o] = v[q];
}
__global__ void k_f32_to_f64(const float* in, double* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

struct rtmi_batch {
    const rtmi_field* field = nullptr;
    rtmi_params p{};
    int64_t R = 0;
    size_t esz = 8;
    hipStream_t stream = nullptr;
    void* state = nullptr;       // one slab: 6 fp64 accumulator arrays, then 3 (+4) dtype arrays
    int* istep = nullptr;
    unsigned char* alive = nullptr;
    void *s_ray = nullptr, *n_ray = nullptr;
    bool own_s = false, own_n = false;
    double* launch = nullptr;    // [3][R] x0, y0, theta0 (slot order)
    int* perm = nullptr;         // device [R] when sort_rays
    void *vstep = nullptr, *vstep2h = nullptr;   // device [R] dtype: per-ray DELTA_S, DELTA_S**2/2 (set_per_ray)
    int* vmax = nullptr;         // device [R]: per-ray max_size
    unsigned long long* counters = nullptr;  // device [2]
    unsigned long long* h_counters = nullptr;  // pinned host [2]
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_used = 0;
    size_t pass_ev_start = 0;    // events[pass_ev_start ..] belong to the present pass (since the last reset); earlier ones to passes before
    double kernel_ms = 0;
    double total_kernel_ms = 0;  // all advance launches since create (a reset does not clear it)
    uint64_t total_launches = 0;
    uint32_t launches = 0;
    const void* kfn = nullptr;
    const void* kfn_refill = nullptr;
    int persistent_blocks = 0;   // resident 256-thread blocks of the refill kernel on this device
    const void* kfn_sliced = nullptr;
    int sliced_blocks = 0;       // resident 256-thread blocks of the sliced kernel on this device
    unsigned long long* sliced_ctl = nullptr;   // launch_mode 2: head, pushed, finished, stalled, queue entries (k_advance_sliced)
    bool dirty = false;          // rows may hold data a re-run will not overwrite (set_state / set_per_ray since the last clear)
    bool dirty_state = false;    // rtmi_batch_set_state ran since create / reset
    // RTMI_LAUNCH_AUTO: kernel time of the last complete run from the launch conditions under each schedule
    // ([0] sliced, [1] plain; < 0: not measured yet), and what the last rtmi_run used
    double auto_ms[2][RTMI_AUTO_SAMPLES] = {{0, 0, 0}, {0, 0, 0}};
    int auto_n[2] = {0, 0};      // timed runs per schedule so far
    int auto_kept = -1;          // -1 while exploring; then 0 (sliced) or 1 (plain) for the rest of the batch's life
    double gold_sup[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // gold_sup_derivatives(gamma_step) for op10/op11
    int lat_simds = 0, lat_waves_per_simd = 0;       // SIMDs of the device (CUs x 4); > 0 once known (pick_advance's latency rule)
    int mode_used = RTMI_LAUNCH_PLAIN;
    uint32_t auto_fallbacks = 0; // RTMI_LAUNCH_AUTO: sliced launches that abandoned a wait and were finished by the plain kernel
    void* staging = nullptr;     // device scratch of the read / metric / set_state paths, grown on demand, freed with the batch
    size_t staging_bytes = 0;
    // rtmi_step_repeat: `count` launches of `nsteps` steps as one hipGraph (a chain of kernel nodes), kept while the same
    // kernel, step count and launch count are asked for
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    const void* graph_kfn = nullptr;
    int graph_nsteps = 0, graph_count = 0, graph_block = 0;
    struct Retrace* rt = nullptr;   // critical rays handed over to a reference-order re-trace (nullptr: this batch hands nothing over)
    bool is_retrace_sub = false;    // this batch IS such a re-trace batch (owned by another batch's Retrace)
};
static void drop_graph(rtmi_batch* b);

// ------------------------------------------------------------------ critical rays: the automatic re-trace in reference order
// The fused forms of op1/2/6/8 are ~1e-13 from the reference over a whole trajectory -- except on the handful of rays per
// million that run ALONG a sharp transition of the medium (the interface scenario's critical angle): those amplify a last-bit
// difference a million times, in the reference itself as much as here, and only the reference's own roundings
// reproduce its rows to 1e-9.  The fused kernels find them on the way (rt::hover_update: the steepness of the cells in which a
// ray runs nearly along the iso-lines, added up), stop them and push (row, slot) into a device queue (push_critical); the host,
// which is waiting for the launch anyway, watches the queue's count in pinned memory and hands the queued rays to a second,
// hidden batch of the same parameters in RTMI_ORDER_REFERENCE -- chunks of them as they arrive, on a high-priority stream
// BESIDE the main kernel: a few hundred rays are a few waves, bound by the latency of their own ~4 000 dependent steps, which
// the main kernel's remaining run time hides.  Rays are independent (RT_bench.py:807), so this changes no other ray's bits;
// when both have finished k_retrace_scatter copies the re-traced rays' rows and final state over the fused ones.
// Result: every ray of a default batch is within 1e-9 of the reference, the critical ones bit for bit (the oracle's bits).
struct Retrace {
    rtmi_batch* sub = nullptr;            // the hidden batch: cap slots of state and rows (its own kernels never run: k_retrace fills it)
    unsigned cap = 0;                     // queue entries = slots that can be re-traced per pass
    unsigned long long* rq = nullptr;     // device: [0] rays pushed, [1 .. cap] entries
    unsigned* host_count = nullptr;       // pinned host: [0] the flags counted so far, [2..3] scratch for reading rq[0], [4 + slot] the kernels' per-slot flags
    float* hov = nullptr;                 // device [R]: the main batch's hover sums (ray state)
    static constexpr int kAux = 4;
    hipStream_t aux[kAux] = {nullptr, nullptr, nullptr, nullptr};   // high priority, non-blocking: successive chunks run side by side
    hipEvent_t ev_main = nullptr, ev_aux[kAux] = {nullptr, nullptr, nullptr, nullptr};
    // RTMI_RETRACE_CUS = n > 0: n compute units are set aside for the re-trace -- the aux streams carry a CU mask of those n, and a
    // run with the host watching (rtmi_run) launches its main kernel on `masked`, a stream of the batch's own with the
    // complementary mask, between two events on the caller's stream
    hipStream_t masked = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    unsigned chunks = 0;                  // chunks launched since the last reset (chunk c goes to aux[c % kAux])
    unsigned launched = 0;                // slots handed to the sub-batch since the last reset
    unsigned scattered = 0;               // ... and copied back
    bool pending = false;                 // advance kernels ran since the queue was last drained
    unsigned long long* dbg = nullptr;    // RTMI_DEBUG: device [8] statistics of k_retrace
    unsigned overflow = 0;                // rays that found the queue full this pass (they stay fused)
    unsigned swept = 0;                   // rays whose fused tail hovered again and that were re-traced in reference order throughout
    uint64_t total = 0;                   // rays re-traced over the batch's life
    // Dispatch order learnt from the batch's first pass with critical rays (like RTMI_LAUNCH_AUTO's schedule: knowledge a re-run batch
    // keeps).  Hardware blocks are dispatched in index order and take the ray bundles in fan order, so critical rays in the
    // middle of a fan are found after 40 % of the kernel's time and their re-trace -- 15-20 ms of dependent steps per ray --
    // outlasts the kernel by as much.  rot = the first bundle behind the largest gap (circular) between bundles that held
    // critical rays: the plain kernel's block 0 starts there, every critical ray is handed over in the first milliseconds.
    unsigned rot = 0;
    bool rot_learned = false;
};
static int retrace_create(rtmi_batch* b, const double* x0, const double* y0, const double* theta0);
static void retrace_destroy(rtmi_batch* b);
static int retrace_reset(rtmi_batch* b);
static int retrace_drain(rtmi_batch* b, bool overlap);
static bool retrace_wanted(const rtmi_batch* b);
// every path that reads results first drains the queue (no-op unless kernels ran since)
#define RETRACE_FLUSH(b)                                             \
    do {                                                             \
        if ((b)->rt && (b)->rt->pending) {                           \
            const int rcf_ = retrace_drain((b), false);              \
            if (rcf_) return rcf_;                                   \
        }                                                            \
    } while (0)

// device scratch owned by the batch (one allocation reused by every read path instead of a hipMalloc/hipFree per call)
static int batch_staging(rtmi_batch* b, size_t bytes, void** out) {
    if (bytes > b->staging_bytes) {
        HIP_TRY(hipStreamSynchronize(b->stream));     // nothing may still use the old buffer
        (void)hipFree(b->staging);
        b->staging = nullptr; b->staging_bytes = 0;
        hipError_t e = hipMalloc(&b->staging, bytes);
        if (e != hipSuccess) return fail(RTMI_ERR_ALLOC, std::string("staging buffer: ") + hipGetErrorString(e));
        b->staging_bytes = bytes;
    }
    *out = b->staging;
    return RTMI_OK;
}

// The reference-order lookup's wave-uniform window (rt::ex::lookup_uniform) from two waves per SIMD on: 1024 SIMDs x 64 lanes x 2
// (FieldDev::window) when rtmi_params.field_path is 0 (auto); 3 asks for it whatever the size, 1 and 2 never use it.
// RTMI_WINDOW_MIN_RAYS overrides the size (0: always, a huge number: never) for A/B runs.
static long window_min_rays() {
    static const long v = [] { const char* e = getenv("RTMI_WINDOW_MIN_RAYS"); return e ? atol(e) : 131072L; }();
    return v;
}
// ... except for the golden-section methods on a fan that enters a new cell on (nearly) every step: the fisheye fan at its calibrated
// DELTA_S = 2 pi / 303 (two cells per step) runs op9 in 165 ms with the window and 48 without, op5 in 78 / 54 -- the other
// reference-order methods gain there as everywhere (op3 53 -> 46 ms, op7 37 -> 29, op6 in reference order 40 -> 32), and op5 / op9
// gain where a wave stays in its cell for several steps (vert_heterogeneous and interface: a quarter of a cell per step).  Measured,
// not understood: the kernels' vector-instruction counts are the same, the waves wait (profiles/r04q_fisheye_op9_none_pmc_summary.txt,
// taken with the window on; asking for all of the window's cache lines at once did not change it, profiles/r04_ab_variants_not_kept.txt).
static bool window_loses(const rtmi_batch* b) {
    const int m = b->p.method;
    if (m != 5 && m != 9 && m != 10 && m != 11) return false;
    const rtmi_field* f = b->field;
    return std::fabs(b->p.step) > 0.5 * std::fmin(f->hx, f->hy);
}
// rt::kHoverLimit (RTMI_HOVER_LIMIT overrides it, for calibration runs)
static float hover_limit() {
    static const float v = [] { const char* e = getenv("RTMI_HOVER_LIMIT"); return e ? (float)atof(e) : rt::kHoverLimit; }();
    return v;
}
template <typename T> static BatchDev<T> batch_dev(const rtmi_batch* b) {
    BatchDev<T> a;
    a.F = field_dev<T>(b->field, b->p.exact_basis);
    a.F.window = (b->p.field_path == 3 || (b->p.field_path == 0 && b->R >= window_min_rays() && !window_loses(b))) ? 1 : 0;
    const rtmi_params& p = b->p;
    a.K.step = (T)p.step;
    a.K.step2h = (T)(libm_square(p.step) / 2.0);    // numpy scalar step**2 is libm pow (:330); /2 is exact
    a.K.step2 = (T)libm_square(p.step);
    a.K.gamma = (T)p.gamma; a.K.g2m1 = (T)(p.gamma * p.gamma - 1.0);
    a.K.gamma_s = (T)p.gamma_step; a.K.g2m1_s = (T)(p.gamma_step * p.gamma_step - 1.0);
    for (int i = 0; i < 4; i++) a.K.box[i] = (T)p.box[i];
    for (int i = 0; i < 8; i++) a.K.gold_sup[i] = (T)b->gold_sup[i];
    a.R = b->R; a.max_size = p.max_size; a.stride = p.record_stride; a.rec_rows = p.rec_rows;
    const size_t R = (size_t)b->R;
    a.st = (double*)b->state; a.has_hist = p.method == 7;
    a.exact = p.dtype == RTMI_F64 && (rt::is_exact_method(p.method) || ref_order(p));
    a.iso = p.gamma == 1.0 && p.method < 10 && !a.exact;
    a.rot = rt::rotates_unit(p.method, p.dtype == RTMI_F64) && !a.exact;
    a.istep = b->istep; a.alive = b->alive;
    a.s_ray = (T*)b->s_ray; a.n_ray = (T*)b->n_ray;
    a.counters = b->counters;
    a.x0 = b->launch; a.y0 = b->launch + R; a.th0 = b->launch + 2 * R;
    a.perm = b->perm;
    a.vstep = (const T*)b->vstep; a.vstep2h = (const T*)b->vstep2h; a.vmax = b->vmax;
    a.hov = b->rt ? b->rt->hov : nullptr;
    a.hov_limit = b->rt ? hover_limit() : INFINITY;
    a.rq_cap = b->rt ? b->rt->cap : 0u;
    a.rq = b->rt ? b->rt->rq : nullptr;
    a.rq_host = b->rt ? b->rt->host_count + 4 : nullptr;
    a.prio = b->is_retrace_sub ? 1 : 0;
    a.gflat = b->field->gmax * 0x1p-72;
    a.blk_rot = (b->rt && !b->is_retrace_sub) ? b->rt->rot : 0u;
    return a;
}

// kernel variant tables: [kernel method][iso][lds].  Kernel methods: the step methods 1..11, then op1/2/6/7/8 in the reference's
// operation order (METHOD = method | rt::kRefOrder, fp64 only: the fp32 tables alias the ordinary builds there).
// Anisotropic-only methods (op10/op11) have no ISO build.
// Index 16: op7 with RTMI_ORDER_FAST_FIELD -- the reference-order step on the fast field lookup (rt::kFastField).
constexpr int kKernelMethods = 17;
constexpr int kmethod_of(int idx) {
    return idx < 11 ? idx + 1 : idx == 16 ? (7 | rt::kRefOrder | rt::kFastField) : (idx == 11 ? 1 : idx == 12 ? 2 : idx == 13 ? 6 : idx == 14 ? 7 : 8) | rt::kRefOrder;
}
static int kernel_index(int method, bool ref_order, bool fast_field) {
    if (!ref_order || rt::is_exact_method(method)) return method - 1;
    if (fast_field) return 16;
    return method == 1 ? 11 : method == 2 ? 12 : method == 6 ? 13 : method == 7 ? 14 : 15;
}
template <typename T> constexpr int km(int idx) { return sizeof(T) == 4 ? rt::base_method(kmethod_of(idx)) : kmethod_of(idx); }
constexpr bool iso_ok(int m) { return rt::base_method(m) < 10; }
#define RTMI_ADV_(T, I) \
    {{(const void*)k_advance<T, km<T>(I), false, false, false>, (const void*)k_advance<T, km<T>(I), false, true, false>}, \
     {(const void*)k_advance<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), false, false>, (const void*)k_advance<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), true, false>}}
// the wave-shared builds for a field without flat cells (NOFLAT; only the polynomial lookup has the map's tests): [kernel method][iso]
#define RTMI_ADVNF_(T, I) \
    {(const void*)k_advance<T, km<T>(I), false, true, false, uses_poly<T, km<T>(I)>()>, \
     (const void*)k_advance<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), true, false, uses_poly<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1)>()>}
#define RTMI_SLICEDNF_(T, I) \
    {(const void*)k_advance_sliced<T, km<T>(I), false, true, uses_poly<T, km<T>(I)>()>, \
     (const void*)k_advance_sliced<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), true, uses_poly<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1)>()>}
// ... and the per-lane-gather builds of op1/2/6/8 in reference order (kernel indices 11, 12, 13, 15) for a field whose map is empty: without
// the flat path of the reference-order step (rt::GlobalGather's FLATMAP).  Every other index: the ordinary build again.
constexpr bool ref1268(int idx) { return idx == 11 || idx == 12 || idx == 13 || idx == 15; }
#define RTMI_ADVNF0_(T, I) \
    {(const void*)k_advance<T, km<T>(I), false, false, false, sizeof(T) == 8 && ref1268(I)>, \
     (const void*)k_advance<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), false, false, sizeof(T) == 8 && ref1268(I) && iso_ok(km<T>(I))>}
#define RTMI_SLICEDNF0_(T, I) \
    {(const void*)k_advance_sliced<T, km<T>(I), false, false, sizeof(T) == 8 && ref1268(I)>, \
     (const void*)k_advance_sliced<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), false, sizeof(T) == 8 && ref1268(I) && iso_ok(km<T>(I))>}
#define RTMI_ADVVAR_(T, I) \
    {(const void*)k_advance<T, km<T>(I), false, false, true>, (const void*)k_advance<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), false, true>}
#define RTMI_REFILL_(T, I) \
    {{(const void*)k_trace_refill<T, km<T>(I), false, false>, (const void*)k_trace_refill<T, km<T>(I), false, true>}, \
     {(const void*)k_trace_refill<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), false>, (const void*)k_trace_refill<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), true>}}
#define RTMI_SLICED_(T, I) \
    {{(const void*)k_advance_sliced<T, km<T>(I), false, false>, (const void*)k_advance_sliced<T, km<T>(I), false, true>}, \
     {(const void*)k_advance_sliced<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), false>, (const void*)k_advance_sliced<T, (iso_ok(km<T>(I)) ? km<T>(I) : 1), iso_ok(km<T>(I)), true>}}
#define RTMI_ALL16_(X, T) X(T, 0), X(T, 1), X(T, 2), X(T, 3), X(T, 4), X(T, 5), X(T, 6), X(T, 7), X(T, 8), X(T, 9), X(T, 10), X(T, 11), X(T, 12), X(T, 13), X(T, 14), X(T, 15), X(T, 16)
template <typename T> static const void* sliced_fn(int ki, bool iso, bool lds, bool noflat) {
    static const void* const tab[kKernelMethods][2][2] = {RTMI_ALL16_(RTMI_SLICED_, T)};
    static const void* const tabnf[kKernelMethods][2] = {RTMI_ALL16_(RTMI_SLICEDNF_, T)};
    static const void* const tabnf0[kKernelMethods][2] = {RTMI_ALL16_(RTMI_SLICEDNF0_, T)};
    if (!lds && noflat) return tabnf0[ki][iso ? 1 : 0];
    return lds && noflat ? tabnf[ki][iso ? 1 : 0] : tab[ki][iso ? 1 : 0][lds ? 1 : 0];
}
template <typename T> static const void* advance_fn(int ki, bool iso, bool lds, bool noflat) {
    static const void* const tab[kKernelMethods][2][2] = {RTMI_ALL16_(RTMI_ADV_, T)};
    static const void* const tabnf[kKernelMethods][2] = {RTMI_ALL16_(RTMI_ADVNF_, T)};
    static const void* const tabnf0[kKernelMethods][2] = {RTMI_ALL16_(RTMI_ADVNF0_, T)};
    if (!lds && noflat) return tabnf0[ki][iso ? 1 : 0];
    return lds && noflat ? tabnf[ki][iso ? 1 : 0] : tab[ki][iso ? 1 : 0][lds ? 1 : 0];
}
// per-ray DELTA_S / max_size builds (global gather only): [kernel method][iso]
template <typename T> static const void* advance_var_fn(int ki, bool iso) {
    static const void* const tab[kKernelMethods][2] = {RTMI_ALL16_(RTMI_ADVVAR_, T)};
    return tab[ki][iso ? 1 : 0];
}
template <typename T> static const void* refill_fn(int ki, bool iso, bool lds) {
    static const void* const tab[kKernelMethods][2][2] = {RTMI_ALL16_(RTMI_REFILL_, T)};
    return tab[ki][iso ? 1 : 0][lds ? 1 : 0];
}
#undef RTMI_SLICED_
#undef RTMI_ADV_
#undef RTMI_ADVNF_
#undef RTMI_ADVNF0_
#undef RTMI_SLICEDNF0_
#undef RTMI_SLICEDNF_
#undef RTMI_ADVVAR_
#undef RTMI_REFILL_
#undef RTMI_ALL16_
// VRCP14PD's table (rt_rcp14_table.h) decoded once per device, for rt::ex::atan2_
__global__ void k_rcp14_init() {
    unsigned short v = RT_RCP14_T0;
    for (int k = 0; k < 65536; k++) {
        v = (unsigned short)(v - (unsigned short)((rt::ex::kRcp14Words[k >> 5] >> (2 * (k & 31))) & 3ull));
        rt::ex::g_rcp14[k] = v;
    }
}
// (per device, once: a mutex around a per-device flag -- two host threads creating batches at the same time decode once, and a
// failed decode is retried by the next create)
static int ensure_rcp14_table(hipStream_t st) {
    static std::mutex mu;
    static std::vector<char> done;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    try {
        if ((size_t)dev >= done.size()) done.resize((size_t)dev + 1, 0);
    } catch (const std::exception& e) {
        return fail(RTMI_ERR_ALLOC, std::string("rcp14 table: ") + e.what());
    }
    if (done[dev]) return RTMI_OK;
    hipLaunchKernelGGL(k_rcp14_init, dim3(1), dim3(1), 0, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    done[dev] = 1;
    return RTMI_OK;
}
// the fp64 batch runs rt_exact.h's arithmetic: always for op3/4/5/9/10/11, for the others when ref_order() says so
static bool batch_exact(const rtmi_batch* b) { return b->p.dtype == RTMI_F64 && (rt::is_exact_method(b->p.method) || ref_order(b->p)); }
static int batch_kernel_index(const rtmi_batch* b) { return kernel_index(b->p.method, ref_order(b->p), fast_field_order(b->p)); }
// field_path 0 (auto): which gather policy the step kernels are built with.
static bool use_lds_tile(const rtmi_batch* b) {
    if (b->p.field_path == 1) return false;
    if (b->p.field_path == 2) return true;
    if (b->p.field_path == 3) return !batch_exact(b) || fast_field_order(b->p);     // reference-order methods: the scalar-cache window (FieldDev::window)
    // Measured: the tile build wins for the fast-form methods in every record mode -- op2/op6 at four waves
    // per SIMD 12.5 vs 15.0 ms without recording and 19 vs 24 ms with the full record, op1/7/8 and fp32 by 0-2 %, a
    // shuffled fan 44 vs 50 ms; the reference-order fp64 methods (op3/4/5/9/10/11: 2-74 cost evaluations per step, more
    // live state) are 2-15 % faster on global gathers.
    return !batch_exact(b) || fast_field_order(b->p);
}
// rows can go out through the wave-uniform descriptor path: rays in lockstep (no rtmi_batch_set_state since the last
// reset) and 6 quantities x R values within 31-bit byte offsets
static bool uniform_rows_ok(const rtmi_batch* b) {
    return b->p.record_stride == 0 || (!b->dirty_state && (double)b->R * (double)b->esz * 6.0 < 2147483647.0);
}
// op2/op6 fp64 tile builds for few waves (k_advance_lat): [method 2 | 6][iso]
static const void* advance_lat_fn(int m, bool iso, bool noflat) {
    static const void* const tab[2][2][2] = {{{(const void*)k_advance_lat<double, 2, false, false>, (const void*)k_advance_lat<double, 2, false, true>},
                                              {(const void*)k_advance_lat<double, 2, true, false>, (const void*)k_advance_lat<double, 2, true, true>}},
                                             {{(const void*)k_advance_lat<double, 6, false, false>, (const void*)k_advance_lat<double, 6, false, true>},
                                              {(const void*)k_advance_lat<double, 6, true, false>, (const void*)k_advance_lat<double, 6, true, true>}}};
    return tab[m == 6 ? 1 : 0][iso ? 1 : 0][noflat ? 1 : 0];
}
static const void* pick_advance(const rtmi_batch* b) {
    const bool iso = b->p.gamma == 1.0 && b->p.method < 10, lds = use_lds_tile(b);
    // at most two waves per SIMD's worth of rays: the latency build (env RTMI_NO_LAT=1 keeps the throughput build, for A/B)
    if (lds && b->p.dtype == RTMI_F64 && (b->p.method == 2 || b->p.method == 6) && !b->vstep && uniform_rows_ok(b) &&
        !ref_order(b->p) && b->lat_waves_per_simd > 0 && (b->R + 63) / 64 <= (int64_t)2 * b->lat_simds && !getenv("RTMI_NO_LAT"))
        return advance_lat_fn(b->p.method, iso, b->field->flat_cells == 0 && b->field->steep_cells == 0);
    // the VAR build: per-ray DELTA_S / max_size when set, and per-lane row bookkeeping always
    if (b->vstep || !uniform_rows_ok(b))
        return b->p.dtype == RTMI_F64 ? advance_var_fn<double>(batch_kernel_index(b), iso) : advance_var_fn<float>(batch_kernel_index(b), iso);
    const bool noflat = b->field->flat_cells == 0 && b->field->steep_cells == 0;      // nothing in this field's map: the builds without its tests
    return b->p.dtype == RTMI_F64 ? advance_fn<double>(batch_kernel_index(b), iso, lds, noflat) : advance_fn<float>(batch_kernel_index(b), iso, lds, noflat);
}
// queue entries beyond the implicit first NB: every bundle is pushed back once per slice it survives
static unsigned long long sliced_capacity(const rtmi_batch* b, int slice) {
    const unsigned long long NB = ((unsigned long long)b->R + 255) / 256;
    const long rest = (long)b->p.max_size - 6L * slice;
    // + slack: at the end every idle block holds one index past the last entry
    return NB * (unsigned long long)(2 + (rest > 0 ? (rest + slice - 1) / slice : 0)) + 8192;
}
static int sliced_steps(const rtmi_batch* b) {
    const int s = b->p.slice_steps > 0 ? b->p.slice_steps : 512;   // 128..1024 measure alike
    return s < (1 << 20) ? s : (1 << 20);
}
// Wall-clock bound (100 MHz ticks) on a wait in which nothing moves: 2 s plus the longest slice any block can be in --
// min(4 * slice, max_size) steps at a generous per-step cost (golden-section methods evaluate 74 costs per step; a
// wave-step of op11 measures ~16 us at three waves per SIMD, of op6 ~0.4 us) -- so a long slice elsewhere is not a stall.
static unsigned long long sliced_timeout_ticks(const rtmi_batch* b, int slice) {
    const double steps = std::min(4.0 * (double)slice, (double)b->p.max_size);
    const int m = b->p.method;
    const double us_per_step = (m == 5 || m >= 9) ? 200.0 : 20.0;
    const double seconds = 2.0 + steps * us_per_step * 1e-6;
    return (unsigned long long)(seconds * 1e8);
}
// can this batch run the time-sliced schedule at all (queue within 256 MB, uniform DELTA_S, rays in lockstep)?
static bool sliced_feasible(const rtmi_batch* b) { return sliced_capacity(b, sliced_steps(b)) <= (1ull << 25); }
static const void* pick_sliced(const rtmi_batch* b) {
    const bool iso = b->p.gamma == 1.0 && b->p.method < 10, lds = use_lds_tile(b);
    const bool noflat = b->field->flat_cells == 0 && b->field->steep_cells == 0;
    return b->p.dtype == RTMI_F64 ? sliced_fn<double>(batch_kernel_index(b), iso, lds, noflat) : sliced_fn<float>(batch_kernel_index(b), iso, lds, noflat);
}
static const void* pick_refill(const rtmi_batch* b) {
    const bool iso = b->p.gamma == 1.0 && b->p.method < 10, lds = use_lds_tile(b);
    return b->p.dtype == RTMI_F64 ? refill_fn<double>(batch_kernel_index(b), iso, lds) : refill_fn<float>(batch_kernel_index(b), iso, lds);
}

// clear_traj: zero the trajectory arrays (np.zeros, :802-803).  A reset with unchanged launch conditions rewrites
// exactly the rows it wrote before, so the zeros past each ray's last row survive and need no second pass.
static int batch_init_state(rtmi_batch* b, bool clear_traj) {
    hipStream_t st = b->stream;
    const size_t R = (size_t)b->R;
    HIP_TRY(hipMemsetAsync(b->counters, 0, 4 * sizeof(unsigned long long), st));
    if (b->rt) { const int rcr = retrace_reset(b); if (rcr) return rcr; }
    if (clear_traj && b->s_ray) HIP_TRY(hipMemsetAsync(b->s_ray, 0, (size_t)b->p.rec_rows * 6 * R * b->esz, st));
    if (clear_traj && b->n_ray) HIP_TRY(hipMemsetAsync(b->n_ray, 0, (size_t)b->p.rec_rows * R * b->esz, st));
    b->dirty = false;
    const dim3 g((unsigned)((R + 255) / 256)), blk(256);
    if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_init<double>, g, blk, 0, st, batch_dev<double>(b));
    else hipLaunchKernelGGL(k_init<float>, g, blk, 0, st, batch_dev<float>(b));
    HIP_TRY(hipGetLastError());
    // the events of the pass before stay on the list (they are folded into total_kernel_ms at the next stats call or when the
    // list is full): a caller timing many passes reads the kernel time of all of them without a host sync per pass
    b->kernel_ms = 0; b->launches = 0; b->pass_ev_start = b->ev_used; b->dirty_state = false;
    return RTMI_OK;
}

RTMI_EXPORT void rtmi_batch_destroy(rtmi_batch* b) {
    if (!b) return;
    (void)hipStreamSynchronize(b->stream);
    (void)hipFree(b->state); (void)hipFree(b->istep); (void)hipFree(b->alive); (void)hipFree(b->launch);
    (void)hipFree(b->perm);
    (void)hipFree(b->vstep); (void)hipFree(b->vstep2h); (void)hipFree(b->vmax);
    (void)hipFree(b->counters); (void)hipFree(b->sliced_ctl); (void)hipFree(b->staging);
    if (b->h_counters) (void)hipHostFree(b->h_counters);
    if (b->own_s) (void)hipFree(b->s_ray);
    if (b->own_n) (void)hipFree(b->n_ray);
    retrace_destroy(b);
    drop_graph(b);
    for (auto& e : b->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete b;
}

RTMI_EXPORT int rtmi_batch_create(const rtmi_field* f, const rtmi_params* p, int64_t R, const double* x0,
                                  const double* y0, const double* theta0, void* stream, rtmi_batch** out) {
    ARG_TRY(f && p && x0 && y0 && theta0 && out, "rtmi_batch_create: null argument");
    ARG_TRY(R > 0 && R < (1ll << 31), "rtmi_batch_create: R must be in [1, 2^31)");
    ARG_TRY(p->method >= RTMI_OP_MIN && p->method <= RTMI_OP_MAX, "rtmi_batch_create: method must be 1..11");
    ARG_TRY(p->dtype == f->dtype, "rtmi_batch_create: params.dtype differs from the field's dtype");
    ARG_TRY(p->step > 0 && std::isfinite(p->step), "rtmi_batch_create: step must be finite and > 0");
    ARG_TRY(p->gamma > 0 && p->gamma_step > 0, "rtmi_batch_create: gamma must be > 0");
    ARG_TRY(p->max_size >= 2, "rtmi_batch_create: max_size must be >= 2");
    ARG_TRY(p->method != 7 || p->max_size >= 4, "rtmi_batch_create: op7 needs max_size >= 4 (two bootstrap rows)");
    ARG_TRY(p->record_stride >= 0, "rtmi_batch_create: record_stride < 0");
    ARG_TRY(p->box[1] > p->box[0] && p->box[3] > p->box[2], "rtmi_batch_create: empty box");
    ARG_TRY(p->launch_mode >= RTMI_LAUNCH_AUTO && p->launch_mode <= RTMI_LAUNCH_PLAIN,
            "rtmi_batch_create: launch_mode must be 0 (auto), 1 (refill), 2 (sliced) or 3 (plain)");
    ARG_TRY(p->slice_steps >= 0, "rtmi_batch_create: slice_steps must be >= 0");
    ARG_TRY(p->refill_min >= 0 && p->refill_min <= 64, "rtmi_batch_create: refill_min must be in [0, 64]");
    ARG_TRY(p->exact_basis == 0 || p->exact_basis == 1, "rtmi_batch_create: exact_basis must be 0 or 1");
    ARG_TRY(p->field_path >= 0 && p->field_path <= 3, "rtmi_batch_create: field_path must be 0 (auto), 1 (global), 2 (wave-shared: LDS tile) or 3 (wave-shared: scalar cache)");
    ARG_TRY(p->sort_rays == 0 || p->sort_rays == 1, "rtmi_batch_create: sort_rays must be 0 or 1");
    ARG_TRY(p->lazy_clear == 0 || p->lazy_clear == 1, "rtmi_batch_create: lazy_clear must be 0 or 1");
    ARG_TRY(p->no_n_ray == 0 || p->no_n_ray == 1, "rtmi_batch_create: no_n_ray must be 0 or 1");
    ARG_TRY(!(p->no_n_ray && p->ext_n_ray), "rtmi_batch_create: no_n_ray set together with ext_n_ray");
    ARG_TRY(p->reference_order >= 0 && p->reference_order <= 3, "rtmi_batch_create: reference_order must be 0 .. 3 (RTMI_ORDER_*)");
    ARG_TRY(!(p->reference_order == RTMI_ORDER_REFERENCE && p->dtype != RTMI_F64), "rtmi_batch_create: reference_order 1 needs an fp64 batch (the reference has no fp32)");
    ARG_TRY(p->no_retrace == 0 || p->no_retrace == 1, "rtmi_batch_create: no_retrace must be 0 or 1");
    DEVICE_TRY(f, "rtmi_batch_create");
    rtmi_batch* b = new (std::nothrow) rtmi_batch();
    if (!b) return fail(RTMI_ERR_ALLOC, "rtmi_batch_create: host allocation failed");
    b->field = f; b->p = *p; b->R = R; b->esz = p->dtype == RTMI_F64 ? 8 : 4; b->stream = (hipStream_t)stream;
    if (p->method >= 10) gold_sup_derivatives(p->gamma_step, b->gold_sup);
    if (batch_exact(b) && (p->method == 1 || p->method == 4 || p->method == 7 || p->method == 8)) {   // they call rt::ex::atan2_
        const int rct = ensure_rcp14_table(b->stream);
        if (rct) { delete b; return rct; }
    }
    if (b->p.record_stride > 0 && b->p.rec_rows <= 0)
        b->p.rec_rows = ((int64_t)p->max_size + p->record_stride - 1) / p->record_stride;
    if (b->p.record_stride == 0) b->p.rec_rows = 0;
    int rc = RTMI_OK;
    auto body = [&]() -> int {
        const size_t Rz = (size_t)R;
        const int naux = p->method == 7 ? 7 : (rt::rotates_unit(p->method, p->dtype == RTMI_F64) && !ref_order(*p)) ? 5 : 3;   // n gx gy + history | unit vector
        HIP_TRY(hipMalloc(&b->state, 6 * Rz * sizeof(double) + naux * Rz * b->esz));
        HIP_TRY(hipMalloc(&b->istep, Rz * sizeof(int)));
        HIP_TRY(hipMalloc(&b->alive, Rz));
        HIP_TRY(hipMalloc(&b->launch, 3 * Rz * sizeof(double)));
        HIP_TRY(hipMalloc(&b->counters, 4 * sizeof(unsigned long long)));
        HIP_TRY(hipHostMalloc(&b->h_counters, 2 * sizeof(unsigned long long)));
        if (b->p.record_stride > 0) {
            if (p->ext_s_ray) b->s_ray = p->ext_s_ray;
            else { HIP_TRY(hipMalloc(&b->s_ray, (size_t)b->p.rec_rows * 6 * Rz * b->esz)); b->own_s = true; }
            if (p->no_n_ray) b->n_ray = nullptr;
            else if (p->ext_n_ray) b->n_ray = p->ext_n_ray;
            else { HIP_TRY(hipMalloc(&b->n_ray, (size_t)b->p.rec_rows * Rz * b->esz)); b->own_n = true; }
        }
        std::vector<double> sorted;   // keeps the permuted launch conditions alive until the copies are done
        if (p->sort_rays) {
            // Coherence sort: rays that start in the same 16x16-cell block of the grid and leave in neighbouring
            // directions become neighbouring lanes, so a wave's lookups share cache lines / one LDS tile.
            std::vector<int> perm(Rz);
            std::iota(perm.begin(), perm.end(), 0);
            const double ibx = 1.0 / (16.0 * f->hx), iby = 1.0 / (16.0 * f->hy);
            auto block = [&](int k) {
                const double u = (x0[k] - f->ax) * ibx, v = (y0[k] - f->ay) * iby;
                const long bu = u >= 0 && u < 65535 ? (long)u : (u < 0 ? -1 : 65535);      // NaN -> 65535 (sorted last)
                const long bv = v >= 0 && v < 65535 ? (long)v : (v < 0 ? -1 : 65535);
                return bv * 65537 + bu;
            };
            std::vector<long> key(Rz);
            for (size_t k = 0; k < Rz; k++) key[k] = block((int)k);
            std::stable_sort(perm.begin(), perm.end(), [&](int i, int j) {
                if (key[i] != key[j]) return key[i] < key[j];
                const double a = theta0[i], c = theta0[j];
                if (a == a && c == c) return a < c;
                return a == a && c != c;       // NaNs last
            });
            sorted.resize(3 * Rz);
            for (size_t k = 0; k < Rz; k++) {
                sorted[k] = x0[perm[k]]; sorted[Rz + k] = y0[perm[k]]; sorted[2 * Rz + k] = theta0[perm[k]];
            }
            HIP_TRY(hipMalloc(&b->perm, Rz * sizeof(int)));
            HIP_TRY(hipMemcpyAsync(b->perm, perm.data(), Rz * sizeof(int), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->launch, sorted.data(), 3 * Rz * 8, hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));   // perm (local) must outlive the copy
        } else {
            HIP_TRY(hipMemcpyAsync(b->launch, x0, Rz * 8, hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->launch + Rz, y0, Rz * 8, hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->launch + 2 * Rz, theta0, Rz * 8, hipMemcpyHostToDevice, b->stream));
        }
        HIP_TRY(hipStreamSynchronize(b->stream));  // caller's host buffers may go away
        int dev = 0, cus = 0, per_cu = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        b->lat_simds = cus * 4; b->lat_waves_per_simd = 1;
        b->kfn = pick_advance(b);
        b->kfn_refill = pick_refill(b);
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, b->kfn_refill, 256, 0));
        b->persistent_blocks = cus * (per_cu > 0 ? per_cu : 1);
        if (b->p.launch_mode == RTMI_LAUNCH_SLICED || b->p.launch_mode == RTMI_LAUNCH_AUTO) {
            b->kfn_sliced = pick_sliced(b);
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, b->kfn_sliced, 256, 0));
            b->sliced_blocks = cus * (per_cu > 0 ? per_cu : 1);
            if (getenv("RTMI_DEBUG")) fprintf(stderr, "rtmi: sliced kernel: %d CUs x %d resident blocks\n", cus, per_cu);
            // one queue entry per slice a bundle survives: tiny slices on a large batch with a large max_size would ask for
            // gigabytes of queue (and a memset of it per run); refuse beyond 256 MB (auto: such a batch runs the plain launch)
            if (b->p.launch_mode == RTMI_LAUNCH_SLICED)
                ARG_TRY(sliced_feasible(b),
                        "rtmi_batch_create: RTMI_LAUNCH_SLICED: slice_steps too small for this batch size and max_size (queue > 256 MB)");
            // auto only ever slices a batch with more bundles than resident blocks (auto_wants_sliced)
            const bool may_slice = b->p.launch_mode == RTMI_LAUNCH_SLICED || (sliced_feasible(b) && (R + 255) / 256 > b->sliced_blocks);
            if (may_slice)
                HIP_TRY(hipMalloc(&b->sliced_ctl, (4 + sliced_capacity(b, sliced_steps(b))) * sizeof(unsigned long long)));
        }
        if (retrace_wanted(b)) {
            const int rcr = retrace_create(b, x0, y0, theta0);
            if (rcr) return rcr;
        }
        return batch_init_state(b, true);
    };
    try {
        rc = body();
    } catch (const std::exception& e) {   // host allocations (sort_rays) must not unwind across the C ABI
        rc = fail(RTMI_ERR_ALLOC, std::string("rtmi_batch_create: ") + e.what());
    }
    if (rc) { rtmi_batch_destroy(b); return rc; }
    *out = b;
    return RTMI_OK;
}

template <typename T>
__global__ void k_set_per_ray(BatchDev<T> a, const double* step, const double* step2h, const int* ms, T* vstep, T* vstep2h, int* vmax) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    const long o = out_index(a, k);
    vstep[k] = (T)step[o]; vstep2h[k] = (T)step2h[o]; vmax[k] = ms[o];
    a.alive[k] = a.istep[k] + 1 < ms[o];
}

RTMI_EXPORT int rtmi_batch_set_per_ray(rtmi_batch* b, const double* step, const int32_t* max_size) {
    ARG_TRY(b && step && max_size, "rtmi_batch_set_per_ray: null");
    ARG_TRY(b->p.launch_mode == RTMI_LAUNCH_PLAIN || b->p.launch_mode == RTMI_LAUNCH_AUTO,
            "rtmi_batch_set_per_ray: per-ray steps run on the one-lane-per-ray kernel (RTMI_LAUNCH_PLAIN or RTMI_LAUNCH_AUTO)");
    DEVICE_TRY(b->field, "rtmi_batch_set_per_ray");
    drop_graph(b);     // a step graph holds the kernel arguments by value
    if (b->launches != 0 || b->dirty_state)
        return fail(RTMI_ERR_STATE, "rtmi_batch_set_per_ray: only valid on a fresh or reset batch (before any rtmi_step / rtmi_run / "
                                    "rtmi_batch_set_state): rays that already stopped would be revived");
    const size_t R = (size_t)b->R;
    std::vector<double> h2;
    try {
        h2.resize(R);
    } catch (const std::exception& e) {
        return fail(RTMI_ERR_ALLOC, std::string("rtmi_batch_set_per_ray: ") + e.what());
    }
    for (size_t k = 0; k < R; k++) {
        ARG_TRY(step[k] > 0 && std::isfinite(step[k]), "rtmi_batch_set_per_ray: every step must be finite and > 0");
        ARG_TRY(max_size[k] >= (b->p.method == 7 ? 4 : 2) && max_size[k] <= b->p.max_size,
                "rtmi_batch_set_per_ray: every max_size must be in [2 (4 for op7), params.max_size]");
        h2[k] = libm_square(step[k]) / 2.0;   // numpy scalar step**2 (:330)
    }
    if (!b->vstep) {   // all three or none: pick_advance keys on vstep alone
        void *v1 = nullptr, *v2 = nullptr;
        int* v3 = nullptr;
        hipError_t ea = hipMalloc(&v1, R * b->esz);
        if (ea == hipSuccess) ea = hipMalloc(&v2, R * b->esz);
        if (ea == hipSuccess) ea = hipMalloc(&v3, R * sizeof(int));
        if (ea != hipSuccess) {
            (void)hipFree(v1); (void)hipFree(v2); (void)hipFree(v3);
            return fail(RTMI_ERR_ALLOC, std::string("rtmi_batch_set_per_ray: ") + hipGetErrorString(ea));
        }
        b->vstep = v1; b->vstep2h = v2; b->vmax = v3;
    }
    void* stg = nullptr;
    int rcs = batch_staging(b, 2 * R * sizeof(double) + R * sizeof(int), &stg);
    if (rcs) return rcs;
    double* d = (double*)stg;
    int* di = (int*)(d + 2 * R);
    hipError_t e = hipMemcpyAsync(d, step, R * 8, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + R, h2.data(), R * 8, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(di, max_size, R * sizeof(int), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        const dim3 g((unsigned)((R + 255) / 256)), blk(256);
        if (b->p.dtype == RTMI_F64)
            hipLaunchKernelGGL(k_set_per_ray<double>, g, blk, 0, b->stream, batch_dev<double>(b), d, d + R, di, (double*)b->vstep, (double*)b->vstep2h, b->vmax);
        else
            hipLaunchKernelGGL(k_set_per_ray<float>, g, blk, 0, b->stream, batch_dev<float>(b), d, d + R, di, (float*)b->vstep, (float*)b->vstep2h, b->vmax);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_batch_set_per_ray: ") + hipGetErrorString(e));
    b->kfn = pick_advance(b);
    b->dirty = true;   // rows written under the previous steps would no longer be rewritten: clear on the next reset
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_batch_reset(rtmi_batch* b) {
    ARG_TRY(b, "rtmi_batch_reset: null");
    DEVICE_TRY(b->field, "rtmi_batch_reset");
    // np.zeros (:802-803): rows past each ray's last written row must read 0.  lazy_clear skips the memset when the
    // re-run is known to rewrite exactly the rows the previous run wrote (same launch conditions, same steps).
    return batch_init_state(b, b->dirty || !b->p.lazy_clear);
}

template <typename T> __global__ void k_set_state(BatchDev<T> a, const double* st, const double* hist, const int* istep, const unsigned char* live) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    const long o = out_index(a, k);
    // state9 order: x, y, theta, n, dn/dx, dn/dy, dist_sim, dist_real, T
    a.acc(0)[k] = st[o]; a.acc(1)[k] = st[(size_t)a.R + o]; a.acc(2)[k] = st[(size_t)2 * a.R + o];
    for (int q = 0; q < 3; q++) a.aux(q)[k] = (T)st[(size_t)(3 + q) * a.R + o];
    for (int q = 0; q < 3; q++) a.acc(3 + q)[k] = st[(size_t)(6 + q) * a.R + o];
    if (a.has_hist && hist)
        for (int q = 0; q < 4; q++) a.aux(3 + q)[k] = (T)hist[(size_t)q * a.R + o];
    if (a.rot) {
        if (hist && live) {                       // the carried unit tangent of a checkpoint (rtmi_batch_get_state)
            a.unit(0)[k] = (T)hist[o]; a.unit(1)[k] = (T)hist[(size_t)a.R + o];
        } else {                                  // a state given from outside starts from its angle's own sin/cos
            T sn, cs;
            rt::M<T>::sincos_(a.acc(2)[k], &sn, &cs);
            a.unit(0)[k] = cs; a.unit(1)[k] = sn;
        }
    }
    if (a.hov) a.hov[k] = (a.rot && hist && live) ? (float)hist[(size_t)2 * a.R + o] : 0.f;     // the hover sum of a checkpoint (aux4 row 2)
    if (istep) a.istep[k] = istep[o];
    // a checkpoint knows which rays had left the box (a position outside it does not say so: op7's bootstrap rows skip the test)
    a.alive[k] = (live ? live[o] != 0 : true) && a.istep[k] + 1 < max_size_of(a, k);
}
// out: state9[9][R] then hist4[4][R] (fp64), oi: istep[R], ol: alive[R]; caller's ray order
template <typename T> __global__ void k_get_state(BatchDev<T> a, double* out, int* oi, unsigned char* ol) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    const long o = out_index(a, k);
    const size_t R = (size_t)a.R;
    out[o] = a.acc(0)[k]; out[R + o] = a.acc(1)[k]; out[2 * R + o] = a.acc(2)[k];
    for (int q = 0; q < 3; q++) out[(3 + q) * R + o] = (double)a.aux(q)[k];
    for (int q = 0; q < 3; q++) out[(6 + q) * R + o] = a.acc(3 + q)[k];
    double h[4] = {0, 0, 0, 0};
    if (a.has_hist) for (int q = 0; q < 4; q++) h[q] = (double)a.aux(3 + q)[k];
    else if (a.rot) { h[0] = (double)a.unit(0)[k]; h[1] = (double)a.unit(1)[k]; h[2] = a.hov ? (double)a.hov[k] : 0.0; }
    for (int q = 0; q < 4; q++) out[(9 + q) * R + o] = h[q];
    oi[o] = a.istep[k];
    ol[o] = a.alive[k];
}

static int set_state_impl(rtmi_batch* b, const double* state9, const double* hist4, const int32_t* istep, const uint8_t* live, const char* who) {
    ARG_TRY(b && state9, std::string(who) + ": null");
    DEVICE_TRY(b->field, who);
    const size_t R = (size_t)b->R;
    b->dirty = true;
    b->dirty_state = true;
    void* stg = nullptr;
    int rc = batch_staging(b, 13 * R * sizeof(double) + R * sizeof(int) + R, &stg);
    if (rc) return rc;
    double* d = (double*)stg;
    int* di = (int*)(d + 13 * R);
    unsigned char* dl = (unsigned char*)(di + R);
    hipError_t e = hipMemcpyAsync(d, state9, 9 * R * 8, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess && hist4) e = hipMemcpyAsync(d + 9 * R, hist4, 4 * R * 8, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess && istep) e = hipMemcpyAsync(di, istep, R * sizeof(int), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess && live) e = hipMemcpyAsync(dl, live, R, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        const dim3 g((unsigned)((R + 255) / 256)), blk(256);
        if (b->p.dtype == RTMI_F64)
            hipLaunchKernelGGL(k_set_state<double>, g, blk, 0, b->stream, batch_dev<double>(b), d, hist4 ? d + 9 * R : nullptr, istep ? di : nullptr, live ? dl : nullptr);
        else
            hipLaunchKernelGGL(k_set_state<float>, g, blk, 0, b->stream, batch_dev<float>(b), d, hist4 ? d + 9 * R : nullptr, istep ? di : nullptr, live ? dl : nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    return RTMI_OK;
}
RTMI_EXPORT int rtmi_batch_set_state(rtmi_batch* b, const double* state9, const double* hist4, const int32_t* istep) {
    return set_state_impl(b, state9, hist4, istep, nullptr, "rtmi_batch_set_state");
}
RTMI_EXPORT int rtmi_batch_restore_state(rtmi_batch* b, const double* state9, const double* aux4, const int32_t* istep, const uint8_t* alive) {
    ARG_TRY(aux4 && istep && alive, "rtmi_batch_restore_state: null (pass what rtmi_batch_get_state returned)");
    return set_state_impl(b, state9, aux4, istep, alive, "rtmi_batch_restore_state");
}

RTMI_EXPORT int rtmi_batch_get_state(rtmi_batch* b, double* state9, double* hist4, int32_t* istep, uint8_t* alive) {
    ARG_TRY(b, "rtmi_batch_get_state: null");
    DEVICE_TRY(b->field, "rtmi_batch_get_state");
    RETRACE_FLUSH(b);
    const size_t R = (size_t)b->R;
    void* stg = nullptr;
    int rc = batch_staging(b, 13 * R * sizeof(double) + R * sizeof(int) + R, &stg);
    if (rc) return rc;
    double* d = (double*)stg;
    int* di = (int*)(d + 13 * R);
    unsigned char* dl = (unsigned char*)(di + R);
    const dim3 g((unsigned)((R + 255) / 256)), blk(256);
    if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_get_state<double>, g, blk, 0, b->stream, batch_dev<double>(b), d, di, dl);
    else hipLaunchKernelGGL(k_get_state<float>, g, blk, 0, b->stream, batch_dev<float>(b), d, di, dl);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && state9) e = hipMemcpyAsync(state9, d, 9 * R * 8, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && hist4) e = hipMemcpyAsync(hist4, d + 9 * R, 4 * R * 8, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && istep) e = hipMemcpyAsync(istep, di, R * sizeof(int), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && alive) e = hipMemcpyAsync(alive, dl, R, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_batch_get_state: ") + hipGetErrorString(e));
    return RTMI_OK;
}

static int fold_events(rtmi_batch* b) {
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (size_t i = 0; i < b->ev_used; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, b->events[i].first, b->events[i].second));
        b->total_kernel_ms += ms;
        if (i >= b->pass_ev_start) b->kernel_ms += ms;
    }
    b->ev_used = 0;
    b->pass_ev_start = 0;
    return RTMI_OK;
}

template <typename T> static void launch_advance(rtmi_batch* b, int nsteps) {
    BatchDev<T> a = batch_dev<T>(b);
    b->kfn = pick_advance(b);   // depends on state that changes after create (set_state, set_per_ray)
    void* args[] = {&a, &nsteps};
    const int bs = b->p.block_size > 0 ? b->p.block_size : 256;
    const dim3 g((unsigned)((b->R + bs - 1) / bs)), blk(bs);
    (void)hipLaunchKernel(b->kfn, g, blk, args, 0, b->stream);
}

static int next_event_pair(rtmi_batch* b, std::pair<hipEvent_t, hipEvent_t>** out) {
    if (b->ev_used == b->events.size()) {
        if (b->events.size() >= 1024) {
            int rc = fold_events(b);
            if (rc) return rc;
        } else {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            try {
                b->events.emplace_back(e0, e1);
            } catch (const std::exception& ex) {   // never unwind across the C ABI
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
                return fail(RTMI_ERR_ALLOC, std::string("event list: ") + ex.what());
            }
        }
    }
    *out = &b->events[b->ev_used++];
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_step(rtmi_batch* b, int32_t nsteps);
static void drop_graph(rtmi_batch* b) {
    if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
    if (b->graph) (void)hipGraphDestroy(b->graph);
    b->graph_exec = nullptr; b->graph = nullptr; b->graph_kfn = nullptr; b->graph_nsteps = b->graph_count = 0;
}
// A chain of `count` kernel nodes, each the advance kernel for `nsteps` steps with the batch's present arguments.
template <typename T> static hipError_t build_step_graph(rtmi_batch* b, int nsteps, int count) {
    drop_graph(b);
    BatchDev<T> a = batch_dev<T>(b);
    b->kfn = pick_advance(b);
    const int bs = b->p.block_size > 0 ? b->p.block_size : 256;
    hipError_t e = hipGraphCreate(&b->graph, 0);
    if (e != hipSuccess) return e;
    void* args[] = {&a, &nsteps};
    hipKernelNodeParams kp{};
    kp.func = const_cast<void*>(b->kfn);
    kp.gridDim = dim3((unsigned)((b->R + bs - 1) / bs));
    kp.blockDim = dim3(bs);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < count; i++) {
        hipGraphNode_t node = nullptr;
        e = hipGraphAddKernelNode(&node, b->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
        if (e != hipSuccess) return e;
        prev = node;
    }
    e = hipGraphInstantiate(&b->graph_exec, b->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) return e;
    b->graph_kfn = b->kfn; b->graph_nsteps = nsteps; b->graph_count = count; b->graph_block = bs;
    return hipSuccess;
}

RTMI_EXPORT int rtmi_step_repeat(rtmi_batch* b, int32_t nsteps, int32_t count) {
    ARG_TRY(b, "rtmi_step_repeat: null");
    ARG_TRY(nsteps > 0 && count > 0 && count <= 65536, "rtmi_step_repeat: nsteps must be > 0 and count in 1..65536");
    DEVICE_TRY(b->field, "rtmi_step_repeat");
    ARG_TRY(b->p.block_size == 0 || (b->p.block_size % 64 == 0 && b->p.block_size <= 256),
            "rtmi_step_repeat: block_size must be a multiple of 64, at most 256");
    const int bs = b->p.block_size > 0 ? b->p.block_size : 256;
    // the graph holds the kernel arguments by value: rebuilt when anything they derive from may have changed (the kernel
    // choice follows set_state / set_per_ray; reset and restore keep the same buffers and parameters)
    if (!b->graph_exec || b->graph_kfn != pick_advance(b) || b->graph_nsteps != nsteps || b->graph_count != count || b->graph_block != bs) {
        const hipError_t e = b->p.dtype == RTMI_F64 ? build_step_graph<double>(b, nsteps, count) : build_step_graph<float>(b, nsteps, count);
        if (e != hipSuccess) {   // no graph: the same launches one by one
            drop_graph(b);
            (void)hipGetLastError();
            for (int i = 0; i < count; i++) { const int rc = rtmi_step(b, nsteps); if (rc) return rc; }
            return RTMI_OK;
        }
    }
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    int rc0 = next_event_pair(b, &evp);
    if (rc0) return rc0;
    HIP_TRY(hipEventRecord(evp->first, b->stream));
    HIP_TRY(hipGraphLaunch(b->graph_exec, b->stream));
    if (b->rt) b->rt->pending = true;
    HIP_TRY(hipEventRecord(evp->second, b->stream));
    b->launches += (uint32_t)count; b->total_launches += (uint64_t)count;
    b->mode_used = RTMI_LAUNCH_PLAIN;
    return RTMI_OK;
}

// drain: the launch runs every ray to its end (rtmi_run) -- critical rays are re-traced beside it, inside its timing events
// A run the host watches (rtmi_run) on a batch with compute units set aside for the re-trace (Retrace::masked): between the
// opening and the closing event on the caller's stream, the batch's launches go to its own stream, whose CU mask is the
// complement of the re-trace streams'.  enter() after the opening event, leave() before the closing one; an error path in
// between only gets the caller's stream put back.
struct OwnStream {
    rtmi_batch* b = nullptr;
    hipStream_t callers = nullptr;
    bool on = false;
    int enter(rtmi_batch* b_);
    int leave();
    ~OwnStream() { if (on) b->stream = callers; }
};

int OwnStream::enter(rtmi_batch* b_) {
    b = b_;
    callers = b->stream;
    if (!b->rt || !b->rt->masked) return RTMI_OK;
    HIP_TRY(hipEventRecord(b->rt->ev_in, callers));
    HIP_TRY(hipStreamWaitEvent(b->rt->masked, b->rt->ev_in, 0));
    b->stream = b->rt->masked;
    on = true;
    return RTMI_OK;
}
int OwnStream::leave() {
    if (!on) return RTMI_OK;
    b->stream = callers;
    on = false;
    HIP_TRY(hipEventRecord(b->rt->ev_out, b->rt->masked));
    HIP_TRY(hipStreamWaitEvent(callers, b->rt->ev_out, 0));
    return RTMI_OK;
}

static int step_impl(rtmi_batch* b, int32_t nsteps, bool drain) {
    ARG_TRY(b, "rtmi_step: null");
    ARG_TRY(nsteps > 0, "rtmi_step: nsteps must be > 0");
    DEVICE_TRY(b->field, "rtmi_step");
    ARG_TRY(b->p.block_size == 0 || (b->p.block_size % 64 == 0 && b->p.block_size <= 256),
            "rtmi_step: block_size must be a multiple of 64, at most 256");
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    int rc0 = next_event_pair(b, &evp);
    if (rc0) return rc0;
    auto& ev = *evp;
    HIP_TRY(hipEventRecord(ev.first, b->stream));
    OwnStream own;
    if (drain) { const int rco = own.enter(b); if (rco) return rco; }
    if (b->p.dtype == RTMI_F64) launch_advance<double>(b, nsteps);
    else launch_advance<float>(b, nsteps);
    HIP_TRY(hipGetLastError());
    if (b->rt) {
        b->rt->pending = true;
        if (drain) { const int rcd = retrace_drain(b, true); if (rcd) return rcd; }
    }
    { const int rco = own.leave(); if (rco) return rco; }
    HIP_TRY(hipEventRecord(ev.second, b->stream));
    b->launches++; b->total_launches++;
    b->mode_used = RTMI_LAUNCH_PLAIN;
    return RTMI_OK;
}
RTMI_EXPORT int rtmi_step(rtmi_batch* b, int32_t nsteps) { return step_impl(b, nsteps, false); }

static int read_counters(rtmi_batch* b) {
    HIP_TRY(hipMemsetAsync(b->counters, 0, 2 * sizeof(unsigned long long), b->stream));
    hipLaunchKernelGGL(k_stats, dim3(256), dim3(256), 0, b->stream, b->istep, b->alive, (long)b->R, b->counters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_counters, b->counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));  // [0..1]
    HIP_TRY(hipStreamSynchronize(b->stream));
    return RTMI_OK;
}

static int next_event_pair(rtmi_batch* b, std::pair<hipEvent_t, hipEvent_t>** out);

template <typename T> static void launch_refill(const rtmi_batch* b) {
    BatchDev<T> a = batch_dev<T>(b);
    int refill_min = b->p.refill_min > 0 ? b->p.refill_min : 32;
    int chunk = 16;
    void* args[] = {&a, &refill_min, &chunk};
    long need = (b->R + 255) / 256;
    const dim3 g((unsigned)(need < b->persistent_blocks ? need : b->persistent_blocks)), blk(256);
    (void)hipLaunchKernel(b->kfn_refill, g, blk, args, 0, b->stream);
}

template <typename T> static void launch_sliced(const rtmi_batch* b, int slice, unsigned long long capacity) {
    BatchDev<T> a = batch_dev<T>(b);
    unsigned long long* ctl = b->sliced_ctl;
    unsigned long long timeout = sliced_timeout_ticks(b, slice);
    void* args[] = {&a, &slice, &capacity, &ctl, &timeout};
    const long need = (b->R + 255) / 256;
    const dim3 g((unsigned)(need < b->sliced_blocks ? need : b->sliced_blocks)), blk(256);
    (void)hipLaunchKernel(b->kfn_sliced, g, blk, args, 0, b->stream);
}

// the time-sliced schedule can take this run: queue allocated, uniform DELTA_S, rows through the wave-uniform descriptor
static bool sliced_ready(const rtmi_batch* b) { return b->sliced_ctl && !b->vstep && uniform_rows_ok(b); }

static int run_sliced(rtmi_batch* b, std::pair<hipEvent_t, hipEvent_t>** evp) {
    // persistent blocks, bundles advanced in time slices (k_advance_sliced)
    int rc = next_event_pair(b, evp);
    if (rc) return rc;
    auto* ev = *evp;
    const int slice = sliced_steps(b);
    const unsigned long long capacity = sliced_capacity(b, slice);
    HIP_TRY(hipMemsetAsync(b->sliced_ctl, 0, (4 + capacity) * sizeof(unsigned long long), b->stream));
    HIP_TRY(hipEventRecord(ev->first, b->stream));
    OwnStream own;
    { const int rco = own.enter(b); if (rco) return rco; }
    if (b->p.dtype == RTMI_F64) launch_sliced<double>(b, slice, capacity);
    else launch_sliced<float>(b, slice, capacity);
    HIP_TRY(hipGetLastError());
    if (b->rt) {       // critical rays: re-traced beside the launch
        b->rt->pending = true;
        const int rcd = retrace_drain(b, true);
        if (rcd) return rcd;
    }
    { const int rco = own.leave(); if (rco) return rco; }
    HIP_TRY(hipEventRecord(ev->second, b->stream));
    b->launches++; b->total_launches++;
    b->mode_used = RTMI_LAUNCH_SLICED;
    rc = read_counters(b);
    if (rc) return rc;
    if (b->h_counters[1] != 0) {   // every wait is bounded: a launch that gave one up reports here instead of hanging
        unsigned long long ctl4[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(ctl4, b->sliced_ctl, sizeof(ctl4), hipMemcpyDeviceToHost));
        return fail(RTMI_ERR_STATE, ctl4[3] == 1 ? "rtmi_run: the sliced launch abandoned a wait for a queue entry"
                                  : ctl4[3] == 2 ? "rtmi_run: the sliced launch ran out of queue entries"
                                                 : "rtmi_run: rays still live after the sliced launch");
    }
    return RTMI_OK;
}

static int run_plain(rtmi_batch* b, std::pair<hipEvent_t, hipEvent_t>** evp) {
    // one lane per ray to completion: a single launch covers every remaining row
    int rc = step_impl(b, b->p.max_size, true);
    if (rc) return rc;
    *evp = &b->events[b->ev_used - 1];
    rc = read_counters(b);
    if (rc) return rc;
    if (b->h_counters[1] != 0) return fail(RTMI_ERR_STATE, "rtmi_run: rays still live after a full-length launch");
    return RTMI_OK;
}


// ------------------------------------------------------------------ critical rays: implementation (see struct Retrace)
static bool retrace_wanted(const rtmi_batch* b) {
    const rtmi_params& p = b->p;
    return !b->is_retrace_sub && !p.no_retrace && p.dtype == RTMI_F64 && b->field->steep_cells > 0 &&
           (p.reference_order == RTMI_ORDER_DEFAULT || p.reference_order == RTMI_ORDER_FAST_FIELD) &&
           rt::rotates_unit(p.method, true) && !getenv("RTMI_NO_RETRACE");
}
// slots of the queue: a few hundred rays of a million are critical on the interface fan -- room for 1/128 of the batch, for 1 024 at
// least (a small batch may be ALL window: 380 of the 4 096 rays around the interface fan's split), never for more than the batch
static unsigned retrace_capacity(int64_t R) {
    const int64_t c = std::min<int64_t>(std::max<int64_t>((R / 128 + 63) / 64 * 64, 1024), 65536);
    return (unsigned)std::min<int64_t>(c, (R + 63) / 64 * 64);
}
// The re-trace of queue slots [lo, hi): one lane per ray, from its launch conditions (:809-826) --
//   in the reference's operation order (rt::ex::ray_step: the oracle's bits, row for row) through the stretch that made the ray
//   critical: until it has reached the row where the fused run stopped it AND has not hovered -- in a steep cell, heading within
//   60 degrees of the iso-lines (rt::hover_weight > 0.5) -- for kRetraceOut steps;
//   then in the fused form again (rt::ray_step, per-lane lookups) to its end: a ray that has turned away from the transition is as
//   well conditioned as any other the fused kernels keep (the same measure says so), and the thousands of steps a refracted grazing
//   ray still has to go to the box's far side (8 700 on the interface fan) cost 0.5 us each instead of 2.5.  Its rows from there on
//   are the fused form's: ~1e-13 from the reference's.  Should the fused tail hover again (its sum starts from 0 at the hand-back)
//   the lane starts over from the launch conditions with the reference-order part extended to the row it had reached.
// Rows and final state go to the hidden batch's arrays s (slot j of every array), never to the main batch's: the main kernel is
// still running.  m: the main batch (launch conditions, per-ray steps).  One wave per block, <= 168 registers: a block fits
// wherever one of the main kernel's has retired.
constexpr int kRetraceOut = 256;
constexpr int kRetraceCus = 8;         // compute units set aside for the re-trace streams (retrace_create)
// Two kernels per chunk, one after the other on the chunk's stream (one wave per block; a lane = a queue slot):
//   k_retrace_ref: the reference-order part.  A lane leaves its loop at the hand-back (state stored, alive = 1) or when its ray ends
//   (alive = 0); no lane steps in two forms in one iteration.  marked_only: the final sweep for rays whose fused tail hovered again
//   (alive = 2, see k_retrace_tail) -- those are taken in reference order to their END.
//   k_retrace_tail: the fused part of the rays that were handed back, with the few-waves lookup (the wave's cell kept in vector
//   registers, rt::kPolyCached): 0.75 us per step where per-lane loads two rows at a time take 3.
template <int METHOD>
__global__ __launch_bounds__(64, 3) void k_retrace_ref(BatchDev<double> s, BatchDev<double> m, const unsigned long long* rq, unsigned lo, unsigned hi,
                                                       int marked_only, unsigned long long* dbg) {
    typedef double T;
    __builtin_amdgcn_s_setprio(3);          // beside the main kernel's waves: a handful of waves on the critical path of the call
    const unsigned j = lo + blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = j < hi && (!marked_only || s.alive[j < hi ? j : lo] == 2);
    const unsigned long long e = valid ? rq[1 + j] : 0ull;
    long k = (long)(e & 0xffffffffull);
    if (k >= m.R) k = 0;
    rt::Consts<T> K = m.K;
    int max_size = m.max_size;
    if (m.vstep) { K.step = m.vstep[k]; K.step2h = m.vstep2h[k]; K.step2 = K.step2h * 2.0; max_size = m.vmax[k]; }
    const int ref_until = marked_only ? 0x7fffffff : (int)(e >> 32);          // the row where the fused run stopped this ray
    rt::GlobalGather<T> gg;
    gg.gflat = s.gflat;
    rt::Ray<T> r;
    int i = 0, out = 0;
    bool alive = valid && max_size > 1, handed = false;
    if (valid) {                             // the initial conditions, as init_ray of a reference-order batch
        r.x = m.x0[k]; r.y = m.y0[k]; r.th = m.th0[k];
        rt::ex::n_gradient(s.F, gg, true, (T)r.x, (T)r.y, r.n, r.gx, r.gy);
        rt::ex::derive(K, r);
        r.dsim = 0; r.dreal = 0; r.tt = 0;
        r.hx0 = r.hy0 = r.hx1 = r.hy1 = 0;
        r.hov = 0.f;
        if (s.stride && s.rec_rows > 0) write_row(s, 0, (long)j, r);
    }
    while (rt_ballot(alive && !handed) != 0ull) {
        if (alive && !handed) {
            ++i;
            const T ux0 = r.ux, uy0 = r.uy;            // the tangent the step starts with
            const bool inside = rt::ex::ray_step<METHOD>(s.F, K, gg, true, r, i);
            // (reference order: Ray::hov is the steepness of the cell the ray arrived in, and the gradient there is current whenever
            // that is not 0)
            const T d = rt::fma_(r.gy, uy0, r.gx * ux0), g2 = rt::fma_(r.gy, r.gy, r.gx * r.gx);
            out = (r.hov != 0.f && rt::hover_weight(d, g2) > 0.5f) ? 0 : out + 1;
            if (s.stride && i % s.stride == 0 && i / s.stride < s.rec_rows) write_row(s, (long)(i / s.stride), (long)j, r);
            alive = inside && (i + 1 < max_size);
            handed = alive && i >= ref_until && out >= kRetraceOut;
        }
    }
    if (valid) {
        rt::ex::finish_state<METHOD>(s.F, gg, r);
        s.acc(0)[j] = r.x; s.acc(1)[j] = r.y; s.acc(2)[j] = r.th; s.aux(0)[j] = r.n; s.aux(1)[j] = r.gx; s.aux(2)[j] = r.gy;
        s.acc(3)[j] = r.dsim; s.acc(4)[j] = r.dreal; s.acc(5)[j] = r.tt;
        s.istep[j] = i;
        s.alive[j] = alive ? 1 : 0;
        if (dbg) { atomicAdd(dbg, (unsigned long long)i); atomicMax(dbg + 3, (unsigned long long)i); }
    }
}
template <int METHOD, bool ISO>
__global__ __launch_bounds__(64, 2) void k_retrace_tail(BatchDev<double> s, BatchDev<double> m, const unsigned long long* rq, unsigned lo, unsigned hi,
                                                        unsigned long long* dbg) {
    typedef double T;
    __builtin_amdgcn_s_setprio(3);
    const unsigned j = lo + blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = j < hi && s.alive[j < hi ? j : lo] == 1;
    long k = valid ? (long)(rq[1 + j] & 0xffffffffull) : 0;
    if (k >= m.R) k = 0;
    rt::Consts<T> K = m.K;
    int max_size = m.max_size;
    if (m.vstep) { K.step = m.vstep[k]; K.step2h = m.vstep2h[k]; K.step2 = K.step2h * 2.0; max_size = m.vmax[k]; }
    rt::PolyLaneKept<T, true> pg;
    pg.init();
    pg.hov_limit = m.hov_limit / (float)K.step;
    rt::Ray<T> r;
    idle_ray(s, r);
    int i = 0, i0 = 0;
    if (valid) {
        r.x = s.acc(0)[j]; r.y = s.acc(1)[j]; r.th = s.acc(2)[j]; r.n = s.aux(0)[j]; r.gx = s.aux(1)[j]; r.gy = s.aux(2)[j];
        r.dsim = s.acc(3)[j]; r.dreal = s.acc(4)[j]; r.tt = s.acc(5)[j];
        i = i0 = s.istep[j];
        // the fused form continues from the reference's own state: the unit tangent it carries from here on is the reference's
        // cos / sin of the angle
        const rt::ex::SinCos u = rt::ex::sincos_(r.th);
        r.ux = u.c; r.uy = u.s;
        rt::derive<T, ISO, true>(K, r);
    }
    bool alive = valid;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // steps until the lane's next recorded row, counted down (an integer division per step is a third of a fused step's instructions)
    const int stride = s.stride;
    int until = stride > 0 ? stride - i % stride : 0;
    long rec = stride > 0 ? i / stride : 0;
    // what a step leaves behind: the row, and the state the moment the ray ends
    auto after = [&](bool active, bool inside, int row) {
        if (active) {
            i = row;
            if (stride > 0 && --until == 0) {
                until = stride;
                ++rec;
                if (rec < s.rec_rows) write_row(s, rec, (long)j, r);
            }
            alive = inside && (i + 1 < max_size);
            const bool again = r.hov == INFINITY;       // hovering again further on (the step ended the ray): the final sweep takes it in reference order throughout
            if (!alive) {
                s.acc(0)[j] = r.x; s.acc(1)[j] = r.y; s.acc(2)[j] = r.th; s.aux(0)[j] = r.n; s.aux(1)[j] = r.gx; s.aux(2)[j] = r.gy;
                s.acc(3)[j] = r.dsim; s.acc(4)[j] = r.dreal; s.acc(5)[j] = r.tt;
                s.istep[j] = i;
                s.alive[j] = again ? 2 : 0;
                if (again) atomicAdd(s.counters + 3, 1ull);          // the hidden batch's counters[3]: rays for the final sweep
                if (dbg) { atomicAdd(dbg + 1, (unsigned long long)(i - i0)); atomicAdd(dbg + 2, again ? 1ull : 0ull); atomicMax(dbg + 4, (unsigned long long)i); }
            }
        }
    };
    // every lane runs every iteration; a lane whose ray has ended evolves a stale state nobody reads: what it leaves behind is
    // stored the moment it ends
    while (rt_ballot(alive) != 0ull) {
        const bool active = alive;
        const int row = i + 1;
        const bool inside = rt::ray_step<T, METHOD, ISO>(s.F, K, pg, active, r, row);
        after(active, inside, row);
        // where the next step will look the field up, if the ray goes on as it goes now: a lane about to enter another cell
        // starts that cell's loads here, a step's arithmetic ahead of their use
        pg.prefetch(s.F, alive, (T)r.x + r.ux * K.step, (T)r.y + r.uy * K.step);
    }
}
static const void* retrace_ref_fn(int method) {
    return method == 1 ? (const void*)k_retrace_ref<1> : method == 2 ? (const void*)k_retrace_ref<2> : method == 6 ? (const void*)k_retrace_ref<6> : (const void*)k_retrace_ref<8>;
}
static const void* retrace_tail_fn(int method, bool iso) {
    static const void* const tab[4][2] = {{(const void*)k_retrace_tail<1, false>, (const void*)k_retrace_tail<1, true>}, {(const void*)k_retrace_tail<2, false>, (const void*)k_retrace_tail<2, true>},
                                          {(const void*)k_retrace_tail<6, false>, (const void*)k_retrace_tail<6, true>}, {(const void*)k_retrace_tail<8, false>, (const void*)k_retrace_tail<8, true>}};
    return tab[method == 1 ? 0 : method == 2 ? 1 : method == 6 ? 2 : 3][iso ? 1 : 0];
}
// ... and back: the re-traced ray's rows and final state over the fused ones.  A row the fused run wrote past the re-traced
// ray's last row (the two may leave the box a row apart) reads 0 like every row past a ray's end (:802).
template <typename T>
__global__ void k_retrace_scatter(BatchDev<T> m, BatchDev<T> s, const unsigned long long* rq, unsigned lo, unsigned hi) {
    const unsigned j = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= hi) return;
    const unsigned long long e = rq[1 + j];
    const long k = (long)(e & 0xffffffffull);
    if (k >= m.R) return;
    const int fused_last = (int)(e >> 32), si = s.istep[j];
    if (m.stride) {
        const long rs = si / m.stride < s.rec_rows - 1 ? si / m.stride : s.rec_rows - 1;       // last recorded row of the re-trace
        long top = fused_last / m.stride > rs ? fused_last / m.stride : rs;
        top = top < m.rec_rows - 1 ? top : m.rec_rows - 1;
        for (long row = blockIdx.y; row <= top; row += gridDim.y) {
#pragma unroll
            for (int q = 0; q < 6; q++)
                m.s_ray[((size_t)row * 6 + q) * m.R + k] = row <= rs ? s.s_ray[((size_t)row * 6 + q) * s.R + j] : T(0);
            if (m.n_ray) m.n_ray[(size_t)row * m.R + k] = (row <= rs && s.n_ray) ? s.n_ray[(size_t)row * s.R + j] : T(0);
        }
    }
    if (blockIdx.y == 0) {
        for (int q = 0; q < 6; q++) m.acc(q)[k] = s.acc(q)[j];
        for (int q = 0; q < 3; q++) m.aux(q)[k] = s.aux(q)[j];
        if (m.rot) {       // the unit tangent the fused batch carries as state: the reference's own cos / sin of the final angle
            m.unit(0)[k] = (T)rt::ex::cos_((double)s.acc(2)[j]);
            m.unit(1)[k] = (T)rt::ex::sin_((double)s.acc(2)[j]);
        }
        m.istep[k] = si;
        m.alive[k] = 0;
    }
}

static void retrace_destroy(rtmi_batch* b) {
    Retrace* t = b->rt;
    if (!t) return;
    for (hipStream_t a : t->aux) if (a) (void)hipStreamSynchronize(a);
    if (t->sub) rtmi_batch_destroy(t->sub);
    (void)hipFree(t->rq); (void)hipFree(t->hov); (void)hipFree(t->dbg);
    if (t->host_count) (void)hipHostFree(t->host_count);
    if (t->ev_main) (void)hipEventDestroy(t->ev_main);
    for (hipEvent_t e : t->ev_aux) if (e) (void)hipEventDestroy(e);
    for (hipStream_t a : t->aux) if (a) (void)hipStreamDestroy(a);
    if (t->masked) { (void)hipStreamSynchronize(t->masked); (void)hipStreamDestroy(t->masked); }
    if (t->ev_in) (void)hipEventDestroy(t->ev_in);
    if (t->ev_out) (void)hipEventDestroy(t->ev_out);
    delete t;
    b->rt = nullptr;
}

static int retrace_create(rtmi_batch* b, const double* x0, const double* y0, const double* theta0) {
    Retrace* t = new (std::nothrow) Retrace();
    if (!t) return fail(RTMI_ERR_ALLOC, "rtmi_batch_create: host allocation failed");
    b->rt = t;                                  // rtmi_batch_destroy frees whatever is there if anything below fails
    t->cap = retrace_capacity(b->R);
    const size_t Rs = (size_t)t->cap;
    HIP_TRY(hipMalloc(&t->rq, (1 + (size_t)t->cap) * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(t->rq, 0, (1 + (size_t)t->cap) * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&t->hov, (size_t)b->R * sizeof(float)));
    if (getenv("RTMI_DEBUG")) { HIP_TRY(hipMalloc(&t->dbg, 8 * sizeof(unsigned long long))); HIP_TRY(hipMemset(t->dbg, 0, 8 * sizeof(unsigned long long))); }
    HIP_TRY(hipHostMalloc(&t->host_count, (4 + (size_t)t->cap) * sizeof(unsigned)));
    memset(t->host_count, 0, (4 + (size_t)t->cap) * sizeof(unsigned));
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));           // hi is the numerically lowest = highest priority
    if (getenv("RTMI_DEBUG")) fprintf(stderr, "rtmi: retrace: stream priorities %d (lowest) .. %d (highest); queue of %u slots\n", lo, hi, t->cap);
    // Compute units of their own for the re-trace (kRetraceCus; RTMI_RETRACE_CUS overrides, 0: none): sharing its SIMD with four waves
    // of the main kernel a reference-order wave takes 7.5 us per step instead of 4.5 (s_setprio wins the arbitration, not the
    // cycles another wave's fp64 instruction already holds the pipeline for).  Eight, because mask bit i is a CU of XCD i % 8 and
    // a queue whose mask leaves an XCD without any CU dispatches no faster than without a mask (measured: 1, 2, 4 -> nothing;
    // 8, 16 -> interface x op8 30.7 -> 21.1 ms).  The main kernel pays 8 / 256 of its rate.  No mask where the runtime refuses one.
    const int own_cus = getenv("RTMI_RETRACE_CUS") ? atoi(getenv("RTMI_RETRACE_CUS")) : kRetraceCus;
    int ncu = 0;
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, b->field->device));
    bool mask_on = own_cus > 0 && own_cus <= ncu / 4;
    std::vector<uint32_t> m_aux((size_t)(ncu + 31) / 32, 0u), m_main((size_t)(ncu + 31) / 32, 0u);
    for (int c = 0; c < ncu; c++) (c < own_cus ? m_aux : m_main)[(size_t)c / 32] |= 1u << (c % 32);
    if (mask_on && hipExtStreamCreateWithCUMask(&t->masked, (uint32_t)m_main.size(), m_main.data()) != hipSuccess) {
        (void)hipGetLastError();
        t->masked = nullptr;
        mask_on = false;
    }
    for (int i = 0; i < Retrace::kAux; i++) {
        if (mask_on && hipExtStreamCreateWithCUMask(&t->aux[i], (uint32_t)m_aux.size(), m_aux.data()) != hipSuccess) {
            (void)hipGetLastError();
            t->aux[i] = nullptr;
        }
        if (!t->aux[i]) HIP_TRY(hipStreamCreateWithPriority(&t->aux[i], hipStreamNonBlocking, hi));
        HIP_TRY(hipEventCreateWithFlags(&t->ev_aux[i], hipEventDisableTiming));
    }
    if (mask_on) {
        HIP_TRY(hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&t->ev_out, hipEventDisableTiming));
    }
    if (getenv("RTMI_DEBUG")) fprintf(stderr, "rtmi: retrace: %d of %d compute units set aside\n", mask_on ? own_cus : 0, ncu);
    HIP_TRY(hipEventCreateWithFlags(&t->ev_main, hipEventDisableTiming));
    // the hidden batch: same field, method, steps, box and record layout, reference order, one plain launch per chunk
    rtmi_params ps = b->p;
    ps.reference_order = RTMI_ORDER_REFERENCE;
    ps.launch_mode = RTMI_LAUNCH_PLAIN;
    ps.sort_rays = 0; ps.ext_s_ray = nullptr; ps.ext_n_ray = nullptr; ps.lazy_clear = 1; ps.no_retrace = 1; ps.field_path = 1; ps.block_size = 0;
    ps.no_n_ray = b->n_ray ? 0 : 1;
    std::vector<double> hx, hy, ht;
    try {
        hx.assign(Rs, x0[0]); hy.assign(Rs, y0[0]); ht.assign(Rs, theta0[0]);    // placeholders: every slot is re-initialised when it is used
    } catch (const std::exception& e) {
        return fail(RTMI_ERR_ALLOC, std::string("rtmi_batch_create: ") + e.what());
    }
    const int rc = rtmi_batch_create(b->field, &ps, (int64_t)Rs, hx.data(), hy.data(), ht.data(), (void*)t->aux[0], &t->sub);
    if (rc) { t->sub = nullptr; return rc; }
    t->sub->is_retrace_sub = true;
    HIP_TRY(hipMemsetAsync(t->sub->alive, 0, Rs, t->aux[0]));        // no slot holds a ray yet
    HIP_TRY(hipStreamSynchronize(t->aux[0]));
    return RTMI_OK;
}

// rtmi_batch_reset / create: an empty queue.  A kernel of the pass before may still be publishing counts (rtmi_step is
// asynchronous): wait for it first.
static int retrace_reset(rtmi_batch* b) {
    Retrace* t = b->rt;
    if (t->pending) HIP_TRY(hipStreamSynchronize(b->stream));
    for (hipStream_t a : t->aux) HIP_TRY(hipStreamSynchronize(a));
    t->chunks = 0;
    HIP_TRY(hipMemsetAsync(t->rq, 0, sizeof(unsigned long long), b->stream));
    memset(t->host_count, 0, (4 + (size_t)t->cap) * sizeof(unsigned));
    HIP_TRY(hipMemsetAsync(t->sub->counters + 3, 0, sizeof(unsigned long long), b->stream));
    t->launched = t->scattered = 0; t->pending = false; t->overflow = 0;
    return RTMI_OK;
}

// queue slots [lo, hi) re-traced to completion on `st`: the reference-order part, then the fused tails
static int retrace_launch_chunk(rtmi_batch* b, unsigned lo, unsigned hi, hipStream_t st, int marked_only = 0) {
    Retrace* t = b->rt;
    BatchDev<double> m = batch_dev<double>(b), s = batch_dev<double>(t->sub);
    const unsigned long long* rq = t->rq;
    unsigned long long* dbg = t->dbg;
    void* args[] = {&s, &m, &rq, &lo, &hi, &marked_only, &dbg};
    const unsigned lanes = 64;         // rays per block = per wave (fewer -- 32, 16, 8 -- was tried: a lone wave's step is not shorter for it)
    HIP_TRY(hipLaunchKernel(retrace_ref_fn(b->p.method), dim3((hi - lo + lanes - 1) / lanes), dim3(lanes), args, 0, st));
    if (!marked_only) {
        void* targs[] = {&s, &m, &rq, &lo, &hi, &dbg};
        HIP_TRY(hipLaunchKernel(retrace_tail_fn(b->p.method, b->p.gamma == 1.0), dim3((hi - lo + lanes - 1) / lanes), dim3(lanes), targs, 0, st));
    }
    return RTMI_OK;
}

// Hand the queued rays to the sub-batch and copy the results back.
// overlap: called between the main kernel's launch and the event that closes its timing -- the host polls the published count
// while the kernel runs and launches chunks on the high-priority stream beside it; otherwise (a read after rtmi_step) the main
// stream is drained first and everything runs in order on it.
static int retrace_drain(rtmi_batch* b, bool overlap) {
    Retrace* t = b->rt;
    unsigned count = 0;
    auto read_count = [&]() -> int {      // the authoritative count (device memory), once the main stream is idle
        HIP_TRY(hipMemcpyAsync(t->host_count + 2, t->rq, sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        unsigned long long c = 0;
        memcpy(&c, t->host_count + 2, sizeof c);
        t->overflow = c > t->cap ? (unsigned)(c - t->cap) : 0u;
        count = (unsigned)std::min<unsigned long long>(c, t->cap);
        return RTMI_OK;
    };
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    if (overlap) {
        HIP_TRY(hipEventRecord(t->ev_main, b->stream));
        auto last_change = std::chrono::steady_clock::now();
        const auto t_start = last_change;
        const bool dbg = getenv("RTMI_DEBUG") != nullptr;
        unsigned seen = t->launched;
        for (;;) {
            const hipError_t q = hipEventQuery(t->ev_main);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(RTMI_ERR_HIP, std::string("rtmi_run: ") + hipGetErrorString(q));
            const bool done = q == hipSuccess;
            if (done) { const int rc = read_count(); if (rc) return rc; }
            else {          // the leading run of per-slot flags
                const volatile unsigned* flags = t->host_count + 4;
                unsigned n = t->host_count[0];
                while (n < t->cap && flags[n] != 0u) ++n;
                t->host_count[0] = n;
                count = n;
            }
            const auto now = std::chrono::steady_clock::now();
            if (count > seen) { seen = count; last_change = now; }
            // A chunk = one launch (a block per 64 rays, the blocks side by side) on one of the aux streams; a stream runs its chunks
            // one after the other and a chunk lasts as long as its slowest ray (10-20 ms), so a chunk must never queue behind
            // another: it goes to an IDLE stream or waits.  While two or more streams are idle a wave's worth of rays is launched as
            // it arrives (the first arrivals start their long reference-order stretch at once); the last idle stream is kept for
            // what has gathered once nothing new has come for 200 us (a burst of arrivals spread over half a millisecond used to end
            // as a fifth chunk behind the first: +11 ms on the full-record interface pass); with no stream idle the rays wait for
            // one.  Once the main kernel is done everything left goes out on the first stream that is or becomes idle.
            if (count > t->launched) {
                int idle = 0, pick = -1;
                for (int a = 0; a < Retrace::kAux; a++) {
                    const hipError_t sq = hipStreamQuery(t->aux[a]);
                    if (sq == hipSuccess) { if (pick < 0) pick = a; ++idle; }
                    else if (sq != hipErrorNotReady) return fail(RTMI_ERR_HIP, std::string("rtmi_run (re-trace): ") + hipGetErrorString(sq));
                }
                const bool paused = now - last_change > std::chrono::microseconds(200);
                const bool go = idle >= 2 ? (done || paused || count - t->launched >= 64) : idle == 1 ? (done || paused) : false;
                if (go) {
                    const int rc = retrace_launch_chunk(b, t->launched, count, t->aux[pick]);
                    if (rc) return rc;
                    ++t->chunks;
                    if (dbg) fprintf(stderr, "rtmi: retrace: slots [%u, %u) launched %.3f ms after the main kernel%s (stream %d, %d idle)\n", t->launched, count,
                                     std::chrono::duration<double, std::milli>(now - t_start).count(), done ? " (which had finished)" : "", pick, idle);
                    t->launched = count;
                }
            }
            if (done && count > t->launched) {          // every stream busy: wait for one (they all finish)
                std::this_thread::sleep_for(std::chrono::microseconds(20));
                continue;
            }
            if (done) {
                if (dbg) fprintf(stderr, "rtmi: retrace: main kernel done after %.3f ms; %u rays handed over, %u found the queue full\n",
                                 std::chrono::duration<double, std::milli>(now - t_start).count(), t->launched, t->overflow);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        if (t->launched > t->scattered) {
            for (int i = 0; i < Retrace::kAux; i++) {
                HIP_TRY(hipEventRecord(t->ev_aux[i], t->aux[i]));
                HIP_TRY(hipStreamWaitEvent(b->stream, t->ev_aux[i], 0));
            }
        }
    } else {
        const int rc = read_count();
        if (rc) return rc;
        if (count > t->launched) {
            const int rc2 = next_event_pair(b, &evp);      // this work is advance-kernel time too
            if (rc2) return rc2;
            HIP_TRY(hipEventRecord(evp->first, b->stream));
            const int rc3 = retrace_launch_chunk(b, t->launched, count, b->stream);
            if (rc3) return rc3;
            t->launched = count;
        }
    }
    if (t->launched > t->scattered) {
        // rays whose fused tail hovered again (expected: none): in reference order to their end, now
        HIP_TRY(hipMemcpyAsync(t->host_count + 2, t->sub->counters + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        unsigned long long again = 0;
        memcpy(&again, t->host_count + 2, sizeof again);
        if (again) {
            HIP_TRY(hipMemsetAsync(t->sub->counters + 3, 0, sizeof(unsigned long long), b->stream));
            const int rca = retrace_launch_chunk(b, t->scattered, t->launched, b->stream, 1);
            if (rca) return rca;
            t->swept += (unsigned)again;
        }
        const unsigned n = t->launched - t->scattered;
        const long rows = b->p.record_stride ? (long)b->p.rec_rows : 1;
        const dim3 g((n + 63) / 64, (unsigned)std::min<long>(rows, 1024)), blk(64);
        hipLaunchKernelGGL(k_retrace_scatter<double>, g, blk, 0, b->stream, batch_dev<double>(b), batch_dev<double>(t->sub),
                           (const unsigned long long*)t->rq, t->scattered, t->launched);
        HIP_TRY(hipGetLastError());
        t->total += n;
        if (!t->rot_learned && overlap && t->scattered == 0 && !getenv("RTMI_NO_DISPATCH_ORDER")) {
            // which bundles held the critical rays of this (whole) pass -> where the next pass's dispatch starts
            std::vector<unsigned long long> ent(t->launched);
            HIP_TRY(hipMemcpyAsync(ent.data(), t->rq + 1, ent.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));
            const unsigned NB = (unsigned)((b->R + 255) / 256);
            std::vector<unsigned> bun;
            for (unsigned long long e : ent) { const unsigned k = (unsigned)(e & 0xffffffffull); if ((int64_t)k < b->R) bun.push_back(k >> 8); }
            std::sort(bun.begin(), bun.end());
            bun.erase(std::unique(bun.begin(), bun.end()), bun.end());
            if (!bun.empty() && NB > 1) {
                unsigned best = bun[0], gap = bun[0] + NB - bun.back();            // the gap that wraps around
                for (size_t i = 1; i < bun.size(); i++)
                    if (bun[i] - bun[i - 1] > gap) { gap = bun[i] - bun[i - 1]; best = bun[i]; }
                t->rot = best;
            }
            t->rot_learned = true;
        }
        t->scattered = t->launched;
        if (t->dbg) {
            unsigned long long h[5] = {0, 0, 0, 0, 0};
            HIP_TRY(hipStreamSynchronize(b->stream));
            HIP_TRY(hipMemcpy(h, t->dbg, sizeof h, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemset(t->dbg, 0, 8 * sizeof(unsigned long long)));
            fprintf(stderr, "rtmi: retrace: %u rays: %llu reference-order steps (at most %llu per ray), %llu fused steps, longest ray %llu steps, %llu hovered again\n", n, h[0], h[3], h[1], h[4], h[2]);
        }
    }
    if (evp) HIP_TRY(hipEventRecord(evp->second, b->stream));
    t->pending = false;
    return RTMI_OK;
}

// ---- RTMI_LAUNCH_AUTO: which schedule a re-run batch keeps (pure host functions; rtmi_debug_auto_rule exposes them to the tests)
// Slicing wins by up to 25 % where whole bundles would be dispatched in rounds (cfg3) and loses a few per cent where they would
// not (the interface fan), so a batch that is re-run from its launch conditions is timed under both.
// What the samples must survive: a recording kernel runs the package into its power limit within a few passes (2.4 -> 1.7 GHz,
// DESIGN.md 5.1), so every pass of a fresh batch is a little slower than the one before and a batch's very first pass is cold
// besides (clocks, caches, page tables).  Hence (a) the two schedules' samples are taken INTERLEAVED and mirrored -- sliced,
// plain, plain, sliced, sliced, plain: positions 1 4 5 against 2 3 6, so a steady drift weighs on both alike -- never a fresh
// sample of one against a stale one of the other (round 4 re-timed only the schedule it had kept: on one box the plain launch
// was kept on the strength of an early sample and then ran 16.3 ms where the sliced one runs 15.3); (b) each schedule is
// judged by the MEDIAN of its three (the cold pass is the one that falls out); (c) the decision is made once and kept --
// launch_mode_used no longer changes inside a caller's timed passes; (d) the plain launch is kept only when it is more than
// 3 % ahead: slicing is the schedule that does not depend on how a fan's lengths fall into dispatch rounds.
#define RTMI_AUTO_RUNS (2 * RTMI_AUTO_SAMPLES)
static int auto_next(const int n[2]) {       // 0 sliced, 1 plain, -1: exploration over
    static const int order[RTMI_AUTO_RUNS] = {0, 1, 1, 0, 0, 1};
    const int k = n[0] + n[1];
    return k < RTMI_AUTO_RUNS ? order[k] : -1;
}
static double auto_median(const double* v, int n) {
    double a[RTMI_AUTO_SAMPLES];
    for (int i = 0; i < n; i++) a[i] = v[i];
    std::sort(a, a + n);
    return n == 0 ? 1e30 : (n & 1) ? a[n / 2] : 0.5 * (a[n / 2 - 1] + a[n / 2]);
}
static int auto_decide(const double* sliced_ms, int ns, const double* plain_ms, int np) {
    return auto_median(plain_ms, np) < 0.97 * auto_median(sliced_ms, ns) ? RTMI_LAUNCH_PLAIN : RTMI_LAUNCH_SLICED;
}

RTMI_EXPORT int rtmi_run(rtmi_batch* b) {
    ARG_TRY(b, "rtmi_run: null");
    DEVICE_TRY(b->field, "rtmi_run");
    int rc;
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (b->p.launch_mode == RTMI_LAUNCH_REFILL) {
        // persistent waves with lane refill: one launch drains the ray queue
        rc = next_event_pair(b, &ev);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(b->counters + 2, 0, sizeof(unsigned long long), b->stream));   // refill queue head
        HIP_TRY(hipEventRecord(ev->first, b->stream));
        OwnStream own;
        { const int rco = own.enter(b); if (rco) return rco; }
        if (b->p.dtype == RTMI_F64) launch_refill<double>(b);
        else launch_refill<float>(b);
        HIP_TRY(hipGetLastError());
        if (b->rt) {
            b->rt->pending = true;
            const int rcd = retrace_drain(b, true);
            if (rcd) return rcd;
        }
        { const int rco = own.leave(); if (rco) return rco; }
        HIP_TRY(hipEventRecord(ev->second, b->stream));
        b->launches++; b->total_launches++;
        b->mode_used = RTMI_LAUNCH_REFILL;
        return read_counters(b);
    }
    // per-ray DELTA_S or rays at rows of their own (the VAR build's cases) always run the plain launch
    if (b->p.launch_mode == RTMI_LAUNCH_SLICED) return sliced_ready(b) ? run_sliced(b, &ev) : run_plain(b, &ev);
    if (b->p.launch_mode == RTMI_LAUNCH_PLAIN || !sliced_ready(b)) return run_plain(b, &ev);
    // RTMI_LAUNCH_AUTO on a batch with more bundles than resident blocks: auto_next / auto_decide above.  Only a run that
    // starts from the launch conditions is a sample.
    const bool fresh = b->launches == 0 && !b->dirty_state;
    int pick = b->auto_kept >= 0 ? b->auto_kept : auto_next(b->auto_n);
    const bool exploring = b->auto_kept < 0;
    rc = pick == 0 ? run_sliced(b, &ev) : run_plain(b, &ev);
    if (rc == RTMI_ERR_STATE && pick == 0) {
        // the sliced launch gave up a wait (it reports instead of hanging); what it advanced is valid state: finish plainly
        b->auto_kept = 1;
        b->auto_fallbacks++;           // countable (rtmi_stats.auto_fallbacks): a wait-bound trip is a scheduler defect signal
        if (getenv("RTMI_DEBUG")) fprintf(stderr, "rtmi: RTMI_LAUNCH_AUTO: the sliced launch gave up a wait (%s); finishing with the plain kernel\n", g_err.c_str());
        return run_plain(b, &ev);
    }
    if (rc == RTMI_OK && exploring && fresh && ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev->first, ev->second) == hipSuccess) {   // the stream is idle (read_counters)
            b->auto_ms[pick][b->auto_n[pick]++] = (double)ms;
            if (auto_next(b->auto_n) < 0) {
                b->auto_kept = auto_decide(b->auto_ms[0], b->auto_n[0], b->auto_ms[1], b->auto_n[1]) == RTMI_LAUNCH_PLAIN ? 1 : 0;
                if (getenv("RTMI_DEBUG"))
                    fprintf(stderr, "rtmi: RTMI_LAUNCH_AUTO: sliced %.3f %.3f %.3f ms, plain %.3f %.3f %.3f ms -> %s\n", b->auto_ms[0][0], b->auto_ms[0][1],
                            b->auto_ms[0][2], b->auto_ms[1][0], b->auto_ms[1][1], b->auto_ms[1][2], b->auto_kept ? "plain" : "sliced");
            }
        }
    }
    return rc;
}

RTMI_EXPORT int rtmi_debug_auto_rule(const double* sliced_ms, int ns, const double* plain_ms, int np, int* next, int* decision) {
    ARG_TRY(ns >= 0 && np >= 0 && ns <= RTMI_AUTO_SAMPLES && np <= RTMI_AUTO_SAMPLES && (sliced_ms || ns == 0) && (plain_ms || np == 0),
            "rtmi_debug_auto_rule: at most RTMI_AUTO_SAMPLES samples per schedule");
    const int n[2] = {ns, np};
    if (next) { const int k = auto_next(n); *next = k < 0 ? -1 : k == 0 ? RTMI_LAUNCH_SLICED : RTMI_LAUNCH_PLAIN; }
    if (decision) *decision = auto_decide(sliced_ms, ns, plain_ms, np);
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_sync(rtmi_batch* b) {
    ARG_TRY(b, "rtmi_sync: null");
    RETRACE_FLUSH(b);
    HIP_TRY(hipStreamSynchronize(b->stream));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_read_d_ray(rtmi_batch* b, double* d_ray) {
    ARG_TRY(b && d_ray, "rtmi_read_d_ray: null");
    DEVICE_TRY(b->field, "rtmi_read_d_ray");
    RETRACE_FLUSH(b);
    const size_t nb = 3 * (size_t)b->R * sizeof(double);
    void* stg = nullptr;
    int rc = batch_staging(b, nb, &stg);
    if (rc) return rc;
    double* d = (double*)stg;
    const dim3 g((unsigned)((b->R + 255) / 256)), blk(256);
    if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_pack_d_ray<double>, g, blk, 0, b->stream, batch_dev<double>(b), d);
    else hipLaunchKernelGGL(k_pack_d_ray<float>, g, blk, 0, b->stream, batch_dev<float>(b), d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(d_ray, d, nb, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_read_d_ray: ") + hipGetErrorString(e));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_read_final(rtmi_batch* b, double* final9) {
    ARG_TRY(b && final9, "rtmi_read_final: null");
    DEVICE_TRY(b->field, "rtmi_read_final");
    RETRACE_FLUSH(b);
    const size_t nb = 9 * (size_t)b->R * sizeof(double);
    void* stg = nullptr;
    int rc = batch_staging(b, nb, &stg);
    if (rc) return rc;
    double* d = (double*)stg;
    const dim3 g((unsigned)((b->R + 255) / 256)), blk(256);
    if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_pack_final<double>, g, blk, 0, b->stream, batch_dev<double>(b), d);
    else hipLaunchKernelGGL(k_pack_final<float>, g, blk, 0, b->stream, batch_dev<float>(b), d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(final9, d, nb, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_read_final: ") + hipGetErrorString(e));
    return RTMI_OK;
}

int rtmi_internal_pack_device(rtmi_batch* b, int what, double* dst, void* stream) {
    ARG_TRY(b && dst, "rtmi_internal_pack_device: null");
    DEVICE_TRY(b->field, "rtmi_shard read-back");
    RETRACE_FLUSH(b);
    HIP_TRY(hipStreamSynchronize(b->stream));
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)((b->R + 255) / 256)), blk(256);
    if (what == 0) {
        if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_pack_d_ray<double>, g, blk, 0, st, batch_dev<double>(b), dst);
        else hipLaunchKernelGGL(k_pack_d_ray<float>, g, blk, 0, st, batch_dev<float>(b), dst);
    } else {
        if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_pack_final<double>, g, blk, 0, st, batch_dev<double>(b), dst);
        else hipLaunchKernelGGL(k_pack_final<float>, g, blk, 0, st, batch_dev<float>(b), dst);
    }
    HIP_TRY(hipGetLastError());
    return RTMI_OK;
}

// rows [n][nq][R] in slot order -> fp64 rows in the caller's ray order
template <typename T> __global__ void k_unpermute_rows(const T* src, double* dst, const int* perm, long R, long nvec) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    const long o = perm ? (long)perm[k] : k;
    for (long v = blockIdx.y; v < nvec; v += gridDim.y) dst[(size_t)v * R + o] = (double)src[(size_t)v * R + k];
}

RTMI_EXPORT int rtmi_read_rows(rtmi_batch* b, int64_t row0, int64_t nrows, double* s_ray, double* n_ray) {
    ARG_TRY(b, "rtmi_read_rows: null");
    ARG_TRY(b->p.record_stride > 0, "rtmi_read_rows: batch keeps no trajectory (record_stride = 0)");
    ARG_TRY(row0 >= 0 && nrows >= 0 && row0 + nrows <= b->p.rec_rows, "rtmi_read_rows: row range outside rec_rows");
    ARG_TRY(!(n_ray && !b->n_ray), "rtmi_read_rows: n_ray requested but the batch keeps none (params.no_n_ray)");
    DEVICE_TRY(b->field, "rtmi_read_rows");
    RETRACE_FLUSH(b);
    if (nrows == 0) return RTMI_OK;
    const size_t R = (size_t)b->R;
    for (int which = 0; which < 2; which++) {
        double* dst = which == 0 ? s_ray : n_ray;
        if (!dst) continue;
        const size_t nq = which == 0 ? 6 : 1, per_row = nq * R;
        const char* src = (const char*)(which == 0 ? b->s_ray : b->n_ray) + (size_t)row0 * per_row * b->esz;
        if (b->p.dtype == RTMI_F64 && !b->perm) {
            HIP_TRY(hipMemcpyAsync(dst, src, per_row * (size_t)nrows * 8, hipMemcpyDeviceToHost, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));
            continue;
        }
        // convert / un-permute on the device through a bounded staging buffer (<= 256 MB)
        const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nrows, (int64_t)((256u << 20) / (per_row * 8))));
        void* stg = nullptr;
        const int rcs = batch_staging(b, (size_t)chunk * per_row * 8, &stg);
        if (rcs) return rcs;
        double* d = (double*)stg;
        hipError_t e = hipSuccess;
        for (int64_t r0 = 0; r0 < nrows && e == hipSuccess; r0 += chunk) {
            const int64_t n = std::min<int64_t>(chunk, nrows - r0);
            const long nvec = (long)(n * (int64_t)nq);
            const dim3 g((unsigned)((R + 255) / 256), (unsigned)std::min<long>(nvec, 1024)), blk(256);
            const char* s0 = src + (size_t)r0 * per_row * b->esz;
            if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_unpermute_rows<double>, g, blk, 0, b->stream, (const double*)s0, d, b->perm, (long)R, nvec);
            else hipLaunchKernelGGL(k_unpermute_rows<float>, g, blk, 0, b->stream, (const float*)s0, d, b->perm, (long)R, nvec);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(dst + (size_t)r0 * per_row, d, (size_t)n * per_row * 8, hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        }
        if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_read_rows: ") + hipGetErrorString(e));
    }
    return RTMI_OK;
}

// ------------------------------------------------------------------ on-device validation metrics
// One lane per ray; outputs fp64 [R].  They read data the trace left in HBM, so a 1 M-ray run is checked
// without copying 100+ GB of trajectories to the host.
template <typename T> __global__ void k_metric_snell(BatchDev<T> a, double* out) {   // RT_bench.py:896-919
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    const int i = a.istep[k];
    const double th = a.th0[k];
    double angreal;
    if (th < M_PI / 4) angreal = 90 - 180 * th / M_PI;                                  // reflection (:903)
    else if (th == M_PI / 4) angreal = 0;
    else angreal = 180 * asin(sqrt(2.0) * sin(M_PI / 2 - th)) / M_PI;                   // refraction (:908)
    const long ra = (long)(9.5 * i / 10), rb = (long)((double)(9 * (long)i) / 10);      // int(9.5*i/10), int(9*i/10) (:913)
    const double distx = (double)a.s_ray[(size_t)ra * 6 * a.R + k] - (double)a.s_ray[(size_t)rb * 6 * a.R + k];
    const double disty = (double)a.s_ray[((size_t)ra * 6 + 1) * a.R + k] - (double)a.s_ray[((size_t)rb * 6 + 1) * a.R + k];
    const double angsim = 180 * atan(fabs(distx / disty)) / M_PI;                       // (:916)
    out[out_index(a, k)] = fabs(angsim - angreal);
}
template <typename T> __global__ void k_metric_closure(BatchDev<T> a, double* out) {  // RT_bench.py:956, :1393
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    // s_ray[-1, 0:2, k]: the last row of the array -- written only if the ray ran all max_size-1 steps
    const bool full = a.istep[k] == max_size_of(a, k) - 1;
    // the row itself is stored in dtype: round the accumulator the way write_row did
    const double dx = 1.0 - (full ? (double)(T)a.acc(0)[k] : 0.0), dy = 0.0 - (full ? (double)(T)a.acc(1)[k] : 0.0);
    out[out_index(a, k)] = 100 * sqrt(fma(dy, dy, dx * dx)) / (2 * M_PI);
}
template <typename T> __global__ void k_metric_px_cv(BatchDev<T> a, double* out) {    // RT_bench.py:1354-1360, :1398-1402
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    const long rows = a.istep[k] / a.stride + 1 < a.rec_rows ? a.istep[k] / a.stride + 1 : a.rec_rows;
    double sum = 0;
    long cnt = 0;
    for (long r = 0; r < rows; r++) {            // np.ma.masked_equal(., 0).compressed(): zeros are skipped
        const double v = (double)a.s_ray[((size_t)r * 6 + 2) * a.R + k];
        if (v != 0) { sum += v; ++cnt; }
    }
    const double mean = sum / (double)cnt;
    double ss = 0;
    for (long r = 0; r < rows; r++) {
        const double v = (double)a.s_ray[((size_t)r * 6 + 2) * a.R + k];
        if (v != 0) ss = fma(v - mean, v - mean, ss);
    }
    out[out_index(a, k)] = 100 * sqrt(ss / (double)cnt) / mean;  // 100*np.std/np.mean
}

RTMI_EXPORT int rtmi_metric(rtmi_batch* b, int kind, double* out) {
    ARG_TRY(b && out, "rtmi_metric: null");
    ARG_TRY(kind >= RTMI_METRIC_SNELL_ERROR && kind <= RTMI_METRIC_PX_CV, "rtmi_metric: unknown metric");
    if (kind == RTMI_METRIC_SNELL_ERROR)
        ARG_TRY(b->p.record_stride == 1 && b->p.rec_rows >= b->p.max_size,
                "rtmi_metric: the exit-angle metric needs the full trajectory (record_stride 1, rec_rows >= max_size)");
    if (kind == RTMI_METRIC_PX_CV) ARG_TRY(b->p.record_stride >= 1, "rtmi_metric: the p_x metric needs recorded rows");
    DEVICE_TRY(b->field, "rtmi_metric");
    RETRACE_FLUSH(b);
    const size_t nb = (size_t)b->R * sizeof(double);
    void* stg = nullptr;
    int rc = batch_staging(b, nb, &stg);
    if (rc) return rc;
    double* d = (double*)stg;
    const dim3 g((unsigned)((b->R + 255) / 256)), blk(256);
#define LAUNCH_(K)                                                                                          \
    do {                                                                                                    \
        if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(K<double>, g, blk, 0, b->stream, batch_dev<double>(b), d); \
        else hipLaunchKernelGGL(K<float>, g, blk, 0, b->stream, batch_dev<float>(b), d);                   \
    } while (0)
    if (kind == RTMI_METRIC_SNELL_ERROR) LAUNCH_(k_metric_snell);
    else if (kind == RTMI_METRIC_CLOSURE) LAUNCH_(k_metric_closure);
    else LAUNCH_(k_metric_px_cv);
#undef LAUNCH_
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, nb, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_metric: ") + hipGetErrorString(e));
    return RTMI_OK;
}

// ------------------------------------------------------------------ isochrones
// Per-ray PCHIP interpolation of (x, y, theta) at fixed traveltimes -- the first stage of the reference's
// wavefront extraction (RT_bench.py:987-1003: scipy PchipInterpolator(t_ray, v_ray) evaluated at travel_time).
// scipy's algorithm restated: shape-preserving derivative estimates (Fritsch-Butland weighted harmonic mean,
// three-point end formula with the two sign guards), cubic Hermite in the power basis of (t - T_j).
namespace {
__device__ __forceinline__ double sgn_(double v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ double pchip_edge(double h0, double h1, double m0, double m1) {
    double d = ((2 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
    if (sgn_(d) != sgn_(m0)) d = 0;
    else if (sgn_(m0) != sgn_(m1) && fabs(d) > 3 * fabs(m0)) d = 3 * m0;
    return d;
}
template <typename T> struct Column {   // s_ray[:, q, k]
    const T* base; size_t pitch;
    __device__ __forceinline__ double operator()(long row) const { return (double)base[(size_t)row * pitch]; }
};
// derivative estimate at point j of an n-point (n >= 3) data set
template <typename T> __device__ __forceinline__ double pchip_deriv(const Column<T>& tt, const Column<T>& yy, long j, long n) {
    if (j == 0) {
        const double h0 = tt(1) - tt(0), h1 = tt(2) - tt(1);
        return pchip_edge(h0, h1, (yy(1) - yy(0)) / h0, (yy(2) - yy(1)) / h1);
    }
    if (j == n - 1) {
        const double h0 = tt(n - 1) - tt(n - 2), h1 = tt(n - 2) - tt(n - 3);
        return pchip_edge(h0, h1, (yy(n - 1) - yy(n - 2)) / h0, (yy(n - 2) - yy(n - 3)) / h1);
    }
    const double ha = tt(j) - tt(j - 1), hb = tt(j + 1) - tt(j);
    const double ma = (yy(j) - yy(j - 1)) / ha, mb = (yy(j + 1) - yy(j)) / hb;
    if (sgn_(ma) != sgn_(mb) || ma == 0 || mb == 0) return 0;
    const double w1 = 2 * hb + ha, w2 = hb + 2 * ha;
    return 1.0 / ((w1 / ma + w2 / mb) / (w1 + w2));
}
}  // namespace

template <typename T> __global__ void k_isochrone(BatchDev<T> a, int ntimes, const double* times, double* out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.R) return;
    // rows 0..last_i of this ray (:993); a batch created with rec_rows < max_size holds only the first rec_rows of them
    const long n = a.istep[k] + 1 < a.rec_rows ? a.istep[k] + 1 : a.rec_rows;
    const size_t pitch = (size_t)6 * a.R;
    const Column<T> tt{a.s_ray + (size_t)4 * a.R + k, pitch};
    const int qsel[3] = {0, 1, 5};                      // x, y, theta (:993)
    for (int it = 0; it < ntimes; it++) {
        const double t = times[it];
        double res[3] = {NAN, NAN, NAN};
        if (n >= 2 && tt(n - 1) >= t && t >= tt(0)) {   // np.max(t_ray) >= travel_time (:997)
            long lo = 0, hi = n - 1;                     // T[lo] <= t < T[hi] (right end closed)
            while (hi - lo > 1) {
                const long mid = (lo + hi) >> 1;
                if (tt(mid) <= t) lo = mid; else hi = mid;
            }
            const double dx = tt(lo + 1) - tt(lo), s = t - tt(lo);
            for (int q = 0; q < 3; q++) {
                const Column<T> yy{a.s_ray + (size_t)qsel[q] * a.R + k, pitch};
                const double y0 = yy(lo), y1 = yy(lo + 1), slope = (y1 - y0) / dx;
                double d0, d1;
                if (n == 2) { d0 = d1 = slope; }
                else { d0 = pchip_deriv(tt, yy, lo, n); d1 = pchip_deriv(tt, yy, lo + 1, n); }
                const double tq = (d0 + d1 - 2 * slope) / dx;           // CubicHermiteSpline coefficients
                const double c0 = tq / dx, c1 = (slope - d0) / dx - tq;
                res[q] = y0 + d0 * s + c1 * (s * s) + c0 * (s * s * s);
            }
        }
        for (int q = 0; q < 3; q++) out[((size_t)it * 3 + q) * a.R + out_index(a, k)] = res[q];
    }
}

int rtmi_internal_fail(int code, const char* msg) { return fail(code, msg); }

// the per-ray isochrone stage with its result left on the device (shared with wavefront.hip)
int rtmi_internal_isochrones_device(rtmi_batch* b, int32_t ntimes, const double* times, double** d_out, long* R, void** stream) {
    ARG_TRY(b && times && d_out, "rtmi_isochrones: null");
    ARG_TRY(ntimes > 0 && ntimes <= 4096, "rtmi_isochrones: ntimes must be in [1, 4096]");
    ARG_TRY(b->p.record_stride == 1, "rtmi_isochrones: needs the full trajectory (record_stride 1)");
    DEVICE_TRY(b->field, "rtmi_isochrones");
    RETRACE_FLUSH(b);
    double *d = nullptr, *dt = nullptr;
    const size_t nb = (size_t)ntimes * 3 * (size_t)b->R * sizeof(double);
    HIP_TRY(hipMalloc(&d, nb));
    hipError_t e = hipMalloc(&dt, ntimes * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(dt, times, ntimes * sizeof(double), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        const dim3 g((unsigned)((b->R + 127) / 128)), blk(128);
        if (b->p.dtype == RTMI_F64) hipLaunchKernelGGL(k_isochrone<double>, g, blk, 0, b->stream, batch_dev<double>(b), (int)ntimes, dt, d);
        else hipLaunchKernelGGL(k_isochrone<float>, g, blk, 0, b->stream, batch_dev<float>(b), (int)ntimes, dt, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(dt);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(RTMI_ERR_HIP, std::string("rtmi_isochrones: ") + hipGetErrorString(e));
    }
    *d_out = d;
    if (R) *R = (long)b->R;
    if (stream) *stream = (void*)b->stream;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_isochrones(rtmi_batch* b, int32_t ntimes, const double* times, double* out) {
    ARG_TRY(out, "rtmi_isochrones: null");
    double* d = nullptr;
    const int rc = rtmi_internal_isochrones_device(b, ntimes, times, &d, nullptr, nullptr);
    if (rc) return rc;
    const hipError_t e = hipMemcpy(out, d, (size_t)ntimes * 3 * (size_t)b->R * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_isochrones: ") + hipGetErrorString(e));
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_batch_view(rtmi_batch* b, rtmi_device_view* v) {
    ARG_TRY(b && v, "rtmi_batch_view: null");
    RETRACE_FLUSH(b);
    const size_t R = (size_t)b->R, e = b->esz;
    double* acc = (double*)b->state;
    char* aux = (char*)(acc + 6 * R);
    v->s_ray = b->s_ray; v->n_ray = b->n_ray;
    v->x = acc; v->y = acc + R; v->theta = acc + 2 * R; v->dist_sim = acc + 3 * R; v->dist_real = acc + 4 * R; v->T = acc + 5 * R;
    v->n = aux; v->gx = aux + R * e; v->gy = aux + 2 * R * e;
    v->perm = b->perm;
    v->istep = b->istep; v->R = b->R; v->rec_rows = b->p.rec_rows; v->dtype = b->p.dtype; v->record_stride = b->p.record_stride;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_batch_stats(rtmi_batch* b, rtmi_stats* s) {
    ARG_TRY(b && s, "rtmi_batch_stats: null");
    RETRACE_FLUSH(b);
    int rc = fold_events(b);
    if (rc) return rc;
    rc = read_counters(b);
    if (rc) return rc;
    s->ray_steps = b->h_counters[0];
    s->live_rays = b->h_counters[1];
    s->kernel_ms = b->kernel_ms;
    s->launches = b->launches;
    s->kernel_ms_total = b->total_kernel_ms;
    s->launches_total = b->total_launches;
    s->auto_fallbacks = b->auto_fallbacks;
    s->retraced = b->rt ? b->rt->scattered : 0u;
    s->retrace_overflow = b->rt ? b->rt->overflow : 0u;
    s->retraced_total = b->rt ? b->rt->total : 0ull;
    s->dispatch_first = b->rt ? b->rt->rot : 0u;
    s->reserved_ = 0;
    s->auto_kept = b->auto_kept < 0 ? 0 : b->auto_kept == 0 ? RTMI_LAUNCH_SLICED : RTMI_LAUNCH_PLAIN;
    for (int k = 0; k < 2; k++) {
        s->auto_n[k] = (uint32_t)b->auto_n[k];
        for (int i = 0; i < RTMI_AUTO_SAMPLES; i++) s->auto_ms[k][i] = i < b->auto_n[k] ? b->auto_ms[k][i] : 0.0;
    }
    hipFuncAttributes fa;
    s->vgprs = s->sgprs = s->lds_bytes = 0;
    s->launch_mode_used = (uint32_t)b->mode_used;
    const void* kfn = b->mode_used == RTMI_LAUNCH_REFILL ? b->kfn_refill : (b->mode_used == RTMI_LAUNCH_SLICED && b->kfn_sliced) ? b->kfn_sliced : b->kfn;
    if (hipFuncGetAttributes(&fa, kfn) == hipSuccess) { s->vgprs = fa.numRegs; s->lds_bytes = (uint32_t)fa.sharedSizeBytes; }
    return RTMI_OK;
}

// ------------------------------------------------------------------ diagnostic: libm-identical sin/cos on the device
__global__ void k_debug_sincos(long n, const double* x, double* s, double* c) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = rt::ex::sin_(x[i]);
    c[i] = rt::ex::cos_(x[i]);
}

RTMI_EXPORT int rtmi_debug_sincos(int64_t n, const double* x, double* s, double* c) {
    ARG_TRY(x && s && c, "rtmi_debug_sincos: null");
    ARG_TRY(n >= 0, "rtmi_debug_sincos: n < 0");
    if (n == 0) return RTMI_OK;
    double* d = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    HIP_TRY(hipMalloc(&d, 3 * nb));
    hipError_t e = hipMemcpy(d, x, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_sincos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (long)n, d, d + n, d + 2 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(s, d + n, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(c, d + 2 * n, nb, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RTMI_ERR_HIP, std::string("rtmi_debug_sincos: ") + hipGetErrorString(e));
    return RTMI_OK;
}
