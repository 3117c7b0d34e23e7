// Test helper (CPU, g++): the per-cell polynomial table of raytracing_amd/csrc/rt_polytab.h built on the host with the very
// functions the library uses (poly_axis_build, poly_cell_convert), and the lookup rt::PolyGather performs on it restated in
// plain C++ (same operation order, explicit fma).  tests/test_polytab_host.py compares it with FITPACK's arithmetic (the
// oracle's n_gradient) -- the mathematics of the conversion, checked without a GPU.
#include <cmath>
#include <cstddef>
#include <vector>

#include "../../raytracing_amd/csrc/rt_polytab.h"

extern "C" int polytab_build(const double* x, int qx, const double* y, int qy, const double* Z, const double* cdx,
                             const double* cdy, double inv_hx, double inv_hy, double* out) {
    const rt::PolyAxis AX = rt::poly_axis_build(std::vector<double>(x, x + qx), x[0], inv_hx);
    const rt::PolyAxis AY = rt::poly_axis_build(std::vector<double>(y, y + qy), y[0], inv_hy);
    // the flat-cell rule of the lookup (rt::poly_cell_flat): threshold from the largest gradient-spline coefficient, like
    // the library's field build; the verdict is kept in the entry's first unused slot here (the library keeps a map of its own)
    double gmax = 0.0;
    for (size_t i = 0; i < (size_t)qx * qy; i++) gmax = std::fmax(gmax, std::fmax(std::fabs(cdx[i]), std::fabs(cdy[i])));
    const double thr = gmax * rt::kPolyFlatRel;
    for (int jy = 0; jy < qy - 1; jy++)
        for (int jx = 0; jx < qx - 1; jx++) {
            double* o = out + ((size_t)jy * (qx - 1) + jx) * rt::kPolyStride;
            rt::poly_cell_convert(Z, cdx, cdy, qx, qy, jx, jy, AX.C.data(), AX.L.data(), AY.C.data(), AY.L.data(), o);
            for (int i = 36; i < rt::kPolyStride; i++) o[i] = 0.0;
            o[36] = rt::poly_cell_flat(o, thr) ? 1.0 : 0.0;
        }
    return 0;
}

static double horner3(const double* a, double u) { return std::fma(std::fma(std::fma(a[3], u, a[2]), u, a[1]), u, a[0]); }

extern "C" void polytab_eval(const double* tab, int qx, int qy, double ax, double bx, double inv_hx, double ay, double by,
                             double inv_hy, long npts, const double* px, const double* py, double* n, double* gx, double* gy) {
    for (long i = 0; i < npts; i++) {
        double x = px[i], y = py[i];
        double xa = x - ax, ya = y - ay;
        double urx = xa * inv_hx, ury = ya * inv_hy;
        double jfx = std::floor(urx), jfy = std::floor(ury);
        if (!(jfx >= 0 && jfx < qx - 1)) {          // FITPACK's argument clamp (Q4)
            x = x < ax ? ax : (x > bx ? bx : x);
            xa = x - ax; urx = xa * inv_hx; jfx = std::floor(urx);
            jfx = jfx < 0 ? 0 : (jfx > qx - 2 ? qx - 2 : jfx);
        }
        if (!(jfy >= 0 && jfy < qy - 1)) {
            y = y < ay ? ay : (y > by ? by : y);
            ya = y - ay; ury = ya * inv_hy; jfy = std::floor(ury);
            jfy = jfy < 0 ? 0 : (jfy > qy - 2 ? qy - 2 : jfy);
        }
        const double u = std::fma(xa, inv_hx, -jfx), v = std::fma(ya, inv_hy, -jfy);   // the exact product minus the cell index
        const double* p = tab + ((size_t)(int)jfy * (qx - 1) + (int)jfx) * rt::kPolyStride;
        double r[4];
        for (int k = 0; k < 4; k++) r[k] = horner3(p + 4 * k, u);
        gx[i] = horner3(r, v);
        for (int k = 0; k < 4; k++) r[k] = horner3(p + 16 + 4 * k, u);
        gy[i] = horner3(r, v);
        n[i] = std::fma(std::fma(p[35], u, p[34]), v, std::fma(p[33], u, p[32]));
        if (p[36] != 0.0) { n[i] = p[32]; gx[i] = 0.0; gy[i] = 0.0; }      // a flat cell answers (b0, 0, 0)
    }
}

// Markstein's division as rt::ex::mdiv does it on the device (raytracing_amd/csrc/rt_exact.h): a / d from r = RN(1 / d).
// Returns how many of n (a, d) pairs give a quotient that is not the IEEE one; d runs over the knot differences of an m-point
// linspace axis [lo, hi] (1, 2 and 3 pitches wide, with linspace's roundings), a over [0, 1] (fpbspl's partial weights).
extern "C" long mdiv_mismatches(double lo, double hi, int m, long n, unsigned long long seed) {
    std::vector<double> x(m);
    const double step = (hi - lo) / (double)(m - 1);
    for (int i = 0; i < m; i++) x[i] = (double)i * step + lo;
    x[m - 1] = hi;
    long bad = 0;
    unsigned long long s = seed;
    auto next = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return s; };
    for (long k = 0; k < n; k++) {
        const int j = (int)(next() >> 33) % (m - 3);
        const int w = 1 + (int)((next() >> 40) % 3);
        const double d = x[j + w] - x[j];
        const double a = (double)(next() >> 11) * 0x1.0p-53;
        const double r = 1.0 / d;
        const double q0 = a * r;
        const double q = std::fma(std::fma(-d, q0, a), r, q0);
        if (q != a / d) bad++;
    }
    return bad;
}
