// rt_polytab.h -- the field as one polynomial per grid cell (the lookup of the fast-form step methods).
//
// FITPACK's tensor-product splines (RT_bench.py:455-457: RectBivariateSpline of n, dn/dx, dn/dy) are piecewise polynomials:
// on a grid cell [x_j, x_j+1] x [y_i, y_i+1] each of them IS a bicubic (n: bilinear) polynomial.  The B-spline form pays for
// that polynomial with two cubic bases (24 flops) and a 4x4 tensor sum per spline (2 x 20 fma) at every lookup; the power
// form in the cell's own coordinates (u, v) in [0, 1)^2 needs no basis and 15 fma per spline (Horner in u, then in v).
// What it costs is memory -- 36 numbers per cell instead of 3 per grid node, 127 MB for the vert_heterogeneous grid in fp64:
// nothing on a 288 GB device -- and that is the trade this file makes.  The rays of a wave travel together (a 1 M-ray fan: 64
// neighbouring rays span 4 % of a cell, 1.02 distinct cells per wave-step on average), so a cell's 36 coefficients are
// wave-uniform almost always and come through the scalar cache into SGPRs (rt_device.h, PolyGather): no LDS tile to stage,
// no window in vector registers.
//
// Conversion, per axis (host, long double): for cell j the four cubic B-splines of the interpolating knot vector
// t = [x0 x4, x[2..m-3], x[m-1] x4] that are non-zero on it, by the de Boor-Cox recursion carried out on polynomial
// coefficients in the cell coordinate u, where x = a + (j + u)/inv_h is exactly the map the device inverts
// (u = (x - a)*inv_h - j); likewise the two linear ones of n's spline.  The TRUE knots (numpy.linspace's roundings, the
// double-width not-a-knot intervals at both ends) enter here, so every cell of the grid has its polynomial and the
// lookup has no rim case other than FITPACK's argument clamp (quirk Q4).
// Conversion, per cell (device, fp64, fixed order): A[k][p] = sum_r My[r][k] * (sum_q c[r][q] * Mx[q][p]).
//
// Entry layout (kPolyStride numbers, 64-byte lines): [0..15] d/dx spline A[k][p] at 4k+p (k: power of v, p: power of u),
// [16..31] d/dy spline, [32..35] n = b0 + b1 u + b2 v + b3 u v, [36..39] unused.
// Evaluation (rt::poly_eval): r_k = ((A[k][3] u + A[k][2]) u + A[k][1]) u + A[k][0];  value = ((r_3 v + r_2) v + r_1) v + r_0.
#pragma once
#include <cmath>
#include <vector>

#if defined(__HIPCC__)
#define RT_PT_HD __host__ __device__
#else
#define RT_PT_HD
#endif

namespace rt {

constexpr int kPolyStride = 40;

// cubic knot interval of cell j on an m-point axis (the 4-coefficient window starts at l - 3)
RT_PT_HD inline int poly_interval(int j, int m) { const int l = j + 2; return l < 3 ? 3 : (l > m - 1 ? m - 1 : l); }

struct PolyAxis {
    std::vector<double> C;   // [m-1][4][4]: C[j][q][p] = coefficient of u^p of the q-th cubic B-spline of cell j's window
    std::vector<double> L;   // [m-1][2][2]: the two linear B-splines (coefficients j, j+1 of n's spline)
};

// x: the axis as numpy.linspace made it (genZ, :429); origin, inv_h: the device's map u = (x - origin)*inv_h - j in the
// lookup's precision (fp32 fields: both rounded to float).
inline PolyAxis poly_axis_build(const std::vector<double>& x, double origin, double inv_h) {
    typedef long double R;
    const int m = (int)x.size();
    std::vector<R> t(m + 4);
    for (int i = 0; i <= 3; i++) { t[i] = x[0]; t[m + 3 - i] = x[m - 1]; }
    for (int i = 4, j = 2; i < m; i++, j++) t[i] = x[j];
    PolyAxis A;
    A.C.assign((size_t)(m - 1) * 16, 0.0);
    A.L.assign((size_t)(m - 1) * 4, 0.0);
    const R s = (R)1 / (R)inv_h;                    // x = origin + (j + u) * s
    for (int j = 0; j < m - 1; j++) {
        const int l = poly_interval(j, m);
        const R xc = (R)origin + (R)j * s;          // x at u = 0
        // B[i][p]: B-spline with global index l - 3 + i at the current degree, coefficient of u^p
        R B[4][4] = {{0}};
        B[3][0] = 1;                                // degree 0: N_l = 1 on [t_l, t_l+1)
        for (int k = 1; k <= 3; k++) {
            R Nw[4][4] = {{0}};
            for (int gi = l - k; gi <= l; gi++) {
                const int li = gi - (l - 3);
                // (x - t_gi) / (t_gi+k - t_gi) * N_{gi,k-1}
                if (gi >= l - k + 1) {
                    const R d = t[gi + k] - t[gi];
                    if (d != 0) {
                        const R c0 = (xc - t[gi]) / d, c1 = s / d;
                        for (int p = 0; p < 4; p++) {
                            Nw[li][p] += c0 * B[li][p];
                            if (p + 1 < 4) Nw[li][p + 1] += c1 * B[li][p];
                        }
                    }
                }
                // (t_gi+k+1 - x) / (t_gi+k+1 - t_gi+1) * N_{gi+1,k-1}
                if (gi + 1 <= l) {
                    const R d = t[gi + k + 1] - t[gi + 1];
                    if (d != 0) {
                        const R c0 = (t[gi + k + 1] - xc) / d, c1 = -s / d;
                        for (int p = 0; p < 4; p++) {
                            Nw[li][p] += c0 * B[li + 1][p];
                            if (p + 1 < 4) Nw[li][p + 1] += c1 * B[li + 1][p];
                        }
                    }
                }
            }
            for (int i = 0; i < 4; i++) for (int p = 0; p < 4; p++) B[i][p] = Nw[i][p];
        }
        for (int q = 0; q < 4; q++) for (int p = 0; p < 4; p++) A.C[(size_t)j * 16 + q * 4 + p] = (double)B[q][p];
        const R d = (R)x[j + 1] - (R)x[j];
        A.L[(size_t)j * 4 + 0] = (double)(((R)x[j + 1] - xc) / d);
        A.L[(size_t)j * 4 + 1] = (double)(-s / d);
        A.L[(size_t)j * 4 + 2] = (double)((xc - (R)x[j]) / d);
        A.L[(size_t)j * 4 + 3] = (double)(s / d);
    }
    return A;
}

// One cell's 36 coefficients from the B-spline coefficient arrays (row-major [qy][qx], y the row index, :455-457) and the two
// axes' tables.  Plain fp64 fma in a fixed order: the same bits on the host (tests/native/polytab_check.cpp) and on the device.
RT_PT_HD inline void poly_cell_convert(const double* Z, const double* cdx, const double* cdy, int qx, int qy, int jx, int jy,
                                    const double* Cx, const double* Lx, const double* Cy, const double* Ly, double out[36]) {
    const int lx = poly_interval(jx, qx), ly = poly_interval(jy, qy);
    const double* mx = Cx + (size_t)jx * 16;
    const double* my = Cy + (size_t)jy * 16;
    for (int s = 0; s < 2; s++) {
        const double* c = (s == 0 ? cdx : cdy) + (size_t)(ly - 3) * qx + (lx - 3);
        double tmp[4][4];
        for (int r = 0; r < 4; r++)
            for (int p = 0; p < 4; p++) {
                double acc = c[(size_t)r * qx] * mx[p];
                for (int q = 1; q < 4; q++) acc = __builtin_fma(c[(size_t)r * qx + q], mx[q * 4 + p], acc);
                tmp[r][p] = acc;
            }
        for (int k = 0; k < 4; k++)
            for (int p = 0; p < 4; p++) {
                double acc = my[k] * tmp[0][p];
                for (int r = 1; r < 4; r++) acc = __builtin_fma(my[r * 4 + k], tmp[r][p], acc);
                out[s * 16 + k * 4 + p] = acc;
            }
    }
    const double* z = Z + (size_t)jy * qx + jx;
    const double* lx2 = Lx + (size_t)jx * 4;     // [q][p]
    const double* ly2 = Ly + (size_t)jy * 4;     // [r][k]
    for (int k = 0; k < 2; k++)
        for (int p = 0; p < 2; p++) {
            double acc = 0.0;
            for (int r = 0; r < 2; r++)
                for (int q = 0; q < 2; q++) acc = __builtin_fma(z[(size_t)r * qx + q] * ly2[r * 2 + k], lx2[q * 2 + p], acc);
            out[32 + 2 * k + p] = acc;        // b0 (1), b1 (u), b2 (v), b3 (u v)
        }
}

// A FLAT cell: the medium is constant there as far as a ray can tell -- every one of the 32 coefficients of the two gradient
// polynomials is at most thr in magnitude (thr = 2^-80 of the largest gradient-spline coefficient of the grid: the global
// spline of a piecewise-constant medium decays geometrically away from the transitions but never reaches zero; 2^-80 of the
// field's gradient scale moves an angle by 1e-30 per step) and n's polynomial varies by less than 2^-50 of its constant term
// across the cell (the rounding residue of the bilinear weights of equal samples).  The lookup then answers (b0, 0, 0) without
// touching the cell's 36 coefficients (rt::PolyGather): both flanks of the interface scenario's sigmoid, i.e. most of its steps.
// The rule is part of the lookup's definition -- every path applies it -- so results do not depend on which path served a lane.
constexpr double kPolyFlatRel = 0x1p-80, kPolyFlatN = 0x1p-50;
RT_PT_HD inline bool poly_cell_flat(const double c[36], double thr) {
    for (int i = 0; i < 32; i++)
        if (!(__builtin_fabs(c[i]) <= thr)) return false;
    const double t = kPolyFlatN * __builtin_fabs(c[32]);
    return __builtin_fabs(c[33]) <= t && __builtin_fabs(c[34]) <= t && __builtin_fabs(c[35]) <= t;
}

}  // namespace rt
