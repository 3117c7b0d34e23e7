// rt_libm.h -- fp64 sin/cos with the host libm's roundings, for the methods whose results hinge on them.
//
// Why this exists.  The golden-section methods (op5/9/10/11, RT_bench.py:175-199) return the midpoint of a
// 6e-8 bracket that depends only on the sequence of comparisons cost(c) < cost(d); the curvature advancement
// (op3/4/5/10, :361-363) divides a cancelled difference of sines by a curvature that can be as small as 1.5e-8.
// Both amplify a last-bit difference in sin/cos to 1e-8 .. 1e-6 of a trajectory, so "within 1 ulp of the
// reference's libm" is not enough there: the bits have to be the same.
//
// The reference calls numpy's sin/cos, which on x86-64 Linux are glibc's (numpy >= 1.25 routes float64 sin/cos
// to libm).  Third-party dependency, absent from /root/reference: glibc 2.35 (Ubuntu 2.35-0ubuntu3.11 in the
// build image), sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library lineage), FMA variant
// (__sin_fma / __cos_fma, selected at run time on CPUs with FMA + AVX2).  Its published algorithm is restated
// here:
//     |x| < 2^-26 (sin) / 2^-27 (cos)        x / 1
//     |x| < 0.855469                          table step directly on x
//     |x| < 2.426265                          through pi/2 - |x| (pi/2 as hp0 + hp1)
//     |x| < 105414350                         x - n*pi/2 in three pieces (mp1, mp2, pp3, pp4), then by n mod 4
//   table step: x = k/128 + r (k by adding 1.5*2^45), sin(k/128) and cos(k/128) from a table as hi + lo, sin r and
//   cos r - 1 from degree-5/6 polynomials, combined so that the hi part is added last; |x| < 0.126 uses a degree-11
//   polynomial instead.
// The FMA variant contracts a*b+c wherever the compiler chose to; those choices decide last bits, so they are
// written out below as explicit fma() (the file is compiled with -ffp-contract=off).  The table is generated
// from first principles (tools/gen_sincos_table.py); glibc's own table differs from it in the LOW words of 19
// entries by a few units of 2^-105 relative, which can change a result only when the exact value lies within
// 1e-32 relative of a rounding boundary.
// Checked bit for bit against the libm of the build image on 2e9 arguments (tools/check_libm_sincos.c; every range
// above, both functions) and on the device in tests/test_gpu_exact.py::test_device_sincos_is_libm_bit_for_bit; |x| >= 105414350 (glibc's __branred) is outside what
// the path produces and returns the library's ordinary sincos there.
//
// Host/device: the same source compiles under gcc for the CPU-side checker (tools/, tests/) and under hipcc for
// the kernels.
#pragma once
#include "rt_sincos_table.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

namespace rt {
namespace gl {

struct Tab { double v[4 * RT_SINCOS_TAB_ENTRIES]; };
#define RT_SINCOS_TAB_INIT {{RT_SINCOS_TAB_VALUES}}

constexpr double kBig = 0x1.8p45;                       // ulp = 1/128: adding it rounds |x| to k/128
constexpr double kHp0 = 0x1.921fb54442d18p+0;           // pi/2 hi
constexpr double kHp1 = 0x1.1a62633145c07p-54;          // pi/2 lo
constexpr double kHpInv = 0x1.45f306dc9c883p-1;         // 2/pi
constexpr double kToInt = 0x1.8p52;
constexpr double kMp1 = 0x1.921fb58p+0, kMp2 = -0x1.dde973cp-27;
constexpr double kPp3 = -0x1.cb3b398p-55, kPp4 = -0x1.d747f23e32ed7p-83;
constexpr double kSn3 = -0x1.5555555555515p-3, kSn5 = 0x1.11110e829872fp-7;
constexpr double kCs2 = 0.5, kCs4 = -0x1.5555555555535p-5, kCs6 = 0x1.6c16bedd9e239p-10;
constexpr double kS1 = -0x1.5555555555555p-3, kS2 = 0x1.1111111110ecep-7, kS3 = -0x1.a01a019db08b8p-13;
constexpr double kS4 = 0x1.71de27b9a7ed9p-19, kS5 = -0x1.addffc2fcdf59p-26;

RT_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
RT_HD unsigned long long bits_(double x) { return __builtin_bit_cast(unsigned long long, x); }
RT_HD unsigned hi_word_(double x) { return (unsigned)(bits_(x) >> 32) & 0x7fffffffu; }

// |a| < 0.126: a - a^3/3! + ... (degree 11) with the low part da folded in
RT_HD double taylor_sin(double a, double da) {
    const double xx = a * a;
    double p = fma_(xx, kS5, kS4);
    p = fma_(xx, p, kS3);
    p = fma_(xx, p, kS2);
    p = fma_(xx, p, kS1);
    const double t = fma_(xx, fma_(p, a, -(da * 0.5)), da);
    return a + t;
}

// sin(x + dx) for 0.126 <= |x| <= 0.86 (dx: low part, |dx| << ulp-scale of x); sign of x is restored at the end
template <typename TP> RT_HD double table_sin(TP tab, double x, double dx) {
    const double ax = __builtin_fabs(x);
    if (ax < 0.126) return taylor_sin(x, dx);
    if (x <= 0) dx = -dx;
    const double u = kBig + ax;
    const double r = ax - (u - kBig);
    const TP e = tab + 4 * (int)(unsigned)bits_(u);
    const double xx = r * r;
    const double s = r + fma_(r * xx, fma_(xx, kSn5, kSn3), dx);
    const double c = fma_(r, dx, xx * fma_(xx, fma_(xx, kCs6, kCs4), kCs2));
    const double sn = e[0], ssn = e[1], cs = e[2], ccs = e[3];
    double t = fma_(s, ccs, ssn);
    t = fma_(-c, sn, t);
    const double cor = fma_(s, cs, t);
    return __builtin_copysign(sn + cor, x);
}

// cos(x + dx), |x| <= 0.86
template <typename TP> RT_HD double table_cos(TP tab, double x, double dx) {
    const double ax = __builtin_fabs(x);
    if (x < 0) dx = -dx;
    const double u = kBig + ax;
    const double r = (ax - (u - kBig)) + dx;
    const TP e = tab + 4 * (int)(unsigned)bits_(u);
    const double xx = r * r;
    const double s = fma_(r * xx, fma_(xx, kSn5, kSn3), r);
    const double c = xx * fma_(xx, fma_(xx, kCs6, kCs4), kCs2);
    const double sn = e[0], ssn = e[1], cs = e[2], ccs = e[3];
    double t = fma_(-s, ssn, ccs);
    t = fma_(-c, cs, t);
    const double cor = fma_(-s, sn, t);
    return cs + cor;
}

// x -> (a, da, n): x = n*pi/2 + a + da, |a| <= pi/4 (+ a little), for 2.426265 <= |x| < 105414350
RT_HD int reduce(double x, double* a, double* da) {
    const double t = fma_(x, kHpInv, kToInt);
    const double xn = t - kToInt;
    double y = fma_(-xn, kMp1, x);
    y = fma_(-xn, kMp2, y);
    const double t2 = fma_(-xn, kPp3, y);
    double db = fma_(-kPp3, xn, y - t2);
    const double b = fma_(-xn, kPp4, t2);
    db = db + fma_(-xn, kPp4, t2 - b);
    *a = b;
    *da = db;
    return (int)(unsigned)bits_(t) & 3;
}

template <typename TP> RT_HD double by_quadrant(TP tab, double a, double da, int n) {
    const double r = (n & 1) ? table_cos(tab, a, da) : table_sin(tab, a, da);
    return (n & 2) ? -r : r;
}

// true when |x| is in the range the functions below reproduce (everything the path produces)
RT_HD bool in_range(double x) { return hi_word_(x) < 0x419921FBu; }

template <typename TP> RT_HD double sin(TP tab, double x) {
    const unsigned k = hi_word_(x);
    if (k < 0x3e500000u) return x;
    if (k < 0x3feb6000u) return table_sin(tab, x, 0.0);
    if (k < 0x400368fdu) return __builtin_copysign(table_cos(tab, kHp0 - __builtin_fabs(x), kHp1), x);
    double a, da;
    const int n = reduce(x, &a, &da);
    return by_quadrant(tab, a, da, n);
}

template <typename TP> RT_HD double cos(TP tab, double x) {
    const unsigned k = hi_word_(x);
    if (k < 0x3e400000u) return 1.0;
    if (k < 0x3feb6000u) return table_cos(tab, x, 0.0);
    if (k < 0x400368fdu) {
        const double y = kHp0 - __builtin_fabs(x);
        const double a = y + kHp1;
        const double da = (y - a) + kHp1;
        return table_sin(tab, a, da);
    }
    double a, da;
    const int n = reduce(x, &a, &da);
    return by_quadrant(tab, a, da, n + 1);
}

}  // namespace gl
}  // namespace rt
