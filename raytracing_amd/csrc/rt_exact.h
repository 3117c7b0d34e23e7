// rt_exact.h -- the step arithmetic of the methods whose results hinge on last bits, in the reference's own
// operation order (fp64 only).  Included by rt_device.h.
//
// For op3/4/5/9/10/11 (curvature advancement and/or golden-section angle search) a 1-ulp difference in an
// intermediate value can move a trajectory by 1e-8 .. 1e-6, so those methods do not use the
// fused / reciprocal / uniform-knot forms of rt_device.h.  Everything here is written operation by operation like
// the reference's scalar numpy code as restated in oracle/rt_oracle.c (same association, no contraction, IEEE
// division and square root, libm-identical sin/cos from rt_libm.h, numpy's SVML arctan2 restated in atan2_ below), so that
// a ray's state is the SAME BITS as the oracle's after every step.  Reference lines are RT_bench.py.
//
// The golden-section search keeps the reference's comparison sequence without paying for 74 such cost evaluations
// per step: see golden_filtered below.
#pragma once
#include "rt_libm.h"
#include "rt_golden_rot.h"
#include "rt_rcp14_table.h"

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
// Both of one angle in one call: the two evaluations are glibc's own (gl::sin, gl::cos: same operations, same bits as sin_ and
// cos_), inlined side by side so that what they have in common -- the range test, |x|, the table index and entry, the
// polynomial pieces in r^2, the reduction by pi/2 -- is computed once, and one call is paid instead of two.
struct SinCos { double s, c; };
__device__ __attribute__((noinline)) SinCos sincos_(double x) {
    SinCos r;
    if (gl::in_range(x)) { r.s = gl::sin(kSinCosTab, x); r.c = gl::cos(kSinCosTab, x); }
    else sincos_k(x, &r.s, &r.c);
    return r;
}
// glibc's sin/cos table (3.5 KB) in LDS for the inline evaluations below: the reference-order kernels stream 18 KB of field
// coefficients per wave-step through the vector L1, which keeps evicting the table's lines (52 of an interface x op9 step's
// 86 vector-memory instructions are table reads).  Staged once per block by stage_sincos_tab() (rtmi.hip calls it at the top of
// every kernel body that can reach sincos_inline); the out-of-line sincos_ / sin_ / cos_ keep reading the table in memory.
#ifndef RTMI_LDS_SINCOS
#define RTMI_LDS_SINCOS 1
#endif
__device__ __forceinline__ RT_LDS double* sincos_tab_lds() {
    __shared__ double t[4 * RT_SINCOS_TAB_ENTRIES];
    return (RT_LDS double*)t;
}
__device__ __forceinline__ void stage_sincos_tab() {       // every thread of the block
#if RTMI_LDS_SINCOS
    RT_LDS double* t = sincos_tab_lds();
    for (int i = (int)threadIdx.x; i < 4 * RT_SINCOS_TAB_ENTRIES; i += (int)blockDim.x) t[i] = kSinCosTab[i];
    __syncthreads();
#endif
}
#if RTMI_LDS_SINCOS
#define RT_EX_SINCOS_TAB ((const RT_LDS double*)sincos_tab_lds())
#else
#define RT_EX_SINCOS_TAB kSinCosTab
#endif
// sin and cos of one angle, glibc's bits, INLINE for the two ranges a golden-section bracket point lies in (theta -+ h with
// h <= 0.6: 2^-26 <= |x| < 0.855469, the table step on x itself, and |x| < 2.426265, through pi/2 - |x|); anything else
// calls sincos_().  Same operations as gl::sin / gl::cos: same bits.  For exact_lt_iso below, which pays this twice per
// undecided comparison -- thirteen times per step where the index is constant (see there) -- and for the step's own sin / cos
// of the methods that gain from having them inline (sincos_sel, inline_sincos).
__device__ __forceinline__ SinCos sincos_inline(double x) {
    const unsigned k = gl::hi_word_(x);
    SinCos r;
    if (k - 0x3e500000u < 0x3feb6000u - 0x3e500000u) {
        r.s = gl::table_sin(RT_EX_SINCOS_TAB, x, 0.0);
        r.c = gl::table_cos(RT_EX_SINCOS_TAB, x, 0.0);
    } else if (k - 0x3feb6000u < 0x400368fdu - 0x3feb6000u) {
        const double y = gl::kHp0 - __builtin_fabs(x);
        r.s = __builtin_copysign(gl::table_cos(RT_EX_SINCOS_TAB, y, gl::kHp1), x);
        const double a = y + gl::kHp1;
        const double da = (y - a) + gl::kHp1;
        r.c = gl::table_sin(RT_EX_SINCOS_TAB, a, da);
    } else {
        r = sincos_(x);
    }
    return r;
}
// Which methods take their step's sin / cos inline: measured per method (profiles/r04_ab_inline_sincos.txt) -- the curvature and
// golden-section methods gain 1-6 %, op1/2/6/7/8 in reference order lose 11-42 % (their kernels are smaller and were not
// short of registers before).  Same operations either way: same bits.
constexpr bool inline_sincos(int method) { return method == 3 || method == 4 || method == 5 || method >= 9; }
template <bool INL> __device__ __forceinline__ SinCos sincos_sel(double x) {
    if constexpr (INL) return sincos_inline(x);
    else return sincos_(x);
}
// np.arctan2 as the reference's numpy evaluates it (AVX512_SKX builds: Intel SVML's __svml_atan28_ha; not libm's atan2 in the
// last bit for 7 % of arguments) -- same text as oracle/rt_oracle.c np_arctan2, see there; tools/check_np_atan2.py: 0
// mismatches against np.arctan2 on 1.6e7 argument pairs.  Its reciprocal starts from the VRCP14PD instruction, which is a
// table of the operand's top 16 mantissa bits (rt_rcp14_table.h; decoded into g_rcp14 once per device by k_rcp14_init).
__device__ unsigned short g_rcp14[65536];
__device__ const unsigned long long kRcp14Words[RT_RCP14_WORDS] = {RT_RCP14_DELTAS};
__device__ __forceinline__ double vrcp14pd(double x) {
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x);
    const unsigned long long e = (xb >> 52) & 0x7ffull, m = xb & 0xfffffffffffffull;
    const unsigned long long rb = m == 0ull ? (0x7feull - e) << 52 : ((0x7fdull - e) << 52) | ((unsigned long long)g_rcp14[m >> 36] << 36);
    return __builtin_bit_cast(double, rb);
}
__device__ __attribute__((noinline)) double atan2_(double y, double x) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const int ix = (int)((unsigned)(__builtin_bit_cast(unsigned long long, ax) >> 32) - 0x80300000u);
    const int iy = (int)((unsigned)(__builtin_bit_cast(unsigned long long, ay) >> 32) - 0x80300000u);
    if (ix >= (int)0xfdd00000u || iy >= (int)0xfdd00000u) return ::atan2(y, x);   // zeros, infinities, 2^-1020 > |.| or |.| >= 2^993
    const bool k5 = 0.4375 * ax < ay, k1 = 0.6875 * ax < ay, k2 = 1.1875 * ax < ay, k3 = 2.4375 * ax < ay;
    const double c = k2 ? (k3 ? 1.0 : 1.5) : (k1 ? 1.0 : 0.5);
    const double ahi = k2 ? (k3 ? 0x1.921fb54442d18p+0 : 0x1.f730bd281f69bp-1) : (k1 ? 0x1.921fb54442d18p-1 : 0x1.dac670561bb4fp-2);
    const double alo = k2 ? (k3 ? 0x1.1a62633145c07p-54 : 0x1.007887af0cbbdp-56) : (k1 ? 0x1.1a62633145c07p-55 : 0x1.a2b7f222f65e2p-56);
    double den = k3 ? 0.0 : ax, num = k3 ? 0.0 : ay;
    if (k5) { den = __builtin_fma(c, ay, den); num = __builtin_fma(-c, ax, num); }
    double r = vrcp14pd(den);
    double e = __builtin_fma(-r, den, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-r, den, 1.0);
    r = __builtin_fma(e, r, r);
    const double q = num * r, q2 = q * q;
    const double res = __builtin_fma(-q, den, num);
    const double q4 = q2 * q2;
    double ql = res * r;
    if (k5) ql = ql + alo;
    double A = __builtin_fma(0x1.be4fbe6733718p-7, q4, 0x1.6ad5558fe19c9p-5), B = __builtin_fma(-0x1.04cd71f92185ep-5, q4, -0x1.a9e755ca13d23p-5);
    A = __builtin_fma(q4, A, 0x1.e12f1edf7c393p-5);  B = __builtin_fma(q4, B, -0x1.1108d326c68edp-4);
    A = __builtin_fma(q4, A, 0x1.3b132b731e73ap-4);  B = __builtin_fma(q4, B, -0x1.745d119677a4fp-4);
    A = __builtin_fma(q4, A, 0x1.c71c719f99f96p-4);  B = __builtin_fma(q4, B, -0x1.2492492441a21p-3);
    A = __builtin_fma(q4, A, 0x1.9999999998f43p-3);  B = __builtin_fma(q4, B, -0x1.5555555555552p-2);
    double t = __builtin_fma(q2, A, B) * q2;
    const bool xneg = x < 0.0;
    if (xneg) ql = ql - 0x1.1a64000000000p-53;
    t = __builtin_fma(q, t, ql);
    double s = q + t;
    if (k5) s = s + ahi;
    if (xneg) s = -s + 0x1.921fb54442d18p+1;
    return __builtin_copysign(s, y);
}
__device__ __forceinline__ double sqrt_(double x) { return __dsqrt_rn(x); }            // correctly rounded
__device__ __forceinline__ double sq(double x) { return x * x; }                        // oracle SQ()
__device__ __forceinline__ double dot2(double a0, double a1, double b0, double b1) { return __builtin_fma(a1, b1, a0 * b0); }
__device__ __forceinline__ double norm2(double a0, double a1) { return sqrt_(dot2(a0, a1, a0, a1)); }
__device__ __forceinline__ double aniso(double s, double c, double gamma) { return sqrt_(sq(gamma * s) + sq(c)); }   // :118-119
// a / d from r = RN(1 / d): Markstein's division (see axis_exact below) -- the correctly rounded quotient in 3 instructions
__device__ __forceinline__ double mdiv(double a, double d, double r) {
    const double q0 = a * r;
    return fma_(fma_(-d, q0, a), r, q0);
}
// moment() (:217-230) with coef = anisotropy(theta, gamma) passed in
__device__ __forceinline__ double moment(double n, double coef, double g2m1, double o0, double o1) {
    return n * coef * o0 * (1.0 + o1 * g2m1 / sq(coef));
}
// both moments of one direction (cos, sin) = (c, s): they divide by the same coef^2 -- one IEEE reciprocal and two Markstein
// divisions give the two IEEE quotients (17 instructions for 22)
__device__ __forceinline__ void moments(double n, double coef, double g2m1, double c, double s, double& mx, double& my) {
    const double d = sq(coef), r = 1.0 / d;
    mx = n * coef * c * (1.0 + mdiv(-sq(s) * g2m1, d, r));
    my = n * coef * s * (1.0 + mdiv(sq(c) * g2m1, d, r));
}
__device__ __forceinline__ double impulse(double a, double b, double step) { return step * (a + b) / 2.0; }          // :214

// ---------------------------------------------------------------- n_gradient in FITPACK's arithmetic
// One axis: fpbisp's clamp + interval search (through rt::locate, exact on the true knots) and fpbspl for k = 1
// and k = 3 with IEEE divisions and separate multiply/add, exactly as oracle/rt_oracle.c fpbspl().
//
// The seven divisions of one axis all divide by a difference of knots, and a cell has only one set of them: the field build
// tabulates their CORRECTLY ROUNDED reciprocals per cell index (rd[j][0..6], fp_axis_tab_build on the host: the same knots by
// the same operations, then an IEEE division).  1.0 / d is then the table value itself, and a / d is Markstein's division
//     q0 = a r,   e = a - d q0  (exact: one fma),   q = q0 + e r  (one fma)      with r = RN(1 / d)
// which returns the correctly rounded quotient (Markstein 1990; Cornea, Harrison, Tang: Scientific Computing on Itanium,
// thm 8.3 -- no overflow or underflow here: 0 <= a <= 1, d ~ the grid pitch): 3 instructions for the 11 of the IEEE sequence
// (v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup), 110 fewer per lookup.  Bits: every test that holds a
// reference-order method to the oracle's bits runs through it (the oracle divides).
// The table (rtmi.hip, fp_axis_tab_build) holds kAxisTab doubles per cell index: the seven reciprocals and a zero (what this
// function reads), then the cell's knots and the seven differences themselves for a wave that shares ONE cell (AxisTab below).
constexpr int kAxisTab = 24;
__device__ __forceinline__ void axis_exact(double v, int q, double a, double h, double b, double ih, const double* rd, int& j, int& l,
                                           double wl[2], double w[4]) {
    double t0, t1;
    j = locate(v, q, a, h, b, ih, t0, t1);      // v is clamped in place (quirk Q4)
    // the cell's seven reciprocals: two 32-byte rows (L1-resident: q x 64 bytes per axis)
    typedef double Quad4 __attribute__((ext_vector_type(4)));
    const Quad4 ra = *reinterpret_cast<const Quad4*>(rd + (size_t)j * kAxisTab), rb = *reinterpret_cast<const Quad4*>(rd + (size_t)j * kAxisTab + 4);
    {   // k = 1 on [t0, t1]
        const double f = ra.x;                  // 1.0 / (t1 - t0)
        wl[0] = 0.0 + f * (t1 - v);
        wl[1] = f * (v - t0);
    }
    l = j + 2;
    l = l < 3 ? 3 : (l > q - 1 ? q - 1 : l);
    const double tm2 = knot3(l - 2, q, a, h, b), tm1 = knot3(l - 1, q, a, h, b), k0 = knot3(l, q, a, h, b);
    const double k1 = knot3(l + 1, q, a, h, b), k2 = knot3(l + 2, q, a, h, b), k3 = knot3(l + 3, q, a, h, b);
    // j = 1
    double f = ra.y;                            // 1.0 / (k1 - k0)
    double h0 = 0.0 + f * (k1 - v), h1 = f * (v - k0), h2, h3;
    // j = 2
    double hh0 = h0, hh1 = h1, hh2;
    f = mdiv(hh0, k1 - tm1, ra.z);
    h0 = 0.0 + f * (k1 - v);
    h1 = f * (v - tm1);
    f = mdiv(hh1, k2 - k0, ra.w);
    h1 = h1 + f * (k2 - v);
    h2 = f * (v - k0);
    // j = 3
    hh0 = h0; hh1 = h1; hh2 = h2;
    f = mdiv(hh0, k1 - tm2, rb.x);
    h0 = 0.0 + f * (k1 - v);
    h1 = f * (v - tm2);
    f = mdiv(hh1, k2 - tm1, rb.y);
    h1 = h1 + f * (k2 - v);
    h2 = f * (v - tm1);
    f = mdiv(hh2, k3 - k0, rb.z);
    h2 = h2 + f * (k3 - v);
    h3 = f * (v - k0);
    w[0] = h0; w[1] = h1; w[2] = h2; w[3] = h3;
}

__device__ __forceinline__ void field_locate(const FieldDev<double>& F, double x, double y, Cell<double>& c) {
    axis_exact(x, F.qx, F.ax, F.hx, F.bx, F.inv_hx, F.rdx, c.jx, c.lx, c.lwx, c.wx);
    axis_exact(y, F.qy, F.ay, F.hy, F.by, F.inv_hy, F.rdy, c.jy, c.ly, c.lwy, c.wy);
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

// ---------------------------------------------------------------- the same lookup for a wave whose live lanes share ONE cell
// The rays of a wave leave one origin with neighbouring angles: on a fan 64 of them sit in one grid cell on 98 % of their steps.
// Everything of the lookup that depends only on the cell is then the same number in every lane -- the 36 coefficients of the
// window, the knots, the seven knot differences of each axis and their reciprocals -- and is read ONCE per wave through the
// scalar cache into scalar registers, where the vector instructions take it as their scalar operand: no vector load (18 + 4
// per lookup in the per-lane form), no address arithmetic, no knot arithmetic, and none of the 72 vector registers the window
// is staged in.  The operations on the per-lane numbers (x, y, the basis values, the sums) are axis_exact's and
// field_combine's, one for one: the same bits.  A wave in several cells takes the per-lane form.
#ifndef RTMI_EXACT_UNIFORM
#define RTMI_EXACT_UNIFORM 1
#endif
// The per-lane fallback reads its 4 x 4 window in this many groups of rows (1: all 36 coefficients in flight, 72 staging registers).
// fpbisp's sums run row by row anyway -- same order, same bits -- and with the window path in front of it the fallback no longer
// has to be the fast one, only not to set the kernel's register count: in two groups op3's kernel has 117 instead of 143 vector
// registers (four waves per SIMD) and is 4.7 % faster, op4 121 / 147, +5.5 %; op5 and op9 lose their scratch; cfg5 +1.6 %, in four
// groups +2.1 %.  Per method, because the register allocation of the kernels at the 168-register cap answers in its own way:
// interface x op9 loses 6-12 % with either (profiles/r04_ab_exact_fallback_phases.txt) and keeps the single group.
constexpr int fallback_phases(int method) { return method == 9 ? 1 : method == 11 ? 4 : 2; }
struct AxisTab { double r[8]; double t0, t1, tm2, tm1, k0, k1, k2, k3; double d[8]; };   // one cell index of one axis
static_assert(sizeof(AxisTab) == kAxisTab * sizeof(double), "fp_axis_tab_build writes this layout");
typedef const AxisTab __attribute__((address_space(4)))* AxisTabS;
__device__ __forceinline__ void axis_uniform(double v, AxisTabS t, double wl[2], double w[4]) {
    {   // k = 1 on [t0, t1]
        const double f = t->r[0];
        wl[0] = 0.0 + f * (t->t1 - v);
        wl[1] = f * (v - t->t0);
    }
    const double tm2 = t->tm2, tm1 = t->tm1, k0 = t->k0, k1 = t->k1, k2 = t->k2, k3 = t->k3;
    double f = t->r[1];
    double h0 = 0.0 + f * (k1 - v), h1 = f * (v - k0), h2, h3;
    double hh0 = h0, hh1 = h1, hh2;
    f = mdiv(hh0, t->d[2], t->r[2]);
    h0 = 0.0 + f * (k1 - v);
    h1 = f * (v - tm1);
    f = mdiv(hh1, t->d[3], t->r[3]);
    h1 = h1 + f * (k2 - v);
    h2 = f * (v - k0);
    hh0 = h0; hh1 = h1; hh2 = h2;
    f = mdiv(hh0, t->d[4], t->r[4]);
    h0 = 0.0 + f * (k1 - v);
    h1 = f * (v - tm2);
    f = mdiv(hh1, t->d[5], t->r[5]);
    h1 = h1 + f * (k2 - v);
    h2 = f * (v - tm1);
    f = mdiv(hh2, t->d[6], t->r[6]);
    h2 = h2 + f * (k3 - v);
    h3 = f * (v - k0);
    w[0] = h0; w[1] = h1; w[2] = h2; w[3] = h3;
}
// (jx, jy): the cell the wave's first live lane estimates for itself, in scalar registers; x, y: per lane, clamped to the grid
// (FITPACK's argument clamp, quirk Q4).  The estimate is (int)((x - a) / h); fpbisp's interval search ends in THE cell with
// t0 <= x < t1 on the true knots -- which are the table entry's t0, t1, so whether the estimate is that cell, for every live lane,
// is two comparisons per axis against scalars that are loaded anyway (rt::locate, the search itself, is ~25 vector instructions
// per axis).  False when a live lane is elsewhere (a wave in several cells; an estimate off by one at a knot; the grid's last
// point, which belongs to the cell on its left; NaN): the caller takes the per-lane path, which searches.
// CHECK false: (jx, jy) is what every live lane's interval search returned (n_gradient's EST false): nothing to check.
template <bool CHECK>
__device__ __forceinline__ bool lookup_uniform(const FieldDev<double>& F, int jx, int jy, unsigned long long live, double x, double y,
                                               double& n, double& gx, double& gy) {
    AxisTabS tx = (AxisTabS)(F.rdx) + jx, ty = (AxisTabS)(F.rdy) + jy;
    asm volatile("" : "+s"(tx), "+s"(ty));
    if constexpr (CHECK) {
        const bool in = tx->t0 <= x && x < tx->t1 && ty->t0 <= y && y < ty->t1;
        if ((rt_ballot(!in) & live) != 0ull) return false;
    }
    int lx = jx + 2, ly = jy + 2;
    lx = lx < 3 ? 3 : (lx > F.qx - 1 ? F.qx - 1 : lx);
    ly = ly < 3 ? 3 : (ly > F.qy - 1 ? F.qy - 1 : ly);
    typedef const double __attribute__((address_space(4)))* SD;
    typedef const Pair<double> __attribute__((address_space(4)))* SP;
    SD zp = (SD)(F.zn) + ((long)jy * F.qx + jx);
    SP gp = (SP)(F.g) + ((long)(ly - 3) * F.qx + (lx - 3));
    asm volatile("" : "+s"(zp), "+s"(gp));
    double lwx[2], lwy[2], wx[4], wy[4];
    axis_uniform(x, tx, lwx, wx);
    axis_uniform(y, ty, lwy, wy);
    double sp = 0.0;
    sp += zp[0] * lwy[0] * lwx[0];
    sp += zp[1] * lwy[0] * lwx[1];
    sp += zp[F.qx] * lwy[1] * lwx[0];
    sp += zp[F.qx + 1] * lwy[1] * lwx[1];
    n = sp;
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        SP row = gp + (long)r * F.qx;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const Pair<double> g = row[q];
            sx += g.x * wy[r] * wx[q];
            sy += g.y * wy[r] * wx[q];
        }
    }
    gx = sx;
    gy = sy;
    return true;
}

// EST: the window path's cell from the first live lane's estimate, checked against the table's knots (lookup_uniform), instead of
// from every lane's interval search and a vote on the result: op7 +3.3 %, op3 +4.4 %, op4 +3.9 %, op6 in reference order +3.8 %,
// interface op3 +5.7 %; cfg5's kernel, at its register cap, answers with -0.8 % and keeps the search (window_estimates;
// profiles/r04_ab_uniform_window.txt, last section).
constexpr bool window_estimates(int method) { return method != 10 && method != 11; }
// VOTE (with EST): the attempt is abandoned on the lanes' OWN estimates -- a ballot -- before the table entry is asked for: a wave
// that straddles cells (8 % of a vert fan's steps at most, a third of an interface fan's near the interface) goes to the per-lane path
// without a scalar load.  vert op7 +1.5 %, op3 +1.5 %, interface op5 +4 %, op7 +2 %; vert op9 -1.5 % (window_votes;
// profiles/r04_ab_variants_not_kept.txt, call 50, where it was tried as a cure for the fisheye case and was not one).
constexpr bool window_votes(int method) { return method != 9; }
// cellp (optional): where the lookup landed, as the flat-cell map's index (jy * ncx + jx on the true knots)
template <int PH = 1, bool EST = true, bool VOTE = false, typename G>
__device__ __forceinline__ void n_gradient(const FieldDev<double>& F, G& gather, bool active, double x, double y,
                                           double& n, double& gx, double& gy, int* cellp = nullptr) {
    if constexpr (RTMI_EXACT_UNIFORM && G::kUniformWindow) {
        const unsigned long long live = rt_ballot(active && F.window != 0);     // no lane is asked when the batch does not use the window
        if (live != 0ull) {
            const int lead = __builtin_ctzll(live);
            if constexpr (EST) {
                double xv = x < F.ax ? F.ax : x, yv = y < F.ay ? F.ay : y;
                xv = xv > F.bx ? F.bx : xv;
                yv = yv > F.by ? F.by : yv;
                const int jx = (int)((xv - F.ax) * F.inv_hx), jy = (int)((yv - F.ay) * F.inv_hy);
                int jx0 = __builtin_amdgcn_readlane(jx, lead), jy0 = __builtin_amdgcn_readlane(jy, lead);
                bool others = false;
                if constexpr (VOTE) others = (rt_ballot(jx != jx0 || jy != jy0) & live) != 0ull;
                jx0 = jx0 < 0 ? 0 : (jx0 > F.qx - 2 ? F.qx - 2 : jx0);
                jy0 = jy0 < 0 ? 0 : (jy0 > F.qy - 2 ? F.qy - 2 : jy0);
                if (!others && lookup_uniform<true>(F, jx0, jy0, live, xv, yv, n, gx, gy)) { if (cellp) *cellp = jy0 * F.ncx + jx0; return; }
            } else {
                double xv = x, yv = y, t0, t1;
                const int jx = locate(xv, F.qx, F.ax, F.hx, F.bx, F.inv_hx, t0, t1);      // clamps in place
                const int jy = locate(yv, F.qy, F.ay, F.hy, F.by, F.inv_hy, t0, t1);
                const int jx0 = __builtin_amdgcn_readlane(jx, lead), jy0 = __builtin_amdgcn_readlane(jy, lead);
                if ((rt_ballot(jx != jx0 || jy != jy0) & live) == 0ull && lookup_uniform<false>(F, jx0, jy0, live, xv, yv, n, gx, gy)) { if (cellp) *cellp = jy0 * F.ncx + jx0; return; }
            }
        }
    }
    Cell<double> c;
    ex::field_locate(F, x, y, c);
    if (cellp) *cellp = c.jy * F.ncx + c.jx;
    if constexpr (RTMI_EXACT_UNIFORM && G::kUniformWindow && PH > 1) {
        // the per-lane gather with the window consumed in groups of rows: fpbisp's sums run row by row anyway (same order, same
        // bits), and the kernel's register count is no longer set by 72 staging registers of a path a coherent wave rarely takes
        Cell<double> cc = c;   // an idle lane reads the grid's first window instead of its stale cell (one shared cache line)
        cc.jx = active ? c.jx : 0; cc.jy = active ? c.jy : 0; cc.lx = active ? c.lx : 3; cc.ly = active ? c.ly : 3;
        const double* zp = F.zn + (size_t)cc.jy * F.qx + cc.jx;
        const double z0 = zp[0], z1 = zp[1], z2 = zp[F.qx], z3 = zp[F.qx + 1];
        const Pair<double>* gp = reinterpret_cast<const Pair<double>*>(F.g) + ((size_t)(cc.ly - 3) * F.qx + (cc.lx - 3));
        constexpr int ROWS = 4 / PH;
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int ph = 0; ph < PH; ph++) {
            Pair<double> a[ROWS][4];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
#pragma unroll
                for (int q = 0; q < 4; q++) a[r][q] = gp[(size_t)(ph * ROWS + r) * F.qx + q];
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    sx += a[r][q].x * c.wy[ph * ROWS + r] * c.wx[q];
                    sy += a[r][q].y * c.wy[ph * ROWS + r] * c.wx[q];
                }
            }
            asm volatile("" : "+v"(sx), "+v"(sy) : : "memory");   // the next rows' loads stay behind these sums
        }
        double sp = 0.0;
        sp += z0 * c.lwy[0] * c.lwx[0];
        sp += z1 * c.lwy[0] * c.lwx[1];
        sp += z2 * c.lwy[1] * c.lwx[0];
        sp += z3 * c.lwy[1] * c.lwx[1];
        n = sp; gx = sx; gy = sy;
        return;
    }
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
    const double den = 2.0 * r.n, rden = 1.0 / den;      // both quotients by 2n: one IEEE reciprocal, two Markstein divisions
    fx = (r.x + r.ux * k.step) + mdiv((r.gx - d * r.ux) * k.step2, den, rden);
    fy = (r.y + r.uy * k.step) + mdiv((r.gy - d * r.uy) * k.step2, den, rden);
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
        const SinCos t = sincos_inline(r.th - dc);
        fx = r.x + (r.uy - t.s) / curv;                 // r.uy == sin(theta), r.ux == cos(theta): same function, same bits
        fy = r.y + (t.c - r.ux) / curv;
    } else {
        const SinCos t = sincos_inline(r.th + dc);
        fx = r.x + (t.s - r.uy) / curv;
        fy = r.y + (-t.c + r.ux) / curv;
    }
    return true;
}

// ---------------------------------------------------------------- angle determination (:370-407)
template <bool INL>
__device__ __forceinline__ double ang_rk2(const Ray<double>& r, double step, double fn, double fgx, double fgy) {
    const double k1 = step * (r.ux * r.gy - r.uy * r.gx) / r.n;
    const SinCos t = sincos_sel<INL>(r.th + k1);
    const double k2 = step * (t.c * fgy - t.s * fgx) / fn;
    return r.th + (k1 + k2) / 2.0;
}
__device__ __forceinline__ double ang_cost(const Ray<double>& r, double step, double fgx, double fgy) {
    return atan2_(r.n * r.uy + impulse(r.gy, fgy, step), r.n * r.ux + impulse(r.gx, fgx, step));
}

// ---------------------------------------------------------------- golden() (:175-199), filtered  (anisotropic cost)
// The reference evaluates func(c) and func(d) afresh in each of its 37 iterations and keeps [a, d] when
// func(c) < func(d), else [c, b]; the returned midpoint depends only on that sequence of outcomes.  Here every
// comparison is first attempted WITHOUT the reference's arithmetic, from values that carry a bound on their distance from
// what the reference's arithmetic gives at the current points; when the bound allows, the outcome is decided.  Otherwise
// (per lane, rare) both costs are recomputed at the current points in the reference's arithmetic (`exact`) and compared like
// the reference does.  Either way the outcome is the reference's, so a, b, c, d -- advanced with the reference's own
// unfused bracket arithmetic -- are its bits, and so is the returned midpoint.
//
//   Phase A (the first kGoldTaylorFrom iterations, bracket width pi .. ~1e-2): one new FAST cost value per iteration.  Of
//     the two points of the new bracket one is the previous iteration's better point ("survivor"; equal up to a few ulps of t
//     to the point the reference would evaluate again), the other is the survivor moved by h_j = pi*GR^(j+4) towards the
//     kept side; its sin/cos come from rotating the survivor's through h_j (table rt_golden_rot.h), no sincos evaluation.
//     FastA(s, c, f, g): cost f and g = |e_x| + |e_y| of the residual vector from sin/cos of the evaluation angle.
//     Bound of one stored value: g*K1 + f*K2 (+ K3 once), K1 = 2 x (absolute error bound of one residual component:
//     reference arithmetic + fast arithmetic + LIP x the tracked angle's drift), K2 = relative error of squaring and adding.
//   Phase T (the remaining ~23 iterations): no cost value at all.  The cost F = |e|^2 is expanded to third order about a
//     centre t0 next to its minimiser (two evaluations of the residual e and of its first three derivatives in closed form:
//     one at theta, a Newton step, one at the centre); with xi = t - t0
//         F(c) - F(d) = (xi_c - xi_d) V + R4(c) - R4(d),   V = F1 + F2 (xi_c + xi_d)/2 + F3 (xi_c^2 + xi_c xi_d + xi_d^2)/6,
//     and since xi_c < xi_d the reference's outcome F(c) < F(d) is V > 0 -- for certain when |V| (d - c) exceeds the remainder
//     bound M4 rho^4/12 (rho = max |xi|, M4 >= |d4F/dt4| from suprema of the momentum curve's derivatives, Consts::gold_sup)
//     plus the rounding error the reference's arithmetic can have made in its two values (the same g*K1 + f*K2 + K3 with
//     g <= g0 + 2 E1 rho) plus the error of V's coefficients.  24 vector instructions per iteration (the threshold is kept between iterations and
//     refreshed only where it does not decide, see below) against 58 in phase A.  The
//     minimiser stays inside every bracket, so rho shrinks with the bracket and the comparison is decided by the sign of V in
//     all but ~4e-3 of the steps (the reference's own rounding noise in its last three iterations).
struct GoldBounds { double K1, LIP, K2, K3; };
// third-order expansion of the cost about one angle: |e_x| + |e_y| there, and the cost's first three derivatives
struct GoldExpansion { double g0, F1, F2, F3; };

#ifndef RTMI_GOLD_TAYLOR_FROM
#define RTMI_GOLD_TAYLOR_FROM 11   // iterations of phase A; bracket width pi*GR^11 = 1.6e-2 when phase T takes over (A/B: 14: 170 ms for cfg5, 12: 159.5, 11: 156.4, 10: 160, 9: 164)
#endif
constexpr int kGoldTaylorFrom = RTMI_GOLD_TAYLOR_FROM;
static_assert(kGoldTaylorFrom <= RT_GOLD_ROT_ENTRIES, "phase A rotates through rt_golden_rot.h");
__device__ const double kGoldRot[2 * RT_GOLD_ROT_ENTRIES] = {RT_GOLD_ROT_VALUES};
constexpr double kU = 1.1102230246251565e-16;   // 2^-53

// fast_a: the phase-A evaluation (may be looser than the bound's K1 by at most EA in one residual component).
// expand(s, c): GoldExpansion at the angle whose (sin, cos) are (s, c).  E[1..4]: bounds on |d^k e / dt^k| (one component);
// e_abs: bound on the absolute error of one residual component as `expand` computes it.
template <typename FastA, typename Expand, typename Exact>
__device__ __forceinline__ double golden_filtered(FastA fast_a, double EA, Expand expand, Exact exact, const GoldBounds& B,
                                                  const double E[5], double e_abs, double th, double s0, double c0) {
    const double GR = kGoldRatio, tol = M<double>::gold_tol;
    double a = th - kHalfPi, b = th + kHalfPi;
    double c = b - (b - a) * GR, d = a + (b - a) * GR;
    int it = 0;
    {   // ---- phase A.  X: survivor, Y: the newer point; x_is_c: X is the lower point c
        // tracked angle vs actual point: rotation arithmetic (<= 6u each, table constants included) and the bracket
        // arithmetic's roundings (<= ulp(t) per update, t up to |theta| + pi/2)
        const double dphi = kU * (8.0 + kGoldTaylorFrom * (6.0 + 2.0 * (__builtin_fabs(th) + 2.0)));
        const double lA = B.LIP * dphi;
        const double K1A = B.K1 + lA + 2.0 * EA, K3A = 4.0 * K1A * K1A;
        const double sk = RT_GOLD_KAPPA_SIN, ck = RT_GOLD_KAPPA_COS;
        double sX = fma_(-c0, sk, s0 * ck), cX = fma_(s0, sk, c0 * ck);   // theta0 - kappa
        double sY = fma_(c0, sk, s0 * ck), cY = fma_(-s0, sk, c0 * ck);   // theta0 + kappa
        double fX, gX, fY, gY;
        fast_a(sX, cX, fX, gX);
        fast_a(sY, cY, fY, gY);
        bool x_is_c = true;
        // the rotation (sin h_j, cos h_j) of iteration j is asked for one iteration ahead: read where it is used it is a
        // scalar load with its wait right behind it, eleven scalar-cache round trips per step
        double sh_next = kGoldRot[0], ch_next = kGoldRot[1];
        static_assert(kGoldTaylorFrom < RT_GOLD_ROT_ENTRIES, "the look-ahead reads entry kGoldTaylorFrom");
        for (; it < kGoldTaylorFrom && __builtin_fabs(c - d) > tol; ++it) {
            const double sh_now = sh_next, ch = ch_next;
            sh_next = kGoldRot[2 * it + 2]; ch_next = kGoldRot[2 * it + 3];
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
            // -sin h_j when the lower part of the bracket is kept: the sign bit flipped by the outcome's lane mask
            const double sh = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, sh_now) ^ (lt ? 0x8000000000000000ull : 0ull));
            sY = fma_(cX, sh, sX * ch);
            cY = fma_(-sX, sh, cX * ch);
            fast_a(sY, cY, fY, gY);
            x_is_c = !lt;
        }
    }
    if (!(__builtin_fabs(c - d) > tol)) return (b + a) / 2.0;
    // ---- phase T.  Centre: theta moved by one Newton step (with the cubic term) of the expansion at theta.
    // (the centre is a choice, not a result: whatever it is, the thresholds below hold for it and the outcomes are the
    // reference's -- so the Newton step needs neither IEEE divisions nor the third derivative: without the cubic term it lands
    // within x0^2 |F3 / 2 F2| of the minimiser, 2e-7 for the 4e-4 a step turns by, where the remainder bound wants < 4e-5)
    const GoldExpansion X0 = expand(s0, c0, false);
    const double xs = -X0.F1 * rcp_full(X0.F2);
    double s1, c1;
    sincos_add_small(s0, c0, xs, &s1, &c1);                     // |xs| < 2^-5 is checked below
    const GoldExpansion T = expand(s1, c1, true);
    const double t0 = th + xs;
    // the centre's angle is known to: the rounding of th + xs and of t - t0, the series of the rotation (angle error < 4u),
    // and theta against its carried (sin, cos) -- a uniform shift eps0 of every xi, i.e. an error F2*eps0 of V
    const double eps0 = kU * (16.0 + 4.0 * __builtin_fabs(th));
    const double F2h = 0.5 * T.F2, F36 = T.F3 * (1.0 / 6.0);
    // |d4F/dt4| <= 4 (E0 E4 + 4 E1 E3 + 3 E2^2), E0 = sup |e| <= g0 + E1 rho, for rho <= kRhoMax
#ifndef RTMI_GOLD_RHO_MAX
#define RTMI_GOLD_RHO_MAX 0.03125
#endif
    constexpr double kRhoMax = RTMI_GOLD_RHO_MAX;
    const double M4c = (1.001 / 12.0) * 4.0 * fma_(T.g0 + E[1] * kRhoMax, E[4], fma_(4.0 * E[1], E[3], 3.0 * E[2] * E[2]));
    // V's coefficients: F1 = 2 e.e1 inherits e's absolute error e_abs (e is a cancelled difference, its derivative is not);
    // F2, F3 and the evaluation of V a relative 64u of their terms (|sigma| <= 2 rho, |pi2| <= 3 rho^2)
    const double errV0 = fma_(8.0 * e_abs, E[1], fma_(__builtin_fabs(T.F2), eps0, 32.0 * kU * __builtin_fabs(T.F1)));
    const double cF2 = 64.0 * kU * __builtin_fabs(T.F2), cF3 = 32.0 * kU * __builtin_fabs(T.F3);
    const double E1x2 = 2.0 * E[1], K3x2 = 2.0 * B.K3;
    // outside what the bounds were derived for (never on the path): every comparison goes to the reference's arithmetic
    const bool ok = X0.F2 > 0.0 && T.F2 > 0.0 && __builtin_fabs(xs) < 0.03125 && __builtin_fabs(T.F3) < 1e300 && M4c < 1e300;
    double invd = 1.002 * rcp_full(d - c);                      // 1/(d - c): slack for the recurrence against the rounded widths (and the reciprocal's ulp)
    // The threshold is thrA / (d - c) + thrB with thrA = M4c rho^4 + noise(rho), thrB = rho (rho cF3 + cF2) + errV0, both
    // increasing in rho, and rho never grows (every c, d lies inside the bracket before): a threshold formed from an EARLIER
    // iteration's thrA, thrB is still a valid one.  So they are refreshed only for a lane whose comparison the kept pair does
    // not decide -- the first two or three iterations, where the remainder term still matters, and the last three, where the
    // reference's own noise does -- and the ~17 iterations between cost one fma for their threshold instead of ten
    // instructions.  Starting values: rho of the whole bracket.
    double thrA, thrB;
    bool rho_ok;
    auto refresh = [&](double rho) {
        const double Gb = fma_(E1x2, rho, T.g0);
        const double noise = fma_(2.0 * Gb, fma_(B.K2, Gb, B.K1), K3x2);
        const double r2 = rho * rho;
        thrA = fma_(M4c * r2, r2, noise);
        thrB = fma_(rho, fma_(rho, cF3, cF2), errV0);
        rho_ok = ok && rho < kRhoMax;
    };
    refresh(__builtin_fmax(__builtin_fabs(a - t0), __builtin_fabs(b - t0)));
    for (; it < kGoldMaxIter && __builtin_fabs(c - d) > tol; ++it) {
        const double xc = c - t0, xd = d - t0;
        const double sg = xc + xd, p2 = fma_(sg, sg, -(xc * xd));
        const double V = fma_(F36, p2, fma_(F2h, sg, T.F1));
        bool lt = V > 0.0;
        if (!(rho_ok && __builtin_fabs(V) > fma_(thrA, invd, thrB))) {      // also when anything is NaN
            refresh(__builtin_fmax(__builtin_fabs(xc), __builtin_fabs(xd)));
            if (!(rho_ok && __builtin_fabs(V) > fma_(thrA, invd, thrB))) lt = exact(c) < exact(d);
        }
        if (lt) b = d; else a = c;
        c = b - (b - a) * GR;
        d = a + (b - a) * GR;
        invd *= 1.618033988749895;
    }
    return (b + a) / 2.0;
}

// One comparison of the isotropic search in the reference's arithmetic: cost(c) < cost(d) (:191 with :595 / :697), both
// costs inline and side by side (two independent chains: the second hides the first's latency).  Where the index is constant
// (both flanks of the interface scenario's sigmoid) the impulse vanishes, psi = theta, and the search's points fall
// symmetrically about its minimum at every third iteration: 13 of a step's 37 comparisons are ties in exact arithmetic,
// decided in the reference by the rounding of glibc's sin / cos at c and d -- and every one of them reaches the returned
// bits (the brackets of the two outcomes differ in their last places).  So this runs 13 times per step there, not 1e-4
// times: with the out-of-line cost (a call into a call, 14 registers spilled around it) such steps cost 3 666 vector
// instructions and 2 268 scalar ones per wave against 1 629 / 700 elsewhere (profiles/r04_iface_op9_base_pmc_summary.txt).
__device__ __forceinline__ bool exact_lt_iso(double c, double d, double fn, double px, double py, double ix, double iy) {
    const SinCos uc = sincos_inline(c), ud = sincos_inline(d);
    const double Fc = sq(fn * uc.c - px - ix) + sq(fn * uc.s - py - iy);
    const double Fd = sq(fn * ud.c - px - ix) + sq(fn * ud.s - py - iy);
    return Fc < Fd;
}
// The anisotropic cost in the reference's arithmetic -- the rare path of golden_filtered, kept out of line.
__device__ __attribute__((noinline)) double exact_cost_aniso(double t, double fn, double gam, double g2, double mix, double miy,
                                                            double cgx, double cgy, double fgx, double fgy, double step) {
    const SinCos u = sincos_(t);
    const double s = u.s, c = u.c;                                                                    // (:728, :761)
    const double a = aniso(s, c, gam);
    double mx, my;
    moments(fn, a, g2, c, s, mx, my);
    return sq(mx - mix - impulse(cgx, a * fgx, step)) + sq(my - miy - impulse(cgy, a * fgy, step));
}

// isotropic cost (:595, :697): (n' cos t - n u_x - I_x)^2 + (n' sin t - n u_y - I_y)^2
//
// golden() for this cost WITHOUT evaluating it.  With P = (n u_x + I_x, n u_y + I_y) the cost is
//     F(t) = |n'(cos t, sin t) - P|^2 = (n' - |P|)^2 + 4 n'|P| sin^2((t - psi)/2),      psi = arg P,
// symmetric about psi and increasing in |t - psi| < pi.  The reference keeps [a, d] when F(c) < F(d), c < d, i.e. when c is
// the closer of the two to psi, i.e. when psi < (c + d)/2: each of its ~37 comparisons is the sign of
//     x = (c + d)/2 - psi        --  two subtractions against one angle, psi - theta = atan2(P x u, P . u), per step.
// What the reference actually compares are the ROUNDED values of F; the sign of x is its outcome for certain only when the
// true difference |F(c) - F(d)| = 4 n'|P| |sin x| sin h (h = (d - c)/2) exceeds the rounding error its arithmetic can
// have made in the two values, (g_c + g_d) K1 + (F_c + F_d) K2 + 2 K3 (GoldBounds, as in golden_filtered).  With
// g <= sqrt(2 F), sqrt F <= |Delta| + sqrt(n'|P|) |t - psi|, Delta = n' - |P|, |t - psi| <= |x| + h <= 2.4 this is implied by
//     |x| > Ca + Cb / h,    Ca = (2 sqrt2 K1 / sqrt(n'|P|) + 4.8 K2) / 1.776 + (the error of x itself, <= 16 u),
//                           Cb = (2 sqrt2 K1 |Delta| + 2 K2 Delta^2 + 2 K3) / (1.776 n'|P|)
// (1.776 = 4 x min sin x / x on [0, 2] x min sin h / h on [0, 0.371]).  Delta is O(step^2): Cb / h stays below 1e-12 even in
// the last iteration (h = 7e-9) and the comparison is decided by the sign of x in all but ~1e-4 of the steps; otherwise --
// per lane, for that one comparison -- both costs are evaluated in the reference's arithmetic (exact_lt_iso) and compared
// like the reference does.  Either way the outcome is the reference's, and a, b, c, d, advanced with its own unfused bracket
// arithmetic, are its bits.  (Round 2 evaluated a fast cost with an error bound at every new point: ~90 instructions per
// iteration, ~20 now.)
__device__ __forceinline__ double ang_golden_iso(const Ray<double>& r, double step, double fn, double fgx, double fgy) {
    const double px = r.n * r.ux, py = r.n * r.uy;
    const double ix = impulse(r.gx, fgx, step), iy = impulse(r.gy, fgy, step);
    const double Px = px + ix, Py = py + iy;
    // P turned by -theta with the carried (cos theta, sin theta): its angle is phi = psi - theta, |phi| << 1
    const double Pc = fma_(Py, r.uy, Px * r.ux), Ps = fma_(Py, r.ux, -(Px * r.uy));
    // phi is small (the turn of one step) and enters the comparisons with an error allowance of its own (Ca): atan of the
    // ratio by a five-term series for |t| < 2^-5 (error < 2^-53 |t|) instead of ocml's atan2; larger turns take atan2
    double phi;
    {
        const double t = Ps * rcp_full(Pc);
        const bool near = Pc > 0.0 && __builtin_fabs(t) < 0.03125;
        auto series = [&]() {
            const double z = t * t;
            double p = fma_(z, 1.0 / 9.0, -1.0 / 7.0);
            p = fma_(z, p, 1.0 / 5.0);
            p = fma_(z, p, -1.0 / 3.0);
            return fma_(t * z, p, t);
        };
        if (rt_ballot(!near) == 0ull) phi = series();
        else phi = near ? series() : ::atan2(Ps, Pc);
    }
    const double P2 = fma_(Ps, Ps, Pc * Pc);
    double rP = __builtin_amdgcn_rsq(P2);
    rP = fma_(rP * 0.5, fma_(-P2 * rP, rP, 1.0), rP);
    rP = fma_(rP * 0.5, fma_(-P2 * rP, rP, 1.0), rP);          // 1/|P| to ~2 ulp
    const double aP = P2 * rP, afn = __builtin_fabs(fn);
    // rounding error of one residual component in the reference's arithmetic (and the margin round 2's fast form needed): 12 u mag
    const double mag = afn + __builtin_fmax(__builtin_fabs(px), __builtin_fabs(py)) + __builtin_fmax(__builtin_fabs(ix), __builtin_fabs(iy));
    const double e1 = 12.0 * kU * mag;
    const double K1 = 2.0 * e1, K2 = 8.0 * kU, K3 = 16.0 * e1 * e1;
    const double A = afn * aP;                                   // n'|P|
    double rsA = __builtin_amdgcn_rsq(A);
    rsA = fma_(rsA * 0.5, fma_(-A * rsA, rsA, 1.0), rsA) * 1.000001;   // >= 1/sqrt(A)
    const double Delta = __builtin_fabs(afn - aP) + 16.0 * kU * (afn + aP);
    const double beta = fma_(2.8284271247461903 * K1, Delta, fma_(2.0 * K2 * Delta, Delta, 2.0 * K3));
    // slack 1.001: the tabulated-by-recurrence 1/h against the bracket's actual (rounded) width, the Newton reciprocals
    double Ca = (1.001 / 1.776) * fma_(2.8284271247461903 * K1, rsA, 4.8 * K2) + 16.0 * kU;
    const double Cb = (1.001 / 1.776) * beta * rsA * rsA;
    // outside the geometry the bound was derived for (never on the path: psi within 1 rad of theta, |P| ~ n' > 0): every
    // comparison goes to the reference's arithmetic
    if (!(fn > 0.0 && aP > 0.25 * afn && __builtin_fabs(phi) < 1.0 && Cb < 1.0)) Ca = __builtin_inf();
    const double GR = kGoldRatio, tol = M<double>::gold_tol, th = r.th, phi2 = 2.0 * phi;
    double a = th - kHalfPi, b = th + kHalfPi;
    double c = b - (b - a) * GR, d = a + (b - a) * GR;
    double invh = 2.6967646315695153;                            // 1 / h_0, h_0 = (2 GR - 1) pi / 2; h_j = h_0 GR^j
    for (int it = 0; it < kGoldMaxIter && __builtin_fabs(c - d) > tol; ++it) {
        const double q2 = ((c - th) + (d - th)) - phi2;          // 2 x; both differences are exact or rounded at their own size
        bool lt = q2 > 0.0;
        if (!(__builtin_fabs(q2) > 2.0 * fma_(Cb, invh, Ca)))    // also taken when anything is NaN
            lt = exact_lt_iso(c, d, fn, px, py, ix, iy);
        if (lt) b = d; else a = c;
        c = b - (b - a) * GR;
        d = a + (b - a) * GR;
        invh *= 1.618033988749895;
    }
    return (b + a) / 2.0;
}

// anisotropic cost (:725-728 / :758-761); the step functions read the module-global gamma (quirk Q12)
__device__ __forceinline__ double ang_golden_aniso(const Ray<double>& r, const Consts<double>& k, double fn, double fgx,
                                                   double fgy) {
    const double gam = k.gamma_s, g2 = k.g2m1_s, step = k.step;
    const double c0 = aniso(r.uy, r.ux, gam);
    double mix, miy;
    moments(r.n, c0, g2, r.ux, r.uy, mix, miy);
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
    // Phase T: e(t) = M(t) - m_i - (coef g + a(t) g1) step/2 (g1: the gradient at the new point) and its derivatives in
    // closed form.  With A = 1/a, G = gamma^2 - 1, and D for d/dt:
    //   M    = n1 (c A, gamma^2 s A)                        D M = n1 gamma^2 A^3 (-s, c)
    //   D2 M = -n1 gamma^2 A^5 (c (1 - 2G s^2), s (1 + 3G - 2G s^2))
    //   D3 M = n1 gamma^2 A^7 ( s [(1 - 2G s^2 + 4G c^2) a^2 + 5G c^2 (1 - 2G s^2)], -c [(1 + 3G - 6G s^2) a^2 - 5G s^2 (1 + 3G - 2G s^2)] )
    //   D a = G s c A     D2 a = G A [(c^2 - s^2) - G s^2 c^2 A^2]     D3 a = G s c A [-4 - 3G (c^2 - s^2) A^2 + 3 G^2 s^2 c^2 A^4]
    // (tools/check_aniso_derivatives.py compares them with mpmath's numerical derivatives.)
    auto expand = [=](double s, double c, const bool third) {
        const double ss = s * s, cc = c * c, sc = s * c;
        const double a2 = fma_(gam2, ss, cc);
        double A = __builtin_amdgcn_rsq(a2);
        A = fma_(A * 0.5, fma_(-a2 * A, A, 1.0), A);
        A = fma_(A * 0.5, fma_(-a2 * A, A, 1.0), A);
        const double a = a2 * A, A2 = A * A, A3 = A2 * A, A5 = A3 * A2, A7 = A5 * A2;
        const double Gss = g2 * ss, Gcc = g2 * cc, kk = fn * gam2, dif = cc - ss;
        const double u1 = fma_(-2.0, Gss, 1.0), u2 = fma_(3.0, g2, u1);            // 1 - 2G s^2, 1 + 3G - 2G s^2
        const double hx = fgx * hstep, hy = fgy * hstep;                            // g1 step/2
        const double a1 = g2 * sc * A, a2d = g2 * A * fma_(-Gss * cc, A2, dif);

        const double e0x = fn * c * A - mix - fma_(a, fgx, cgx) * hstep, e0y = kk * s * A - miy - fma_(a, fgy, cgy) * hstep;
        const double k3 = kk * A3, k5 = kk * A5;
        const double e1x = fma_(-a1, hx, -(k3 * s)), e1y = fma_(-a1, hy, k3 * c);
        const double e2x = fma_(-a2d, hx, -(k5 * c * u1)), e2y = fma_(-a2d, hy, -(k5 * s * u2));
        GoldExpansion X;
        X.g0 = __builtin_fabs(e0x) + __builtin_fabs(e0y);
        X.F1 = 2.0 * fma_(e0y, e1y, e0x * e1x);
        X.F2 = 2.0 * (fma_(e1y, e1y, e1x * e1x) + fma_(e0y, e2y, e0x * e2x));
        X.F3 = 0.0;
        if (third) {       // a compile-time constant at both call sites
            const double k7 = kk * A7;
            const double a3 = a1 * fma_(3.0 * Gss * Gcc * A2, A2, fma_(-3.0 * g2 * dif, A2, -4.0));
            const double m3x = k7 * s * fma_(fma_(4.0, Gcc, u1), a2, 5.0 * Gcc * u1);
            const double m3y = -(k7 * c) * fma_(fma_(-4.0, Gss, u2), a2, -5.0 * Gss * u2);
            const double e3x = fma_(-a3, hx, m3x), e3y = fma_(-a3, hy, m3y);
            X.F3 = 2.0 * fma_(3.0, fma_(e1y, e2y, e1x * e2x), fma_(e0y, e3y, e0x * e3x));
        }
        return X;
    };
    // |D^k e| <= n1 D_k + (step/2) max|g1| A_k with D_k, A_k the suprema over all angles of the k-th derivatives of the unit
    // momentum curve's components and of a(t) (host: gold_sup_derivatives, 10 % on top of an 8192-point sampling)
    const double gmx = __builtin_fmax(__builtin_fabs(fgx), __builtin_fabs(fgy)) * hstep;
    const double E[5] = {0.0, fma_(afn, k.gold_sup[0], gmx * k.gold_sup[4]), fma_(afn, k.gold_sup[1], gmx * k.gold_sup[5]),
                         fma_(afn, k.gold_sup[2], gmx * k.gold_sup[6]), fma_(afn, k.gold_sup[3], gmx * k.gold_sup[7])};
    return golden_filtered(fast_a, EA, expand, exact, B, E, e1, r.th, r.uy, r.ux);   // expand refines 1/a twice: e1 covers it
}

// ---------------------------------------------------------------- opN around the field lookup
// METHOD here is the step method itself, 1..11: op1/2/6/7/8 take this path only when the batch asks for the reference's
// operation order throughout (rtmi_params.reference_order); op3/4/5/9/10/11 always do.
template <int METHOD>
__device__ __forceinline__ bool op_advance(const Consts<double>& k, const Ray<double>& r, double& fx, double& fy) {
    if constexpr (METHOD == 1 || METHOD == 2) { ex::adv_first(r, k.step, fx, fy); return true; }
    else if constexpr (METHOD == 3 || METHOD == 4 || METHOD == 5 || METHOD == 10) return ex::adv_curv(r, k, fx, fy);
    else { ex::adv_second(r, k, fx, fy); return true; }
}
// i: the row being produced (op7's bootstrap rows 1 and 2, :833-864)
template <int METHOD>
__device__ __forceinline__ double op_angle(const Consts<double>& k, const Ray<double>& r, bool flag, double fx, double fy,
                                           double fn, double fgx, double fgy, int i) {
    if constexpr (METHOD == 1 || METHOD == 8) return ex::ang_cost(r, k.step, fgx, fgy);
    else if constexpr (METHOD == 2 || METHOD == 6) return ex::ang_rk2<inline_sincos(METHOD)>(r, k.step, fn, fgx, fgy);
    else if constexpr (METHOD == 7) {
        // finite_diff (:370-372) over [P0, P1, P2, P3] = [h0, h1, (x, y), f], left to right like the reference; rows 1 and 2
        // use the first- and second-order backward differences (:843, :856)
        double vx, vy;
        if (i == 1) { vx = fx - r.x; vy = fy - r.y; }
        else if (i == 2) { vx = 3.0 * fx - 4.0 * r.x + r.hx1; vy = 3.0 * fy - 4.0 * r.y + r.hy1; }
        else { vx = 11.0 * fx - 18.0 * r.x + 9.0 * r.hx1 - 2.0 * r.hx0; vy = 11.0 * fy - 18.0 * r.y + 9.0 * r.hy1 - 2.0 * r.hy0; }
        return atan2_(vy, vx);
    }
    else if constexpr (METHOD == 3) return flag ? ex::ang_rk2<inline_sincos(METHOD)>(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 4) return flag ? ex::ang_cost(r, k.step, fgx, fgy) : r.th;
    else if constexpr (METHOD == 5) return flag ? ex::ang_golden_iso(r, k.step, fn, fgx, fgy) : r.th;
    else if constexpr (METHOD == 9) return ex::ang_golden_iso(r, k.step, fn, fgx, fgy);
    else if constexpr (METHOD == 10) return flag ? ex::ang_golden_aniso(r, k, fn, fgx, fgy) : r.th;
    else return ex::ang_golden_aniso(r, k, fn, fgx, fgy);   // 11
}

// store_update_results (:783-790) + the row bookkeeping of the loop body (:871-875)
template <bool INL>
__device__ __forceinline__ void store_update(const Consts<double>& k, Ray<double>& r, double fx, double fy, double fth,
                                             double fn, double fgx, double fgy) {
    const double dist = norm2(r.x - fx, r.y - fy);
    r.dsim += dist;
    r.dreal += k.step;
    const SinCos u = sincos_sel<INL>(fth);
    const double c = u.c, s = u.s;
    const double coef = aniso(s, c, k.gamma);
    moments(fn, coef, k.g2m1, c, s, r.mx, r.my);
    r.hx0 = r.hx1; r.hy0 = r.hy1; r.hx1 = r.x; r.hy1 = r.y;
    r.x = fx; r.y = fy; r.th = fth; r.n = fn; r.gx = fgx; r.gy = fgy;
    r.ux = c; r.uy = s; r.coef = coef;
    const double nray = coef * fn;                              // (:873)
    r.tt = r.tt + dist * (r.nray + nray) / 2.0;                 // (:874) quirk Q6
    r.nray = nray;
}

// derived quantities from the stored state: the same functions of the same bits as when they were first formed
__device__ __forceinline__ void derive(const Consts<double>& k, Ray<double>& r) {
    const SinCos u = sincos_(r.th);
    r.ux = u.c; r.uy = u.s;
    r.coef = aniso(r.uy, r.ux, k.gamma);
    r.nray = r.coef * r.n;
    r.rn = 0;
    moments(r.n, r.coef, k.g2m1, r.ux, r.uy, r.mx, r.my);
}

// ---------------------------------------------------------------- the reference-order step where the medium is constant
// op1/2/6/8 in the reference's operation order (rtmi_params.reference_order = 1; the automatic re-trace of critical rays, rtmi.hip)
// on a field with FLAT cells (rt::poly_cell_flat: every gradient-spline coefficient of the cell at most 2^-80 of the grid's largest
// -- both flanks of the interface scenario's sigmoid, three quarters of a ray's steps there).  In such a cell FITPACK's gradient
// evaluates to at most gflat = 2^-72 of that scale (32 coefficients of at most 2^-80, weights in [0, 1], and the rounding of its sums),
// and the reference's step then does nothing with it that survives a rounding:
//   * the advancement (:330) adds (grad n - (grad n . u) u) DELTA_S^2 / 2n, at most gflat DELTA_S^2 / n, to r + u DELTA_S: below a
//     quarter ulp of each coordinate the sum is r + u DELTA_S itself;
//   * the new angle theta + (k1 + k2) / 2 (:374-391) with |k| <= 2 gflat DELTA_S / n is theta itself below a quarter ulp of theta --
//     and then its sin / cos, the anisotropy factor and the unit tangent are the ones the ray already carries (same function, same
//     argument, same bits).
// What is left of the step is the index at the new point -- FITPACK's BILINEAR sum (fpbspl k = 1 on the true knots, fpbisp's
// order: the four samples of a flat cell are equal, but their weights sum to 1 +- ulp and the reference's n, momenta and traveltime
// carry that) -- the chord, the moments and the traveltime, in store_update's order.  No cubic basis (12 Markstein divisions), no
// 4 x 4 window (18 loads, 96 flops), no sin / cos, no division by n: ~150 instructions off a 2-instruction dependent chain instead
// of ~1 500 on a chain of ~400 -- what a lone wave of re-traced rays is bound by.  The three conditions are tested per lane and per
// step with the ray's own numbers; a lane that fails one (a coordinate within 1e-9 of zero, a cell that is not flat) takes the
// full step, which is the same bits by construction -- tests/test_gpu_exact.py holds both to the oracle's.
// The gradient at the new point is not evaluated on this path (gstale): the full step evaluates it at the point it starts from when
// it finds it missing, and so does whoever stores the ray's state (finish_state).  op1 / op8 form their new angle as
// arctan2(n u_y + I_y, n u_x + I_x) (:407), which is not theta in the last bits even where the impulse vanishes: they keep numpy's
// arctan2 and the sin / cos of its result and skip the rest.
// Two flags of the flat path travel in Ray::rn, which this order of stepping has no other use for (ex::derive clears it; the fused
// forms keep 1/n there): 1 = the gradient has not been evaluated at (x, y) yet, 2 = (x, y) lies in a flat cell.  (Not members of
// their own: the kernels at their register cap answer to every change of the structs they share, see rt_device.h, Consts.)
__device__ __forceinline__ bool grad_stale(const Ray<double>& r) { return r.rn == 1.0 || r.rn == 3.0; }
__device__ __forceinline__ bool in_flat_cell(const Ray<double>& r) { return r.rn >= 2.0; }
__device__ __forceinline__ void set_flat_flags(Ray<double>& r, bool stale, bool flat) { r.rn = (flat ? 2.0 : 0.0) + (stale ? 1.0 : 0.0); }
template <int METHOD> constexpr bool flat_shortcut() { return RTMI_FLAT_MAP && (METHOD == 1 || METHOD == 2 || METHOD == 6 || METHOD == 8); }
template <typename G> struct IsGlobalGather { static constexpr bool value = false; };
template <> struct IsGlobalGather<GlobalGather<double, true>> { static constexpr bool value = true; };

// fpbspl for k = 1 on the cell's true knots (the linear part of axis_exact): j, and the two weights
__device__ __forceinline__ int axis_linear(double v, int q, double a, double h, double b, double ih, const double* rd, double wl[2]) {
    double t0, t1;
    const int j = locate(v, q, a, h, b, ih, t0, t1);      // v is clamped in place (quirk Q4)
    const double f = rd[(size_t)j * kAxisTab];             // 1.0 / (t1 - t0), correctly rounded (fp_axis_tab_build)
    wl[0] = 0.0 + f * (t1 - v);
    wl[1] = f * (v - t0);
    return j;
}
// is the point's cell flat (the flat-cell map), and FITPACK's bilinear n there
__device__ __forceinline__ bool flat_point(const FieldDev<double>& F, double x, double y, double& n) {
    double lwx[2], lwy[2];
    const int jx = axis_linear(x, F.qx, F.ax, F.hx, F.bx, F.inv_hx, F.rdx, lwx);
    const int jy = axis_linear(y, F.qy, F.ay, F.hy, F.by, F.inv_hy, F.rdy, lwy);
    double cf; float lam;
    const bool flat = flat_lane(F, jy * F.ncx + jx, cf, lam);
    const double* zp = F.zn + (size_t)jy * F.qx + jx;
    double sp = 0.0;                                        // field_combine's first four lines
    sp += zp[0] * lwy[0] * lwx[0];
    sp += zp[1] * lwy[0] * lwx[1];
    sp += zp[F.qx] * lwy[1] * lwx[0];
    sp += zp[F.qx + 1] * lwy[1] * lwx[1];
    n = sp;
    return flat;
}
// the gradient at the point the ray is at, when the flat path left it out (the index there is r.n already: same function, same bits)
template <int METHOD, typename G>
__device__ __forceinline__ void finish_state(const FieldDev<double>& F, G& gather, Ray<double>& r) {
    if constexpr (flat_shortcut<METHOD>() && IsGlobalGather<G>::value) {
        if (grad_stale(r)) {
            double nn;
            ex::n_gradient<fallback_phases(METHOD), window_estimates(METHOD), window_votes(METHOD)>(F, gather, true, (double)r.x, (double)r.y, nn, r.gx, r.gy);
            set_flat_flags(r, false, in_flat_cell(r));
        }
    }
}

template <int METHOD, typename G>
__device__ __forceinline__ bool ray_step(const FieldDev<double>& F, const Consts<double>& k, G& gather, bool active,
                                         Ray<double>& r, int i) {
    if constexpr (flat_shortcut<METHOD>() && IsGlobalGather<G>::value) {
        // tried only when some live lane of the wave stands in a flat cell (wave-uniform test on state the lanes carry: inside a
        // transition band nothing is looked up twice)
        if (F.flat != 0 && rt_ballot(active && in_flat_cell(r)) != 0ull) {
            // the step if nothing of the gradient survives: r + u DELTA_S (adv_first; adv_second's first bracket), theta kept
            const double fx = r.x + r.ux * k.step, fy = r.y + r.uy * k.step;
            double fn;
            const bool newflat = flat_point(F, fx, fy, fn);
            const double q = gather.gflat * k.step2, sc = 0x1p-56;           // a quarter ulp of v is at least 2^-55 |v|; half of that in hand
            bool ok = in_flat_cell(r) && newflat && (METHOD == 1 || METHOD == 2 || (q < sc * __builtin_fabs(fx) * r.n && q < sc * __builtin_fabs(fy) * r.n));
            if constexpr (METHOD == 2 || METHOD == 6)
                ok = ok && gather.gflat * k.step * (r.n + fn) < sc * __builtin_fabs(r.th) * r.n * fn;
            else   // op1 / op8: the impulse DELTA_S (g + g') / 2 (:214) against n u_x and n u_y (:407)
                ok = ok && gather.gflat * k.step < sc * r.n * __builtin_fmin(__builtin_fabs(r.ux), __builtin_fabs(r.uy));
            if (!active || ok) {       // (an idle lane: its stale state takes the short way too)
                const double dist = norm2(r.x - fx, r.y - fy);
                r.dsim += dist;
                r.dreal += k.step;
                double c = r.ux, s = r.uy, coef = r.coef, fth = r.th;
                if constexpr (METHOD == 1 || METHOD == 8) {
                    fth = atan2_(r.n * r.uy, r.n * r.ux);
                    const SinCos u = sincos_(fth);
                    c = u.c; s = u.s;
                    coef = aniso(s, c, k.gamma);
                }
                moments(fn, coef, k.g2m1, c, s, r.mx, r.my);
                r.hx0 = r.hx1; r.hy0 = r.hy1; r.hx1 = r.x; r.hy1 = r.y;
                r.x = fx; r.y = fy; r.th = fth; r.n = fn;
                r.ux = c; r.uy = s; r.coef = coef;
                const double nray = coef * fn;
                r.tt = r.tt + dist * (r.nray + nray) / 2.0;
                r.nray = nray;
                set_flat_flags(r, true, true);
                r.hov = 0.f;           // a flat cell is not a steep one
                return !outside(k, r);
            }
            finish_state<METHOD>(F, gather, r);      // the full step starts from the gradient at (r.x, r.y)
        }
    }
    double fx, fy, fn, fgx, fgy;
    [[maybe_unused]] bool arrived_flat = false;
    const bool flag = ex::op_advance<METHOD>(k, r, fx, fy);
    if constexpr (IsPoly<G>::value) rt::n_gradient(F, gather, active, fx, fy, fn, fgx, fgy);   // kFastField: the cell's polynomial
    else if constexpr (flat_shortcut<METHOD>() && IsGlobalGather<G>::value) {
        // ... and whether the cell the lookup landed in is flat (one more load, in the shadow of the window's): the next step's
        // short way starts from a flat cell
        int cell = 0;
        ex::n_gradient<fallback_phases(METHOD), window_estimates(METHOD), window_votes(METHOD)>(F, gather, active, fx, fy, fn, fgx, fgy, &cell);
        r.hov = 0.f;               // (in this order of stepping Ray::hov is free: it carries the steepness of the cell the ray arrived in)
        if (F.flat != 0) { double cf; arrived_flat = flat_lane(F, cell, cf, r.hov); }
    }
    else ex::n_gradient<fallback_phases(METHOD), window_estimates(METHOD), window_votes(METHOD)>(F, gather, active, fx, fy, fn, fgx, fgy);
    const double fth = ex::op_angle<METHOD>(k, r, flag, fx, fy, fn, fgx, fgy, i);
    ex::store_update<inline_sincos(METHOD) || IsPoly<G>::value>(k, r, fx, fy, fth, fn, fgx, fgy);   // (IsPoly: op7 with RTMI_ORDER_FAST_FIELD, kFastField)
    if constexpr (flat_shortcut<METHOD>() && IsGlobalGather<G>::value) set_flat_flags(r, false, arrived_flat);
    return (METHOD == 7 && i <= 2) || !outside(k, r);     // no boundary test in op7's bootstrap rows
}

}  // namespace ex
}  // namespace rt
