// check_libm_sincos.cpp -- brute-force check of rt::gl::sin/cos (raytracing_amd/csrc/rt_libm.h) against the host
// libm's sin() and cos(), bit for bit, over every argument range of the algorithm.  Also counts how often libm's
// sincos() returns other bits than its sin() and cos(): on glibc 2.35 it does for 0.14 % of arguments, which is
// why oracle/Makefile passes -fno-builtin-sin -fno-builtin-cos (gcc would otherwise merge the reference's two
// separate calls into one sincos()).  Build with the same two flags, or the "libm" column is sincos() as well:
//   g++ -O2 -mfma -ffp-contract=off -fno-builtin-sin -fno-builtin-cos -fopenmp tools/check_libm_sincos.cpp \
//       -o /tmp/check_libm_sincos && /tmp/check_libm_sincos [n_per_range]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../raytracing_amd/csrc/rt_libm.h"

static const rt::gl::Tab TAB = RT_SINCOS_TAB_INIT;
static inline uint64_t rng_next(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double u01(uint64_t& s) { return (double)(rng_next(s) >> 11) * 0x1.0p-53; }

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 20000000L;
    struct { double lo, hi; const char* name; } R[] = {
        {0.0, 0.126, "taylor"}, {0.126, 0.855469, "table"}, {0.855469, 2.426265, "pi/2-x"},
        {2.426265, 8.0, "reduce small"}, {8.0, 200.0, "reduce mid"}, {200.0, 1.0e5, "reduce large"},
        {1.0e5, 105414350.0, "reduce huge"}, {1e-9, 1e-7, "tiny"}, {0.125, 0.127, "edge .126"},
        {0.8554, 0.8556, "edge .855"}, {2.4262, 2.4263, "edge 2.426"}};
    long bad_total = 0, sc_total = 0;
    for (auto& r : R) {
        long bad_s = 0, bad_c = 0, bad_sc = 0;
#pragma omp parallel for reduction(+ : bad_s, bad_c, bad_sc)
        for (int t = 0; t < 64; t++) {
            uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1) + (uint64_t)(r.lo * 1e6);
            for (long i = 0; i < n / 64; i++) {
                double x = r.lo + (r.hi - r.lo) * u01(s);
                if (rng_next(s) & 1) x = -x;
                const double s0 = std::sin(x), c0 = std::cos(x);
                const double s1 = rt::gl::sin(TAB.v, x), c1 = rt::gl::cos(TAB.v, x);
                double s2, c2;
                sincos(x, &s2, &c2);
                bad_s += memcmp(&s0, &s1, 8) != 0;
                bad_c += memcmp(&c0, &c1, 8) != 0;
                bad_sc += (memcmp(&s0, &s2, 8) != 0) + (memcmp(&c0, &c2, 8) != 0);
            }
        }
        printf("%-14s [%g, %g): %ld args, sin mismatches %ld, cos mismatches %ld, libm sincos != sin/cos %ld\n", r.name,
               r.lo, r.hi, n / 64 * 64, bad_s, bad_c, bad_sc);
        bad_total += bad_s + bad_c;
        sc_total += bad_sc;
    }
    // multiples of 1/128 and of pi/2 (table nodes, reduction boundaries)
    long bad = 0;
    for (int k = -20000; k <= 20000; k++) {
        for (int j = -2; j <= 2; j++) {
            double xs[2] = {k / 128.0, k * M_PI / 2};
            for (double x : xs) {
                for (int q = 0; q < j * (j > 0 ? 1 : -1); q++) x = nextafter(x, j > 0 ? 1e300 : -1e300);
                const double s0 = std::sin(x), c0 = std::cos(x), s1 = rt::gl::sin(TAB.v, x), c1 = rt::gl::cos(TAB.v, x);
                bad += (memcmp(&s0, &s1, 8) != 0) + (memcmp(&c0, &c1, 8) != 0);
            }
        }
    }
    printf("nodes k/128 and k*pi/2 (+-2 ulp): mismatches %ld\n", bad);
    bad_total += bad;
    printf("libm sincos() differs from libm sin()/cos() on %ld arguments (information only)\n", sc_total);
    printf(bad_total ? "FAIL\n" : "OK: rt::gl::sin/cos bit-identical to the host libm's sin()/cos()\n");
    return bad_total != 0;
}
