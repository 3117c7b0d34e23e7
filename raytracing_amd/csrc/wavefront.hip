// wavefront.hip -- the across-ray stage of the reference's wavefront extraction (RT_bench.py:1005-1026, 1043-1044) on
// the device: per traveltime, the isochrone points of the rays that reach it (rtmi_isochrones' per-ray PCHIP stage,
// :987-1003) are sorted by y (np.argsort, :1016), scipy's PchipInterpolator x(y) is built through them (:1020), and its
// derivative at the points (:1021-1022), the tangent / normal angles (:1025-1026), |ray angle - normal angle| (:1032) and
// the interpolant on nfine equally spaced y (:1043-1044) are evaluated.  One lane per point; the sort is rocPRIM's radix
// sort through hipCUB (library code: this stage is a consumer of the hot path, not part of it).  Every traveltime of a call is
// handled in one pass -- the reference's animation (:1066-1102) re-does the whole extraction per frame, 45 times.
// Third-party arithmetic restated: scipy.interpolate.PchipInterpolator (scipy 1.15.3 in the build image):
// _find_derivatives (Fritsch-Butland weighted harmonic mean, three-point end rule), CubicHermiteSpline's power-basis
// coefficients, PPoly's Horner evaluation and .derivative().
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_internal.h"

namespace {
__device__ __forceinline__ double sgn_(double v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ double pchip_edge(double h0, double h1, double m0, double m1) {
    double d = ((2 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
    if (sgn_(d) != sgn_(m0)) d = 0;
    else if (sgn_(m0) != sgn_(m1) && fabs(d) > 3 * fabs(m0)) d = 3 * m0;
    return d;
}
// scipy's derivative estimate at point j of the n-point data set (t, v), n >= 2
__device__ double pchip_deriv(const double* t, const double* v, long j, long n) {
    if (n == 2) return (v[1] - v[0]) / (t[1] - t[0]);
    if (j == 0) {
        const double h0 = t[1] - t[0], h1 = t[2] - t[1];
        return pchip_edge(h0, h1, (v[1] - v[0]) / h0, (v[2] - v[1]) / h1);
    }
    if (j == n - 1) {
        const double h0 = t[n - 1] - t[n - 2], h1 = t[n - 2] - t[n - 3];
        return pchip_edge(h0, h1, (v[n - 1] - v[n - 2]) / h0, (v[n - 2] - v[n - 3]) / h1);
    }
    const double ha = t[j] - t[j - 1], hb = t[j + 1] - t[j];
    const double ma = (v[j] - v[j - 1]) / ha, mb = (v[j + 1] - v[j]) / hb;
    if (sgn_(ma) != sgn_(mb) || ma == 0 || mb == 0) return 0;
    const double w1 = 2 * hb + ha, w2 = hb + 2 * ha;
    return 1.0 / ((w1 / ma + w2 / mb) / (w1 + w2));
}

// All the kernels below work on a CHUNK of traveltimes at once (blockIdx.y = traveltime within the chunk): the reference's movie
// path (RT_bench.py:1066-1102) asks for 45 wavefronts of one trajectory set, frame after frame; here they are one pass.
// keys for the sort: y of the rays that reach the traveltime, +inf for the others (they sort last); counts the valid ones.
// vals: the item's index t*R + k.
__global__ void k_keys(const double* iso, long R, double* keys, int* vals, unsigned long long* count) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    bool ok = false;
    if (k < R) {
        const double y = iso[((size_t)t * 3 + 1) * R + k];                 // iso: [nt][3][R] = x, y, theta per traveltime
        ok = y == y;
        keys[(size_t)t * R + k] = ok ? y : HUGE_VAL;
        vals[(size_t)t * R + k] = (int)(t * R + k);
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count + t, (unsigned long long)__popcll(m));
}
// second sort key: the traveltime an item belongs to (a stable sort on it regroups the y-sorted items per traveltime)
__global__ void k_frame_keys(const int* vals, long R, long n, unsigned* fk) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) fk[j] = (unsigned)(vals[j] / R);
}
// sorted order -> ys, xs, angle, ray index rows of `nodes` ([nt][7][R]); NaN beyond the valid count
__global__ void k_gather(const double* iso, long R, const int* order, const unsigned long long* count, double* nodes) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (j >= R) return;
    const bool ok = (unsigned long long)j < count[t];
    const long k = (long)order[(size_t)t * R + j] - t * R;
    const double* it = iso + (size_t)t * 3 * R;
    double* nd = nodes + (size_t)t * 7 * R;
    nd[j] = ok ? it[R + k] : NAN;                // y
    nd[R + j] = ok ? it[k] : NAN;                // x
    nd[2 * R + j] = ok ? it[2 * R + k] : NAN;    // ray angle
    nd[6 * R + j] = ok ? (double)k : NAN;        // ray index (caller's order)
}
// derivative of the interpolant at its own points, angles (:1021-1026, :1032)
__global__ void k_nodes(long R, const unsigned long long* count, double* nodes_all, double* deriv_all) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    const long n = (long)count[t];
    if (j >= R) return;
    double* nodes = nodes_all + (size_t)t * 7 * R;
    double d = NAN, slope = NAN, normal = NAN, diff = NAN;
    if (j < n && n >= 2) {
        const double *y = nodes, *x = nodes + R;
        d = pchip_deriv(y, x, j, n);
        slope = d;
        if (j == n - 1) {
            // PPoly.derivative() evaluated at the last breakpoint uses the last interval at its right end:
            // ((3 c0) s + 2 c1) s + c2 with CubicHermiteSpline's coefficients of interval n-2
            const double dx = y[n - 1] - y[n - 2], m = (x[n - 1] - x[n - 2]) / dx;
            const double d0 = pchip_deriv(y, x, n - 2, n);
            const double tq = (d0 + d - 2 * m) / dx;
            const double c0 = tq / dx, c1 = (m - d0) / dx - tq;
            slope = (3 * c0 * dx + 2 * c1) * dx + d0;
        }
        const double tangent = M_PI / 2 - atan(slope);       // (:1025)
        normal = tangent - M_PI / 2;                          // (:1026)
        diff = fabs(nodes[2 * R + j] - normal);               // (:1032) with the ray angle of the SAME sorted point
    }
    deriv_all[(size_t)t * R + j] = d;
    nodes[3 * R + j] = slope;
    nodes[4 * R + j] = normal;
    nodes[5 * R + j] = diff;
}
// the interpolant on nfine equally spaced y between the first and the last point (:1043-1044, :1096-1097)
__global__ void k_fine(long R, const unsigned long long* count, const double* nodes_all, const double* deriv_all, int nfine, double* fine_all) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const long t = blockIdx.y;
    if (q >= nfine) return;
    const long n = (long)count[t];
    const double* nodes = nodes_all + (size_t)t * 7 * R;
    const double* deriv = deriv_all + (size_t)t * R;
    double* fine = fine_all + (size_t)t * 2 * nfine;
    double xf = NAN, yf = NAN;
    if (n >= 2) {
        const double *y = nodes, *x = nodes + R;
        const double a = y[0], b = y[n - 1];
        const double step = (b - a) / (double)(nfine - 1);
        yf = q == nfine - 1 ? b : (double)q * step + a;                     // numpy.linspace
        long lo = 0, hi = n - 1;
        while (hi - lo > 1) {
            const long mid = (lo + hi) >> 1;
            if (y[mid] <= yf) lo = mid; else hi = mid;
        }
        const double dx = y[lo + 1] - y[lo], s = yf - y[lo], m = (x[lo + 1] - x[lo]) / dx;
        const double d0 = deriv[lo], d1 = deriv[lo + 1];
        const double tq = (d0 + d1 - 2 * m) / dx;
        const double c0 = tq / dx, c1 = (m - d0) / dx - tq;
        xf = ((c0 * s + c1) * s + d0) * s + x[lo];                           // PPoly: Horner in (y - y_lo)
    }
    fine[q] = xf;
    fine[nfine + q] = yf;
}
}  // namespace

#define WF_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { rc = rtmi_internal_fail(RTMI_ERR_HIP, (std::string("rtmi_wavefronts: ") + #expr + ": " + hipGetErrorString(e_)).c_str()); goto done; } \
    } while (0)

extern "C" __attribute__((visibility("default"))) int rtmi_wavefronts(rtmi_batch* b, int32_t ntimes, const double* times, int32_t nfine,
                                                                      int64_t* count, double* nodes, double* fine) {
    if (!b || !times || !count || !nodes) return rtmi_internal_fail(RTMI_ERR_ARG, "rtmi_wavefronts: null");
    if (nfine < 0 || nfine == 1 || (nfine > 0 && !fine)) return rtmi_internal_fail(RTMI_ERR_ARG, "rtmi_wavefronts: nfine must be 0 or >= 2 (with a fine buffer)");
    double* iso = nullptr;
    long R = 0;
    hipStream_t st = nullptr;
    int rc = rtmi_internal_isochrones_device(b, ntimes, times, &iso, &R, (void**)&st);
    if (rc) return rc;
    // Traveltimes are processed in chunks of `tc` (all of them, unless that needs more than ~1 GB of work arrays: 92 bytes per
    // point): per chunk ONE stable radix sort of every point by y, one more by the traveltime it belongs to (which regroups the
    // y-sorted points per wavefront: each has exactly R of them, the rays that do not reach it at the end with y = +inf), the
    // PCHIP stage for all wavefronts at once, one copy to the host.
    const size_t Rz = (size_t)R;
    int tc = (int)std::max<size_t>(1, std::min<size_t>((size_t)ntimes, ((size_t)1 << 30) / (92 * Rz)));
    if (const char* e = getenv("RTMI_WF_CHUNK")) tc = std::max(1, std::min((int)ntimes, atoi(e)));   // tests: force several chunks
    while ((size_t)tc * Rz >= ((size_t)1 << 31)) tc /= 2;              // item indices are 32-bit
    const size_t N = (size_t)tc * Rz;
    double *keys = nullptr, *keys2 = nullptr, *dn = nullptr, *dd = nullptr, *df = nullptr;
    int *vals = nullptr, *vals2 = nullptr;
    unsigned *fk = nullptr, *fk2 = nullptr;
    unsigned long long* dcount = nullptr;
    std::vector<unsigned long long> hcount;
    void* tmp = nullptr;
    size_t tmp1 = 0, tmp2 = 0;
    const dim3 blk(256);
    int fbits = 1;
    while ((1 << fbits) < tc) fbits++;
    try {
        hcount.resize((size_t)tc);
    } catch (const std::exception&) {
        rc = rtmi_internal_fail(RTMI_ERR_ALLOC, "rtmi_wavefronts: host allocation failed");
        goto done;
    }
    WF_TRY(hipMalloc(&keys, N * 8)); WF_TRY(hipMalloc(&keys2, N * 8));
    WF_TRY(hipMalloc(&vals, N * 4)); WF_TRY(hipMalloc(&vals2, N * 4));
    WF_TRY(hipMalloc(&fk, N * 4)); WF_TRY(hipMalloc(&fk2, N * 4));
    WF_TRY(hipMalloc(&dn, 7 * N * 8)); WF_TRY(hipMalloc(&dd, N * 8));
    WF_TRY(hipMalloc(&dcount, (size_t)tc * 8));
    if (nfine) WF_TRY(hipMalloc(&df, (size_t)tc * 2 * (size_t)nfine * 8));
    WF_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp1, keys, keys2, vals, vals2, (int)N, 0, 64, st));
    WF_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp2, fk, fk2, vals2, vals, (int)N, 0, fbits, st));
    WF_TRY(hipMalloc(&tmp, std::max(tmp1, tmp2)));
    for (int t0 = 0; t0 < ntimes; t0 += tc) {
        const int nt = std::min(tc, ntimes - t0);
        const size_t n = (size_t)nt * Rz;
        const double* iso_c = iso + (size_t)t0 * 3 * Rz;
        const dim3 grd((unsigned)((Rz + 255) / 256), (unsigned)nt);
        WF_TRY(hipMemsetAsync(dcount, 0, (size_t)nt * 8, st));
        hipLaunchKernelGGL(k_keys, grd, blk, 0, st, iso_c, R, keys, vals, dcount);
        WF_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp1, keys, keys2, vals, vals2, (int)n, 0, 64, st));           // every point by y
        if (nt > 1) {
            hipLaunchKernelGGL(k_frame_keys, dim3((unsigned)((n + 255) / 256)), blk, 0, st, vals2, R, (long)n, fk);
            WF_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp2, fk, fk2, vals2, vals, (int)n, 0, fbits, st));          // stable: regrouped per traveltime
        }
        const int* order = nt > 1 ? vals : vals2;
        hipLaunchKernelGGL(k_gather, grd, blk, 0, st, iso_c, R, order, dcount, dn);
        hipLaunchKernelGGL(k_nodes, grd, blk, 0, st, R, dcount, dn, dd);
        if (nfine) hipLaunchKernelGGL(k_fine, dim3((nfine + 255) / 256, (unsigned)nt), blk, 0, st, R, dcount, dn, dd, (int)nfine, df);
        WF_TRY(hipGetLastError());
        WF_TRY(hipMemcpyAsync(hcount.data(), dcount, (size_t)nt * 8, hipMemcpyDeviceToHost, st));
        WF_TRY(hipMemcpyAsync(nodes + (size_t)t0 * 7 * Rz, dn, 7 * n * 8, hipMemcpyDeviceToHost, st));
        if (nfine) WF_TRY(hipMemcpyAsync(fine + (size_t)t0 * 2 * nfine, df, (size_t)nt * 2 * (size_t)nfine * 8, hipMemcpyDeviceToHost, st));
        WF_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < nt; i++) count[t0 + i] = (int64_t)hcount[i];
    }
done:
    (void)hipFree(iso); (void)hipFree(keys); (void)hipFree(keys2); (void)hipFree(vals); (void)hipFree(vals2);
    (void)hipFree(fk); (void)hipFree(fk2);
    (void)hipFree(dn); (void)hipFree(dd); (void)hipFree(df); (void)hipFree(dcount); (void)hipFree(tmp);
    return rc;
}
