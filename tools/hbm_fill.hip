// Write-only HBM ceiling: fills a buffer the size of the full trajectory record with 8- and 16-byte stores, plain and
// nontemporal, in the recording kernel's own pattern (a block writes 6 x 2 KB pieces of a row, row after row).
// hipcc --offload-arch=gfx950 -O3 -o build/hbm_fill tools/hbm_fill.hip ; build/hbm_fill [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NT> __global__ void fill_b64(double* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) { if (NT) __builtin_nontemporal_store(1.0, p + i); else p[i] = 1.0; }
}
typedef double rt_d2 __attribute__((ext_vector_type(2)));
template <int NT> __global__ void fill_b128(rt_d2* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    const rt_d2 v = {1.0, 1.0};
    for (; i < n; i += s) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
// rows x 6 x R doubles; block b owns columns [256 b, 256 b + 256) of every row (the k_advance pattern)
template <int NT> __global__ void fill_rows(double* p, long R, long rows) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long r = 0; r < rows; r++)
        for (int q = 0; q < 6; q++) { double* d = p + (r * 6 + q) * R + k; if (NT) __builtin_nontemporal_store(1.0, d); else *d = 1.0; }
}
template <typename F> static void timed(const char* name, double gb, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-28s %7.2f ms  %.2f TB/s\n", name, ms, gb / ms);
}
int main(int argc, char** argv) {
    const long R = 1 << 20, rows = argc > 1 ? atol(argv[1]) : 1848;
    const size_t n = (size_t)rows * 6 * R; const double gb = n * 8 / 1e9;
    double* p; if (hipMalloc(&p, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("%.1f GB\n", gb);
    timed("hipMemsetAsync", gb, [&] { hipMemsetAsync(p, 0, n * 8, 0); });
    for (int blocks : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "b64 grid-stride %d", blocks); timed(nm, gb, [&] { fill_b64<0><<<blocks, 256>>>(p, n); });
        snprintf(nm, 64, "b64 nt grid-stride %d", blocks); timed(nm, gb, [&] { fill_b64<1><<<blocks, 256>>>(p, n); });
        snprintf(nm, 64, "b128 grid-stride %d", blocks); timed(nm, gb, [&] { fill_b128<0><<<blocks, 256>>>((rt_d2*)p, n / 2); });
        snprintf(nm, 64, "b128 nt grid-stride %d", blocks); timed(nm, gb, [&] { fill_b128<1><<<blocks, 256>>>((rt_d2*)p, n / 2); });
    }
    timed("rows pattern", gb, [&] { fill_rows<0><<<R / 256, 256>>>(p, R, rows); });
    timed("rows pattern nt", gb, [&] { fill_rows<1><<<R / 256, 256>>>(p, R, rows); });
    hipFree(p); return 0;
}
