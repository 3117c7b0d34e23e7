/* A C caller of include/rtmi.h, linked against librtmi.so: what a non-Python host of the reference's loop would write.
 * genZ + interpolacion (rtmi_field_build), trazar's preamble (rtmi_batch_create), the loop (rtmi_run), the returned arrays
 * (rtmi_read_d_ray / rtmi_read_final / rtmi_read_rows), n_gradient (rtmi_field_eval), checkpoint / resume, error reporting.
 *
 *   cabi_smoke <expect.bin>     expect.bin (written by tests/test_cabi_native.py from the ORACLE): int64 R, int64 max_size,
 *                               double step, double theta[R], double d_ray[3][R], double final[9][R], double row_T[R]
 *                               (traveltime column of recorded row 64)
 * Prints what it compared; exit status 0 only if every step count is equal and every value within 1e-9 of its quantity's
 * scale.  (Built and run by tests/test_cabi_native.py; on a box without a GPU rtmi_field_build must fail with RTMI_ERR_HIP.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtmi.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != RTMI_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rtmi_last_error()); return 2; } \
    } while (0)

static double worst(const double *a, const double *b, int64_t n) {   /* max |a - b| / max |b| over one quantity group */
    double scale = 0, diff = 0;
    for (int64_t i = 0; i < n; i++) {
        if (fabs(b[i]) > scale) scale = fabs(b[i]);
        if (fabs(a[i] - b[i]) > diff) diff = fabs(a[i] - b[i]);
    }
    return diff == 0 ? 0 : diff / scale;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: cabi_smoke expect.bin\n"); return 64; }
    if (rtmi_abi_version() != RTMI_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 3; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 64; }
    int64_t R = 0, max_size = 0;
    double step = 0;
    if (fread(&R, 8, 1, f) != 1 || fread(&max_size, 8, 1, f) != 1 || fread(&step, 8, 1, f) != 1 || R <= 0 || R > 1 << 20) return 64;
    double *theta = malloc(8 * R), *ed = malloc(24 * R), *ef = malloc(72 * R), *eT = malloc(8 * R);
    double *x0 = malloc(8 * R), *y0 = malloc(8 * R), *d = malloc(24 * R), *fin = malloc(72 * R), *row = malloc(48 * R);
    if (fread(theta, 8, R, f) != (size_t)R || fread(ed, 8, 3 * R, f) != (size_t)(3 * R) || fread(ef, 8, 9 * R, f) != (size_t)(9 * R) ||
        fread(eT, 8, R, f) != (size_t)R) return 64;
    fclose(f);
    for (int64_t k = 0; k < R; k++) { x0[k] = -2.0; y0[k] = -2.0; }

    int ndev = -1;
    int rc = rtmi_device_count(&ndev);
    if (rc != RTMI_OK || ndev < 1) {       /* no GPU: the library must say so, not compute on the CPU */
        rtmi_field *none = NULL;
        rc = rtmi_field_build(RTMI_VERT_HETEROGENEOUS, -2, 5, -2.5, 1, 0.05293304824724534 / 3, RTMI_F64, NULL, &none);
        printf("no HIP device: rtmi_field_build -> %d (%s)\n", rc, rtmi_last_error());
        return rc == RTMI_ERR_HIP ? 77 : 4;
    }
    CHECK(rtmi_set_device(0));
    rtmi_field *fld = NULL;
    CHECK(rtmi_field_build(RTMI_VERT_HETEROGENEOUS, -2, 5, -2.5, 1, 0.05293304824724534 / 3, RTMI_F64, NULL, &fld));   /* :1587-1588 */
    int qx = 0, qy = 0;
    CHECK(rtmi_field_dims(fld, &qx, &qy));
    double px = -2.0, py = -2.0, n0, gx0, gy0;
    CHECK(rtmi_field_eval(fld, 1, &px, &py, &n0, &gx0, &gy0));                /* n_gradient at the launch point (:815) */
    printf("grid %d x %d, n(-2,-2) = %.17g (SURVEY anchor 0.07142864686293911)\n", qx, qy, n0);
    if (fabs(n0 - 0.07142864686293911) > 1e-15) return 5;

    rtmi_params p;
    memset(&p, 0, sizeof p);                /* every field not set below: the library's default (launch_mode RTMI_LAUNCH_AUTO ...) */
    p.method = 6; p.dtype = RTMI_F64; p.gamma = 1; p.gamma_step = 1; p.step = step; p.max_size = (int32_t)max_size;
    p.record_stride = 64; p.no_n_ray = 1;
    p.box[0] = -2; p.box[1] = 5; p.box[2] = -2.5; p.box[3] = 1;
    rtmi_batch *b = NULL;
    CHECK(rtmi_batch_create(fld, &p, R, x0, y0, theta, NULL, &b));

    /* half the way, checkpoint, the rest on a second batch: the loop (:866-879) cut in two gives the same bits */
    CHECK(rtmi_step(b, 700));
    double *st = malloc(72 * R), *aux = malloc(32 * R);
    int32_t *is = malloc(4 * R);
    uint8_t *al = malloc(R);
    CHECK(rtmi_batch_get_state(b, st, aux, is, al));
    rtmi_batch *b2 = NULL;
    CHECK(rtmi_batch_create(fld, &p, R, x0, y0, theta, NULL, &b2));
    CHECK(rtmi_batch_restore_state(b2, st, aux, is, al));
    CHECK(rtmi_run(b2));
    CHECK(rtmi_run(b));
    double *d2 = malloc(24 * R), *fin2 = malloc(72 * R);
    CHECK(rtmi_read_d_ray(b, d)); CHECK(rtmi_read_final(b, fin));
    CHECK(rtmi_read_d_ray(b2, d2)); CHECK(rtmi_read_final(b2, fin2));
    if (memcmp(d, d2, 24 * R) || memcmp(fin, fin2, 72 * R)) { fprintf(stderr, "resumed run differs from the uninterrupted one\n"); return 6; }
    CHECK(rtmi_read_rows(b, 1, 1, row, NULL));                                   /* recorded row 1 = step 64: x y p_x p_y T theta */

    rtmi_stats s;
    CHECK(rtmi_batch_stats(b, &s));
    int64_t steps = 0, bad_steps = 0;
    for (int64_t k = 0; k < R; k++) { steps += (int64_t)d[2 * R + k]; bad_steps += d[2 * R + k] != ed[2 * R + k]; }
    /* per quantity: d_ray (dist_real, dist_sim), final x y | theta | n | grad | p | T, and the traveltime column of row 64 */
    double e = worst(d, ed, 2 * R), w;
    const int g0[6] = {0, 2, 3, 4, 6, 8}, gl[6] = {2, 1, 1, 2, 2, 1};
    for (int g = 0; g < 6; g++) if ((w = worst(fin + g0[g] * R, ef + g0[g] * R, gl[g] * R)) > e) e = w;
    if ((w = worst(row + 4 * R, eT, R)) > e) e = w;
    printf("%lld rays, %lld ray-steps (stats: %llu), %lld step counts differ, max relative error %.3g, kernel %.3f ms, schedule %u, fallbacks %u\n",
           (long long)R, (long long)steps, (unsigned long long)s.ray_steps, (long long)bad_steps, e, s.kernel_ms, s.launch_mode_used,
           s.auto_fallbacks);
    /* an argument error is a status code and a message, never a crash */
    p.method = 12;
    rtmi_batch *bad = NULL;
    rc = rtmi_batch_create(fld, &p, R, x0, y0, theta, NULL, &bad);
    printf("method 12 -> %d (%s)\n", rc, rtmi_last_error());
    rtmi_batch_destroy(b); rtmi_batch_destroy(b2); rtmi_field_destroy(fld);
    return (bad_steps == 0 && (uint64_t)steps == s.ray_steps && e < 1e-9 && rc == RTMI_ERR_ARG && s.auto_fallbacks == 0) ? 0 : 1;
}
