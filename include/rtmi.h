/*
 * rtmi.h -- C ABI of librtmi.so: MI355X (gfx950) ray propagation for the
 * shooting-method hot path of neyuru/RayTracing's RT_bench.py.
 *
 * This header is the drop-in boundary.  The reference has no FFI layer; its
 * seam is the Python call surface, so each entry point below names the
 * reference interface (RT_bench.py file:line) it stands in for.  Plain
 * pointers and sizes only; every function returns 0 on success or a negative
 * rtmi_status, never throws, and rtmi_last_error() describes the last failure
 * on the calling thread.  Handles are not thread-safe (the reference's trazar
 * is not re-entrant either: RT_bench.py:73, 646-648).
 *
 * There is NO CPU fallback behind this ABI: with no HIP device the calls fail
 * with RTMI_ERR_HIP.  The CPU restatement used by the tests lives in oracle/.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_ABI_VERSION 7

typedef enum {
    RTMI_OK = 0,
    RTMI_ERR_ARG = -1,      /* bad argument (null pointer, size, enum out of range) */
    RTMI_ERR_HIP = -2,      /* a HIP runtime call failed / no device */
    RTMI_ERR_ALLOC = -3,    /* device or host allocation failed */
    RTMI_ERR_STATE = -4,    /* call not valid in the handle's state */
    RTMI_ERR_UNSUPPORTED = -5
} rtmi_status;

/* scenario <-> user_choice "1".."4" (RT_bench.py:1565-1580); 4 reuses field 3 with gamma=3 (:1579) */
typedef enum { RTMI_INTERFACE = 1, RTMI_FISHEYE = 2, RTMI_VERT_HETEROGENEOUS = 3, RTMI_ANISOTROPY = 4 } rtmi_scenario;

/* method m <-> step function op<m> (RT_bench.py:469-764; menus :1238-1264 and :1286-1291) */
typedef enum { RTMI_OP_MIN = 1, RTMI_OP_MAX = 11 } rtmi_method_range;

typedef enum { RTMI_F64 = 0, RTMI_F32 = 1 } rtmi_dtype;

/* Device schedules of the loop at RT_bench.py:866-879 (rtmi_params.launch_mode).  The reference runs rays one after another
 * (:807); every schedule here gives each ray the same arithmetic, hence the same bits. */
typedef enum {
    RTMI_LAUNCH_AUTO = 0,    /* the library chooses between SLICED and PLAIN (see rtmi_params.launch_mode) */
    RTMI_LAUNCH_REFILL = 1,  /* persistent waves; terminated lanes are refilled from a device queue (ballot + prefix compaction) */
    RTMI_LAUNCH_SLICED = 2,  /* persistent blocks advance 256-ray bundles in time slices of slice_steps rows: balances fans
                                whose rays differ much in length */
    RTMI_LAUNCH_PLAIN = 3    /* one lane per ray to completion, one block slot per 256 rays */
} rtmi_launch_mode;

/* rtmi_params.reference_order */
typedef enum { RTMI_ORDER_DEFAULT = 0, RTMI_ORDER_REFERENCE = 1, RTMI_ORDER_FUSED = 2, RTMI_ORDER_FAST_FIELD = 3 } rtmi_order;

typedef struct rtmi_field rtmi_field;   /* z + grd of interpolacion() (:435-464), resident in HBM */
typedef struct rtmi_batch rtmi_batch;   /* one trazar() call's ray batch (:766-948), resident in HBM */

/* ------------------------------------------------------------------ library */
int rtmi_abi_version(void);
const char *rtmi_last_error(void);
/* Select the HIP device for subsequent creates on this thread (one process per GPU: pass LOCAL_RANK). */
int rtmi_set_device(int device);
int rtmi_device_count(int *count);

/* -------------------------------------------------------------------- field
 * rtmi_field_build == genZ(xi,xs,yi,ys) (:412-433) followed by interpolacion() (:435-464):
 * samples the scenario's n(x,y) on linspace(xi-3, xs+3, int((xs-xi+6)/delta+1)) x likewise in y,
 * np.gradient(Z, delta, edge_order=2) (:450), bilinear n and not-a-knot bicubic fits of the two
 * gradient components (:455-457), all on the device.  The Hessian fits (:459-462) are never read by
 * the path and are not built.  `stream` is a hipStream_t (NULL = default stream). */
int rtmi_field_build(int scenario, double xi, double xs, double yi, double ys, double delta,
                     int dtype, void *stream, rtmi_field **out);
/* interpolacion(x, y, Z, X, Y) (:435) for caller-provided samples Z[qy][qx] (host pointers).
 * x and y must be the linspace axes genZ produces (checked bit for bit). */
int rtmi_field_from_samples(const double *x, int qx, const double *y, int qy, const double *Z,
                            double delta, int dtype, void *stream, rtmi_field **out);
int rtmi_field_dims(const rtmi_field *f, int *qx, int *qy);
/* Copy the fp64 build products to host buffers (any may be NULL): axes, n samples, and the spline
 * coefficients of GradX (=d/dy) and GradY (=d/dx), each [qy][qx] -- what get_coeffs() returns. */
int rtmi_field_read(const rtmi_field *f, double *x, double *y, double *Z, double *coef_dy, double *coef_dx);
/* n_gradient(vector, grd, z) (:141-156) for npts host points -> n, dn/dx, dn/dy (host buffers). */
int rtmi_field_eval(const rtmi_field *f, int64_t npts, const double *x, const double *y,
                    double *n, double *gx, double *gy);
void rtmi_field_destroy(rtmi_field *f);

/* -------------------------------------------------------------------- batch */
typedef struct {
    int32_t method;          /* 1..11 -> op1..op11 */
    int32_t dtype;           /* must equal the field's dtype */
    double gamma;            /* trazar's local gamma from constants() (:793) */
    double gamma_step;       /* module-global gamma read by op10/op11 (:725, :758); normally == gamma */
    double step;             /* DELTA_S */
    int32_t max_size;        /* rows incl. row 0: int(ceil(s/step)+1) (:799) or N*divisor (:797) */
    int32_t record_stride;   /* 0: keep no trajectory; s>=1: store rows i with i % s == 0 (1 = reference layout) */
    int64_t rec_rows;        /* rows allocated for s_ray/n_ray (row r holds step r*stride); 0 -> derived from max_size */
    double box[4];           /* limx_i, limx_s, limy_i, limy_s (:878) */
    int32_t launch_mode;     /* how rtmi_run schedules the loop on the device (rtmi_launch_mode); results are bit-identical in
                                all of them.  0 = RTMI_LAUNCH_AUTO, the value a zero-initialised struct gets: time-sliced bundles
                                when the batch has more 256-ray bundles than the device holds resident blocks, else the plain
                                launch; a batch that is re-run (rtmi_batch_reset + rtmi_run) is timed RTMI_AUTO_SAMPLES times under
                                each schedule, interleaved (sliced, plain, plain, sliced, sliced, plain), and then keeps ONE for
                                good: the plain launch if its median is more than 3 % ahead, else slicing (rtmi_stats.auto_ms /
                                auto_kept hold the record, launch_mode_used what the last run used) */
    int32_t block_size;      /* 0 -> default */
    int32_t refill_min;      /* RTMI_LAUNCH_REFILL: compact when this many lanes of a wave are idle (0 -> 32) */
    int32_t exact_basis;     /* kept for ABI compatibility (0 or 1; it used to select fpbspl on the true knots), no effect now: the fast-form methods evaluate the
                                field as one polynomial per grid cell, converted from FITPACK's splines on the TRUE knots of
                                every cell (< 1e-15 of the field's scale from FITPACK's own evaluation, rim cells included);
                                rtmi_field_eval and the reference-order methods use FITPACK's arithmetic itself */
    int32_t field_path;      /* how a wave gets at the field; identical results.  0 auto; 1 every lane reads for itself
                                (L1/L2); 2 wave-shared: the fast-form methods read a wave-uniform cell's polynomial through
                                the scalar cache into scalar registers (lanes in other cells fall back to 1), the
                                reference-order methods stage a wave-private LDS tile of B-spline coefficients; 3 as 2, but
                                the reference-order methods read a wave-uniform cell's B-spline window, knots and knot
                                reciprocals through the scalar cache (a wave in several cells falls back to 1).  Auto: 2 for
                                the fast-form methods and fp32; for fp64 op3/4/5/7/9/10/11 and reference_order 3 from
                                131 072 rays on (two waves per SIMD), 1 below -- and 1 for the golden-section methods when a
                                step is longer than half a grid cell (measured: the fisheye fan at DELTA_S = 2 pi / 303) */
    int32_t sort_rays;       /* 1: reorder rays inside the batch by launch cell block and angle so that lanes of a wave stay
                                coherent; every read call still answers in the caller's ray order (see rtmi_device_view.perm) */
    /* optional caller-owned DEVICE buffers (e.g. torch tensors); NULL -> the library allocates */
    void *ext_s_ray;         /* [rec_rows][6][R] of dtype: x, y, p_x, p_y, T, theta (:802, :871-875) */
    void *ext_n_ray;         /* [rec_rows][R]   of dtype: coef*n (:803, :873) */
    int32_t lazy_clear;      /* 0 (default): rtmi_batch_reset zeroes s_ray/n_ray like the reference's np.zeros (:802-803), so rows
                                past a ray's last written row read 0 at every moment.  1: reset skips that memset when the re-run
                                rewrites exactly the rows the previous run wrote (same launch conditions and steps; set_state /
                                set_per_ray force a clear); until the re-run has finished, rows beyond a ray's CURRENT row still
                                hold the previous pass's (identical) values -- for timed re-runs of one batch (bench.py) */
    int32_t no_n_ray;        /* 1: keep no n_ray rows.  n_ray (coef*n per row, :803) is an internal array of trazar -- it feeds the
                                traveltime recurrence (:874) and is not among trazar's return values (:948) -- so a caller of the
                                reference's call surface never sees it; dropping it saves 1/7 of the recorded bytes */
    int32_t slice_steps;     /* time-sliced schedule: DELTA_S steps per time slice of a bundle (0 -> 512); a bundle's first two
                                slices are 4 and 2 times as long */
    int32_t reference_order; /* rtmi_order.  RTMI_ORDER_DEFAULT (0): op1/2/6/8 step in fused forms (~1e-13 from the reference per
                                trajectory).  op7's new angle differentiates POSITIONS (:370-372), so their last bits enter it: it
                                steps in the reference's own operation order throughout, like op3/4/5/9/10/11 always do (the
                                oracle's bits; 3.3 times the fused cost).  RTMI_ORDER_REFERENCE (1, fp64 only): op1/2/6/8 too: the
                                oracle's bits (the reference's, within 1 ulp where numpy's scalar pow(x, 2) is not x*x), at a
                                quarter to a third of the fused speed.  RTMI_ORDER_FUSED (2): fused forms wherever there is one,
                                op7 too (up to 8e-9 from the reference on the interface scenario); fp32 batches always run fused
                                forms.  RTMI_ORDER_FAST_FIELD (3): as DEFAULT, but op7 takes its reference-order step -- the
                                advancement's operation order, numpy's arctan2, glibc's sin / cos -- on the fused field lookup:
                                2.3 times faster, <= 8e-11 from the reference except on rays grazing a sharp interface at its
                                critical angle (2.6e-9 on one sampled ray of the 1 M-ray interface fan).  Such rays -- a handful
                                of a million, running along a sharp interface -- are ill-conditioned in the reference itself
                                (its rows move 1e-6 for a 1e-12 change of the launch angle): there only the reference-order
                                forms agree with it to 1e-9 -- which is why a DEFAULT fp64 op1/2/6/8 batch finds those rays on the way
                                and re-traces them in reference order by itself (no_retrace below): every ray of a default batch is
                                within 1e-9 of the reference */
    int32_t no_retrace;      /* 0 (default): a fused fp64 op1/2/6/8 batch on a field with a sharp transition (cells whose Hessian of n
                                is large against the size of the grid: the interface scenario; none in fisheye / vert_heterogeneous)
                                adds up, per ray and step, the steepness of the cell times |v . g| (v the ray's normal, g the
                                unit gradient): the exponent of the factor by which the trajectory amplifies a rounding
                                difference along a wall.  A ray whose sum says more than ~1e3 times is stopped, queued
                                and re-traced from its launch conditions in the reference's operation order (a hidden batch of the
                                same parameters, launched beside the main kernel); its rows and final state replace the fused
                                ones -- the oracle's bits.  A few hundred rays of a million on the interface fan; the re-trace runs on
                                eight compute units of its own beside the main kernel (which gets a stream of the batch's own
                                between two events on the caller's), and a batch that is run again starts with the bundles that
                                held its critical rays (rtmi_stats.dispatch_first): the interface fan then takes LESS time than
                                with no_retrace = 1 (the critical rays are the fused kernel's stragglers), 4 - 8 % more on a
                                batch's first run; rays are independent (RT_bench.py:807), so no other ray's bits change.  rtmi_run does
                                this before it returns; after rtmi_step it happens at the next call that reads results (rays handed
                                over are no longer live, and are then at their END, ahead of the others).  1: never (A/B runs).
                                The measure is calibrated on walls (the interface scenario's, also tilted against the grid);
                                ill-conditioning of another kind -- e.g. the focusing of a strongly bent wall in a field given
                                by samples -- is not seen by it: RTMI_ORDER_REFERENCE is the answer there.
                                rtmi_stats.retraced counts them */
} rtmi_params;

/* Upload R launch conditions (host pointers; x0/y0 per ray -- pos_x[k], -2 or the fisheye start, :809-813),
 * run the initial conditions (:814-826) on the device and write row 0. */
int rtmi_batch_create(const rtmi_field *f, const rtmi_params *p, int64_t R, const double *x0, const double *y0,
                      const double *theta0, void *stream, rtmi_batch **out);
/* Overwrite the ray state with caller-provided values (host, fp64): state9[9][R] = x, y, theta, n, dn/dx,
 * dn/dy, dist_sim, dist_real, T; hist4[4][R] = the two positions before (x,y), oldest first (op7's
 * VECTOR_LIST, :73; may be NULL for other methods); istep[R] = last written row (NULL keeps it).  Every ray
 * with istep+1 < max_size becomes live.  This is the explicit-argument form of one selected_func call
 * (:868): opN(i_angle, init_n, i_grad, i_unitv, i_vpos, coef_i, grd, z, step) with caller-chosen inputs.
 * fp64 op1/op2/op6/op8 batches carry the unit tangent (cos theta, sin theta) as ray state (it is advanced by rotation, not
 * recomputed from theta every step); a state set here restarts it from sin/cos of theta -- a state of the caller's own
 * making has no other.  To continue a run from a checkpoint use rtmi_batch_get_state / rtmi_batch_restore_state. */
int rtmi_batch_set_state(rtmi_batch *b, const double *state9, const double *hist4, const int32_t *istep);
/* Checkpoint: copy the current ray state to host buffers (any may be NULL), caller's ray order: state9 and istep as in
 * rtmi_batch_set_state; aux4[4][R] = the method's private state -- op7: the position history (hist4); fp64 op1/2/6/8: rows 0-1
 * the carried unit tangent (cos, sin), rows 2-3 zero; otherwise zero; alive[R] = 1 while the ray would still step (0 once it
 * left the box, :878, or ran out of rows). */
int rtmi_batch_get_state(rtmi_batch *b, double *state9, double *aux4, int32_t *istep, uint8_t *alive);
/* Resume: rtmi_batch_set_state with the method's private state taken from aux4 as rtmi_batch_get_state returned it.  On a
 * batch with the same parameters the run continues bit for bit (the reference has no counterpart: :866 runs to the end). */
int rtmi_batch_restore_state(rtmi_batch *b, const double *state9, const double *aux4, const int32_t *istep,
                             const uint8_t *alive);
/* Give every ray its own DELTA_S and max_size (host arrays [R], caller's ray order): one batch then holds the whole
 * DELTA_S calibration sweep, candidate x ray (search_delta over delta_s_options, RT_bench.py:950-958, 1317-1318).
 * max_size[k] <= params.max_size (which sizes the trajectory arrays).  Survives rtmi_batch_reset.  Only valid on a
 * fresh or reset batch (RTMI_ERR_STATE after rtmi_step / rtmi_run / rtmi_batch_set_state): changing the steps of rays
 * that are under way, or reviving rays that already left the box, has no counterpart in the reference. */
int rtmi_batch_set_per_ray(rtmi_batch *b, const double *step, const int32_t *max_size);
/* Back to row 0 with the same launch conditions: re-runs the initial conditions and zeroes the trajectory arrays
 * (caller-owned ext_s_ray / ext_n_ray included) unless params.lazy_clear is set, see there. */
int rtmi_batch_reset(rtmi_batch *b);
/* One launch that advances every live ray by at most nsteps DELTA_S steps (the body of the loop at :866-879;
 * nsteps = 1 is exactly one call of selected_func + store_update_results per ray). */
int rtmi_step(rtmi_batch *b, int32_t nsteps);
/* `count` times rtmi_step(b, nsteps), submitted as ONE hipGraph (a chain of `count` kernel nodes, built on first use and kept
 * while nsteps / count / the kernel stay the same): a host that advances the batch a few steps at a time pays one call and
 * one event pair per group instead of three runtime calls per launch.  Same results as the single calls. */
int rtmi_step_repeat(rtmi_batch *b, int32_t nsteps, int32_t count);
/* Run every ray to termination (:866-879 to exhaustion/break). */
int rtmi_run(rtmi_batch *b);
/* Block until the batch's stream is idle. */
int rtmi_sync(rtmi_batch *b);

/* d_ray[3][R] (:801, :888-890): expected arclength, simulated arclength, last written row i. Host buffer, fp64. */
int rtmi_read_d_ray(rtmi_batch *b, double *d_ray);
/* final[9][R] fp64: x, y, theta, n, dn/dx, dn/dy, p_x, p_y, T of each ray's last written row. */
int rtmi_read_final(rtmi_batch *b, double *final9);
/* Copy recorded rows [row0, row0+nrows) to host as fp64: s_ray[nrows][6][R] and/or n_ray[nrows][R] (may be NULL). */
int rtmi_read_rows(rtmi_batch *b, int64_t row0, int64_t nrows, double *s_ray, double *n_ray);

/* The reference's in-script physical checks, evaluated on the device from what the trace left in HBM
 * (SURVEY.md section 4).  out[R], host, fp64:
 *   SNELL_ERROR  |exit angle - Snell/reflection angle| in degrees per ray (RT_bench.py:896-919); needs record_stride 1
 *   CLOSURE      100*|(1,0) - s_ray[-1,0:2,k]|/(2 pi) per ray, the fisheye closure error (:956, :1393)
 *   PX_CV        100*std/mean of the recorded non-zero p_x per ray (:1354-1360); the reference averages rays 1..R-2 */
typedef enum { RTMI_METRIC_SNELL_ERROR = 1, RTMI_METRIC_CLOSURE = 2, RTMI_METRIC_PX_CV = 3 } rtmi_metric_kind;
int rtmi_metric(rtmi_batch *b, int kind, double *out);

/* Isochrone points: per-ray PCHIP interpolation (scipy PchipInterpolator's algorithm) of x, y, theta at the given
 * traveltimes, from the recorded T column -- the per-ray stage of the reference's wavefront extraction
 * (RT_bench.py:987-1003).  out[ntimes][3][R], host, fp64; NaN where a ray never reaches that traveltime
 * (the reference skips such rays, :997).  Needs record_stride 1. */
int rtmi_isochrones(rtmi_batch *b, int32_t ntimes, const double *times, double *out);

/* Wavefronts: the across-ray stage of the reference's wavefront extraction (RT_bench.py:1005-1026, 1043-1044).  For each
 * traveltime the isochrone points of the rays that reach it (as rtmi_isochrones) are sorted by y (np.argsort, :1016),
 * scipy's PchipInterpolator x(y) is built through them (:1020) and evaluated: its derivative at the points (:1021-1022),
 * the normal angle (:1025-1026), |ray angle - normal angle| (:1032) and the curve itself on nfine equally spaced y between the
 * first and the last point (:1043-1044; the reference uses 100).  All on the device; host outputs, fp64:
 *   count[ntimes]            points on each wavefront (rays that reach the traveltime)
 *   nodes[ntimes][7][R]      per sorted position j < count: y, x, ray angle, dx/dy, normal angle, |ray angle - normal angle|,
 *                            ray index (caller's order); NaN at j >= count, and in the derived columns when count < 2
 *   fine[ntimes][2][nfine]   x, y of the interpolated wavefront (NaN when count < 2); nfine = 0 skips it (fine may be NULL)
 * Needs record_stride 1.  Points with equal y make scipy raise; here they give NaN in the derived columns.
 * All `ntimes` wavefronts are made in ONE pass (one sort of every point by y, a stable regrouping per traveltime, one PCHIP
 * stage, one copy to the host): the 45 frames of the reference's animation (travel_time = 0.01 + 0.01 frame, :1066-1102) are
 * one call. */
int rtmi_wavefronts(rtmi_batch *b, int32_t ntimes, const double *times, int32_t nfine, int64_t *count, double *nodes,
                    double *fine);

typedef struct {
    void *s_ray, *n_ray;                 /* device, dtype, layouts above */
    double *x, *y, *theta;               /* device SoA ray state, length R: the accumulated quantities are fp64 in */
    void *n, *gx, *gy;                   /*   BOTH precisions (fp32 batches add fp32 increments onto fp64 sums); */
    double *dist_sim, *dist_real, *T;    /*   n and its gradient are of the batch's dtype */
    int32_t *istep;                      /* device, last written row per ray */
    const int32_t *perm;                 /* device [R] or NULL: with sort_rays, slot k of every array above holds the
                                            caller's ray perm[k] */
    int64_t R, rec_rows;
    int32_t dtype, record_stride;
} rtmi_device_view;
/* Raw device pointers for zero-copy consumers (torch / RCCL gather of the read-back). */
int rtmi_batch_view(rtmi_batch *b, rtmi_device_view *v);

#define RTMI_AUTO_SAMPLES 3   /* timed runs per schedule before RTMI_LAUNCH_AUTO settles (rtmi_params.launch_mode) */
typedef struct {
    uint64_t ray_steps;      /* sum over rays of the last written row (= sum of d_ray[2]): steps taken since create/reset */
    uint64_t live_rays;      /* rays that would still step */
    double kernel_ms;        /* sum of advance-kernel durations since create/reset (HIP events on the batch's stream) */
    uint32_t launches;       /* advance-kernel launches since create/reset */
    uint32_t vgprs, sgprs, lds_bytes;   /* of the advance kernel in use */
    uint32_t launch_mode_used;          /* rtmi_launch_mode of the last rtmi_run (RTMI_LAUNCH_PLAIN after rtmi_step) */
    double kernel_ms_total;             /* the same sum over the batch's whole life: rtmi_batch_reset does not clear it, so a
                                           caller that times many passes (reset + run each) gets the kernel time of all of
                                           them from two stats calls, one before and one after, without a host sync per pass */
    uint64_t launches_total;            /* advance-kernel launches since create */
    uint32_t auto_fallbacks;            /* RTMI_LAUNCH_AUTO only: time-sliced launches of this batch that gave up a bounded wait and
                                           were finished by the plain kernel (results unaffected).  Expected 0: a non-zero count is a
                                           scheduler defect signal, and the batch stays on the plain schedule afterwards */
    uint32_t auto_kept;                 /* RTMI_LAUNCH_AUTO: 0 while the batch is still exploring (or never had the choice: fewer bundles
                                           than resident blocks, per-ray steps), else the schedule it keeps: RTMI_LAUNCH_SLICED / _PLAIN */
    double auto_ms[2][RTMI_AUTO_SAMPLES];   /* the exploration record: kernel time of each timed run under [0] the time-sliced and
                                           [1] the plain schedule, in the order they were taken; auto_n[k] of them are valid */
    uint32_t auto_n[2];
    uint32_t retraced;                  /* critical rays re-traced in reference order since create/reset (rtmi_params.no_retrace) */
    uint32_t retrace_overflow;          /* ... and rays that qualified but found the hand-over queue full (1/128 of the batch, at least
                                           256): they stay in the fused form.  Expected 0 */
    uint64_t retraced_total;            /* re-traced over the batch's whole life */
    uint32_t dispatch_first;            /* the 256-ray bundle the plain kernel's first hardware block takes: 0, or -- learnt from the batch's
                                           first rtmi_run that handed critical rays over, kept for its re-runs like the AUTO schedule --
                                           the first of the bundles that held them (their re-trace then starts with the kernel) */
    uint32_t reserved_;
} rtmi_stats;
/* Synchronises the stream, then fills *s. */
int rtmi_batch_stats(rtmi_batch *b, rtmi_stats *s);
void rtmi_batch_destroy(rtmi_batch *b);

/* ---------------------------------------------------------------- one call's rays over the GPUs of a node
 * The reference's outer loop over rays (RT_bench.py:807) carries nothing from ray to ray, and the only parallelism it has is
 * a pool of replica processes that pickle whole results back (:1317-1318, :1521-1523).  rtmi_shard is one trazar() call whose
 * rays are dealt to `ndev` devices round-robin (ray k -> devices[k % ndev]: every device gets the same mix of short and long
 * rays); each device builds the field itself (rtmi_field_build) and runs its rays (rtmi_batch_create / rtmi_run): no
 * collective on the data path.  One host thread drives every GPU (the library starts one worker per device for the runs).
 * The read-back functions gather to devices[0] DEVICE TO DEVICE -- RCCL's ncclGather over xGMI between the communicators
 * of ncclCommInitAll, or peer copies -- and answer in the caller's ray order; results are the bits of an unsharded batch.
 * Every rtmi_shard_* call leaves the calling thread's current HIP device as it found it.
 * (One process per GPU is the other way to spread a call: raytracing_amd/dist.py over torch.distributed.) */
typedef struct rtmi_shard rtmi_shard;
typedef enum {
    RTMI_SHARD_AUTO = 0,   /* RCCL when librccl.so.1 loads and the devices are distinct, else copies */
    RTMI_SHARD_RCCL = 1,   /* ncclCommInitAll + ncclGather; an error if that is not possible */
    RTMI_SHARD_COPY = 2    /* hipMemcpyPeerAsync into devices[0] (also: the same device listed several times, to rehearse an
                              N-way split on one GPU) */
} rtmi_shard_transport;
/* Arguments as rtmi_field_build + rtmi_batch_create; x0 / y0 / theta0 [R] are the whole call's launch conditions.
 * p->ext_s_ray / ext_n_ray must be NULL.  R >= ndev. */
int rtmi_shard_create(int scenario, double xi, double xs, double yi, double ys, double delta, const rtmi_params *p, int64_t R,
                      const double *x0, const double *y0, const double *theta0, const int *devices, int ndev, int transport,
                      rtmi_shard **out);
/* rtmi_run / rtmi_batch_reset on every device at once; returns when all have finished. */
int rtmi_shard_run(rtmi_shard *s);
int rtmi_shard_reset(rtmi_shard *s);
/* d_ray[3][R], final[9][R] as rtmi_read_d_ray / rtmi_read_final, of the whole call (host, fp64). */
int rtmi_shard_read_d_ray(rtmi_shard *s, double *d_ray);
int rtmi_shard_read_final(rtmi_shard *s, double *final9);
/* Recorded rows row0, row0 + every, ... (nrows of them) of every ray, gathered on devices[0]: *rows_dev = DEVICE pointer to
 * fp64 [nrows][6][R], valid until the next rtmi_shard_* call on s (for consumers that stay on the GPU); rtmi_shard_read_rows
 * copies the same to the host. */
int rtmi_shard_gather_rows(rtmi_shard *s, int64_t row0, int64_t nrows, int64_t every, double **rows_dev);
int rtmi_shard_read_rows(rtmi_shard *s, int64_t row0, int64_t nrows, int64_t every, double *s_ray);
typedef struct {
    int32_t ndev, transport;     /* transport in use: RTMI_SHARD_RCCL or RTMI_SHARD_COPY */
    int64_t R;
    uint64_t ray_steps, live_rays;   /* summed over the devices */
    double run_seconds;          /* wall time of the last rtmi_shard_run (all devices, host clock) */
    double kernel_ms_max;        /* the slowest device's advance-kernel time since its last reset */
    uint32_t auto_fallbacks, reserved_;
} rtmi_shard_stats;
int rtmi_shard_info(rtmi_shard *s, rtmi_shard_stats *st);
/* Shard i's batch (owned by s) and its device: every rtmi_batch function applies (make *device current first). */
int rtmi_shard_batch(rtmi_shard *s, int i, rtmi_batch **b, int *device);
void rtmi_shard_destroy(rtmi_shard *s);

/* Diagnostic: the library's libm-identical fp64 sin and cos (the functions op3/4/5/9/10/11 step with; they reproduce
 * glibc 2.35's sin()/cos(), i.e. numpy's np.sin/np.cos, bit for bit for |x| < 105414350) evaluated on the device
 * for n host values.  s[n], c[n]: host, fp64. */
int rtmi_debug_sincos(int64_t n, const double *x, double *s, double *c);
/* Diagnostic: the lookup the fast-form step methods (op1/2/6/7/8, every fp32 batch) make -- the grid cell's polynomial
 * (one per cell, converted from FITPACK's splines on the true knots at field build) -- for npts host points
 * -> n, dn/dx, dn/dy (host, fp64).  Within 1e-15 of the field's scale of rtmi_field_eval (FITPACK's own arithmetic,
 * n_gradient :141-156) in every cell; tests compare it bit for bit with the host restatement of the same table. */
int rtmi_debug_field_lookup(const rtmi_field *f, int64_t npts, const double *x, const double *y, double *n,
                            double *gx, double *gy);
/* Diagnostic (host only, no device needed): RTMI_LAUNCH_AUTO's rule on a recorded sequence.  Given the kernel times taken so far
 * under the time-sliced (sliced_ms[ns]) and the plain (plain_ms[np]) schedule, ns, np <= RTMI_AUTO_SAMPLES: *next = the schedule
 * the next exploration run takes (RTMI_LAUNCH_SLICED / RTMI_LAUNCH_PLAIN; -1 once exploration is over), *decision = the schedule
 * that would be kept on these samples (medians; the plain launch only if more than 3 % ahead).  Either pointer may be NULL. */
int rtmi_debug_auto_rule(const double *sliced_ms, int ns, const double *plain_ms, int np, int *next, int *decision);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
