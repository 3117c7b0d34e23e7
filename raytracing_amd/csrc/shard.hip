// shard.hip -- one trazar() call's rays split over the GPUs of a node behind the C ABI (include/rtmi.h, rtmi_shard_*).
//
// The reference's outer loop over rays (RT_bench.py:807) carries nothing from one ray to the next, so the rays of one call are
// dealt to the devices round-robin (ray k -> devices[k % ndev], slot k / ndev: every device gets the same mix of short and long
// rays) and each device runs the whole path -- genZ + interpolacion (rtmi_field_build), trazar's preamble (rtmi_batch_create),
// the loop (rtmi_run) -- on its own rays.  No data-path collective.  The read-back gathers the per-device results to
// devices[0] DEVICE TO DEVICE: RCCL's ncclGather over xGMI between communicators made by ncclCommInitAll -- one host thread
// drives every GPU -- or, where RCCL cannot be used (the library is not installed, or the same device is
// listed twice to rehearse the N-way split on one GPU), peer copies; an interleave kernel on devices[0] puts the rays back in
// the caller's order.  The reference's counterpart is its replica fan-out (RT_bench.py:1317-1318, 1521-1523: a
// ProcessPoolExecutor that pickles whole results back).
//
// librccl is opened at run time (dlopen "librccl.so.1"): librtmi.so carries no link-time dependency on it, and a process that
// already holds a copy (torch's) shares it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_internal.h"

#define RTMI_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

int fail(int code, const std::string& msg) { return rtmi_internal_fail(code, msg.c_str()); }
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

// Every rtmi_shard_* entry walks over the shard's devices with hipSetDevice; the caller's current device is put back on every
// exit path (the rest of the ABI rejects handles whose field lives on another device, and a caller's torch tensors would land
// on whichever device the last call happened to leave current).
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---- the few RCCL entry points used, resolved at run time (rccl.h: ncclCommInitAll :236, ncclGather :745)
typedef void* nccl_comm;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(nccl_comm*, int, const int*) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return lib && CommInitAll && CommDestroy && GroupStart && GroupEnd && Gather && GetErrorString; }
};
constexpr int kNcclFloat64 = 8;       // ncclDataType_t ncclFloat64 (rccl.h:467)

Rccl& rccl() {
    static Rccl R = [] {
        Rccl r;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
            r.Gather = (decltype(r.Gather))dlsym(r.lib, "ncclGather");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        }
        return r;
    }();
    return R;
}

// rays of shard i of an interleaved N-way split of R rays
int64_t shard_rays(int64_t R, int i, int n) { return (R - i + n - 1) / n; }

// recorded rows [row0 + j*every] (j < nrows) of a batch, from its device view: fp64, the shard's ray order -> dst[nrows*6][Rloc]
template <typename T>
__global__ void k_rows_pack(const T* s_ray, const int* perm, long Rloc, long row0, long every, long nvec, double* dst) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Rloc) return;
    const long o = perm ? (long)perm[k] : k;
    for (long v = blockIdx.y; v < nvec; v += gridDim.y) {
        const long row = row0 + (v / 6) * every, q = v % 6;
        dst[(size_t)v * Rloc + o] = (double)s_ray[((size_t)row * 6 + q) * Rloc + k];
    }
}
// gathered[i][v][slot] (block i: nvec x Rmax doubles, its first nvec x Rloc_i packed [v][Rloc_i]) -> out[v][slot*ndev + i]
__global__ void k_interleave(const double* gathered, long R, int ndev, long nvec, long block_elems, double* out) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int i = (int)(r % ndev);
    const long slot = r / ndev, Rloc = (R - i + ndev - 1) / ndev;
    const double* src = gathered + (size_t)i * block_elems;
    for (long v = blockIdx.y; v < nvec; v += gridDim.y) out[(size_t)v * R + r] = src[(size_t)v * Rloc + slot];
}

}  // namespace

// One worker thread per device for the life of the shard (started by rtmi_shard_create when ndev > 1): it makes its device
// current once and then runs the jobs rtmi_shard_run / rtmi_shard_reset hand it.
struct ShardWorkers {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<int(rtmi_batch*)> job;
    uint64_t generation = 0;      // bumped per job
    int pending = 0;              // workers that have not finished the present job
    bool stop = false;
    std::vector<std::thread> threads;
    std::vector<int> rcs;
    std::vector<std::string> msgs;
};

struct rtmi_shard {
    int ndev = 0;
    int64_t R = 0, Rmax = 0;
    rtmi_params p{};
    std::vector<int> devices;
    std::vector<rtmi_field*> fields;
    std::vector<rtmi_batch*> batches;
    std::vector<hipStream_t> streams;        // one per shard, for the read-back
    std::vector<double*> send;               // per shard, on its device: [nvec][Rloc] packed, room for nvec x Rmax
    std::vector<size_t> send_elems;
    double* recv = nullptr;                  // devices[0]: [ndev] blocks of nvec x Rmax
    double* out = nullptr;                   // devices[0]: [nvec][R], the caller's ray order
    size_t recv_elems = 0, out_elems = 0;
    std::vector<nccl_comm> comms;            // RCCL communicators (transport 1)
    int transport = RTMI_SHARD_COPY;
    double run_seconds = 0;                  // wall time of the last rtmi_shard_run
    ShardWorkers* workers = nullptr;         // ndev > 1: one thread per device, kept until rtmi_shard_destroy
};

namespace {

int grow(double** p, size_t* have, size_t want) {
    if (want <= *have) return RTMI_OK;
    (void)hipFree(*p);
    *p = nullptr; *have = 0;
    const hipError_t e = hipMalloc(p, want * sizeof(double));
    if (e != hipSuccess) return fail(RTMI_ERR_ALLOC, std::string("shard staging: ") + hipGetErrorString(e));
    *have = want;
    return RTMI_OK;
}

// `pack(i, dst)` leaves shard i's [nvec][Rloc_i] fp64 block in dst (on device i, enqueued on streams[i]); the blocks are
// gathered to devices[0] and interleaved there: s->out = [nvec][R] in the caller's ray order.
template <typename Pack> int gather_vecs(rtmi_shard* s, long nvec, Pack pack) {
    const size_t block = (size_t)nvec * (size_t)s->Rmax;
    for (int i = 0; i < s->ndev; i++) {
        HIP_TRY(hipSetDevice(s->devices[i]));
        int rc = grow(&s->send[i], &s->send_elems[i], block);
        if (rc) return rc;
        rc = pack(i, s->send[i]);
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(s->devices[0]));
    int rc = grow(&s->recv, &s->recv_elems, block * (size_t)s->ndev);
    if (rc) return rc;
    rc = grow(&s->out, &s->out_elems, (size_t)nvec * (size_t)s->R);
    if (rc) return rc;
    if (s->transport == RTMI_SHARD_RCCL) {
        Rccl& N = rccl();
        // nothing returns between GroupStart and GroupEnd: a group left open would swallow every later RCCL call of the thread
        int nrc = N.GroupStart();
        hipError_t he = hipSuccess;
        if (nrc == 0) {
            for (int i = 0; i < s->ndev && nrc == 0 && he == hipSuccess; i++) {
                he = hipSetDevice(s->devices[i]);
                if (he == hipSuccess) nrc = N.Gather(s->send[i], s->recv, block, kNcclFloat64, 0, s->comms[i], s->streams[i]);
            }
            const int erc = N.GroupEnd();
            if (nrc == 0) nrc = erc;
        }
        if (he != hipSuccess) return fail(RTMI_ERR_HIP, std::string("ncclGather: hipSetDevice: ") + hipGetErrorString(he));
        if (nrc != 0) return fail(RTMI_ERR_HIP, std::string("ncclGather: ") + N.GetErrorString(nrc));
    } else {
        for (int i = 0; i < s->ndev; i++) {
            HIP_TRY(hipSetDevice(s->devices[i]));
            double* dst = s->recv + (size_t)i * block;
            if (s->devices[i] == s->devices[0]) HIP_TRY(hipMemcpyAsync(dst, s->send[i], block * sizeof(double), hipMemcpyDeviceToDevice, s->streams[i]));
            else HIP_TRY(hipMemcpyPeerAsync(dst, s->devices[0], s->send[i], s->devices[i], block * sizeof(double), s->streams[i]));
        }
    }
    for (int i = 0; i < s->ndev; i++) {
        HIP_TRY(hipSetDevice(s->devices[i]));
        HIP_TRY(hipStreamSynchronize(s->streams[i]));
    }
    HIP_TRY(hipSetDevice(s->devices[0]));
    const dim3 g((unsigned)((s->R + 255) / 256), (unsigned)(nvec < 1024 ? nvec : 1024)), blk(256);
    hipLaunchKernelGGL(k_interleave, g, blk, 0, s->streams[0], s->recv, (long)s->R, s->ndev, nvec, (long)block, s->out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->streams[0]));
    return RTMI_OK;
}

int to_host(rtmi_shard* s, double* host, size_t n) {
    HIP_TRY(hipSetDevice(s->devices[0]));
    HIP_TRY(hipMemcpy(host, s->out, n * sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

}  // namespace

static int start_workers(rtmi_shard* s);
static void stop_workers(rtmi_shard* s);

RTMI_EXPORT void rtmi_shard_destroy(rtmi_shard* s) {
    if (!s) return;
    DeviceGuard guard;
    stop_workers(s);
    for (int i = 0; i < (int)s->batches.size(); i++) {
        if ((size_t)i < s->devices.size()) (void)hipSetDevice(s->devices[i]);
        rtmi_batch_destroy(s->batches[i]);
    }
    for (int i = 0; i < (int)s->fields.size(); i++) {
        if ((size_t)i < s->devices.size()) (void)hipSetDevice(s->devices[i]);
        rtmi_field_destroy(s->fields[i]);
    }
    for (int i = 0; i < (int)s->send.size(); i++) {
        (void)hipSetDevice(s->devices[i]);
        (void)hipFree(s->send[i]);
        if ((size_t)i < s->streams.size() && s->streams[i]) (void)hipStreamDestroy(s->streams[i]);
    }
    if (!s->devices.empty()) (void)hipSetDevice(s->devices[0]);
    (void)hipFree(s->recv);
    (void)hipFree(s->out);
    if (s->transport == RTMI_SHARD_RCCL)
        for (nccl_comm c : s->comms)
            if (c) (void)rccl().CommDestroy(c);
    delete s;
}

RTMI_EXPORT int rtmi_shard_create(int scenario, double xi, double xs, double yi, double ys, double delta, const rtmi_params* p,
                                  int64_t R, const double* x0, const double* y0, const double* theta0, const int* devices, int ndev,
                                  int transport, rtmi_shard** out) {
    ARG_TRY(p && x0 && y0 && theta0 && devices && out, "rtmi_shard_create: null argument");
    ARG_TRY(ndev >= 1 && ndev <= 64, "rtmi_shard_create: ndev must be in [1, 64]");
    ARG_TRY(R >= ndev, "rtmi_shard_create: fewer rays than devices");
    ARG_TRY(transport >= RTMI_SHARD_AUTO && transport <= RTMI_SHARD_COPY, "rtmi_shard_create: transport must be an rtmi_shard_transport");
    ARG_TRY(!p->ext_s_ray && !p->ext_n_ray, "rtmi_shard_create: caller-owned trajectory buffers belong to one device (use rtmi_batch_create)");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    bool distinct = true;
    for (int i = 0; i < ndev; i++) {
        ARG_TRY(devices[i] >= 0 && devices[i] < count, "rtmi_shard_create: device index out of range");
        for (int j = 0; j < i; j++) distinct = distinct && devices[j] != devices[i];
    }
    ARG_TRY(!(transport == RTMI_SHARD_RCCL && !distinct), "rtmi_shard_create: RCCL needs one distinct GPU per shard");
    DeviceGuard guard;
    rtmi_shard* s = new (std::nothrow) rtmi_shard();
    if (!s) return fail(RTMI_ERR_ALLOC, "rtmi_shard_create: host allocation failed");
    int rc = RTMI_OK;
    auto body = [&]() -> int {
        s->ndev = ndev; s->R = R; s->Rmax = shard_rays(R, 0, ndev); s->p = *p;
        s->devices.assign(devices, devices + ndev);
        s->send.assign(ndev, nullptr); s->send_elems.assign(ndev, 0); s->streams.assign(ndev, nullptr);
        std::vector<double> sx, sy, st;
        for (int i = 0; i < ndev; i++) {
            const int64_t Rl = shard_rays(R, i, ndev);
            sx.resize(Rl); sy.resize(Rl); st.resize(Rl);
            for (int64_t k = 0; k < Rl; k++) { sx[k] = x0[k * ndev + i]; sy[k] = y0[k * ndev + i]; st[k] = theta0[k * ndev + i]; }
            HIP_TRY(hipSetDevice(devices[i]));
            HIP_TRY(hipStreamCreateWithFlags(&s->streams[i], hipStreamNonBlocking));
            rtmi_field* f = nullptr;
            int r = rtmi_field_build(scenario, xi, xs, yi, ys, delta, p->dtype, nullptr, &f);   // each device builds its own (<= 26 MB, a few ms)
            if (r) return r;
            s->fields.push_back(f);
            rtmi_batch* b = nullptr;
            r = rtmi_batch_create(f, p, Rl, sx.data(), sy.data(), st.data(), nullptr, &b);
            if (r) return r;
            s->batches.push_back(b);
        }
        // transport of the read-back gather
        s->transport = RTMI_SHARD_COPY;
        if (transport != RTMI_SHARD_COPY && distinct) {
            Rccl& N = rccl();
            if (N.ok()) {
                s->comms.assign(ndev, nullptr);
                const int nrc = N.CommInitAll(s->comms.data(), ndev, devices);
                if (nrc == 0) s->transport = RTMI_SHARD_RCCL;
                else if (transport == RTMI_SHARD_RCCL) return fail(RTMI_ERR_HIP, std::string("ncclCommInitAll: ") + N.GetErrorString(nrc));
                else s->comms.clear();
            } else if (transport == RTMI_SHARD_RCCL) {
                return fail(RTMI_ERR_UNSUPPORTED, "rtmi_shard_create: librccl.so.1 could not be loaded");
            }
        }
        if (s->transport == RTMI_SHARD_COPY) {
            for (int i = 1; i < ndev; i++) {      // peer access for the copies (already enabled / same device: fine)
                if (devices[i] == devices[0]) continue;
                HIP_TRY(hipSetDevice(devices[i]));
                const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                // "already enabled" (a second shard on these devices, or torch got there first) and "not possible" (the runtime
                // then stages the copy) are both fine -- but either leaves the thread's last error set, and the next
                // hipGetLastError() after a kernel launch on this thread would report it: cleared in every case
                if (e != hipSuccess) (void)hipGetLastError();
            }
        }
        return start_workers(s);
    };
    try {
        rc = body();
    } catch (const std::exception& e) {
        rc = fail(RTMI_ERR_ALLOC, std::string("rtmi_shard_create: ") + e.what());
    }
    if (rc) {
        const std::string msg = rtmi_last_error();      // rtmi_shard_destroy's calls must not overwrite it
        rtmi_shard_destroy(s);
        return fail(rc, msg);
    }
    *out = s;
    return RTMI_OK;
}

// `fn(batch)` on every shard at once, each on its device's worker thread (started once, in rtmi_shard_create); returns when
// every shard has finished.  One shard: the calling thread itself.
static void worker_main(rtmi_shard* s, int i) {
    ShardWorkers& W = *s->workers;
    const hipError_t e0 = hipSetDevice(s->devices[i]);       // once: the thread keeps its device
    uint64_t seen = 0;
    for (;;) {
        std::function<int(rtmi_batch*)> job;
        {
            std::unique_lock<std::mutex> lk(W.mu);
            W.cv_job.wait(lk, [&] { return W.stop || W.generation != seen; });
            if (W.stop) return;
            seen = W.generation;
            job = W.job;
        }
        int rc = RTMI_OK;
        std::string msg;
        if (e0 != hipSuccess) { rc = RTMI_ERR_HIP; msg = hipGetErrorString(e0); }
        else {
            rc = job(s->batches[i]);
            if (rc) msg = rtmi_last_error();                  // thread-local: copied out here
        }
        {
            std::lock_guard<std::mutex> lk(W.mu);
            W.rcs[i] = rc; W.msgs[i] = msg;
            if (--W.pending == 0) W.cv_done.notify_all();
        }
    }
}
static int start_workers(rtmi_shard* s) {
    if (s->ndev <= 1) return RTMI_OK;
    s->workers = new (std::nothrow) ShardWorkers();
    if (!s->workers) return fail(RTMI_ERR_ALLOC, "rtmi_shard_create: host allocation failed");
    s->workers->rcs.assign(s->ndev, RTMI_OK);
    s->workers->msgs.assign(s->ndev, std::string());
    for (int i = 0; i < s->ndev; i++) s->workers->threads.emplace_back(worker_main, s, i);     // may throw: caught by the caller
    return RTMI_OK;
}
static void stop_workers(rtmi_shard* s) {
    if (!s->workers) return;
    {
        std::lock_guard<std::mutex> lk(s->workers->mu);
        s->workers->stop = true;
    }
    s->workers->cv_job.notify_all();
    for (auto& t : s->workers->threads) t.join();
    delete s->workers;
    s->workers = nullptr;
}
template <typename Fn> static int on_every_device(rtmi_shard* s, const char* who, Fn fn) {
    std::vector<int> rcs(s->ndev, RTMI_OK);
    std::vector<std::string> msgs(s->ndev);
    if (s->ndev == 1 || !s->workers) {
        for (int i = 0; i < s->ndev; i++) {
            const hipError_t e = hipSetDevice(s->devices[i]);
            if (e != hipSuccess) { rcs[i] = RTMI_ERR_HIP; msgs[i] = hipGetErrorString(e); continue; }
            rcs[i] = fn(s->batches[i]);
            if (rcs[i]) msgs[i] = rtmi_last_error();
        }
    } else {
        ShardWorkers& W = *s->workers;
        std::unique_lock<std::mutex> lk(W.mu);
        W.job = fn;
        W.pending = s->ndev;
        W.generation++;
        W.cv_job.notify_all();
        W.cv_done.wait(lk, [&] { return W.pending == 0; });
        rcs = W.rcs; msgs = W.msgs;
    }
    for (int i = 0; i < s->ndev; i++)
        if (rcs[i]) return fail(rcs[i], std::string(who) + ": shard " + std::to_string(i) + " (device " + std::to_string(s->devices[i]) + "): " + msgs[i]);
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_shard_run(rtmi_shard* s) {
    ARG_TRY(s, "rtmi_shard_run: null");
    DeviceGuard guard;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = on_every_device(s, "rtmi_shard_run", [](rtmi_batch* b) { return rtmi_run(b); });
    s->run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

RTMI_EXPORT int rtmi_shard_reset(rtmi_shard* s) {
    ARG_TRY(s, "rtmi_shard_reset: null");
    DeviceGuard guard;
    return on_every_device(s, "rtmi_shard_reset", [](rtmi_batch* b) { return rtmi_batch_reset(b); });
}

RTMI_EXPORT int rtmi_shard_read_d_ray(rtmi_shard* s, double* d_ray) {
    ARG_TRY(s && d_ray, "rtmi_shard_read_d_ray: null");
    DeviceGuard guard;
    const int rc = gather_vecs(s, 3, [&](int i, double* dst) { return rtmi_internal_pack_device(s->batches[i], 0, dst, s->streams[i]); });
    return rc ? rc : to_host(s, d_ray, 3 * (size_t)s->R);
}

RTMI_EXPORT int rtmi_shard_read_final(rtmi_shard* s, double* final9) {
    ARG_TRY(s && final9, "rtmi_shard_read_final: null");
    DeviceGuard guard;
    const int rc = gather_vecs(s, 9, [&](int i, double* dst) { return rtmi_internal_pack_device(s->batches[i], 1, dst, s->streams[i]); });
    return rc ? rc : to_host(s, final9, 9 * (size_t)s->R);
}

RTMI_EXPORT int rtmi_shard_gather_rows(rtmi_shard* s, int64_t row0, int64_t nrows, int64_t every, double** rows_dev) {
    ARG_TRY(s && rows_dev, "rtmi_shard_gather_rows: null");
    ARG_TRY(s->p.record_stride > 0, "rtmi_shard_gather_rows: the batches keep no trajectory (record_stride = 0)");
    ARG_TRY(every >= 1 && nrows >= 1 && row0 >= 0, "rtmi_shard_gather_rows: need row0 >= 0, nrows >= 1, every >= 1");
    DeviceGuard guard;
    rtmi_device_view v0;
    int rc = rtmi_batch_view(s->batches[0], &v0);
    if (rc) return rc;
    ARG_TRY(row0 + (nrows - 1) * every < v0.rec_rows, "rtmi_shard_gather_rows: row range outside rec_rows");
    const long nvec = (long)nrows * 6;
    rc = gather_vecs(s, nvec, [&](int i, double* dst) -> int {
        rtmi_device_view v;
        int r = rtmi_batch_view(s->batches[i], &v);
        if (r) return r;
        r = rtmi_sync(s->batches[i]);                      // the rows are complete (rtmi_run is synchronous; rtmi_step is not)
        if (r) return r;
        const dim3 g((unsigned)((v.R + 255) / 256), (unsigned)(nvec < 1024 ? nvec : 1024)), blk(256);
        if (v.dtype == RTMI_F64)
            hipLaunchKernelGGL(k_rows_pack<double>, g, blk, 0, s->streams[i], (const double*)v.s_ray, v.perm, (long)v.R, (long)row0, (long)every, nvec, dst);
        else
            hipLaunchKernelGGL(k_rows_pack<float>, g, blk, 0, s->streams[i], (const float*)v.s_ray, v.perm, (long)v.R, (long)row0, (long)every, nvec, dst);
        HIP_TRY(hipGetLastError());
        return RTMI_OK;
    });
    if (rc) return rc;
    *rows_dev = s->out;
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_shard_read_rows(rtmi_shard* s, int64_t row0, int64_t nrows, int64_t every, double* s_ray) {
    ARG_TRY(s_ray, "rtmi_shard_read_rows: null");
    DeviceGuard guard;
    double* dev = nullptr;
    const int rc = rtmi_shard_gather_rows(s, row0, nrows, every, &dev);
    return rc ? rc : to_host(s, s_ray, (size_t)nrows * 6 * (size_t)s->R);
}

RTMI_EXPORT int rtmi_shard_info(rtmi_shard* s, rtmi_shard_stats* st) {
    ARG_TRY(s && st, "rtmi_shard_info: null");
    DeviceGuard guard;
    memset(st, 0, sizeof *st);
    st->ndev = s->ndev; st->transport = s->transport; st->run_seconds = s->run_seconds; st->R = s->R;
    for (int i = 0; i < s->ndev; i++) {
        HIP_TRY(hipSetDevice(s->devices[i]));
        rtmi_stats b;
        const int rc = rtmi_batch_stats(s->batches[i], &b);
        if (rc) return rc;
        st->ray_steps += b.ray_steps; st->live_rays += b.live_rays;
        st->kernel_ms_max = b.kernel_ms > st->kernel_ms_max ? b.kernel_ms : st->kernel_ms_max;
        st->auto_fallbacks += b.auto_fallbacks;
    }
    return RTMI_OK;
}

RTMI_EXPORT int rtmi_shard_batch(rtmi_shard* s, int i, rtmi_batch** b, int* device) {
    ARG_TRY(s && b, "rtmi_shard_batch: null");
    ARG_TRY(i >= 0 && i < s->ndev, "rtmi_shard_batch: shard index out of range");
    *b = s->batches[i];
    if (device) *device = s->devices[i];
    return RTMI_OK;
}
