// cmf_comm.hip.h -- the collectives of the sharded solvers inside the C ABI (included by cmf_api.hip).
//
// "one stream + one RCCL communicator per device"; 8(e): one rank per GPU, the V update needs ONE all-reduce of the
// (d + k) x k partial buffer per iteration (pycmf/cmf_solvers.py:242-246 summed over row blocks), the row-sharded Newton three
// in-place all-gathers of factor rows.  RCCL is loaded at run time (dlopen of librccl.so.1, RTLD_LOCAL): libcmfhip.so has no
// link-time dependency on it, a single-GPU process never touches it, and a process that also holds PyTorch's own copy of RCCL sees
// no symbol clash.  Every collective is enqueued on the context's stream: ordered behind the kernels that produce its buffer and
// in front of the ones that consume it, no host synchronisation.  The unique id travels out of band (the launcher's file / pipe:
// pycmf_amd/comm.py).
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return CMF_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(CMF_ERCCL, "librccl.so.1 not found (%s): multi-GPU runs need RCCL", dlerror());
    RcclApi a;
    a.handle = h;
#define CMF_RCCL_SYM(field, sym)                                                                    \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym));                                   \
    if (!a.field) { dlclose(h); return fail(CMF_ERCCL, "librccl: missing symbol %s", sym); }
    CMF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    CMF_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    CMF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    CMF_RCCL_SYM(AllReduce, "ncclAllReduce")
    CMF_RCCL_SYM(AllGather, "ncclAllGather")
    CMF_RCCL_SYM(ReduceScatter, "ncclReduceScatter")
    CMF_RCCL_SYM(CommCount, "ncclCommCount")
    CMF_RCCL_SYM(CommUserRank, "ncclCommUserRank")
    CMF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    CMF_RCCL_SYM(GroupStart, "ncclGroupStart")
    CMF_RCCL_SYM(GroupEnd, "ncclGroupEnd")
#undef CMF_RCCL_SYM
    g_rccl = a;
    return CMF_OK;
}
#define RCCLCHK(expr)                                                                                             \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(CMF_ERCCL, "%s: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
    } while (0)

struct CmfComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool timed = false;
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> events;     // around every collective while `timed`
    int64_t calls = 0, bytes = 0;
    int64_t kcalls[CMF_COMM_KINDS] = {0}, kbytes[CMF_COMM_KINDS] = {0}; // per kind (CMF_COMM_ALLREDUCE_F32 ...)
    double *dscratch = nullptr; // 16 doubles on the device for the host-value reductions
    // background collectives (cmf_comm_allreduce_f32_bg): a side stream ordered behind / in front of the context's stream by events
    hipStream_t side = nullptr;
    hipEvent_t ev_ready[8] = {nullptr}, ev_done = nullptr;
    int ev_next = 0;
    bool bg_pending = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> joins; // on the context's stream around every join: the EXPOSED wait
    // cmf_comm_group_start .. _end: the collectives in between are handed to RCCL as ONE group (one launch point on the stream);
    // while timed, ONE event pair around the group (events recorded inside would bracket nothing: RCCL enqueues at ncclGroupEnd)
    bool in_group = false;
    hipEvent_t grp_a = nullptr, grp_b = nullptr;
    int64_t launch_points = 0;  // collectives outside groups + groups
};

extern "C" int cmf_comm_unique_id(char *id128) {
    if (!id128) return fail(CMF_EINVAL, "null id buffer");
    CHK(rccl_load());
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == CMF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return CMF_OK;
}

extern "C" int cmf_comm_init(cmf_ctx *c, int rank, int world, const char *id128) {
    if (!c || !id128) return fail(CMF_EINVAL, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(CMF_EINVAL, "rank %d out of range for %d ranks", rank, world);
    if (c->comm) return fail(CMF_EINVAL, "the context already holds a communicator");
    CHK(rccl_load());
    DeviceGuard dg(c->device);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    CmfComm *cm = new CmfComm();
    cm->rank = rank; cm->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&cm->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete cm;
        return fail(CMF_ERCCL, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, c->device, g_rccl.GetErrorString(r));
    }
    // RCCL prints a version banner through C stdio on its first communicator; on a pipe that text would sit in the stdio buffer until
    // the process exits and land BEHIND whatever the host program prints through its own buffers (bench.py's one JSON line): out now
    (void)fflush(stdout);
    (void)fflush(stderr);
    if (hipMalloc((void **)&cm->dscratch, 16 * sizeof(double)) != hipSuccess) {
        (void)g_rccl.CommDestroy(cm->comm);
        delete cm;
        return fail(CMF_ENOMEM, "out of device memory");
    }
    c->comm = cm;
    return CMF_OK;
}

extern "C" int cmf_comm_destroy(cmf_ctx *c) {
    if (!c || !c->comm) return CMF_OK;
    DeviceGuard dg(c->device);
    CmfComm *cm = c->comm;
    (void)hipStreamSynchronize(c->stream);
    if (cm->side) (void)hipStreamSynchronize(cm->side);
    for (auto &e : cm->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : cm->joins) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : cm->ev_ready) if (e) (void)hipEventDestroy(e);
    if (cm->ev_done) (void)hipEventDestroy(cm->ev_done);
    if (cm->side) (void)hipStreamDestroy(cm->side);
    if (cm->dscratch) (void)hipFree(cm->dscratch);
    if (cm->comm) (void)g_rccl.CommDestroy(cm->comm);
    delete cm;
    c->comm = nullptr;
    return CMF_OK;
}

struct CommTimed { // events on the collective's stream around it (bench.py: bytes and ms per iteration)
    cmf_ctx *c; CmfComm *cm; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; int kind;
    CommTimed(cmf_ctx *c_, CmfComm *cm_, int64_t nbytes, int kind_, hipStream_t st_ = nullptr) : c(c_), cm(cm_), st(st_ ? st_ : c_->stream), kind(kind_) {
        cm->calls += 1; cm->bytes += nbytes;
        cm->kcalls[kind] += 1; cm->kbytes[kind] += nbytes;
        if (!cm->in_group) cm->launch_points += 1;
        if (!cm->in_group && cm->timed && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, st);
        else a = b = nullptr;
    }
    ~CommTimed() {
        if (a && b) { (void)hipEventRecord(b, st); cm->events.push_back({a, b, kind}); }
    }
};
#define NEED_COMM(c)                                                                                  \
    do {                                                                                              \
        if (!(c)) return fail(CMF_EINVAL, "null context");                                            \
        if (!(c)->comm) return fail(CMF_EINVAL, "cmf_comm_init has not been called on this context"); \
    } while (0)

// The collectives enqueued between cmf_comm_group_start and cmf_comm_group_end go to RCCL as one group (ncclGroupStart / ncclGroupEnd):
// one launch point on the context's stream instead of one per call -- the row-blocked MU protocol pairs its k_pad^2 all-reduce of
// U^T U + Z^T Z with the reduce-scatter of the numerator, and the k_pad^2 all-reduce of V^T V with the all-gather of V.
extern "C" int cmf_comm_group_start(cmf_ctx *c) {
    NEED_COMM(c);
    CmfComm *cm = c->comm;
    if (cm->in_group) return fail(CMF_EINVAL, "cmf_comm_group_start: a group is already open");
    DeviceGuard dg(c->device);
    cm->grp_a = cm->grp_b = nullptr;
    if (cm->timed && hipEventCreate(&cm->grp_a) == hipSuccess && hipEventCreate(&cm->grp_b) == hipSuccess) (void)hipEventRecord(cm->grp_a, c->stream);
    else cm->grp_a = cm->grp_b = nullptr;
    RCCLCHK(g_rccl.GroupStart());
    cm->in_group = true;
    cm->launch_points += 1;
    return CMF_OK;
}
extern "C" int cmf_comm_group_end(cmf_ctx *c) {
    NEED_COMM(c);
    CmfComm *cm = c->comm;
    if (!cm->in_group) return fail(CMF_EINVAL, "cmf_comm_group_end without cmf_comm_group_start");
    DeviceGuard dg(c->device);
    cm->in_group = false;
    RCCLCHK(g_rccl.GroupEnd());
    if (cm->grp_a && cm->grp_b) { (void)hipEventRecord(cm->grp_b, c->stream); cm->events.push_back({cm->grp_a, cm->grp_b, CMF_COMM_GROUP}); }
    cm->grp_a = cm->grp_b = nullptr;
    cm->kcalls[CMF_COMM_GROUP] += 1;
    return CMF_OK;
}

// in-place sum over the ranks of n float32 / float64 values in device memory, on the context's stream
extern "C" int cmf_comm_allreduce_f32(cmf_ctx *c, float *dev_buf, int64_t n) {
    NEED_COMM(c);
    if (!dev_buf || n < 0) return fail(CMF_EINVAL, "bad buffer");
    DeviceGuard dg(c->device);
    CommTimed tm(c, c->comm, n * 4, CMF_COMM_ALLREDUCE_F32);
    RCCLCHK(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)n, ncclFloat32, ncclSum, c->comm->comm, c->stream));
    return CMF_OK;
}
extern "C" int cmf_comm_allreduce_f64(cmf_ctx *c, double *dev_buf, int64_t n) {
    NEED_COMM(c);
    if (!dev_buf || n < 0) return fail(CMF_EINVAL, "bad buffer");
    DeviceGuard dg(c->device);
    CommTimed tm(c, c->comm, n * 8, CMF_COMM_ALLREDUCE_F64);
    RCCLCHK(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)n, ncclFloat64, ncclSum, c->comm->comm, c->stream));
    return CMF_OK;
}
// The same all-reduce in the BACKGROUND: enqueued on the communicator's side stream behind everything the context's stream holds
// now, so that kernels launched on the context's stream afterwards overlap with it.  cmf_comm_join makes the context's stream
// wait for every background collective issued so far (call it before anything reads the reduced buffers).
extern "C" int cmf_comm_allreduce_f32_bg(cmf_ctx *c, float *dev_buf, int64_t n) {
    NEED_COMM(c);
    if (!dev_buf || n < 0) return fail(CMF_EINVAL, "bad buffer");
    DeviceGuard dg(c->device);
    CmfComm *cm = c->comm;
    if (!cm->side) {
        HIPCHK(hipStreamCreateWithFlags(&cm->side, hipStreamNonBlocking));
        for (auto &e : cm->ev_ready) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&cm->ev_done, hipEventDisableTiming));
    }
    hipEvent_t ready = cm->ev_ready[cm->ev_next++ & 7];
    HIPCHK(hipEventRecord(ready, c->stream));
    HIPCHK(hipStreamWaitEvent(cm->side, ready, 0));
    {
        CommTimed tm(c, cm, n * 4, CMF_COMM_ALLREDUCE_F32, cm->side);
        RCCLCHK(g_rccl.AllReduce(dev_buf, dev_buf, (size_t)n, ncclFloat32, ncclSum, cm->comm, cm->side));
    }
    HIPCHK(hipEventRecord(cm->ev_done, cm->side));
    cm->bg_pending = true;
    return CMF_OK;
}
extern "C" int cmf_comm_join(cmf_ctx *c) {
    NEED_COMM(c);
    CmfComm *cm = c->comm;
    if (!cm->bg_pending) return CMF_OK;
    DeviceGuard dg(c->device);
    hipEvent_t a = nullptr, b = nullptr;
    if (cm->timed && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, c->stream);
    else a = b = nullptr;
    HIPCHK(hipStreamWaitEvent(c->stream, cm->ev_done, 0));
    if (a && b) { (void)hipEventRecord(b, c->stream); cm->joins.push_back({a, b}); }
    cm->bg_pending = false;
    return CMF_OK;
}
// milliseconds the context's stream spent WAITING in cmf_comm_join since the last reset (while timed): the exposed part of the
// background collectives; cmf_comm_stats' ms is their total duration, the difference was hidden under compute
extern "C" int cmf_comm_exposed_ms(cmf_ctx *c, double *ms, int reset) {
    NEED_COMM(c);
    DeviceGuard dg(c->device);
    CmfComm *cm = c->comm;
    HIPCHK(hipStreamSynchronize(c->stream));
    double total = 0.0;
    for (auto &e : cm->joins) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) total += t;
    }
    if (ms) *ms = total;
    if (reset) {
        for (auto &e : cm->joins) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        cm->joins.clear();
    }
    return CMF_OK;
}

// in-place all-gather of equal chunks: rank r's `elems_per_rank` floats already sit at dev_full + r * elems_per_rank
extern "C" int cmf_comm_allgather_f32(cmf_ctx *c, float *dev_full, int64_t elems_per_rank) {
    NEED_COMM(c);
    if (!dev_full || elems_per_rank < 0) return fail(CMF_EINVAL, "bad buffer");
    DeviceGuard dg(c->device);
    CommTimed tm(c, c->comm, elems_per_rank * 4 * c->comm->world, CMF_COMM_ALLGATHER_F32);
    RCCLCHK(g_rccl.AllGather(dev_full + (int64_t)c->comm->rank * elems_per_rank, dev_full, (size_t)elems_per_rank, ncclFloat32, c->comm->comm, c->stream));
    return CMF_OK;
}
// in-place reduce-scatter of equal chunks: every rank holds world * elems_per_rank floats at dev_full; afterwards rank r's chunk
// [r * elems_per_rank, (r + 1) * elems_per_rank) of ITS buffer holds the sum over the ranks of that chunk (the other chunks are
// left as they were).  First half of the decomposed all-reduce of the row-blocked MU V update (cmf_mu_v_apply_rows); the second
// half is cmf_comm_allgather_f32 on the updated factor.
extern "C" int cmf_comm_reduce_scatter_f32(cmf_ctx *c, float *dev_full, int64_t elems_per_rank) {
    NEED_COMM(c);
    if (!dev_full || elems_per_rank < 0) return fail(CMF_EINVAL, "bad buffer");
    DeviceGuard dg(c->device);
    CommTimed tm(c, c->comm, elems_per_rank * 4 * c->comm->world, CMF_COMM_REDUCE_SCATTER_F32);
    RCCLCHK(g_rccl.ReduceScatter(dev_full, dev_full + (int64_t)c->comm->rank * elems_per_rank, (size_t)elems_per_rank, ncclFloat32, ncclSum, c->comm->comm, c->stream));
    return CMF_OK;
}
// what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): the bench line reports these, not the launcher's
// environment, so a run in which the ranks did not find each other cannot pass for an N-GPU run
extern "C" int cmf_comm_count(cmf_ctx *c, int *ranks_seen, int *rank_seen) {
    NEED_COMM(c);
    int n = 0, r = -1;
    RCCLCHK(g_rccl.CommCount(c->comm->comm, &n));
    RCCLCHK(g_rccl.CommUserRank(c->comm->comm, &r));
    if (ranks_seen) *ranks_seen = n;
    if (rank_seen) *rank_seen = r;
    return CMF_OK;
}
// a few host scalars (convergence test: two squared residuals; bench: the slowest rank's time): op 0 = sum, 1 = max.  Waits.
extern "C" int cmf_comm_allreduce_host_f64(cmf_ctx *c, double *vals, int n, int op) {
    NEED_COMM(c);
    if (!vals || n < 0 || n > 16 || (op != 0 && op != 1)) return fail(CMF_EINVAL, "bad argument (at most 16 values; op 0 sum, 1 max)");
    DeviceGuard dg(c->device);
    CmfComm *cm = c->comm;
    HIPCHK(hipMemcpyAsync(cm->dscratch, vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCLCHK(g_rccl.AllReduce(cm->dscratch, cm->dscratch, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, cm->comm, c->stream));
    HIPCHK(hipMemcpyAsync(vals, cm->dscratch, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}
extern "C" int cmf_comm_barrier(cmf_ctx *c) {
    double one = 1.0;
    return cmf_comm_allreduce_host_f64(c, &one, 1, 0);
}
extern "C" int cmf_comm_info(cmf_ctx *c, int *rank, int *world) {
    NEED_COMM(c);
    if (rank) *rank = c->comm->rank;
    if (world) *world = c->comm->world;
    return CMF_OK;
}
// collective accounting since the last reset: calls, payload bytes, milliseconds on the stream (only while timed)
extern "C" int cmf_comm_timing(cmf_ctx *c, int enable) {
    NEED_COMM(c);
    c->comm->timed = enable != 0;
    return CMF_OK;
}
extern "C" int cmf_comm_stats(cmf_ctx *c, int64_t *calls, int64_t *bytes, double *ms, int reset) {
    NEED_COMM(c);
    DeviceGuard dg(c->device);
    CmfComm *cm = c->comm;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (cm->side) HIPCHK(hipStreamSynchronize(cm->side));
    double total = 0.0;
    for (auto &e : cm->events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) total += t;
    }
    if (calls) *calls = cm->calls;
    if (bytes) *bytes = cm->bytes;
    if (ms) *ms = total;
    if (reset) {
        for (auto &e : cm->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        cm->events.clear();
        cm->calls = 0; cm->bytes = 0; cm->launch_points = 0;
        for (int k = 0; k < CMF_COMM_KINDS; ++k) cm->kcalls[k] = cm->kbytes[k] = 0;
    }
    return CMF_OK;
}
// launch points on the stream since the last reset: collectives outside groups + groups (read before cmf_comm_stats(reset))
extern "C" int cmf_comm_launch_points(cmf_ctx *c, int64_t *n) {
    NEED_COMM(c);
    if (!n) return fail(CMF_EINVAL, "null argument");
    *n = c->comm->launch_points;
    return CMF_OK;
}
// the same accounting for ONE kind of collective (CMF_COMM_ALLREDUCE_F32 ...), never resets: call before cmf_comm_stats(reset)
extern "C" int cmf_comm_stats_kind(cmf_ctx *c, int kind, int64_t *calls, int64_t *bytes, double *ms) {
    NEED_COMM(c);
    if (kind < 0 || kind >= CMF_COMM_KINDS) return fail(CMF_EINVAL, "unknown collective kind %d", kind);
    DeviceGuard dg(c->device);
    CmfComm *cm = c->comm;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (cm->side) HIPCHK(hipStreamSynchronize(cm->side));
    double total = 0.0;
    for (auto &e : cm->events) {
        float t = 0.f;
        if (e.kind == kind && hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) total += t;
    }
    if (calls) *calls = cm->kcalls[kind];
    if (bytes) *bytes = cm->kbytes[kind];
    if (ms) *ms = total;
    return CMF_OK;
}

// dev[0..n) *= factor on the context's stream: the measurement double of the collectives (CMF_COMM_BACKEND=null) stands in for
// the peers' contributions with the rank's own partial times the world size, so that the iterates stay finite
extern "C" int cmf_scale_f32(cmf_ctx *c, float *dev, int64_t n, double factor) {
    if (!c || !dev || n < 0) return fail(CMF_EINVAL, "bad argument");
    if (n == 0) return CMF_OK;
    DeviceGuard dg(c->device);
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, c->stream, dev, (const float *)dev, (float)factor, (const float *)nullptr, 0.f, n);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// raw copies between device memory the caller holds a pointer to (scratch, partial buffers) and host memory, ordered on the
// context's stream; both wait.  Used by the host-staged test double of the collectives (pycmf_amd/comm.py) and by tests.
extern "C" int cmf_copy_to_host(cmf_ctx *c, const void *dev, void *host, int64_t bytes) {
    if (!c || !dev || !host || bytes < 0) return fail(CMF_EINVAL, "bad argument");
    DeviceGuard dg(c->device);
    HIPCHK(hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}
extern "C" int cmf_copy_from_host(cmf_ctx *c, void *dev, const void *host, int64_t bytes) {
    if (!c || !dev || !host || bytes < 0) return fail(CMF_EINVAL, "bad argument");
    DeviceGuard dg(c->device);
    HIPCHK(hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}
