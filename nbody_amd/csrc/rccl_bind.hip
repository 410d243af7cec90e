// rccl_bind.hip -- RCCL for the sharded pipeline: the library bound lazily, the communicator, the watchdog.
//
// The reference has no counterpart (one device, vulkan_ctx.c:83-84).  librccl is ~0.5 GB, so it is dlopen'ed on first
// use and single-GPU users never load it.  The soname resolves to whatever copy the process already holds: /opt/rocm's in
// nbody-bench and in bench.py (neither imports torch), the torch wheel's in a process that imported torch first
// (DESIGN.md section 4).  Only five entry points of the data path are used: ncclGetUniqueId, ncclCommInitRank,
// ncclAllGather (in place, float32), ncclCommDestroy, plus the introspection calls a multi-GPU run reports.
#include <dlfcn.h>
#include <unistd.h>

#include <chrono>

#include "pipeline_internal.h"

namespace {

typedef struct {
    char internal[NB_HIP_UNIQUE_ID_BYTES];
} ncclUniqueId;
enum { NCCL_FLOAT32 = 7 };  // ncclDataType_t value of ncclFloat32 (rccl.h)

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    int (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    char path[256] = {0};  // file the symbols came from (dladdr), for the record
};

Rccl &rccl() {
    static Rccl r;
    if (r.handle) return r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    NB_ASSERT(r.handle, "cannot load librccl.so.1 (%s): the sharded pipeline needs RCCL", dlerror());
#define NB_SYM(field, name)                                                \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));  \
    NB_ASSERT(r.field, "librccl lacks %s", name)
    NB_SYM(GetUniqueId, "ncclGetUniqueId");
    NB_SYM(CommInitRank, "ncclCommInitRank");
    NB_SYM(CommDestroy, "ncclCommDestroy");
    NB_SYM(AllGather, "ncclAllGather");
    NB_SYM(CommCount, "ncclCommCount");
    NB_SYM(CommUserRank, "ncclCommUserRank");
    NB_SYM(CommCuDevice, "ncclCommCuDevice");
    NB_SYM(GetVersion, "ncclGetVersion");
    NB_SYM(GetErrorString, "ncclGetErrorString");
#undef NB_SYM
    Dl_info where;
    if (dladdr(reinterpret_cast<void *>(r.AllGather), &where) && where.dli_fname)
        snprintf(r.path, sizeof r.path, "%s", where.dli_fname);
    return r;
}

#define ASSERT_NCCL(X, ...)                                                                                   \
    do {                                                                                                      \
        int nb_r_ = (X);                                                                                      \
        if (nb_r_ != 0) {                                                                                     \
            fprintf(stderr, "%s:%d [%s] ncclResult_t = %d, str = %s\n", __FILE__, __LINE__, __func__, nb_r_,  \
                    rccl().GetErrorString(nb_r_));                                                            \
            NB_FAIL(__VA_ARGS__);                                                                             \
        }                                                                                                     \
    } while (0)

}  // namespace

namespace nbi {

namespace {

// the process' one watcher thread: holds one entry per thread that is inside a bounded wait and sleeps until the
// EARLIEST deadline (or a change of the list), so waits on several threads -- several sharded pipelines, local groups
// driven from threads -- are each watched, not only the first one
struct Watcher {
    struct Entry {
        uint64_t seq;
        std::thread::id owner;
        std::chrono::steady_clock::time_point deadline;
        const char *what;
        int rank, nranks, seconds;
    };
    std::mutex m;
    std::condition_variable cv;
    std::thread th;
    bool started = false;
    uint64_t next_seq = 1;
    std::vector<Entry> armed;

    void run() {
        std::unique_lock<std::mutex> l(m);
        for (;;) {
            if (armed.empty()) {
                cv.wait(l, [this] { return !armed.empty(); });
                continue;
            }
            size_t first = 0;
            for (size_t i = 1; i < armed.size(); i++)
                if (armed[i].deadline < armed[first].deadline) first = i;
            const Entry e = armed[first];
            // woken by any arm / disarm: look again; on time-out the entry is still there (same seq) -> it never completed
            if (cv.wait_until(l, e.deadline) == std::cv_status::timeout) {
                for (const Entry &x : armed)
                    if (x.seq == e.seq) give_up(x);
            }
        }
    }

    [[noreturn]] void give_up(const Entry &e) {
        fprintf(stderr, "%s [watchdog] rank %d of %d: %s did not complete within %d s; giving up (exit 3)\n", __FILE__, e.rank, e.nranks, e.what,
                e.seconds);
        const char *log = getenv("NCCL_DEBUG_FILE");
        if (log && !strchr(log, '%')) {
            if (FILE *f = fopen(log, "r")) {
                fseek(f, 0, SEEK_END);
                long sz = ftell(f);
                fseek(f, sz > 4096 ? sz - 4096 : 0, SEEK_SET);
                char buf[4097];
                size_t got = fread(buf, 1, 4096, f);
                buf[got] = 0;
                fprintf(stderr, "---- tail of %s ----\n%s\n", log, buf);
                fclose(f);
            }
        }
        fflush(stderr);
        _exit(3);
    }
};

Watcher &watcher() {
    static Watcher *w = new Watcher();  // never destroyed: the thread may outlive static destruction
    return *w;
}

}  // namespace

Watchdog::Watchdog(const char *what, int rank, int nranks) {
    const char *t = getenv("NB_HIP_COMM_TIMEOUT_S");
    const int seconds = t ? atoi(t) : 180;
    if (seconds <= 0) return;
    Watcher &w = watcher();
    {
        std::lock_guard<std::mutex> l(w.m);
        const std::thread::id me = std::this_thread::get_id();
        for (const Watcher::Entry &e : w.armed)
            if (e.owner == me) return;  // an outer wait of THIS thread is already being watched: its deadline stands
        if (!w.started) {
            w.started = true;
            w.th = std::thread([&w] { w.run(); });
            w.th.detach();
        }
        seq_ = w.next_seq++;
        w.armed.push_back({seq_, me, std::chrono::steady_clock::now() + std::chrono::seconds(seconds), what, rank, nranks, seconds});
    }
    w.cv.notify_all();
}

Watchdog::~Watchdog() {
    if (seq_ == 0) return;
    Watcher &w = watcher();
    {
        std::lock_guard<std::mutex> l(w.m);
        for (size_t i = 0; i < w.armed.size(); i++)
            if (w.armed[i].seq == seq_) {
                w.armed.erase(w.armed.begin() + (long)i);
                break;
            }
    }
    w.cv.notify_all();
}

void comm_create(SimPipeline *s, const void *unique_id128) {
    const int rank = s->rank, nranks = s->nranks;
    RandGuard keep_callers_rand_stream;  // RCCL's bootstrap and HIP's first set-up draw from libc's rand()
    use_device();  // the communicator binds to the current device
    ncclUniqueId id;
    memcpy(&id, unique_id128, NB_HIP_UNIQUE_ID_BYTES);
    {
        Watchdog dog("ncclCommInitRank", rank, nranks);
        const auto t0 = std::chrono::steady_clock::now();
        ASSERT_NCCL(rccl().CommInitRank(&s->comm, nranks, id, rank), "ncclCommInitRank(rank %d of %d)", rank, nranks);
        s->comm_init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // The communicator's own view must agree with what the caller said: this is what tells N real ranks from N
    // independent replicas.
    int seen_n = -1, seen_r = -1;
    ASSERT_NCCL(rccl().CommCount(s->comm, &seen_n), "ncclCommCount");
    ASSERT_NCCL(rccl().CommUserRank(s->comm, &seen_r), "ncclCommUserRank");
    NB_ASSERT(seen_n == nranks && seen_r == rank, "communicator reports rank %d of %d, expected %d of %d", seen_r, seen_n,
              rank, nranks);
    // First collective, bounded: a 256-byte-per-rank all-gather of (rank + 1) tags, checked on arrival.  Pays RCCL's
    // lazy channel setup here instead of inside the first timed step.
    Watchdog dog("the first ncclAllGather", rank, nranks);
    const size_t per = 64;
    float *probe = dev_alloc<float>(per * (size_t)nranks);
    std::vector<float> host(per * (size_t)nranks, 0.0f);
    for (size_t i = 0; i < per; i++) host[(size_t)rank * per + i] = (float)(rank + 1);
    hipStream_t st;
    ASSERT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "probe stream");
    ASSERT_HIP(hipMemcpyAsync(probe, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, st), "probe H2D");
    hipEvent_t e0, e1;
    ASSERT_HIP(hipEventCreate(&e0), "event");
    ASSERT_HIP(hipEventCreate(&e1), "event");
    ASSERT_HIP(hipEventRecord(e0, st), "record");
    ASSERT_NCCL(rccl().AllGather(probe + (size_t)rank * per, probe, per, NCCL_FLOAT32, s->comm, st), "first ncclAllGather");
    ASSERT_HIP(hipEventRecord(e1, st), "record");
    ASSERT_HIP(hipMemcpyAsync(host.data(), probe, host.size() * sizeof(float), hipMemcpyDeviceToHost, st), "probe D2H");
    ASSERT_HIP(hipStreamSynchronize(st), "probe sync");
    for (int q = 0; q < nranks; q++)
        for (size_t i = 0; i < per; i++)
            NB_ASSERT(host[(size_t)q * per + i] == (float)(q + 1), "first all-gather: slot of rank %d holds %g", q,
                      (double)host[(size_t)q * per + i]);
    float ms = 0.0f;
    ASSERT_HIP(hipEventElapsedTime(&ms, e0, e1), "elapsed");
    s->first_gather_ms = (double)ms;
    // ... and what a WARM small all-gather costs in-stream: 16 back-to-back 8-byte-per-rank gathers between one event pair.
    // This is the fixed cost of the per-step gather -- the one term of the scaling curve a single-GPU box cannot measure.
    const int reps = 16;
    ASSERT_HIP(hipEventRecord(e0, st), "record");
    for (int i = 0; i < reps; i++)
        ASSERT_NCCL(rccl().AllGather(probe + (size_t)rank * 2, probe, 2, NCCL_FLOAT32, s->comm, st), "8-byte ncclAllGather");
    ASSERT_HIP(hipEventRecord(e1, st), "record");
    ASSERT_HIP(hipStreamSynchronize(st), "8-byte gathers");
    ASSERT_HIP(hipEventElapsedTime(&ms, e0, e1), "elapsed");
    s->small_gather_us = (double)ms * 1e3 / reps;
    ASSERT_HIP(hipEventDestroy(e0), "event");
    ASSERT_HIP(hipEventDestroy(e1), "event");
    ASSERT_HIP(hipStreamDestroy(st), "probe stream");
    dev_free(probe);
}

void comm_destroy(SimPipeline *s) {
    if (s->comm) ASSERT_NCCL(rccl().CommDestroy(s->comm), "ncclCommDestroy");
    s->comm = nullptr;
}

void comm_allgather_f32(SimPipeline *s, void *base, size_t count_per_rank, hipStream_t st, const char *what) {
    float *b = static_cast<float *>(base);
    ASSERT_NCCL(rccl().AllGather(b + (size_t)s->rank * count_per_rank, b, count_per_rank, NCCL_FLOAT32, s->comm, st),
                "ncclAllGather of %zu floats per rank (%s)", count_per_rank, what);
}

}  // namespace nbi

extern "C" {

void nb_hip_comm_unique_id(void *out128) {
    NB_ASSERT(out128 != nullptr, "NULL id buffer");
    nbi::RandGuard keep_callers_rand_stream;
    ncclUniqueId id;
    ASSERT_NCCL(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, &id, NB_HIP_UNIQUE_ID_BYTES);
}

int nb_hip_comm_bringup(const SimPipeline *s, double *init_ms, double *first_gather_ms, double *small_gather_us) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (init_ms) *init_ms = s->comm_init_ms;
    if (first_gather_ms) *first_gather_ms = s->first_gather_ms;
    if (small_gather_us) *small_gather_us = s->small_gather_us;
    return s->comm != nullptr;
}

int nb_hip_comm_info(const SimPipeline *s, int *nranks, int *rank, int *device, int *rccl_version, double *first_gather_ms,
                     char *lib_path, uint32_t len) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (nranks) *nranks = s->nranks;
    if (rank) *rank = s->rank;
    if (device) *device = nbi::g_dev.ordinal;
    if (rccl_version) *rccl_version = 0;
    if (first_gather_ms) *first_gather_ms = s->first_gather_ms;
    if (lib_path && len) snprintf(lib_path, len, "%s", s->direct ? "direct device-to-device pushes (IPC-mapped peers) + host step barrier" : s->host_gather ? "caller-supplied host all-gather" : s->group ? "local group" : "");
    if (s->comm == nullptr) return 0;  // unsharded, a local-group member or a caller-supplied transport: no communicator
    // everything below is what the COMMUNICATOR says, not what the caller passed at creation
    if (nranks) ASSERT_NCCL(rccl().CommCount(s->comm, nranks), "ncclCommCount");
    if (rank) ASSERT_NCCL(rccl().CommUserRank(s->comm, rank), "ncclCommUserRank");
    if (device) ASSERT_NCCL(rccl().CommCuDevice(s->comm, device), "ncclCommCuDevice");
    if (rccl_version) ASSERT_NCCL(rccl().GetVersion(rccl_version), "ncclGetVersion");
    if (lib_path && len) snprintf(lib_path, len, "%s", rccl().path);
    return 1;
}

}  // extern "C"
