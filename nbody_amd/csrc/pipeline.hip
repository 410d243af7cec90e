// pipeline.hip -- the C-ABI of include/nbody_hip.h: device state, step chains as hipGraphs,
// AoS<->SoA hand-over, and the N/P sharded pipeline with its per-step RCCL all-gather.
//
// Stands where the reference has src/lib/sim_gpu.c (SimPipeline, command-buffer recording,
// staging copies) and src/lib/vulkan_ctx.c (device pick, allocator).  Differences by design:
//   * SoA in HBM (float2 pos/vel/acc, float radius/mass, float G*m) instead of 32-byte AoS records;
//   * ping-pong position buffers instead of a full device-to-device copy per step (sim_gpu.c:316-324);
//   * n-step chains are cached hipGraphs of kernel nodes (keyed on length and ping-pong phase) instead of a command
//     buffer re-recorded per call (sim_gpu.c:262-344); the step size sits in device memory like the reference's
//     uniform (sim_gpu.c:268-284), so a new dt never rebuilds a chain;
//   * device-to-host copy only when GetSimulationData asks (the reference copies after every call,
//     sim_gpu.c:336-341);
//   * nothing is created on the GPU until SetSimulationData, so CPU-only worlds never touch a device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "kernels.h"
#include "nbody_hip.h"

#define NB_HIP_VERSION 100  // 0.1.0

// ---- error convention: print where, abort (reference src/lib/util.h:17-29,47-60) -------------------------

#define NB_FAIL(...)                                                          \
    do {                                                                      \
        fprintf(stderr, "%s:%d [%s] ", __FILE__, __LINE__, __func__);         \
        fprintf(stderr, __VA_ARGS__);                                         \
        fprintf(stderr, "\n");                                                \
        abort();                                                              \
    } while (0)

#define NB_ASSERT(COND, ...)               \
    do {                                   \
        if (!(COND)) NB_FAIL(__VA_ARGS__); \
    } while (0)

#define ASSERT_HIP(X, ...)                                                                              \
    do {                                                                                                \
        hipError_t nb_e_ = (X);                                                                         \
        if (nb_e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "%s:%d [%s] hipError_t = %d, str = %s\n", __FILE__, __LINE__, __func__,     \
                    (int)nb_e_, hipGetErrorString(nb_e_));                                              \
            NB_FAIL(__VA_ARGS__);                                                                       \
        }                                                                                               \
    } while (0)

// ---- RCCL, bound lazily (librccl is ~0.5 GB; single-GPU users never load it) ------------------------------

namespace {

typedef struct ncclComm *ncclComm_t;
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

// A collective that never completes (a rank that died, a fabric that does not come up) must not hang the job: the
// calls that wait on other ranks -- ncclCommInitRank and the first all-gather -- run under a watchdog that prints
// what was being waited for, the tail of RCCL's own log when NCCL_DEBUG_FILE names one, and _exit(3)s.  No retry and
// no re-exec: the process has initialised the GPU.  NB_HIP_COMM_TIMEOUT_S (default 180) sets the bound; 0 disables it.
class Watchdog {
  public:
    Watchdog(const char *what, int rank, int nranks) : what_(what), rank_(rank), nranks_(nranks) {
        const char *t = getenv("NB_HIP_COMM_TIMEOUT_S");
        seconds_ = t ? atoi(t) : 180;
        if (seconds_ > 0) th_ = std::thread([this] { run(); });
    }
    ~Watchdog() {
        if (!th_.joinable()) return;
        {
            std::lock_guard<std::mutex> l(m_);
            done_ = true;
        }
        cv_.notify_all();
        th_.join();
    }

  private:
    void run() {
        std::unique_lock<std::mutex> l(m_);
        if (cv_.wait_for(l, std::chrono::seconds(seconds_), [this] { return done_; })) return;
        fprintf(stderr, "%s [watchdog] rank %d of %d: %s did not complete within %d s; giving up (exit 3)\n", __FILE__,
                rank_, nranks_, what_, seconds_);
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
    const char *what_;
    int rank_, nranks_, seconds_ = 0;
    bool done_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::thread th_;
};

// ---- process-wide device context (the reference keeps one global vulkan_ctx, vulkan_ctx.c:11) -------------

struct DeviceCtx {
    bool ready = false;
    int ordinal = -1;  // -1: not chosen yet
    int compute_units = 0;
    char info[256] = {0};
};

DeviceCtx g_dev;
int g_requested_ordinal = -1;

void ensure_device() {
    if (g_dev.ready) return;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    NB_ASSERT(e == hipSuccess && count > 0,
              "no HIP device visible (hipGetDeviceCount: %s, count %d); the GPU path has no CPU fallback",
              hipGetErrorString(e), count);
    int ord = g_requested_ordinal >= 0 ? g_requested_ordinal : 0;
    NB_ASSERT(ord < count, "device ordinal %d requested, %d visible", ord, count);
    ASSERT_HIP(hipSetDevice(ord), "hipSetDevice(%d)", ord);
    // PerformSimUpdate is synchronous by contract (the reference blocks on its fence, sim_gpu.c:353) and interactive
    // callers step a few thousand particles per frame: how fast the host notices completion is part of the step time.
    // NB_HIP_WAIT=spin|yield|block picks the runtime's wait policy before the context exists; default: the runtime's.
    if (const char *wp = getenv("NB_HIP_WAIT")) {
        const unsigned flag = !strcmp(wp, "spin") ? hipDeviceScheduleSpin
                              : !strcmp(wp, "yield") ? hipDeviceScheduleYield
                              : !strcmp(wp, "block") ? hipDeviceScheduleBlockingSync
                                                     : hipDeviceScheduleAuto;
        if (hipSetDeviceFlags(flag) != hipSuccess) (void)hipGetLastError();  // context already live: keep its policy
    }
    hipDeviceProp_t prop;
    ASSERT_HIP(hipGetDeviceProperties(&prop, ord), "hipGetDeviceProperties(%d)", ord);
    NB_ASSERT(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
              "device %d is %s; this library ships gfx950 (MI355X) code objects only", ord, prop.gcnArchName);
    g_dev.ordinal = ord;
    g_dev.compute_units = prop.multiProcessorCount;
    snprintf(g_dev.info, sizeof g_dev.info, "%s %s %d %d", prop.name[0] ? prop.name : "AMD-GPU", prop.gcnArchName,
             prop.multiProcessorCount, prop.clockRate / 1000);
    g_dev.ready = true;
}

// HIP's current device is per THREAD: every entry point that allocates, launches or copies re-selects the
// process' device, so a call from another thread than the first one lands on the same GPU (ordinal > 0 matters:
// sharded ranks use LOCAL_RANK).
void use_device() {
    ensure_device();
    ASSERT_HIP(hipSetDevice(g_dev.ordinal), "hipSetDevice(%d)", g_dev.ordinal);
}

template <typename T>
T *dev_alloc(size_t count) {
    T *p = nullptr;
    if (count == 0) count = 1;
    ASSERT_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)), "hipMalloc of %zu bytes", count * sizeof(T));
    return p;
}

void dev_free(void *p) {
    if (p) ASSERT_HIP(hipFree(p), "hipFree");
}

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// ---- a cached chain of step launches ------------------------------------------------------------------------

struct StepGraph {
    uint32_t n = 0;           // steps in the chain
    uint32_t passes = 1;      // source passes per step
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;  // one kernel node per launch, in order
    std::vector<nb::StepParams> params; // what each node currently holds
    int phase = -1;                     // which pos buffer the chain reads first
    nb::LaunchShape shape = {0, 0, 0, 0, 0};
    uint64_t last_use = 0;              // for eviction: the cache holds at most GRAPH_CACHE_MAX chains
};

// Event pairs around the kernels and the gathers of a sharded chain (the plain-launch path), so that a multi-GPU
// run can say how much of a step was the all-gather.  Grown on demand, reused by every call.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t next() {
        if (used == ev.size()) {
            hipEvent_t e;
            ASSERT_HIP(hipEventCreate(&e), "event");
            ev.push_back(e);
        }
        return ev[used++];
    }
    void destroy() {
        for (hipEvent_t e : ev) ASSERT_HIP(hipEventDestroy(e), "event");
        ev.clear();
        used = 0;
    }
};

}  // namespace

struct LocalGroup {
    std::vector<SimPipeline *> members;
    hipStream_t stream = nullptr;  // every member enqueues here, so program order is the only ordering needed
};

struct SimPipeline {
    WorldData data;
    // sharding (nranks == 1: the whole world on one device)
    int rank = 0, nranks = 1;
    bool sharded = false;  // RCCL path (nranks > 1, or forced for single-GPU testing of that path)
    NbShardPlan plan;
    ncclComm_t comm = nullptr;
    struct LocalGroup *group = nullptr;  // test transport: all ranks are pipelines of this process (no RCCL)
    // caller-supplied transport (CreateSimPipelineShardedWith): an in-place all-gather over HOST memory; the pipeline
    // stages each exchange through one page-locked buffer (D2H own slot, wait, callback, H2D all slots)
    NbAllGatherFn host_gather = nullptr;
    void *host_gather_ctx = nullptr;
    void *stage = nullptr;        // page-locked staging, max(gathered sources, gathered particle slices) bytes
    size_t stage_bytes = 0;

    bool on_device = false;  // buffers exist and hold data
    uint32_t slots = 0;      // receiver slots on this device (allocation; includes a shard's pad slots)
    uint32_t n_real = 0;     // receivers actually computed (== slots when unsharded)
    uint32_t n_src = 0;      // sources every receiver sees (mass_len, or the padded gathered length)

    // SoA streams (DESIGN.md "Layout in HBM")
    float2 *pos[2] = {nullptr, nullptr};
    float2 *vel = nullptr;
    float2 *acc = nullptr;
    float *radius = nullptr;
    float *mass = nullptr;
    float2 *src_pos[2] = {nullptr, nullptr};  // sharded only: gathered source positions (ping-pong)
    float *src_gm = nullptr;
    // the step size lives in device memory, like the reference's uniform block (sim_gpu.h:8-12): kernels read it
    // through StepParams::dt, a new value is written in stream order when PerformSimUpdate's dt differs from the last
    // one enqueued (the reference's re-upload, sim_gpu.c:268-284), and no cached hipGraph ever needs re-patching
    float *dt_dev = nullptr;
    float dt_enqueued = 0.0f;
    bool dt_valid = false;
    uint32_t dt_uploads = 0;
    void *aos = nullptr;     // device AoS staging for Set/Get (whole world)
    void *aos_shard = nullptr;  // sharded only: this rank's slice, uniform size
    void *host_array = nullptr;  // caller's long-lived particle array (nb_hip_note_host_array), page-locked lazily
    size_t host_bytes = 0;
    bool host_pinned = false;
    void *host_dev = nullptr;    // device-side address of the page-locked array (kernels store to it over PCIe)
    // eager read-back (knob "readback"): in a frame loop -- every blocking update followed by a Get into the noted
    // array -- the merge kernel of the NEXT Get is appended to the update's own submission and stores straight into the
    // noted array, so the Get finds its data already there: one submission + one wait per frame instead of two
    // (step + wait, then merge + D2H copy + wait).
    int readback = 2;             // 0 never, 1 after every blocking update, 2 auto (after two update->Get pairs in a row)
    int zero_copy_upload = 1;     // SetSimulationData from the noted array: the split kernel reads host memory directly
    bool host_current = false;    // the noted array already holds the device's latest state
    uint32_t updates_since_get = 0, frame_streak = 0;
    // record the ev_begin / ev_end pair around every chain (nb_hip_last_step_ms).  Off unless asked for: the two
    // records cost an interactive caller 3-7 us per call (profiles/r02_frame_loop_latency.txt).
    int timing = 0;
    float2 *parts = nullptr;    // split steps only: [split][n_real] partial sums
    uint32_t parts_cap = 0;     // float2 elements allocated in parts
    int cur = 0;             // pos[cur] is the latest state

    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_local = nullptr, ev_gather = nullptr;
    bool timed = false;
    uint32_t timed_launches = 0;         // step-kernel launches between ev_begin and ev_end
    uint32_t timed_finish_launches = 0;  // finish-kernel launches in the same interval (split shapes only)
    // sharded plain-launch chains: [begin, end) event pairs of each step's kernels and of each gather
    EventPool pool;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kernel_iv, comm_iv;
    uint32_t detail_steps = 0;  // steps the intervals above cover (capped)
    uint64_t use_clock = 0;     // ticks once per graph lookup (LRU)

    // knobs
    int want_variant = nb::VARIANT_SMEM, want_k = 0, want_w = 0, want_split = 0;  // SMEM measures 2.5 % faster than LDS tiles
    int use_graph = 2, overlap = 0, sharded_graph = 0;  // use_graph: 0 never, 1 always, 2 from a chain length's second use
    std::vector<uint32_t> seen_chains;                  // chain lengths already run once as plain launches
    int want_passes = 0;  // source passes per step (0 = auto: keep each pass's sources within one XCD's L2)
    double first_gather_ms = 0.0;  // sharded: device time of the probe all-gather at creation (includes lazy setup)
    nb::LaunchShape last_shape = {0, 0, 0, 0, 0};
    int want_unit = 0;  // source-slice granule: 0 = auto, else 64 / 32 / 16 / 8
    uint32_t last_groups = 0;

    std::vector<StepGraph> graphs;
};

namespace {

constexpr uint32_t GRAPH_CHAIN_MAX = 64;  // longer requests replay an even-length chain
constexpr size_t GRAPH_CACHE_MAX = 8;     // cached chains per pipeline; the least recently used one is evicted
// graph = 2 (auto): chains shorter than this stay plain launches.  A hipGraphLaunch costs the host ~12 us more than
// a few plain launches and a replayed node saves 1-2 us, so a graph pays from a dozen steps on
// (profiles/r02_frame_loop_latency.txt: 2-step frames 33 -> 44 us with a graph, 8-step frames still 98 -> 102 us).
constexpr uint32_t GRAPH_AUTO_MIN_CHAIN = 16;

void destroy_graph(StepGraph &g) {
    if (g.exec) ASSERT_HIP(hipGraphExecDestroy(g.exec), "hipGraphExecDestroy");
    if (g.graph) ASSERT_HIP(hipGraphDestroy(g.graph), "hipGraphDestroy");
    g.exec = nullptr;
    g.graph = nullptr;
    g.nodes.clear();
    g.params.clear();
}

void unpin_host(SimPipeline *s) {
    if (s->host_pinned) {
        // best effort: the array is the caller's; failing to unregister must not take the process down
        (void)hipHostUnregister(s->host_array);
        s->host_pinned = false;
        s->host_dev = nullptr;
        s->host_current = false;
    }
}

void pin_host(SimPipeline *s) {
    if (s->host_pinned || s->host_array == nullptr || s->host_bytes == 0) return;
    // page-lock the caller's array so that H2D / D2H run at PCIe speed instead of through a pageable bounce
    if (hipHostRegister(s->host_array, s->host_bytes, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
        s->host_pinned = true;
        if (hipHostGetDevicePointer(&s->host_dev, s->host_array, 0) != hipSuccess) {
            (void)hipGetLastError();
            s->host_dev = nullptr;  // no zero-copy stores then: Get keeps using the D2H copy
        }
    } else {
        (void)hipGetLastError();  // not fatal: copies stay correct, only slower
    }
}

// Make room for one more cached chain: the least recently used one goes (a frame loop with a varying chain length
// or dt must not grow device-side graph execs without bound).
void evict_for_one_more(SimPipeline *s) {
    while (s->graphs.size() >= GRAPH_CACHE_MAX) {
        size_t victim = 0;
        for (size_t i = 1; i < s->graphs.size(); i++)
            if (s->graphs[i].last_use < s->graphs[victim].last_use) victim = i;
        ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before evicting a cached chain");
        destroy_graph(s->graphs[victim]);
        s->graphs.erase(s->graphs.begin() + (long)victim);
    }
}

void release_device(SimPipeline *s) {
    if (!s->on_device) return;
    use_device();
    ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before release");
    if (s->comm_stream) ASSERT_HIP(hipStreamSynchronize(s->comm_stream), "sync before release");
    s->pool.destroy();
    s->kernel_iv.clear();
    s->comm_iv.clear();
    unpin_host(s);
    for (auto &g : s->graphs) destroy_graph(g);
    s->graphs.clear();
    for (int b = 0; b < 2; b++) {
        dev_free(s->pos[b]);
        if (s->sharded) dev_free(s->src_pos[b]);
        s->pos[b] = s->src_pos[b] = nullptr;
    }
    dev_free(s->vel);
    dev_free(s->acc);
    dev_free(s->radius);
    dev_free(s->mass);
    dev_free(s->src_gm);
    dev_free(s->dt_dev);
    s->dt_dev = nullptr;
    s->dt_valid = false;
    dev_free(s->aos);
    dev_free(s->aos_shard);
    if (s->stage) ASSERT_HIP(hipHostFree(s->stage), "hipHostFree staging");
    s->stage = nullptr;
    s->stage_bytes = 0;
    dev_free(s->parts);
    s->parts = nullptr;
    s->parts_cap = 0;
    ASSERT_HIP(hipEventDestroy(s->ev_begin), "event");
    ASSERT_HIP(hipEventDestroy(s->ev_end), "event");
    ASSERT_HIP(hipEventDestroy(s->ev_local), "event");
    ASSERT_HIP(hipEventDestroy(s->ev_gather), "event");
    if (s->comm_stream) ASSERT_HIP(hipStreamDestroy(s->comm_stream), "stream");
    if (!s->group) ASSERT_HIP(hipStreamDestroy(s->stream), "stream");
    s->on_device = false;
}

// First touch of the GPU for this pipeline: stream, events, HBM buffers.
void materialize(SimPipeline *s) {
    use_device();
    if (s->on_device) return;
    if (s->group)
        s->stream = s->group->stream;
    else
        ASSERT_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "stream");
    ASSERT_HIP(hipEventCreate(&s->ev_begin), "event");
    ASSERT_HIP(hipEventCreate(&s->ev_end), "event");
    ASSERT_HIP(hipEventCreateWithFlags(&s->ev_local, hipEventDisableTiming), "event");
    ASSERT_HIP(hipEventCreateWithFlags(&s->ev_gather, hipEventDisableTiming), "event");

    const uint32_t N = s->data.total_len, M = s->data.mass_len;
    if (!s->sharded) {
        s->slots = N;
        s->n_real = N;
        s->n_src = M;
    } else {
        s->n_real = s->plan.mass_count + s->plan.zero_count;
        if (!s->group) ASSERT_HIP(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking), "comm stream");
        s->slots = s->plan.mass_chunk + s->plan.zero_chunk;
        s->n_src = s->plan.src_padded;
    }
    const uint32_t cap = s->slots ? s->slots : 1;
    for (int b = 0; b < 2; b++) s->pos[b] = dev_alloc<float2>(cap);
    s->vel = dev_alloc<float2>(cap);
    s->acc = dev_alloc<float2>(cap);
    s->radius = dev_alloc<float>(cap);
    s->mass = dev_alloc<float>(cap);
    s->src_gm = dev_alloc<float>(s->n_src);
    s->dt_dev = dev_alloc<float>(1);
    s->aos = dev_alloc<Particle>(N);
    if (!s->sharded) {
        // the first mass_len receivers ARE the sources: no separate source array
        s->src_pos[0] = s->pos[0];
        s->src_pos[1] = s->pos[1];
    } else {
        for (int b = 0; b < 2; b++) s->src_pos[b] = dev_alloc<float2>(s->n_src);
        s->aos_shard = dev_alloc<Particle>((size_t)s->slots * (size_t)s->nranks);
        if (s->host_gather) {
            const size_t a = (size_t)s->n_src * sizeof(float2), b = (size_t)s->slots * (size_t)s->nranks * sizeof(Particle);
            s->stage_bytes = a > b ? a : b;
            ASSERT_HIP(hipHostMalloc(&s->stage, s->stage_bytes ? s->stage_bytes : 1, hipHostMallocDefault), "staging of %zu bytes",
                       s->stage_bytes);
        }
    }
    s->on_device = true;
    pin_host(s);
    // The first hipGraph a process instantiates and launches costs ~9 ms of runtime set-up (seen as a 146 us/step
    // 100-step call at N = 20 000 where the next one took 54).  Pay it here, with a one-node graph captured from this
    // pipeline's stream, instead of inside whichever step call first replays a chain.
    static bool graph_machinery_warm = false;
    if (!graph_machinery_warm && s->use_graph != 0) {
        graph_machinery_warm = true;
        hipGraph_t g = nullptr;
        hipGraphExec_t e = nullptr;
        if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            nb::launch_set_scalar(s->stream, s->dt_dev, 0.0f);  // dt is uploaded afresh before any step (dt_valid == false)
            if (hipStreamEndCapture(s->stream, &g) == hipSuccess && g != nullptr &&
                hipGraphInstantiate(&e, g, nullptr, nullptr, 0) == hipSuccess) {
                (void)hipGraphLaunch(e, s->stream);
                (void)hipStreamSynchronize(s->stream);
            }
        }
        if (e) (void)hipGraphExecDestroy(e);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();  // best effort: a failure here only postpones the set-up cost
    }
}

uint32_t passes_for(const SimPipeline *s, const nb::StepParams &p);

nb::LaunchShape resolve_shape(SimPipeline *s) {
    nb::LaunchShape want = {s->want_k, s->want_w, s->want_variant, s->want_split, s->want_unit};
    // the model sees one launch: with source passes that is 1/passes of the sources
    nb::StepParams probe;
    memset(&probe, 0, sizeof probe);
    probe.src_end[0] = s->n_src;
    const uint32_t passes = (s->sharded && s->overlap) ? 1 : passes_for(s, probe);
    nb::LaunchShape sh = nb::choose_shape(want, s->n_real, (s->n_src + passes - 1) / passes, g_dev.compute_units);
    NB_ASSERT(nb::step_kernel_fn(sh) != nullptr, "no step kernel for k=%d w=%d variant=%d", sh.k, sh.w, sh.variant);
    if (sh.split > 1) {
        const size_t need = (size_t)sh.split * s->n_real;
        if (need > s->parts_cap) {
            ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before growing the parts buffer");
            dev_free(s->parts);
            s->parts = dev_alloc<float2>(need);
            s->parts_cap = (uint32_t)need;
            for (auto &g : s->graphs) destroy_graph(g);  // cached nodes point at the old buffer
            s->graphs.clear();
        }
    }
    s->last_shape = sh;
    s->last_groups = nb::step_grid(sh, s->n_real).x * nb::step_grid(sh, s->n_real).y;
    return sh;
}

// Parameters of the single-kernel step that reads phase `in` and writes phase `in ^ 1`.
nb::StepParams whole_step(const SimPipeline *s, int in, float dt) {
    nb::StepParams p;
    memset(&p, 0, sizeof p);
    p.src_pos = s->src_pos[in];
    p.src_gm = s->src_gm;
    p.src_begin[0] = 0;
    p.src_end[0] = s->n_src;
    p.src_begin[1] = p.src_end[1] = 0;
    p.pos_in = s->pos[in];
    p.pos_out = s->pos[in ^ 1];
    p.vel = s->vel;
    p.acc = s->acc;
    p.radius = s->radius;
    p.n_recv = s->n_real;
    p.recv_split = s->n_real;
    p.recv_gap = 0;
    if (s->sharded) {
        // slots [0, mass_count) massive, [mass_count, Mc) pads (never computed), [Mc, Mc + zero_count) massless
        p.recv_split = s->plan.mass_count;
        p.recv_gap = s->plan.mass_chunk - s->plan.mass_count;
        p.mirror = s->src_pos[in ^ 1] + (size_t)s->rank * s->plan.mass_chunk;
        p.n_mirror = s->plan.mass_count;
    }
    (void)dt;  // the value travels through device memory (upload_dt), the parameter block only points at it
    p.dt = s->dt_dev;
    p.flags = 0;
    p.parts = nullptr;
    p.split = 1;
    p.unit = 64;
    return p;
}

nb::StepParams shaped(const SimPipeline *s, nb::StepParams p, nb::LaunchShape sh) {
    p.split = sh.split > 1 ? (uint32_t)sh.split : 1u;
    p.parts = p.split > 1 ? s->parts : nullptr;
    // finer slice granules only for single-range steps (the overlapped sharded step walks two ranges: 64 there)
    p.unit = (sh.unit >= 8 && sh.unit <= 64 && p.src_end[1] == p.src_begin[1]) ? (uint32_t)sh.unit : 64u;
    return p;
}

// Source passes: a step over sources [0, n) can run as P launches over consecutive sub-ranges chained through
// acc[] (STEP_NO_FINALIZE / STEP_ACC_IN).  All workgroups of a pass then stream the same <= ~3 MB of sources, which
// stay resident in each XCD's 4 MiB L2 across the pass's rounds instead of being re-fetched every round.
constexpr size_t L2_SOURCE_BUDGET = 3u << 20;  // bytes of (x, y, G*m) per pass

uint32_t passes_for(const SimPipeline *s, const nb::StepParams &p) {
    if (p.flags != 0 || p.src_end[1] != p.src_begin[1]) return 1;  // only whole, unchained steps are cut up
    const uint32_t n = p.src_end[0] - p.src_begin[0];
    uint32_t want = (uint32_t)s->want_passes;
    if (want == 0) want = (uint32_t)(((size_t)n * 12 + L2_SOURCE_BUDGET - 1) / L2_SOURCE_BUDGET);
    const uint32_t chunks = (n + 63) / 64;
    if (want > chunks) want = chunks;
    return want ? want : 1;
}

// The launches of one step: P passes, each = step kernel (+ finish kernel when the shape is split).
std::vector<nb::StepParams> step_passes(const SimPipeline *s, const nb::StepParams &whole, nb::LaunchShape sh) {
    std::vector<nb::StepParams> out;
    const uint32_t P = passes_for(s, whole);
    const uint32_t lo = whole.src_begin[0], n = whole.src_end[0] - lo;
    const uint32_t per = ((n + 63) / 64 + P - 1) / P * 64;  // whole 64-source chunks per pass
    for (uint32_t q = 0; q < P; q++) {
        nb::StepParams p = shaped(s, whole, sh);
        if (P > 1) {
            p.src_begin[0] = lo + (q * per < n ? q * per : n);
            p.src_end[0] = lo + ((q + 1) * per < n ? (q + 1) * per : n);
            p.flags = (q > 0 ? nb::STEP_ACC_IN : 0u) | (q + 1 < P ? nb::STEP_NO_FINALIZE : 0u);
        }
        out.push_back(p);
    }
    return out;
}

void launch_step(SimPipeline *s, nb::LaunchShape sh, const nb::StepParams &p, hipStream_t st) {
    if (s->n_real == 0) return;  // a rank without receivers still takes part in the gathers
    for (nb::StepParams &copy : step_passes(s, p, sh)) {
        void *args[] = {&copy};
        ASSERT_HIP(hipLaunchKernel(nb::step_kernel_fn(sh), nb::step_grid(sh, s->n_real), nb::step_block(sh), args, 0, st),
                   "step kernel launch (k=%d w=%d variant=%d split=%d, %u receivers)", sh.k, sh.w, sh.variant, sh.split,
                   s->n_real);
        if (copy.split > 1)
            ASSERT_HIP(hipLaunchKernel(nb::finish_kernel_fn(), nb::finish_grid(s->n_real), nb::finish_block(), args, 0, st),
                       "finish kernel launch (%u receivers, %u parts)", s->n_real, copy.split);
    }
}

// The step size of everything enqueued from here on.  Written in stream order, so steps already queued keep theirs.
void upload_dt(SimPipeline *s, float dt) {
    if (s->dt_valid && memcmp(&dt, &s->dt_enqueued, sizeof dt) == 0) return;
    nb::launch_set_scalar(s->stream, s->dt_dev, dt);
    s->dt_enqueued = dt;
    s->dt_valid = true;
    s->dt_uploads++;
}

// ---- single-device chains ------------------------------------------------------------------------------------

void fill_node(hipKernelNodeParams &kp, void **args, const void *fn, dim3 grid, dim3 block) {
    memset(&kp, 0, sizeof kp);
    kp.func = const_cast<void *>(fn);
    kp.gridDim = grid;
    kp.blockDim = block;
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
}

// A cached chain is keyed on (length, passes, shape, PHASE): an odd chain length flips the ping-pong phase, so a
// frame loop that asks for the same odd n alternates between two phases -- with the phase in the key it gets two
// instantiated graphs and replays them untouched, instead of re-patching every node of one graph on every call.
// dt is not part of the key and never forces a rebuild or a patch: the nodes read it from device memory (upload_dt).
StepGraph *find_graph(SimPipeline *s, uint32_t n, uint32_t passes, nb::LaunchShape sh, int phase) {
    for (auto &c : s->graphs)
        if (c.n == n && c.passes == passes && c.phase == phase && c.shape.k == sh.k && c.shape.w == sh.w &&
            c.shape.variant == sh.variant && c.shape.split == sh.split && c.shape.unit == sh.unit)
            return &c;
    return nullptr;
}

StepGraph *find_or_build_graph(SimPipeline *s, uint32_t n, float dt, nb::LaunchShape sh) {
    const uint32_t passes = passes_for(s, whole_step(s, s->cur, dt));
    StepGraph *g = find_graph(s, n, passes, sh, s->cur);
    const uint32_t per_pass = sh.split > 1 ? 2 : 1;  // step kernel (+ finish kernel)
    const uint32_t per_step = passes * per_pass;
    const bool fresh = g == nullptr;
    if (fresh) {
        evict_for_one_more(s);
        s->graphs.emplace_back();
        g = &s->graphs.back();
        g->n = n;
        g->passes = passes;
        g->shape = sh;
        ASSERT_HIP(hipGraphCreate(&g->graph, 0), "hipGraphCreate");
        g->nodes.resize((size_t)n * per_step);
        g->params.resize((size_t)n * passes);
    }
    g->last_use = ++s->use_clock;
    if (!fresh) return g;
    hipGraphNode_t prev = nullptr;
    for (uint32_t i = 0; i < n; i++) {
        const std::vector<nb::StepParams> launches = step_passes(s, whole_step(s, (s->cur + i) & 1, dt), sh);
        for (uint32_t q = 0; q < passes; q++) {
            g->params[(size_t)i * passes + q] = launches[q];
            void *args[] = {&g->params[(size_t)i * passes + q]};
            for (uint32_t j = 0; j < per_pass; j++) {
                hipKernelNodeParams kp;
                if (j == 0)
                    fill_node(kp, args, nb::step_kernel_fn(sh), nb::step_grid(sh, s->n_real), nb::step_block(sh));
                else
                    fill_node(kp, args, nb::finish_kernel_fn(), nb::finish_grid(s->n_real), nb::finish_block());
                hipGraphNode_t &node = g->nodes[(size_t)i * per_step + q * per_pass + j];
                ASSERT_HIP(hipGraphAddKernelNode(&node, g->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp),
                           "hipGraphAddKernelNode step %u/%u", i, n);
                prev = node;
            }
        }
    }
    ASSERT_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0), "hipGraphInstantiate (%u steps)", n);
    g->phase = s->cur;
    return g;
}

// graph = 2 (auto) on small worlds: ONE canonical chain of CANON_STEPS steps starting at phase 0, built when the data
// first reaches the device (outside any step call) and replayed by every call of 32+ steps: one plain step if needed
// to reach phase 0, whole replays (even length: the phase stays 0), the remainder as plain launches.  A replayed node
// is a little cheaper than a plain launch while steps are short -- the first 100-step call of a fresh pipeline runs
// 4.56 vs 4.71 us per step at N = 250, 5.00 vs 5.11 at 1 000, 7.29 vs 7.53 at 4 000, 20.7 vs 20.9 at 10 000, and
// slightly SLOWER at 20 000 (50.0 vs 48.4): profiles/r02_first_call_probe.txt -- and building the 32-step chain costs
// 95-150 us once (profiles/r02_graph_chunk_probe.txt), which a one-off call could never win back.  Prebuilt, the
// reference's nbody-bench -- ONE 100-step call per world (bench.c:30-33) -- runs 96 of its 100 steps at the replay rate.
constexpr uint32_t CANON_STEPS = 32;
constexpr double CANON_MAX_PAIRS = 6.0e7;  // N x M up to which a replay still pays (N ~ 11 000 with galaxy.h ICs)

bool wants_canonical(const SimPipeline *s) {
    return !s->sharded && s->use_graph == 2 && s->n_real > 0 &&
           (double)s->n_real * (double)(s->n_src ? s->n_src : 1) <= CANON_MAX_PAIRS;
}

void enqueue_single(SimPipeline *s, uint32_t n, float dt) {
    const nb::LaunchShape sh = resolve_shape(s);
    if (!s->use_graph || n == 1 || (s->use_graph == 2 && n < GRAPH_AUTO_MIN_CHAIN)) {
        for (uint32_t i = 0; i < n; i++) {
            launch_step(s, sh, whole_step(s, s->cur, dt), s->stream);
            s->cur ^= 1;
        }
        return;
    }
    uint32_t left = n;
    if (wants_canonical(s)) {
        if (s->cur == 1 && left > 0) {  // reach phase 0
            launch_step(s, sh, whole_step(s, s->cur, dt), s->stream);
            s->cur ^= 1;
            left--;
        }
        while (left >= CANON_STEPS) {
            StepGraph *g = find_or_build_graph(s, CANON_STEPS, dt, sh);  // prebuilt at SetSimulationData unless a knob moved
            ASSERT_HIP(hipGraphLaunch(g->exec, s->stream), "hipGraphLaunch (canonical %u steps)", CANON_STEPS);
            left -= CANON_STEPS;
        }
        for (; left > 0; left--) {
            launch_step(s, sh, whole_step(s, s->cur, dt), s->stream);
            s->cur ^= 1;
        }
        return;
    }
    while (left > 0) {
        // full chains have even length so that replaying them keeps the ping-pong phase
        const uint32_t chunk = left > GRAPH_CHAIN_MAX ? GRAPH_CHAIN_MAX : left;
        if (s->use_graph == 2 && !find_graph(s, chunk, passes_for(s, whole_step(s, s->cur, dt)), sh, s->cur)) {
            // Building and instantiating a chain costs ~3 us per node, more than it saves in one run (a graph
            // replay saves 1-2 us per step below N ~ 10 000 and nothing above: profiles/r01_graph_build_vs_replay.txt).
            // A caller that steps the same n again and again -- a frame loop -- gets the graph from its second
            // call; a one-off call (the reference's nbody-bench times exactly one) never pays for it.
            bool seen = false;
            for (uint32_t c : s->seen_chains) seen = seen || c == chunk;
            if (!seen) {
                if (s->seen_chains.size() >= 64) s->seen_chains.clear();
                s->seen_chains.push_back(chunk);
                for (uint32_t i = 0; i < chunk; i++) {
                    launch_step(s, sh, whole_step(s, s->cur, dt), s->stream);
                    s->cur ^= 1;
                }
                left -= chunk;
                continue;
            }
        }
        StepGraph *g = find_or_build_graph(s, chunk, dt, sh);
        ASSERT_HIP(hipGraphLaunch(g->exec, s->stream), "hipGraphLaunch (%u steps)", chunk);
        if (chunk & 1) s->cur ^= 1;
        left -= chunk;
    }
}

// ---- sharded chains --------------------------------------------------------------------------------------------

// In-place all-gather of a device array of nranks slots through the caller's host transport: own slot down, wait,
// callback (blocks until every rank's slot is in the staging buffer), everything up.  The stream stays ordered: what
// was enqueued before has completed when the callback runs, what is enqueued after sees the gathered array.
void host_allgather(SimPipeline *s, void *dev_base, size_t bytes_per_rank, hipStream_t st) {
    NB_ASSERT(bytes_per_rank * (size_t)s->nranks <= s->stage_bytes, "staging too small: %zu x %d > %zu", bytes_per_rank, s->nranks,
              s->stage_bytes);
    char *host = static_cast<char *>(s->stage);
    char *dev = static_cast<char *>(dev_base);
    const size_t mine = (size_t)s->rank * bytes_per_rank;
    ASSERT_HIP(hipMemcpyAsync(host + mine, dev + mine, bytes_per_rank, hipMemcpyDeviceToHost, st), "D2H of the own slot");
    ASSERT_HIP(hipStreamSynchronize(st), "sync before the host all-gather");
    s->host_gather(s->host_gather_ctx, host, (uint64_t)bytes_per_rank, s->rank, s->nranks);
    ASSERT_HIP(hipMemcpyAsync(dev, host, bytes_per_rank * (size_t)s->nranks, hipMemcpyHostToDevice, st), "H2D of the gathered slots");
}

void allgather_sources(SimPipeline *s, int buf, hipStream_t st) {
    // in place: this rank's slice already sits at rank * Mc (written by the step kernel's mirror store)
    const size_t per_rank = (size_t)s->plan.mass_chunk * 2;  // floats
    if (per_rank == 0) return;
    float *base = reinterpret_cast<float *>(s->src_pos[buf]);
    if (s->group) {
        // local transport: push the slice into every peer's gathered array (same device, same stream)
        for (SimPipeline *peer : s->group->members) {
            if (peer == s) continue;
            float *dst = reinterpret_cast<float *>(peer->src_pos[buf]);
            ASSERT_HIP(hipMemcpyAsync(dst + (size_t)s->rank * per_rank, base + (size_t)s->rank * per_rank,
                                      per_rank * sizeof(float), hipMemcpyDeviceToDevice, st),
                       "local push of rank %d's sources", s->rank);
        }
        return;
    }
    if (s->host_gather) {
        host_allgather(s, base, per_rank * sizeof(float), st);
        return;
    }
    ASSERT_NCCL(rccl().AllGather(base + (size_t)s->rank * per_rank, base, per_rank, NCCL_FLOAT32, s->comm, st),
                "ncclAllGather of %zu floats per rank", per_rank);
}

constexpr uint32_t DETAIL_STEPS_MAX = 256;  // steps per call whose kernels / gathers get their own event pairs

// One sharded step of one rank.  `cs` carries the gather (the comm stream with RCCL; the group stream locally).
// `detail`: bracket the step's kernels and its gather with event pairs (plain launches only, not under capture).
void sharded_step(SimPipeline *s, nb::LaunchShape sh, float dt, hipStream_t cs, bool detail = false) {
    const uint32_t Mc = s->plan.mass_chunk;
    const uint32_t own_lo = (uint32_t)s->rank * Mc, own_hi = own_lo + Mc;
    const int in = s->cur;
    auto mark = [&](hipStream_t st) -> hipEvent_t {
        if (!detail) return nullptr;
        hipEvent_t e = s->pool.next();
        ASSERT_HIP(hipEventRecord(e, st), "record interval event");
        return e;
    };
    if (!s->overlap) {
        // one kernel over all gathered sources, then gather the positions it produced
        const hipEvent_t k0 = mark(s->stream);
        launch_step(s, sh, whole_step(s, in, dt), s->stream);
        const hipEvent_t k1 = mark(s->stream);  // end of the kernels == begin of the gather (same stream)
        allgather_sources(s, in ^ 1, s->stream);
        const hipEvent_t g1 = mark(s->stream);
        if (detail) {
            s->kernel_iv.emplace_back(k0, k1);
            s->comm_iv.emplace_back(k1, g1);
        }
    } else {
        // own-shard sources are already here: start on them while the other P-1 slices of
        // src_pos[in] are still arriving on the comm stream, then finish with the remote ones
        nb::StepParams a = whole_step(s, in, dt);
        a.src_begin[0] = own_lo;
        a.src_end[0] = own_hi;
        a.flags = nb::STEP_NO_FINALIZE;
        a.n_mirror = 0;
        const hipEvent_t a0 = mark(s->stream);
        launch_step(s, sh, a, s->stream);
        const hipEvent_t a1 = mark(s->stream);
        ASSERT_HIP(hipStreamWaitEvent(s->stream, s->ev_gather, 0), "wait gather");
        const hipEvent_t b0 = mark(s->stream);  // stamped once the previous step's gather has landed
        nb::StepParams b = whole_step(s, in, dt);
        b.src_begin[0] = 0;
        b.src_end[0] = own_lo;
        b.src_begin[1] = own_hi;
        b.src_end[1] = s->n_src;
        b.flags = nb::STEP_ACC_IN;
        launch_step(s, sh, b, s->stream);
        const hipEvent_t b1 = mark(s->stream);
        ASSERT_HIP(hipEventRecord(s->ev_local, s->stream), "record local");
        ASSERT_HIP(hipStreamWaitEvent(cs, s->ev_local, 0), "comm waits for the new slice");
        const hipEvent_t g0 = mark(cs);
        allgather_sources(s, in ^ 1, cs);
        const hipEvent_t g1 = mark(cs);
        ASSERT_HIP(hipEventRecord(s->ev_gather, cs), "record gather");
        if (detail) {
            s->kernel_iv.emplace_back(a0, a1);
            s->kernel_iv.emplace_back(b0, b1);
            s->comm_iv.emplace_back(g0, g1);
        }
    }
    s->cur ^= 1;
}

// Opt-in ("sharded_graph"): the chain of {step kernel(s), in-place all-gather} x n captured from the stream into a
// hipGraph and replayed, like the single-GPU chains.  Only the in-stream (non-overlapped) step is captured; an even
// chain length keeps the ping-pong phase so a cached graph can be replayed as is.  Off by default: RCCL inside
// stream capture is the least-travelled path of this library (exercised with one rank only, tests).
StepGraph *capture_sharded_chain(SimPipeline *s, uint32_t n, float dt, nb::LaunchShape sh) {
    for (auto &c : s->graphs)
        if (c.n == n && c.phase == s->cur && c.shape.k == sh.k && c.shape.w == sh.w &&
            c.shape.variant == sh.variant && c.shape.split == sh.split && c.shape.unit == sh.unit) {
            c.last_use = ++s->use_clock;
            return &c;
        }
    evict_for_one_more(s);  // captured RCCL nodes pin communicator resources: the cache stays small
    s->graphs.emplace_back();
    StepGraph *g = &s->graphs.back();
    g->last_use = ++s->use_clock;
    g->n = n;
    g->phase = s->cur;
    g->shape = sh;
    const int cur0 = s->cur;
    ASSERT_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    for (uint32_t i = 0; i < n; i++) sharded_step(s, sh, dt, s->stream);
    ASSERT_HIP(hipStreamEndCapture(s->stream, &g->graph), "hipStreamEndCapture");
    ASSERT_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0), "hipGraphInstantiate (sharded, %u steps)", n);
    s->cur = cur0;  // capture only recorded the work; the replay below advances the phase
    return g;
}

void enqueue_sharded(SimPipeline *s, uint32_t n, float dt) {
    NB_ASSERT(s->group == nullptr, "members of a local group step through nb_hip_local_group_step");
    const nb::LaunchShape sh = resolve_shape(s);
    if (s->sharded_graph && !s->overlap && n > 1 && !s->host_gather) {  // a host callback cannot run inside a captured graph
        uint32_t left = n;
        while (left > 0) {
            const uint32_t chunk = left > GRAPH_CHAIN_MAX ? GRAPH_CHAIN_MAX : left;
            StepGraph *g = capture_sharded_chain(s, chunk, dt, sh);
            ASSERT_HIP(hipGraphLaunch(g->exec, s->stream), "hipGraphLaunch (sharded, %u steps)", chunk);
            if (chunk & 1) s->cur ^= 1;
            left -= chunk;
        }
        return;
    }
    for (uint32_t i = 0; i < n; i++) sharded_step(s, sh, dt, s->comm_stream, i < DETAIL_STEPS_MAX);
    s->detail_steps = n < DETAIL_STEPS_MAX ? n : DETAIL_STEPS_MAX;
    if (s->overlap) ASSERT_HIP(hipStreamWaitEvent(s->stream, s->ev_gather, 0), "join comm stream");
}

void enqueue_steps(SimPipeline *s, uint32_t n, float dt) {
    NB_ASSERT(s->on_device, "PerformSimUpdate before SetSimulationData");
    if (s->slots == 0 || n == 0) return;
    use_device();
    s->pool.used = 0;
    s->kernel_iv.clear();
    s->comm_iv.clear();
    s->detail_steps = 0;
    s->host_current = false;
    upload_dt(s, dt);
    if (s->timing) ASSERT_HIP(hipEventRecord(s->ev_begin, s->stream), "record begin");
    if (!s->sharded)
        enqueue_single(s, n, dt);
    else
        enqueue_sharded(s, n, dt);
    if (s->timing) ASSERT_HIP(hipEventRecord(s->ev_end, s->stream), "record end");
    s->timed = s->timing != 0;
    s->timed_launches = (s->sharded && s->overlap) ? 2 * n : n * passes_for(s, whole_step(s, s->cur, dt));
    s->timed_finish_launches = s->last_shape.split > 1 ? s->timed_launches : 0;
    s->data.dt = dt;
}

}  // namespace

// ============================================================================================================
// C-ABI
// ============================================================================================================

extern "C" {

int nb_hip_version(void) { return NB_HIP_VERSION; }

int nb_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

void nb_hip_set_device(int ordinal) {
    NB_ASSERT(!g_dev.ready || g_dev.ordinal == ordinal, "device %d already in use, cannot switch to %d", g_dev.ordinal,
              ordinal);
    g_requested_ordinal = ordinal;
}

void nb_hip_device_info(char *buf, uint32_t len) {
    ensure_device();
    if (buf && len) snprintf(buf, len, "%s", g_dev.info);
}

NbShardPlan nb_hip_shard_plan(uint32_t total_len, uint32_t mass_len, int rank, int nranks) {
    NB_ASSERT(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    NB_ASSERT(mass_len <= total_len, "mass_len %u > total_len %u", mass_len, total_len);
    NbShardPlan p;
    const uint32_t P = (uint32_t)nranks;
    const uint32_t Z = total_len - mass_len;
    // Massive slices: uniform, wave-aligned chunks, so every rank contributes the same count to the all-gather
    // and gathered index == global massive index.  The last ranks may own fewer (or no) real sources.
    const uint32_t Mc = mass_len ? round_up((mass_len + P - 1) / P, 64) : 0;
    auto mass_of = [&](uint32_t q) -> uint32_t {
        const uint64_t b = (uint64_t)q * Mc;
        return b < mass_len ? (mass_len - (uint32_t)b < Mc ? mass_len - (uint32_t)b : Mc) : 0;
    };
    // Massless slices: every receiver costs the same (all sources), so they are dealt out to level the
    // per-rank totals ("water filling"): find the lowest level L with sum_q max(0, L - mass_q) >= Z, give
    // rank q max(0, L - mass_q), and take the surplus back one by one from the highest ranks that got any.
    // An equal total per rank keeps the workgroup count on a round boundary (see choose_shape).
    uint64_t lo = 0, hi = (uint64_t)total_len + 1;
    while (lo < hi) {
        const uint64_t L = (lo + hi) / 2;
        uint64_t got = 0;
        for (uint32_t q = 0; q < P; q++) got += L > mass_of(q) ? L - mass_of(q) : 0;
        if (got >= Z)
            hi = L;
        else
            lo = L + 1;
    }
    const uint64_t level = lo;
    uint64_t surplus = 0;
    for (uint32_t q = 0; q < P; q++) surplus += level > mass_of(q) ? level - mass_of(q) : 0;
    surplus -= Z;
    uint32_t zero_begin = mass_len, zero_max = 0, my_zero_begin = mass_len, my_zero = 0;
    // surplus < number of ranks at the level: rank q gives one back if it is among the last `surplus` takers
    uint32_t takers = 0;
    for (uint32_t q = 0; q < P; q++) takers += level > mass_of(q);
    uint32_t seen = 0;
    for (uint32_t q = 0; q < P; q++) {
        uint32_t z = level > mass_of(q) ? (uint32_t)(level - mass_of(q)) : 0;
        if (level > mass_of(q)) {
            if (seen >= takers - (uint32_t)surplus) z -= 1;
            seen++;
        }
        if (q == (uint32_t)rank) {
            my_zero_begin = zero_begin;
            my_zero = z;
        }
        zero_begin += z;
        zero_max = z > zero_max ? z : zero_max;
    }
    p.mass_chunk = Mc;
    p.zero_chunk = round_up(zero_max, 64);
    const uint64_t mb = (uint64_t)rank * Mc;
    p.mass_begin = mb < mass_len ? (uint32_t)mb : mass_len;
    p.mass_count = mass_of((uint32_t)rank);
    p.zero_begin = my_zero_begin;
    p.zero_count = my_zero;
    p.src_padded = P * Mc;
    return p;
}

SimPipeline *CreateSimPipeline(WorldData data) {
    NB_ASSERT(data.mass_len <= data.total_len, "mass_len %u > total_len %u", data.mass_len, data.total_len);
    SimPipeline *s = new SimPipeline();
    s->data = data;
    s->plan = nb_hip_shard_plan(data.total_len, data.mass_len, 0, 1);
    const char *v = getenv("NB_HIP_VARIANT");
    if (v) s->want_variant = atoi(v) ? nb::VARIANT_SMEM : nb::VARIANT_LDS;
    const char *k = getenv("NB_HIP_K");
    if (k) s->want_k = atoi(k);
    const char *w = getenv("NB_HIP_W");
    if (w) s->want_w = atoi(w);
    const char *ps = getenv("NB_HIP_PASSES");
    if (ps) s->want_passes = atoi(ps);
    const char *un = getenv("NB_HIP_UNIT");
    if (un) s->want_unit = atoi(un);
    const char *sp = getenv("NB_HIP_SPLIT");
    if (sp) s->want_split = atoi(sp);
    const char *rb = getenv("NB_HIP_READBACK");
    if (rb) s->readback = atoi(rb) < 0 || atoi(rb) > 2 ? 2 : atoi(rb);
    const char *zc = getenv("NB_HIP_ZERO_COPY_UPLOAD");
    if (zc) s->zero_copy_upload = atoi(zc) ? 1 : 0;
    const char *tm = getenv("NB_HIP_TIMING");
    if (tm) s->timing = atoi(tm) ? 1 : 0;
    const char *gr = getenv("NB_HIP_GRAPH");
    if (gr) s->use_graph = atoi(gr) < 0 || atoi(gr) > 2 ? 2 : atoi(gr);
    return s;
}

void nb_hip_comm_unique_id(void *out128) {
    NB_ASSERT(out128 != nullptr, "NULL id buffer");
    ncclUniqueId id;
    ASSERT_NCCL(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, &id, NB_HIP_UNIQUE_ID_BYTES);
}

SimPipeline *CreateSimPipelineSharded(WorldData data, int rank, int nranks, const void *unique_id128) {
    NB_ASSERT(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    // one rank normally means the plain pipeline; NB_HIP_FORCE_SHARDED=1 keeps the RCCL path (used to
    // exercise it on a single-GPU box)
    const char *force = getenv("NB_HIP_FORCE_SHARDED");
    if (nranks == 1 && !(force && atoi(force))) return CreateSimPipeline(data);
    NB_ASSERT(unique_id128 != nullptr, "sharded pipeline needs the RCCL unique id");
    SimPipeline *s = CreateSimPipeline(data);
    s->rank = rank;
    s->nranks = nranks;
    s->sharded = true;
    s->plan = nb_hip_shard_plan(data.total_len, data.mass_len, rank, nranks);
    const char *ov = getenv("NB_HIP_OVERLAP");
    if (ov) s->overlap = atoi(ov) ? 1 : 0;
    const char *sg = getenv("NB_HIP_SHARDED_GRAPH");
    if (sg) s->sharded_graph = atoi(sg) ? 1 : 0;
    use_device();  // the communicator binds to the current device
    ncclUniqueId id;
    memcpy(&id, unique_id128, NB_HIP_UNIQUE_ID_BYTES);
    {
        Watchdog dog("ncclCommInitRank", rank, nranks);
        ASSERT_NCCL(rccl().CommInitRank(&s->comm, nranks, id, rank), "ncclCommInitRank(rank %d of %d)", rank, nranks);
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
    {
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
        ASSERT_HIP(hipEventDestroy(e0), "event");
        ASSERT_HIP(hipEventDestroy(e1), "event");
        ASSERT_HIP(hipStreamDestroy(st), "probe stream");
        dev_free(probe);
    }
    return s;
}

SimPipeline *CreateSimPipelineShardedWith(WorldData data, int rank, int nranks, NbAllGatherFn allgather, void *ctx) {
    NB_ASSERT(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    NB_ASSERT(allgather != nullptr, "NULL all-gather callback");
    SimPipeline *s = CreateSimPipeline(data);
    s->rank = rank;
    s->nranks = nranks;
    s->sharded = true;
    s->plan = nb_hip_shard_plan(data.total_len, data.mass_len, rank, nranks);
    s->host_gather = allgather;
    s->host_gather_ctx = ctx;
    const char *ov = getenv("NB_HIP_OVERLAP");
    if (ov) s->overlap = atoi(ov) ? 1 : 0;
    return s;
}

void nb_hip_plan_launch(uint32_t n_recv, uint32_t n_src, int compute_units, int *k, int *w, int *split, uint32_t *workgroups) {
    const nb::LaunchShape sh = nb::choose_shape({0, 0, nb::VARIANT_SMEM, 0, 0}, n_recv, n_src, compute_units);
    if (k) *k = sh.k;
    if (w) *w = sh.w;
    if (split) *split = sh.split;
    if (workgroups) {
        const dim3 g = nb::step_grid(sh, n_recv);
        *workgroups = g.x * g.y;
    }
}

int nb_hip_plan_launch_unit(uint32_t n_recv, uint32_t n_src, int compute_units) {
    return nb::choose_shape({0, 0, nb::VARIANT_SMEM, 0, 0}, n_recv, n_src, compute_units).unit;
}

int nb_hip_local_group_create(WorldData data, int nranks, SimPipeline **out) {
    NB_ASSERT(nranks >= 1 && out != nullptr, "bad local group request");
    use_device();
    LocalGroup *g = new LocalGroup();
    ASSERT_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking), "group stream");
    for (int r = 0; r < nranks; r++) {
        SimPipeline *s = CreateSimPipeline(data);
        s->rank = r;
        s->nranks = nranks;
        s->sharded = true;
        s->group = g;
        s->plan = nb_hip_shard_plan(data.total_len, data.mass_len, r, nranks);
        g->members.push_back(s);
        out[r] = s;
    }
    return nranks;
}

void nb_hip_local_group_step(SimPipeline **sims, int nranks, uint32_t n, float dt) {
    NB_ASSERT(sims != nullptr && nranks >= 1 && sims[0]->group != nullptr, "not a local group");
    LocalGroup *g = sims[0]->group;
    NB_ASSERT((int)g->members.size() == nranks, "group has %zu members, %d passed", g->members.size(), nranks);
    for (int r = 0; r < nranks; r++) NB_ASSERT(sims[r]->on_device, "member %d has no data", r);
    use_device();
    for (int r = 0; r < nranks; r++) upload_dt(sims[r], dt);
    for (uint32_t i = 0; i < n; i++)
        for (int r = 0; r < nranks; r++) {
            SimPipeline *s = sims[r];
            sharded_step(s, resolve_shape(s), dt, g->stream);
        }
    ASSERT_HIP(hipStreamSynchronize(g->stream), "group sync");
}

void DestroySimPipeline(SimPipeline *sim) {
    if (sim == nullptr) return;
    release_device(sim);
    if (sim->group) {
        LocalGroup *g = sim->group;
        for (auto &m : g->members)
            if (m == sim) m = nullptr;
        bool empty = true;
        for (auto m : g->members) empty = empty && m == nullptr;
        if (empty) {
            ASSERT_HIP(hipStreamDestroy(g->stream), "group stream");
            delete g;
        }
    }
    if (sim->comm) ASSERT_NCCL(rccl().CommDestroy(sim->comm), "ncclCommDestroy");
    delete sim;
}

void SetSimulationData(SimPipeline *s, const Particle *ps) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    NB_ASSERT(ps != nullptr || s->data.total_len == 0, "NULL particle array");
    materialize(s);
    const uint32_t N = s->data.total_len, M = s->data.mass_len;
    if (N == 0) return;
    hipStream_t st = s->stream;
    s->cur = 0;
    s->host_current = false;
    s->updates_since_get = 0;
    // The noted, page-locked array is readable from the device: the split kernel pulls the records over PCIe itself
    // (one launch) instead of a DMA copy into the device staging followed by the kernel (two submissions' latency).
    const bool zero_copy = !s->sharded && ps == s->host_array && s->host_dev != nullptr && s->zero_copy_upload &&
                           s->host_bytes >= (size_t)N * sizeof(Particle);
    if (!zero_copy)
        ASSERT_HIP(hipMemcpyAsync(s->aos, ps, (size_t)N * sizeof(Particle), hipMemcpyHostToDevice, st), "H2D of %u particles", N);
    if (!s->sharded) {
        nb::launch_split(st, zero_copy ? s->host_dev : s->aos, 0, N, s->pos[0], s->vel, s->acc, s->radius, s->mass, 0);
        nb::launch_make_gm(st, s->mass, s->src_gm, M);
    } else {
        const NbShardPlan &pl = s->plan;
        // receivers: [0, Mc) this rank's massive slice (tail padded), [Mc, Mc+Zc) its massless slice
        nb::launch_fill_pad(st, s->pos[0], s->vel, s->acc, s->radius, s->mass, 0, s->slots);
        nb::launch_split(st, s->aos, pl.mass_begin, pl.mass_count, s->pos[0], s->vel, s->acc, s->radius, s->mass, 0);
        nb::launch_split(st, s->aos, pl.zero_begin, pl.zero_count, s->pos[0], s->vel, s->acc, s->radius, s->mass,
                         pl.mass_chunk);
        // sources: every rank holds the whole world in `aos`, so the first gathered array and the static
        // G*m need no communication: rank q's slice is aos[q*Mc ..) padded
        nb::launch_split_sources(st, s->aos, M, s->n_src, s->src_pos[0], s->src_pos[1], s->src_gm);
        if (s->overlap) ASSERT_HIP(hipEventRecord(s->ev_gather, s->group ? s->stream : s->comm_stream), "prime gather event");
    }
    ASSERT_HIP(hipStreamSynchronize(st), "sync after SetSimulationData");
    // small worlds in auto mode: have the canonical chain ready before the first step call (see wants_canonical)
    if (wants_canonical(s)) (void)find_or_build_graph(s, CANON_STEPS, 0.0f, resolve_shape(s));
}

void GetSimulationData(const SimPipeline *cs, Particle *ps) {
    SimPipeline *s = const_cast<SimPipeline *>(cs);
    NB_ASSERT(s != nullptr && ps != nullptr, "NULL argument");
    NB_ASSERT(s->on_device, "GetSimulationData before SetSimulationData");
    const uint32_t N = s->data.total_len;
    if (N == 0) return;
    if (!s->sharded) {
        if (ps == s->host_array && s->updates_since_get == 1)
            s->frame_streak++;  // one update, then a Get into the noted array: a frame
        else
            s->frame_streak = 0;
        s->updates_since_get = 0;
        if (s->host_current) {
            // the update's own submission already stored this state into the noted array (eager read-back)
            if (ps != s->host_array) memcpy(ps, s->host_array, (size_t)N * sizeof(Particle));
            return;
        }
    }
    use_device();
    hipStream_t st = s->stream;
    if (!s->sharded) {
        nb::launch_merge(st, s->aos, 0, N, s->pos[s->cur], s->vel, s->acc, s->radius, s->mass, 0);
    } else {
        // every rank merges its slots, the slices are all-gathered (uniform size), then unpacked into
        // partitioned order: massive slices first, massless slices after them
        const NbShardPlan &pl = s->plan;
        Particle *shard = static_cast<Particle *>(s->aos_shard);
        Particle *mine = shard + (size_t)s->rank * s->slots;
        if (s->group) {
            for (SimPipeline *q : s->group->members)
                nb::launch_merge(st, shard + (size_t)q->rank * s->slots, 0, q->slots, q->pos[q->cur], q->vel, q->acc,
                                 q->radius, q->mass, 0);
        } else {
            nb::launch_merge(st, mine, 0, s->slots, s->pos[s->cur], s->vel, s->acc, s->radius, s->mass, 0);
            const size_t floats = (size_t)s->slots * (sizeof(Particle) / sizeof(float));
            if (s->host_gather)
                host_allgather(s, shard, floats * sizeof(float), st);
            else
                ASSERT_NCCL(rccl().AllGather(mine, shard, floats, NCCL_FLOAT32, s->comm, st),
                            "ncclAllGather of particle slices");
        }
        for (int q = 0; q < s->nranks; q++) {
            const NbShardPlan pq = nb_hip_shard_plan(N, s->data.mass_len, q, s->nranks);
            const Particle *from = shard + (size_t)q * s->slots;
            if (pq.mass_count)
                ASSERT_HIP(hipMemcpyAsync(static_cast<Particle *>(s->aos) + pq.mass_begin, from,
                                          (size_t)pq.mass_count * sizeof(Particle), hipMemcpyDeviceToDevice, st),
                           "unpack massive slice of rank %d", q);
            if (pq.zero_count)
                ASSERT_HIP(hipMemcpyAsync(static_cast<Particle *>(s->aos) + pq.zero_begin, from + pl.mass_chunk,
                                          (size_t)pq.zero_count * sizeof(Particle), hipMemcpyDeviceToDevice, st),
                           "unpack massless slice of rank %d", q);
        }
    }
    ASSERT_HIP(hipMemcpyAsync(ps, s->aos, (size_t)N * sizeof(Particle), hipMemcpyDeviceToHost, st), "D2H of %u particles", N);
    ASSERT_HIP(hipStreamSynchronize(st), "sync after GetSimulationData");
}

void nb_hip_step_async(SimPipeline *s, uint32_t n, float dt) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    enqueue_steps(s, n, dt);
}

void nb_hip_sync(SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (!s->on_device) return;
    use_device();
    ASSERT_HIP(hipStreamSynchronize(s->stream), "stream sync");
    if (s->comm_stream) ASSERT_HIP(hipStreamSynchronize(s->comm_stream), "comm stream sync");
}

void PerformSimUpdate(SimPipeline *s, uint32_t n, float dt) {
    nb_hip_step_async(s, n, dt);
    if (n > 0 && s->on_device && !s->sharded && s->slots > 0) {
        // an update that follows an update (no Get in between) ends a frame-loop streak
        if (s->updates_since_get > 0) s->frame_streak = 0;
        s->updates_since_get++;
        const bool eager = s->host_dev != nullptr && s->host_bytes >= (size_t)s->data.total_len * sizeof(Particle) &&
                           (s->readback == 1 || (s->readback == 2 && s->frame_streak >= 2));
        if (eager) {
            nb::launch_merge(s->stream, s->host_dev, 0, s->data.total_len, s->pos[s->cur], s->vel, s->acc, s->radius, s->mass, 0);
            s->host_current = true;  // true once the wait below returns
        }
    }
    nb_hip_sync(s);
}

double nb_hip_last_step_ms(SimPipeline *s, uint32_t *launches) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (launches) *launches = 0;
    if (!s->on_device || !s->timed) return 0.0;
    ASSERT_HIP(hipEventSynchronize(s->ev_end), "event sync");
    float ms = 0.0f;
    ASSERT_HIP(hipEventElapsedTime(&ms, s->ev_begin, s->ev_end), "hipEventElapsedTime");
    if (launches) *launches = s->timed_launches;
    return (double)ms;
}

uint32_t nb_hip_last_finish_launches(const SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    return s->timed ? s->timed_finish_launches : 0;
}

uint32_t nb_hip_last_step_breakdown(SimPipeline *s, double *kernel_ms, double *comm_ms) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (kernel_ms) *kernel_ms = 0.0;
    if (comm_ms) *comm_ms = 0.0;
    if (!s->on_device || !s->timed || s->detail_steps == 0) return 0;
    use_device();
    ASSERT_HIP(hipStreamSynchronize(s->stream), "stream sync");
    if (s->comm_stream) ASSERT_HIP(hipStreamSynchronize(s->comm_stream), "comm stream sync");
    auto total = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>> &iv) {
        double sum = 0.0;
        for (const auto &p : iv) {
            float ms = 0.0f;
            ASSERT_HIP(hipEventElapsedTime(&ms, p.first, p.second), "hipEventElapsedTime");
            sum += (double)ms;
        }
        return sum;
    };
    if (kernel_ms) *kernel_ms = total(s->kernel_iv);
    if (comm_ms) *comm_ms = total(s->comm_iv);
    return s->detail_steps;
}

int nb_hip_comm_info(const SimPipeline *s, int *nranks, int *rank, int *device, int *rccl_version, double *first_gather_ms,
                     char *lib_path, uint32_t len) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (nranks) *nranks = s->nranks;
    if (rank) *rank = s->rank;
    if (device) *device = g_dev.ordinal;
    if (rccl_version) *rccl_version = 0;
    if (first_gather_ms) *first_gather_ms = s->first_gather_ms;
    if (lib_path && len) snprintf(lib_path, len, "%s", s->host_gather ? "caller-supplied host all-gather" : s->group ? "local group" : "");
    if (s->comm == nullptr) return 0;  // unsharded, a local-group member or a caller-supplied transport: no communicator
    // everything below is what the COMMUNICATOR says, not what the caller passed at creation
    if (nranks) ASSERT_NCCL(rccl().CommCount(s->comm, nranks), "ncclCommCount");
    if (rank) ASSERT_NCCL(rccl().CommUserRank(s->comm, rank), "ncclCommUserRank");
    if (device) ASSERT_NCCL(rccl().CommCuDevice(s->comm, device), "ncclCommCuDevice");
    if (rccl_version) ASSERT_NCCL(rccl().GetVersion(rccl_version), "ncclGetVersion");
    if (lib_path && len) snprintf(lib_path, len, "%s", rccl().path);
    return 1;
}

uint32_t nb_hip_graph_stats(const SimPipeline *s, uint32_t *dt_uploads) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (dt_uploads) *dt_uploads = s->dt_uploads;
    return (uint32_t)s->graphs.size();
}

int nb_hip_launch_unit(const SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    return s->last_shape.unit;
}

int nb_hip_runtime_version(void) {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) return 0;
    return v;
}

void nb_hip_note_host_array(SimPipeline *s, void *array, uint64_t bytes) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (s->on_device) use_device();
    if (s->on_device) ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before re-registering the host array");
    unpin_host(s);
    s->host_array = array;
    s->host_bytes = (size_t)bytes;
    if (s->on_device) pin_host(s);
}

int nb_hip_configure(SimPipeline *s, const char *key, int value) {
    NB_ASSERT(s != nullptr && key != nullptr, "NULL argument");
    int old = 0;
    if (!strcmp(key, "variant")) {
        NB_ASSERT(value == 0 || value == 1, "variant must be 0 (lds) or 1 (smem), got %d", value);
        old = s->want_variant;
        s->want_variant = value;
    } else if (!strcmp(key, "k")) {
        NB_ASSERT(value == 0 || value == 1 || value == 2 || value == 4, "k must be 0, 1, 2 or 4, got %d", value);
        old = s->want_k;
        s->want_k = value;
    } else if (!strcmp(key, "w")) {
        NB_ASSERT(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16,
                  "w must be 0, 1, 2, 4, 8 or 16, got %d", value);
        old = s->want_w;
        s->want_w = value;
    } else if (!strcmp(key, "split")) {
        NB_ASSERT(value >= 0 && value <= nb::MAX_SPLIT, "split must be 0 (auto) .. %d, got %d", nb::MAX_SPLIT, value);
        old = s->want_split;
        s->want_split = value;
    } else if (!strcmp(key, "unit")) {
        NB_ASSERT(value == 0 || value == 8 || value == 16 || value == 32 || value == 64, "unit must be 0, 8, 16, 32 or 64, got %d", value);
        old = s->want_unit;
        s->want_unit = value;
    } else if (!strcmp(key, "graph")) {
        NB_ASSERT(value >= 0 && value <= 2, "graph must be 0 (never), 1 (always) or 2 (from the second use), got %d", value);
        old = s->use_graph;
        s->use_graph = value;
    } else if (!strcmp(key, "passes")) {
        NB_ASSERT(value >= 0 && value <= 64, "passes must be 0 (auto) .. 64, got %d", value);
        old = s->want_passes;
        s->want_passes = value;
    } else if (!strcmp(key, "readback")) {
        NB_ASSERT(value >= 0 && value <= 2, "readback must be 0 (lazy), 1 (eager) or 2 (auto), got %d", value);
        old = s->readback;
        s->readback = value;
        s->frame_streak = 0;
    } else if (!strcmp(key, "zero_copy_upload")) {
        old = s->zero_copy_upload;
        s->zero_copy_upload = value ? 1 : 0;
    } else if (!strcmp(key, "timing")) {
        old = s->timing;
        s->timing = value ? 1 : 0;
    } else if (!strcmp(key, "sharded_graph")) {
        old = s->sharded_graph;
        s->sharded_graph = value ? 1 : 0;
    } else if (!strcmp(key, "overlap")) {
        old = s->overlap;
        if (s->on_device && s->sharded) nb_hip_sync(s);
        s->overlap = value ? 1 : 0;
        if (s->on_device && s->overlap && s->sharded)
            ASSERT_HIP(hipEventRecord(s->ev_gather, s->group ? s->stream : s->comm_stream), "prime gather event");
    } else {
        NB_FAIL("unknown knob \"%s\"", key);
    }
    return old;
}

void nb_hip_launch_shape(const SimPipeline *s, int *k, int *w, int *variant, int *split, uint32_t *workgroups) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (split) *split = s->last_shape.split;
    if (k) *k = s->last_shape.k;
    if (w) *w = s->last_shape.w;
    if (variant) *variant = s->last_shape.variant;
    if (workgroups) *workgroups = s->last_groups;
}

}  // extern "C"
