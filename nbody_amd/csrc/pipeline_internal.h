// pipeline_internal.h -- what the translation units behind include/nbody_hip.h share (C++, not part of the C-ABI):
//
//   device_ctx.hip   the process-wide device context (the reference keeps one global vulkan_ctx, vulkan_ctx.c:11)
//   rccl_bind.hip    RCCL bound lazily with dlopen, the communicator of a sharded pipeline, the watchdog
//   shard_plan.hip   nb_hip_shard_plan: who owns which receivers and sources (pure host arithmetic)
//   step_chain.hip   one step / n steps as launches, cached hipGraph chains, the sharded step with its all-gather
//   pipeline.hip     SimPipeline life cycle and the C-ABI entry points (Create/Destroy/Set/Get/PerformSimUpdate, knobs)
//   kernels.hip      the gfx950 kernels (kernels.h)
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "kernels.h"
#include "nbody_hip.h"

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

typedef struct ncclComm *ncclComm_t;  // opaque; rccl_bind.hip holds the rest of the binding

namespace nbi {

// ---- device_ctx.hip ------------------------------------------------------------------------------------------

struct DeviceCtx {
    bool ready = false;
    int ordinal = -1;  // -1: not chosen yet
    int compute_units = 0;
    char info[256] = {0};
};
extern DeviceCtx g_dev;
extern int g_requested_ordinal;

void ensure_device();  // first touch of the GPU by this process; aborts unless a gfx950 device answers (no CPU fallback)
void use_device();     // ensure_device + hipSetDevice on the calling thread (HIP's current device is per THREAD)
void *dev_alloc_bytes(size_t bytes);
void dev_free(void *p);

// libc's rand() stream belongs to the caller: the reference harness seeds it ONCE and draws every universe of its table
// from it (src/bench.c:42,53), between GPU calls.  The HIP runtime's first stream / allocation / code-object set-up
// draws from (or reseeds) that same process-global state -- measured: tools/rand_probe.py, profiles/r04_rand_probe.txt --
// so the first device set-up of a pipeline and RCCL's bootstrap run with a private state swapped in.
// initstate / setstate swap ONE process-global pointer, so two guards that each saved "the previous state" would hand each
// other's private state back to the caller.  The guard is therefore reference-counted process-wide: the FIRST guard to
// come alive saves the caller's state and installs the private one, the LAST to go restores it; the mutex is held only
// for that bookkeeping, never across the guarded region.  That matters because a guarded region can WAIT ON OTHER RANKS
// (the direct exchange's IPC-handle hand-over and "ready" barrier run through the caller's all-gather callback inside the
// first SetSimulationData; ncclCommInitRank blocks until every rank has joined): ranks driven from threads of one process
// would deadlock on a lock held across such a wait (round 5 held one).  Guards nest freely on one thread.
// A caller's own rand() racing a live guard on another thread is the caller's race: libc's rand() stream is one
// process-global object to begin with (tests/test_abi.py pins the two-thread and the wait-inside-a-guard cases).
class RandGuard {
  public:
    RandGuard() {
        std::lock_guard<std::mutex> l(gate());
        if (alive()++ == 0) saved() = initstate(1u, private_state(), 128);
    }
    ~RandGuard() {
        std::lock_guard<std::mutex> l(gate());
        if (--alive() == 0 && saved()) {
            setstate(saved());
            saved() = nullptr;
        }
    }
    RandGuard(const RandGuard &) = delete;
    RandGuard &operator=(const RandGuard &) = delete;

  private:
    static std::mutex &gate() {
        static std::mutex m;
        return m;
    }
    static int &alive() {   // guards alive in the process, any thread; guarded by gate()
        static int n = 0;
        return n;
    }
    static char *&saved() {   // the caller's state while alive() > 0
        static char *p = nullptr;
        return p;
    }
    static char *private_state() {
        static char buf[128];
        return buf;
    }
};

template <typename T>
T *dev_alloc(size_t count) {
    return static_cast<T *>(dev_alloc_bytes((count ? count : 1) * sizeof(T)));
}

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// ---- rccl_bind.hip -------------------------------------------------------------------------------------------

// A collective that never completes (a rank that died, a fabric that does not come up) must not hang the job: every
// call that waits on other ranks -- ncclCommInitRank, the first all-gather, and the blocking waits of a sharded
// pipeline (PerformSimUpdate's sync, the collective GetSimulationData) -- runs under a watchdog that prints what was
// being waited for, the tail of RCCL's own log when NCCL_DEBUG_FILE names one, and _exit(3)s.  No retry and no
// re-exec: the process has initialised the GPU.  NB_HIP_COMM_TIMEOUT_S (default 180) sets the bound; 0 disables it.
// One long-lived thread per process does the watching; a Watchdog object only adds its deadline to the watcher's list and
// takes it off again (a mutex and a notify: a frame loop of short sharded calls does not pay a thread spawn + join per
// sync).  Waits nest per THREAD (an inner wait keeps the outer deadline); waits on different threads are each watched.
class Watchdog {
  public:
    Watchdog(const char *what, int rank, int nranks);
    ~Watchdog();
    Watchdog(const Watchdog &) = delete;
    Watchdog &operator=(const Watchdog &) = delete;

  private:
    uint64_t seq_ = 0;   // 0: not armed by this object (disabled, or an outer wait of the same thread is being watched)
};

// ncclCommInitRank + cross-check of the communicator's own rank count + a verified probe all-gather, all bounded
void comm_create(::SimPipeline *s, const void *unique_id128);
void comm_destroy(::SimPipeline *s);
// in-place ncclAllGather of `count` floats per rank (sendbuff = recvbuff + rank * count) on `st`
void comm_allgather_f32(::SimPipeline *s, void *base, size_t count_per_rank, hipStream_t st, const char *what);

// ---- a cached chain of step launches ---------------------------------------------------------------------------

struct StepGraph {
    uint32_t n = 0;           // steps in the chain
    uint32_t passes = 1;      // source passes per step
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;  // one kernel node per launch, in order
    std::vector<nb::StepParams> params; // what each node currently holds
    int phase = -1;                     // which pos buffer the chain reads first
    nb::LaunchShape shape = {0, 0, 0, 0, 0, 0, 0};
    uint64_t last_use = 0;              // for eviction: the cache holds at most GRAPH_CACHE_MAX chains
};

// Event pairs around the kernels and the gathers of a sharded chain (the plain-launch path, "timing" knob on), so that
// a multi-GPU run can say how much of a step was the all-gather.  Grown on demand, reused by every call.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t next();
    void destroy();
};

}  // namespace nbi

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
    // stages each exchange through one page-locked buffer (D2H own slot, wait, callback, H2D the other ranks' slots)
    NbAllGatherFn host_gather = nullptr;
    void *host_gather_ctx = nullptr;
    void *stage = nullptr;        // page-locked staging, max(gathered sources, gathered particle slices) bytes
    size_t stage_bytes = 0;
    // direct transport (CreateSimPipelineShardedDirect): the per-step exchange is P - 1 device-to-device copies of this
    // rank's slice straight into every peer's gathered array (mapped with hipIpcOpenMemHandle: over xGMI each copy rides
    // its own link), then a host-side step barrier through the caller's callback.  peer_src[b][q] = rank q's src_pos[b]
    // as seen from this process (own entry = the local array).
    bool direct = false;
    std::vector<float2 *> peer_src[2];
    uint64_t direct_steps = 0;    // exchanges done: the ranks compare it at every barrier

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
    // (step + wait, then merge + D2H copy + wait).  Frozen since round 2: the GUI it serves is out of scope.
    int readback = 2;             // 0 never, 1 after every blocking update, 2 auto (after two update->Get pairs in a row)
    int zero_copy_upload = 1;     // SetSimulationData from the noted array: the split kernel reads host memory directly
    bool host_current = false;    // the noted array already holds the device's latest state
    uint32_t updates_since_get = 0, frame_streak = 0;
    // record the ev_begin / ev_end pair around every chain (nb_hip_last_step_ms) and, on the sharded plain-launch path,
    // the per-step kernel / gather event pairs (nb_hip_last_step_breakdown).  Off unless asked for: the records cost an
    // interactive caller 3-7 us per call (profiles/r02_frame_loop_latency.txt).
    int timing = 0;
    float2 *parts = nullptr;    // split steps only: [split][n_real] partial sums
    uint32_t parts_cap = 0;     // float2 elements allocated in parts
    uint32_t *tickets = nullptr;  // fused finish: one arrival counter per receiver tile, zero between launches (re-zeroed at every upload
    uint32_t tickets_len = 0;     // and before a chain is built: a launch that ended part-way must not leave a tile unfinished for ever)
    int fused_finish = 2;         // 0: step kernel + finish kernel; 1: the last workgroup of a tile finishes it, whenever
                                  // the shape allows; 2 (default): auto (step_chain.hip fused_finish_rule)
    int cur = 0;             // pos[cur] is the latest state

    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_local = nullptr, ev_gather = nullptr;
    bool timed = false;
    uint32_t timed_launches = 0;         // step-kernel launches between ev_begin and ev_end
    uint32_t timed_finish_launches = 0;  // finish-kernel launches in the same interval (split shapes only)
    // sharded plain-launch chains: [begin, end) event pairs of each step's kernels and of each gather
    nbi::EventPool pool;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kernel_iv, comm_iv;
    uint32_t detail_steps = 0;  // steps the intervals above cover (capped)
    uint64_t use_clock = 0;     // ticks once per graph lookup (LRU)

    // knobs
    int want_variant = nb::VARIANT_SMEM, want_k = 0, want_w = 0, want_split = 0;  // the scalar-cache route is 8 % faster than LDS tiles at N = 2^20 (roofline.alt_lds)
    int use_graph = 2, overlap = 0, sharded_graph = 0;  // use_graph: 0 never, 1 always, 2 from a chain length's second use
    int fused_chain = 2;                                // one-launch n-step chains for one-workgroup worlds: 0 never, 2 auto
    std::vector<uint32_t> seen_chains;                  // chain lengths already run once as plain launches
    int want_passes = 0;  // source passes per step (0 = auto: keep each pass's sources within one XCD's L2)
    double first_gather_ms = 0.0;  // sharded: device time of the probe all-gather at creation (includes lazy setup)
    double comm_init_ms = 0.0;     // sharded over RCCL: host time of ncclCommInitRank
    double small_gather_us = 0.0;  // sharded over RCCL: device time of one warm 8-byte-per-rank all-gather (mean of 16 in-stream)
    nb::LaunchShape last_shape = {0, 0, 0, 0, 0, 0, 0};
    int want_unit = 0;  // source-slice granule: 0 = auto, else 64 / 32 / 16 / 8
    int want_persist = 0;  // experiment: work items per workgroup of a persistent launch (0 / 1 = classic)
    int want_lanes = 0; // lane groups per wave: 0 = auto, 1 = never, 2 / 4 = lane-split kernel (kernels.h LaunchShape::lanes)
    uint32_t last_groups = 0;
    uint32_t fused_steps = 0;   // steps the last update ran inside fused (one-launch) chains

    std::vector<nbi::StepGraph> graphs;
};

namespace nbi {

// ---- step_chain.hip ------------------------------------------------------------------------------------------

constexpr uint32_t CANON_STEPS = 32;  // length of the prebuilt chain of small worlds (see wants_canonical)
constexpr uint32_t DETAIL_STEPS_MAX = 256;  // steps per call whose kernels / gathers get their own event pairs

void destroy_graph(StepGraph &g);
nb::LaunchShape resolve_shape(SimPipeline *s);
bool wants_canonical(const SimPipeline *s);
StepGraph *find_or_build_graph(SimPipeline *s, uint32_t n, float dt, nb::LaunchShape sh);
void upload_dt(SimPipeline *s, float dt);
void zero_tickets(SimPipeline *s);   // fused finish: all tile tickets back to 0, in stream order
// one sharded step of one rank; `cs` carries the gather (the comm stream with RCCL; the group stream locally)
void sharded_step(SimPipeline *s, nb::LaunchShape sh, float dt, hipStream_t cs, bool detail = false);
// in-place all-gather of a device array of nranks slots through the caller's host transport
void host_allgather(SimPipeline *s, void *dev_base, size_t bytes_per_rank, hipStream_t st);
void enqueue_steps(SimPipeline *s, uint32_t n, float dt);  // what PerformSimUpdate / nb_hip_step_async enqueue
bool fused_finish_rule(uint32_t n_recv, uint32_t n_src);   // the auto rule of the "fused_finish" knob (pure host)

}  // namespace nbi
