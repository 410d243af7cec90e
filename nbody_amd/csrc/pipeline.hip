// pipeline.hip -- SimPipeline life cycle and the C-ABI entry points of include/nbody_hip.h.
//
// Stands where the reference has src/lib/sim_gpu.c (SimPipeline creation, staging, Get/Set/Perform).  Differences by
// design:
//   * SoA in HBM (float2 pos/vel/acc, float radius/mass, float G*m) instead of 32-byte AoS records;
//   * device-to-host copy only when GetSimulationData asks (the reference copies after every call, sim_gpu.c:336-341);
//   * nothing is created on the GPU until SetSimulationData, so CPU-only worlds never touch a device.
// The rest of the seam lives next door: device_ctx.hip (device pick), step_chain.hip (what a step call enqueues),
// rccl_bind.hip + shard_plan.hip (the sharded pipeline's communicator and ownership plan), kernels.hip (gfx950 code).
#include "pipeline_internal.h"
#include "nbody_hip_tuning.h"

using namespace nbi;

namespace {

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

void release_device(SimPipeline *s) {
    if (!s->on_device) return;
    use_device();
    ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before release");
    if (s->comm_stream) ASSERT_HIP(hipStreamSynchronize(s->comm_stream), "sync before release");
    if (s->direct) {
        // every rank has finished writing into its peers before anyone unmaps or frees (DestroySimPipeline is collective
        // for direct pipelines); then the mappings go, then -- below -- the memory they pointed at
        uint64_t tag[64] = {0};
        tag[s->rank < 64 ? s->rank : 0] = ~0ull;
        s->host_gather(s->host_gather_ctx, tag, sizeof(uint64_t), s->rank, s->nranks);
        for (int b = 0; b < 2; b++) {
            for (int q = 0; q < (int)s->peer_src[b].size(); q++)
                if (q != s->rank && s->peer_src[b][q]) ASSERT_HIP(hipIpcCloseMemHandle(s->peer_src[b][q]), "hipIpcCloseMemHandle(rank %d)", q);
            s->peer_src[b].clear();
        }
        tag[s->rank < 64 ? s->rank : 0] = ~1ull;
        s->host_gather(s->host_gather_ctx, tag, sizeof(uint64_t), s->rank, s->nranks);
    }
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
    dev_free(s->tickets);
    s->tickets = nullptr;
    s->tickets_len = 0;
    ASSERT_HIP(hipEventDestroy(s->ev_begin), "event");
    ASSERT_HIP(hipEventDestroy(s->ev_end), "event");
    ASSERT_HIP(hipEventDestroy(s->ev_local), "event");
    ASSERT_HIP(hipEventDestroy(s->ev_gather), "event");
    if (s->comm_stream) ASSERT_HIP(hipStreamDestroy(s->comm_stream), "stream");
    if (!s->group) ASSERT_HIP(hipStreamDestroy(s->stream), "stream");
    s->on_device = false;
}

void set_simulation_data(SimPipeline *s, const Particle *ps);

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
        if (s->direct) {
            // every rank publishes IPC handles of its two gathered arrays and maps everybody else's
            NB_ASSERT(s->nranks <= 64, "direct transport: at most 64 ranks");
            struct Handles {
                hipIpcMemHandle_t h[2];
            };
            std::vector<Handles> all((size_t)s->nranks);
            for (int b = 0; b < 2; b++)
                ASSERT_HIP(hipIpcGetMemHandle(&all[(size_t)s->rank].h[b], s->src_pos[b]), "hipIpcGetMemHandle of the gathered array %d", b);
            s->host_gather(s->host_gather_ctx, all.data(), sizeof(Handles), s->rank, s->nranks);
            for (int b = 0; b < 2; b++) {
                s->peer_src[b].assign((size_t)s->nranks, nullptr);
                for (int q = 0; q < s->nranks; q++) {
                    if (q == s->rank) {
                        s->peer_src[b][(size_t)q] = s->src_pos[b];
                        continue;
                    }
                    void *mapped = nullptr;
                    ASSERT_HIP(hipIpcOpenMemHandle(&mapped, all[(size_t)q].h[b], hipIpcMemLazyEnablePeerAccess),
                               "hipIpcOpenMemHandle of rank %d's gathered array %d", q, b);
                    s->peer_src[b][(size_t)q] = static_cast<float2 *>(mapped);
                }
            }
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

}  // namespace

extern "C" {

SimPipeline *CreateSimPipeline(WorldData data) {
    NB_ASSERT(data.mass_len <= data.total_len, "mass_len %u > total_len %u", data.mass_len, data.total_len);
    SimPipeline *s = new SimPipeline();
    s->data = data;
    s->plan = nb_hip_shard_plan(data.total_len, data.mass_len, 0, 1);
    // presets from the environment: the knobs of include/nbody_hip.h only; launch-shape and experiment knobs can be preset
    // in TUNING=1 builds (tools/ sweeps), never in the library that ships
    const char *v = getenv("NB_HIP_VARIANT");
    if (v) s->want_variant = atoi(v) ? nb::VARIANT_SMEM : nb::VARIANT_LDS;
    const char *gr = getenv("NB_HIP_GRAPH");
    if (gr) s->use_graph = atoi(gr) < 0 || atoi(gr) > 2 ? 2 : atoi(gr);
#ifdef NB_TUNING_SHAPES
    struct { const char *env, *key; } presets[] = {
        {"NB_HIP_K", "k"}, {"NB_HIP_W", "w"}, {"NB_HIP_SPLIT", "split"}, {"NB_HIP_UNIT", "unit"}, {"NB_HIP_PASSES", "passes"},
        {"NB_HIP_LANES", "lanes"}, {"NB_HIP_FUSED_CHAIN", "fused_chain"}, {"NB_HIP_FUSED_FINISH", "fused_finish"},
        {"NB_HIP_READBACK", "readback"}, {"NB_HIP_ZERO_COPY_UPLOAD", "zero_copy_upload"},
    };
    for (const auto &p : presets)
        if (const char *e = getenv(p.env)) nb_hip_tune(s, p.key, atoi(e));
    if (const char *tm = getenv("NB_HIP_TIMING")) s->timing = atoi(tm) ? 1 : 0;
#endif
    return s;
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
    comm_create(s, unique_id128);  // ncclCommInitRank + a verified probe all-gather, both under the watchdog
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

SimPipeline *CreateSimPipelineShardedDirect(WorldData data, int rank, int nranks, NbAllGatherFn control, void *ctx) {
    SimPipeline *s = CreateSimPipelineShardedWith(data, rank, nranks, control, ctx);
    s->direct = true;
    return s;
}

void nb_hip_plan_launch(uint32_t n_recv, uint32_t n_src, int compute_units, int *k, int *w, int *split, uint32_t *workgroups) {
    const nb::LaunchShape sh = nb::choose_shape({0, 0, nb::VARIANT_SMEM, 0, 0, 1, 0}, n_recv, n_src, compute_units);
    if (k) *k = sh.k;
    if (w) *w = sh.w;
    if (split) *split = sh.split;
    if (workgroups) {
        const dim3 g = nb::step_grid(sh, n_recv);
        *workgroups = g.x * g.y;
    }
}

int nb_hip_plan_launch_lanes(uint32_t n_recv, uint32_t n_src, int *w) {
    int ww = 16;
    const int lanes = n_src <= nb::LANE_SPLIT_MAX_SRC ? nb::lane_split_rule(n_recv, n_src, &ww) : 1;
    if (w) *w = ww;
    return lanes;
}

int nb_hip_plan_fused_finish(uint32_t n_recv, uint32_t n_src, int compute_units) {
    const nb::LaunchShape sh = nb::choose_shape({0, 0, nb::VARIANT_SMEM, 0, 0, 1, 0}, n_recv, n_src, compute_units);
    return sh.split > 1 && nb_hip_plan_launch_lanes(n_recv, n_src, nullptr) <= 1 && fused_finish_rule(n_recv, n_src) ? 1 : 0;
}

int nb_hip_plan_launch_unit(uint32_t n_recv, uint32_t n_src, int compute_units) {
    return nb::choose_shape({0, 0, nb::VARIANT_SMEM, 0, 0, 1, 0}, n_recv, n_src, compute_units).unit;
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
    for (int r = 0; r < nranks; r++) {
        SimPipeline *s = sims[r];
        s->pool.used = 0;
        s->kernel_iv.clear();
        s->comm_iv.clear();
        s->detail_steps = 0;
        upload_dt(s, dt);
    }
    // with the "timing" knob every member's kernels and pushes get their own event pairs, so that
    // nb_hip_last_step_breakdown(member) says what ONE rank's shard step costs (bench.py S2 / S4 / S8)
    for (uint32_t i = 0; i < n; i++)
        for (int r = 0; r < nranks; r++) {
            SimPipeline *s = sims[r];
            sharded_step(s, resolve_shape(s), dt, g->stream, s->timing != 0 && i < DETAIL_STEPS_MAX);
        }
    for (int r = 0; r < nranks; r++) {
        SimPipeline *s = sims[r];
        s->detail_steps = s->timing ? (n < DETAIL_STEPS_MAX ? n : DETAIL_STEPS_MAX) : 0;
        s->timed = false;   // no whole-chain event pair for group members; the breakdown has its own
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
    comm_destroy(sim);
    delete sim;
}

void SetSimulationData(SimPipeline *s, const Particle *ps) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    NB_ASSERT(ps != nullptr || s->data.total_len == 0, "NULL particle array");
    if (!s->on_device) {
        // first touch: streams, HBM, page-locking, code objects -- none of it may move the caller's rand() stream
        RandGuard keep_callers_rand_stream;
        materialize(s);
        set_simulation_data(s, ps);
        return;
    }
    use_device();
    set_simulation_data(s, ps);
}

}  // extern "C"

namespace {

void set_simulation_data(SimPipeline *s, const Particle *ps) {
    const uint32_t N = s->data.total_len, M = s->data.mass_len;
    if (N == 0) return;
    hipStream_t st = s->stream;
    s->cur = 0;
    s->host_current = false;
    s->updates_since_get = 0;
    zero_tickets(s);   // a new state starts from clean tile tickets whatever the previous launches did
    // The noted, page-locked array is readable from the device: the split kernel pulls the records over PCIe itself
    // (one launch) instead of a DMA copy into the device staging followed by the kernel (two submissions' latency).
    const bool zero_copy = !s->sharded && ps == s->host_array && s->host_dev != nullptr && s->zero_copy_upload &&
                           s->host_bytes >= (size_t)N * sizeof(Particle);
    if (!zero_copy)
        ASSERT_HIP(hipMemcpyAsync(s->aos, ps, (size_t)N * sizeof(Particle), hipMemcpyHostToDevice, st), "H2D of %u particles", N);
    if (!s->sharded) {
        nb::launch_split(st, zero_copy ? s->host_dev : s->aos, 0, N, s->pos[0], s->vel, s->acc, s->radius, s->mass, 0);
        nb::launch_make_gm(st, s->mass, s->src_gm, M, NB_G);
    } else {
        const NbShardPlan &pl = s->plan;
        // receivers: [0, Mc) this rank's massive slice (tail padded), [Mc, Mc+Zc) its massless slice
        nb::launch_fill_pad(st, s->pos[0], s->vel, s->acc, s->radius, s->mass, 0, s->slots);
        nb::launch_split(st, s->aos, pl.mass_begin, pl.mass_count, s->pos[0], s->vel, s->acc, s->radius, s->mass, 0);
        nb::launch_split(st, s->aos, pl.zero_begin, pl.zero_count, s->pos[0], s->vel, s->acc, s->radius, s->mass,
                         pl.mass_chunk);
        // sources: every rank holds the whole world in `aos`, so the first gathered array and the static
        // G*m need no communication: rank q's slice is aos[q*Mc ..) padded
        nb::launch_split_sources(st, s->aos, M, s->n_src, s->src_pos[0], s->src_pos[1], s->src_gm, NB_G);
        if (s->overlap) ASSERT_HIP(hipEventRecord(s->ev_gather, s->group ? s->stream : s->comm_stream), "prime gather event");
    }
    ASSERT_HIP(hipStreamSynchronize(st), "sync after SetSimulationData");
    if (s->direct) {
        // nobody may push a step's slice into a peer's gathered arrays before that peer has finished initialising them
        // (its split_sources launch above rewrites both arrays whole): one barrier, after which every rank's upload is done
        uint64_t ready[64] = {0};
        ready[s->rank] = 1;
        s->host_gather(s->host_gather_ctx, ready, sizeof(uint64_t), s->rank, s->nranks);
    }
    // small worlds in auto mode: have the canonical chain ready before the first step call (see wants_canonical)
    if (wants_canonical(s)) (void)find_or_build_graph(s, CANON_STEPS, 0.0f, resolve_shape(s));
}

}  // namespace

extern "C" {

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
                comm_allgather_f32(s, shard, floats, st, "particle slices");
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
    if (s->comm) {
        // collective: a rank that died mid-run must not hang its peers forever
        Watchdog dog("the collective GetSimulationData", s->rank, s->nranks);
        ASSERT_HIP(hipStreamSynchronize(st), "sync after GetSimulationData");
    } else {
        ASSERT_HIP(hipStreamSynchronize(st), "sync after GetSimulationData");
    }
}

void nb_hip_step_async(SimPipeline *s, uint32_t n, float dt) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    enqueue_steps(s, n, dt);
}

void nb_hip_sync(SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (!s->on_device) return;
    use_device();
    if (s->comm) {
        // the steps of an RCCL pipeline wait on every other rank's all-gather contribution: bounded like the
        // communicator's creation (NB_HIP_COMM_TIMEOUT_S), diagnostic + _exit(3), no retry
        Watchdog dog("the step chain's all-gathers (stream sync)", s->rank, s->nranks);
        ASSERT_HIP(hipStreamSynchronize(s->stream), "stream sync");
        if (s->comm_stream) ASSERT_HIP(hipStreamSynchronize(s->comm_stream), "comm stream sync");
        return;
    }
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
    if (!s->on_device || (!s->timed && !s->group) || s->detail_steps == 0) return 0;
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

uint32_t nb_hip_graph_stats(const SimPipeline *s, uint32_t *dt_uploads) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    if (dt_uploads) *dt_uploads = s->dt_uploads;
    return (uint32_t)s->graphs.size();
}

uint32_t nb_hip_last_fused_steps(const SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    return s->fused_steps;
}

int nb_hip_launch_lanes(const SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    return s->last_shape.lanes > 1 ? s->last_shape.lanes : 1;
}

int nb_hip_launch_unit(const SimPipeline *s) {
    NB_ASSERT(s != nullptr, "NULL pipeline");
    return s->last_shape.unit;
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
    } else if (!strcmp(key, "graph")) {
        NB_ASSERT(value >= 0 && value <= 2, "graph must be 0 (never), 1 (always) or 2 (from the second use), got %d", value);
        old = s->use_graph;
        s->use_graph = value;
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
        NB_FAIL("unknown knob \"%s\" (include/nbody_hip.h lists the knobs; launch-shape hooks: nbody_hip_tuning.h)", key);
    }
    return old;
}

// Test / tooling hooks (nbody_hip_tuning.h): not part of the C-ABI.
int nb_hip_tuning_build(void) {
#ifdef NB_TUNING_SHAPES
    return 1;
#else
    return 0;
#endif
}

int nb_hip_tune(SimPipeline *s, const char *key, int value) {
    NB_ASSERT(s != nullptr && key != nullptr, "NULL argument");
    int old = 0;
    if (!strcmp(key, "k")) {
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
    } else if (!strcmp(key, "lanes")) {
        NB_ASSERT(value == 0 || value == 1 || value == 2 || value == 4 || value == 8, "lanes must be 0 (auto), 1, 2, 4 or 8, got %d", value);
        old = s->want_lanes;
        s->want_lanes = value;
    } else if (!strcmp(key, "fused_chain")) {
        NB_ASSERT(value >= 0 && value <= 2, "fused_chain must be 0 (never), 1 (whenever the world fits one workgroup) or 2 (auto), got %d", value);
        old = s->fused_chain;
        s->fused_chain = value;
    } else if (!strcmp(key, "fused_finish")) {
        NB_ASSERT(value >= 0 && value <= 2, "fused_finish must be 0 (never), 1 (whenever the shape allows) or 2 (auto), got %d", value);
        old = s->fused_finish;
        if (s->on_device && old != value) {
            ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before switching the finish mode");
            for (auto &g : s->graphs) destroy_graph(g);   // cached chains hold the other kernel set
            s->graphs.clear();
        }
        s->fused_finish = value;
    } else if (!strcmp(key, "persist")) {
        NB_ASSERT(value >= 0 && value <= 64, "persist must be 0 (classic) .. 64 work items per workgroup, got %d", value);
#ifndef NB_TUNING_SHAPES
        NB_ASSERT(value <= 1, "the persistent-launch experiment kernels are built with make TUNING=1 only (persist = %d)", value);
#endif
        old = s->want_persist;
        s->want_persist = value;
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
    } else {
        NB_FAIL("unknown tuning hook \"%s\"", key);
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
