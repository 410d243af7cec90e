// step_chain.hip -- n steps as device work: the launches of one step, cached hipGraph chains, the sharded step.
//
// Stands where the reference records its command buffer (src/lib/sim_gpu.c:258-361: copy -> barrier -> dispatch x n ->
// copy, re-recorded on every call).  Differences by design:
//   * ping-pong position buffers instead of a full device-to-device copy per step (sim_gpu.c:316-324);
//   * n-step chains are cached hipGraphs of kernel nodes, keyed on (length, passes, shape, ping-pong phase); the step
//     size sits in device memory like the reference's uniform (sim_gpu.c:268-284), so a new dt never rebuilds a chain;
//   * worlds that fit ONE workgroup run their chain inside one launch (nb::launch_chain), positions in LDS;
//   * sharded pipelines add the per-step in-place all-gather of the new source positions (RCCL, a local group, or
//     a caller-supplied host transport), in-stream or overlapped with the own-shard launch.
#include "pipeline_internal.h"

namespace nbi {

hipEvent_t EventPool::next() {
    if (used == ev.size()) {
        hipEvent_t e;
        ASSERT_HIP(hipEventCreate(&e), "event");
        ev.push_back(e);
    }
    return ev[used++];
}

void EventPool::destroy() {
    for (hipEvent_t e : ev) ASSERT_HIP(hipEventDestroy(e), "event");
    ev.clear();
    used = 0;
}

constexpr uint32_t GRAPH_CHAIN_MAX = 64;  // longer requests replay an even-length chain
constexpr size_t GRAPH_CACHE_MAX = 8;     // cached chains per pipeline; the least recently used one is evicted
// graph = 2 (auto): chains shorter than this stay plain launches.  A hipGraphLaunch costs the host ~12 us more than
// a few plain launches and a replayed node saves 1-2 us, so a graph pays from a dozen steps on
// (profiles/r02_frame_loop_latency.txt: 2-step frames 33 -> 44 us with a graph, 8-step frames still 98 -> 102 us).
constexpr uint32_t GRAPH_AUTO_MIN_CHAIN = 16;


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

uint32_t passes_for(const SimPipeline *s, const nb::StepParams &p);

void destroy_graph(StepGraph &g) {
    if (g.exec) ASSERT_HIP(hipGraphExecDestroy(g.exec), "hipGraphExecDestroy");
    if (g.graph) ASSERT_HIP(hipGraphDestroy(g.graph), "hipGraphDestroy");
    g.exec = nullptr;
    g.graph = nullptr;
    g.nodes.clear();
    g.params.clear();
}

nb::LaunchShape resolve_shape(SimPipeline *s) {
    nb::LaunchShape want = {s->want_k, s->want_w, s->want_variant, s->want_split, s->want_unit, s->want_lanes, s->want_persist};
    // the model sees one launch: with source passes that is 1/passes of the sources
    nb::StepParams probe;
    memset(&probe, 0, sizeof probe);
    probe.src_end[0] = s->n_src;
    const uint32_t passes = (s->sharded && s->overlap) ? 1 : passes_for(s, probe);
    // lane-split launches walk ONE source range of an unsharded pipeline (one launch = the whole step)
    if (s->sharded || passes != 1 || s->n_src > nb::LANE_SPLIT_MAX_SRC || s->n_src == 0) want.lanes = 1;   // never, not "auto"
    nb::LaunchShape sh = nb::choose_shape(want, s->n_real, (s->n_src + passes - 1) / passes, g_dev.compute_units);
    NB_ASSERT(nb::step_kernel_fn(sh) != nullptr, "no step kernel for k=%d w=%d variant=%d", sh.k, sh.w, sh.variant);
    if (sh.split > 1) {
        const size_t need = (size_t)sh.split * s->n_real;
        if (need > s->parts_cap) {
            ASSERT_HIP(hipStreamSynchronize(s->stream), "sync before growing the parts buffer");
            dev_free(s->parts);
            s->parts = dev_alloc<float2>(need);
            s->parts_cap = (uint32_t)need;
            if (!s->tickets) {
                s->tickets_len = (uint32_t)((size_t)s->n_real / 64 + 2);
                s->tickets = dev_alloc<uint32_t>(s->tickets_len);
                zero_tickets(s);
                ASSERT_HIP(hipStreamSynchronize(s->stream), "sync after zeroing the tickets");
            }
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

// Fused finish ("fused_finish" knob): a split step WITHOUT its second, dependent kernel.  The parts of a receiver tile
// are written with agent-scope (sc1) stores; every workgroup then takes a ticket of its tile, and the one that arrives
// last -- all other parts are complete by then: each workgroup waits for its own stores (vmcnt(0)) before it draws --
// reads the parts back with the same scope, adds them in part order exactly like finish_kernel and integrates.  Same
// bits as the two-kernel form (the GPU suite runs green with it forced on; tools/fused_finish_soak.py: 20 000 steps at
// N = 10 000 + 300 random shapes, 0 differences).  What it buys is the finish kernel's boundary minus the ticket's
// round trip (profiles/r04_fused_finish.txt, us per step, cached graph replays | plain launches):
//   N = 5 000 +0.6 | -0.6    8 000 0.0 | -0.5    10 000 -0.4 | -1.0    14 000 -1.2 | -0.8    20 000 -0.7 | -0.9
//   50 000 -1.5 | -1.6    100 000 -2.3 | -2.3
// Auto (2, default): unsharded steps on the scalar-cache route with N x M >= 4e7 (N >~ 9 000: from where it also wins
// inside a hipGraph) and at most 200 000 receivers (beyond that the finish kernel is < 0.3 % of a step, and the kernels
// of the BASELINE sizes stay the ones the PMC profiles describe).  Sharded steps keep the two-kernel form.
constexpr double FUSED_FINISH_MIN_PAIRS = 4.0e7;
constexpr uint32_t FUSED_FINISH_MAX_RECV = 200000;

bool fused_finish_rule(uint32_t n_recv, uint32_t n_src) {
    return (double)n_recv * (double)n_src >= FUSED_FINISH_MIN_PAIRS && n_recv <= FUSED_FINISH_MAX_RECV;
}

bool fused_finish_applies(const SimPipeline *s, nb::LaunchShape sh) {
    if (s->fused_finish == 0 || s->sharded || sh.split <= 1 || sh.lanes > 1 || s->tickets == nullptr) return false;
    if (nb::step_kernel_fused_fn(sh) == nullptr) return false;   // scalar-cache route, W >= 4
    return s->fused_finish == 1 || fused_finish_rule(s->n_real, s->n_src);
}

nb::StepParams shaped(const SimPipeline *s, nb::StepParams p, nb::LaunchShape sh) {
    p.split = sh.split > 1 ? (uint32_t)sh.split : 1u;
    p.parts = p.split > 1 ? s->parts : nullptr;
    p.tickets = s->tickets;
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
    const bool fused = fused_finish_applies(s, sh);
    for (nb::StepParams &copy : step_passes(s, p, sh)) {
        void *args[] = {&copy};
        ASSERT_HIP(hipLaunchKernel(fused ? nb::step_kernel_fused_fn(sh) : nb::step_kernel_fn(sh), nb::step_grid(sh, s->n_real), nb::step_block(sh), args,
                                   nb::step_lds_bytes(sh, copy.src_end[0] - copy.src_begin[0]), st),
                   "step kernel launch (k=%d w=%d variant=%d split=%d, %u receivers)", sh.k, sh.w, sh.variant, sh.split,
                   s->n_real);
        if (copy.split > 1 && !fused)
            ASSERT_HIP(hipLaunchKernel(nb::finish_kernel_fn(), nb::finish_grid(s->n_real), nb::finish_block(), args, 0, st),
                       "finish kernel launch (%u receivers, %u parts)", s->n_real, copy.split);
    }
}

// The fused finish counts arrivals per receiver tile and its last arriver puts the ticket back to zero -- which only holds
// while every launch runs to its end.  A launch that ended part-way (a failed graph replay, a device error the caller
// survived) would leave non-zero tickets, and every later step would silently skip that tile's integration: so the tickets
// are re-zeroed, in stream order, whenever a new state is uploaded and before a chain is built.
void zero_tickets(SimPipeline *s) {
    if (s->tickets && s->tickets_len)
        ASSERT_HIP(hipMemsetAsync(s->tickets, 0, (size_t)s->tickets_len * sizeof(uint32_t), s->stream), "zero the tile tickets");
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

void fill_node(hipKernelNodeParams &kp, void **args, const void *fn, dim3 grid, dim3 block, size_t lds_bytes = 0) {
    memset(&kp, 0, sizeof kp);
    kp.func = const_cast<void *>(fn);
    kp.gridDim = grid;
    kp.blockDim = block;
    kp.sharedMemBytes = (unsigned int)lds_bytes;
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
            c.shape.variant == sh.variant && c.shape.split == sh.split && c.shape.unit == sh.unit && c.shape.lanes == sh.lanes &&
            c.shape.persist == sh.persist)
            return &c;
    return nullptr;
}

StepGraph *find_or_build_graph(SimPipeline *s, uint32_t n, float dt, nb::LaunchShape sh) {
    const uint32_t passes = passes_for(s, whole_step(s, s->cur, dt));
    StepGraph *g = find_graph(s, n, passes, sh, s->cur);
    const bool fused = fused_finish_applies(s, sh);
    const uint32_t per_pass = sh.split > 1 && !fused ? 2 : 1;  // step kernel (+ finish kernel)
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
    if (fused) zero_tickets(s);
    hipGraphNode_t prev = nullptr;
    for (uint32_t i = 0; i < n; i++) {
        const std::vector<nb::StepParams> launches = step_passes(s, whole_step(s, (s->cur + i) & 1, dt), sh);
        for (uint32_t q = 0; q < passes; q++) {
            g->params[(size_t)i * passes + q] = launches[q];
            void *args[] = {&g->params[(size_t)i * passes + q]};
            for (uint32_t j = 0; j < per_pass; j++) {
                hipKernelNodeParams kp;
                if (j == 0)
                    fill_node(kp, args, fused ? nb::step_kernel_fused_fn(sh) : nb::step_kernel_fn(sh), nb::step_grid(sh, s->n_real), nb::step_block(sh),
                              nb::step_lds_bytes(sh, launches[q].src_end[0] - launches[q].src_begin[0]));
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
constexpr double CANON_MAX_PAIRS = 6.0e7;  // N x M up to which a replay still pays (N ~ 11 000 with galaxy.h ICs)

// A world that fits one workgroup runs its chains inside ONE launch (kernels.hip chain_kernel).  "fused_chain" = 2
// (auto): only while a step is cheaper on one compute unit than the kernel boundary it saves -- N <= 256 (two receiver
// tiles of eight waves each; with four tiles a wave's source slice is 2.5x longer) and N x M <= 3.6e4.  Measured
// (profiles/r03_fused_chain.txt): 2.25 us per step at N = 200 / 250 against 3.1 as per-step launches, but 4.95 against
// 3.4 at N = 300 and 7.6 against 3.7 at N = 512 -- one CU runs the interaction body at ~63 % of its issue rate with four
// latency-bound waves per SIMD.  For calls of two steps or more, and only with the launch shape left on auto (an explicit k / w / split / unit / passes asks for the
// per-step kernel).  1 = whenever the world fits (N <= 512), whatever the other knobs say; 0 = never.
constexpr double CHAIN_MAX_PAIRS = 3.6e4;
constexpr uint32_t CHAIN_AUTO_MAX_RECV = 256;
constexpr uint32_t CHAIN_MAX_STEPS_PER_LAUNCH = 1u << 16;

bool wants_fused_chain(const SimPipeline *s) {
    if (s->sharded || s->fused_chain == 0 || s->n_real == 0 || nb::chain_tiles(s->n_real) == 0) return false;
    if (s->fused_chain == 1) return true;
    // an explicit k / w / split / unit / passes / lanes / route asks for the per-step kernel (choose_shape treats them so too)
    const bool shape_on_auto = s->want_k == 0 && s->want_w == 0 && s->want_split == 0 && s->want_unit == 0 && s->want_passes == 0 &&
                               s->want_lanes == 0 && s->want_persist == 0 && s->want_variant == nb::VARIANT_SMEM;
    return shape_on_auto && s->n_real <= CHAIN_AUTO_MAX_RECV && (double)s->n_real * (double)(s->n_src ? s->n_src : 1) <= CHAIN_MAX_PAIRS;
}

void enqueue_fused(SimPipeline *s, uint32_t n) {
    nb::ChainParams p;
    memset(&p, 0, sizeof p);
    p.pos = s->pos[s->cur];   // updated in place: the ping-pong phase does not move
    p.vel = s->vel;
    p.acc = s->acc;
    p.radius = s->radius;
    p.src_gm = s->src_gm;
    p.n_recv = s->n_real;
    p.n_src = s->n_src;
    p.tiles = nb::chain_tiles(s->n_real);
    p.dt = s->dt_dev;
    for (uint32_t left = n; left > 0;) {
        p.steps = left > CHAIN_MAX_STEPS_PER_LAUNCH ? CHAIN_MAX_STEPS_PER_LAUNCH : left;
        nb::launch_chain(s->stream, p);
        left -= p.steps;
    }
    s->fused_steps = n;
    s->last_shape = {2, (int)(16u / p.tiles), nb::VARIANT_LDS, 1, 8, 1, 0};   // the per-step shape it is bit-equal to
    s->last_groups = 1;
}

bool wants_canonical(const SimPipeline *s) {
    if (s->fused_chain == 2 && wants_fused_chain(s)) return false;   // its chains never reach the graph path
    return !s->sharded && s->use_graph == 2 && s->n_real > 0 &&
           (double)s->n_real * (double)(s->n_src ? s->n_src : 1) <= CANON_MAX_PAIRS;
}

void enqueue_single(SimPipeline *s, uint32_t n, float dt) {
    if (n >= 2 && wants_fused_chain(s)) {
        enqueue_fused(s, n);
        return;
    }
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
    // the own slot never left the device: upload the slots before and after it only (in overlap mode the next step's
    // own-shard launch may already be reading it on the compute stream)
    if (s->rank > 0)
        ASSERT_HIP(hipMemcpyAsync(dev, host, mine, hipMemcpyHostToDevice, st), "H2D of the slots before rank %d's", s->rank);
    if (s->rank + 1 < s->nranks)
        ASSERT_HIP(hipMemcpyAsync(dev + mine + bytes_per_rank, host + mine + bytes_per_rank,
                                  bytes_per_rank * (size_t)(s->nranks - 1 - s->rank), hipMemcpyHostToDevice, st),
                   "H2D of the slots after rank %d's", s->rank);
}

// The direct exchange's data movement: ONE launch reads this rank's slice once and stores it into every peer's gathered
// array (IPC-mapped; over xGMI each peer's stores ride that peer's own link), instead of P - 1 separately submitted
// copies.  16-byte accesses; the slice length is a multiple of 64 float2.
constexpr int DIRECT_PUSH_MAX_PEERS = 15;
struct PushTargets {
    float4 *dst[DIRECT_PUSH_MAX_PEERS];
    int n;
};

__global__ __launch_bounds__(256) void direct_push_kernel(const float4 *__restrict__ src, PushTargets to, uint32_t count4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count4) return;
    const float4 v = src[i];
    for (int q = 0; q < to.n; q++) to.dst[q][i] = v;
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
    if (s->direct) {
        // the direct all-gather: this rank's slice goes straight into every peer's gathered array, device to device --
        // over xGMI one copy per peer, each on its own link (no ring, no staging) -- then one host-side barrier: when it
        // opens, every slice of this array has landed everywhere (each rank synchronised its stream before entering it).
        // The ping-pong makes the writes safe: peers write into src_pos[buf] while every rank still reads src_pos[buf ^ 1].
        const size_t off = (size_t)s->rank * per_rank, bytes = per_rank * sizeof(float);
        if (s->nranks - 1 <= DIRECT_PUSH_MAX_PEERS && s->nranks > 1) {
            PushTargets to;
            to.n = 0;
            for (int q = 0; q < s->nranks; q++)
                if (q != s->rank) to.dst[to.n++] = reinterpret_cast<float4 *>(reinterpret_cast<float *>(s->peer_src[buf][(size_t)q]) + off);
            const uint32_t count4 = (uint32_t)(per_rank / 4);   // per_rank floats = Mc float2, Mc a multiple of 64
            hipLaunchKernelGGL(direct_push_kernel, dim3((count4 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float4 *>(base + off), to, count4);
            ASSERT_HIP(hipGetLastError(), "direct push launch (rank %d, %d peers)", s->rank, to.n);
        } else {
            for (int q = 0; q < s->nranks; q++) {
                if (q == s->rank) continue;
                float *dst = reinterpret_cast<float *>(s->peer_src[buf][(size_t)q]);
                ASSERT_HIP(hipMemcpyAsync(dst + off, base + off, bytes, hipMemcpyDeviceToDevice, st), "direct push of rank %d's sources to rank %d", s->rank, q);
            }
        }
        ASSERT_HIP(hipStreamSynchronize(st), "sync before the step barrier of the direct exchange");
        uint64_t seen[64];
        seen[s->rank] = ++s->direct_steps;
        s->host_gather(s->host_gather_ctx, seen, sizeof(uint64_t), s->rank, s->nranks);
        for (int q = 0; q < s->nranks; q++)
            NB_ASSERT(seen[q] == s->direct_steps, "direct exchange: rank %d is at step %llu, rank %d at %llu", q, (unsigned long long)seen[q], s->rank,
                      (unsigned long long)s->direct_steps);
        return;
    }
    if (s->host_gather) {
        host_allgather(s, base, per_rank * sizeof(float), st);
        return;
    }
    comm_allgather_f32(s, base, per_rank, st, "source positions");
}

// One sharded step of one rank.  `cs` carries the gather (the comm stream with RCCL; the group stream locally).
// `detail`: bracket the step's kernels and its gather with event pairs (plain launches only, not under capture).
void sharded_step(SimPipeline *s, nb::LaunchShape sh, float dt, hipStream_t cs, bool detail) {
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
    // per-step event pairs only when the caller asked for timing: up to ~1000 hipEventRecord per call otherwise for nothing
    for (uint32_t i = 0; i < n; i++) sharded_step(s, sh, dt, s->comm_stream, s->timing != 0 && i < DETAIL_STEPS_MAX);
    s->detail_steps = s->timing ? (n < DETAIL_STEPS_MAX ? n : DETAIL_STEPS_MAX) : 0;
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
    s->fused_steps = 0;
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
    if (s->fused_steps) s->timed_launches = (n + CHAIN_MAX_STEPS_PER_LAUNCH - 1) / CHAIN_MAX_STEPS_PER_LAUNCH;
    s->timed_finish_launches = s->last_shape.split > 1 && !fused_finish_applies(s, s->last_shape) ? s->timed_launches : 0;
    s->data.dt = dt;
}

}  // namespace nbi
