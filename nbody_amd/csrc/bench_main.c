/*
 * bench_main.c -- nbody-bench for the MI355X build.
 *
 * With no size flags it reproduces the reference harness (src/bench.c:21-74):
 * srand(11037) once, then for each N of its table MakeGalaxies(N, 2), one World
 * per backend, 10 warm-up steps and 100 timed steps at dt = 1 in a single
 * UpdateWorld_* call, microseconds per step printed per backend.  `--cpu` /
 * `--gpu` restrict the backends exactly as there (bench.c:44-48).
 *
 * Extras the BASELINE.json configurations need:
 *   --n N          one size instead of the table (repeatable)
 *   --steps K      timed steps            (default 100)
 *   --warmup W     untimed steps          (default 10; 0 still runs one: it carries the upload)
 *   --dt DT        step size              (default 1.0, the reference's)
 *   --galaxies G   galaxies per universe  (default 2)
 *   --seed S       srand seed             (default 11037)
 *   --own-rng      draw the universes from MakeGalaxiesSeeded(seed + row) instead of libc rand()
 *   --repeats R    time the K-step call R times on the same world and report the fastest (default 1 = the
 *                  reference's single timed call; a row of a few hundred microseconds is at the mercy of one OS hiccup)
 *   --cpu-best     one more column, INFORMATIONAL: the fastest CPU variant this host runs (libnbody_cpu_best.so: the same
 *                  sim_cpu.c built AVX2+FMA / AVX-512, with sqrt + div or the rsqrt estimate + one Newton step; picked by a
 *                  short calibration, named on stderr).  Not the reference's bits -- the CPU column stays the -mavx path that is
 *                  (the reference's SIMD matrix: src/lib/CMakeLists.txt:24-33)
 * and more columns: interactions/s = N * mass_len * steps / time per backend, and for the GPU the share of the
 * fp32 roofline that is (14 flop per interaction, 157.3 TFLOP/s), then what the chip allows at THIS size:
 *   GPU floor = N * mass_len / R interactions/s       (R = the rate THIS run measures on THIS box at N = 100 000 before
 *               the table starts -- a calibration world drawn from MakeGalaxiesSeeded, so libc's rand() stream and with
 *               it the table's universes stay the reference's; boxes of the pool differ by up to 8 % on this kernel, so a
 *               constant would silently shift %floor; printed on stderr; --floor-rate R overrides it)
 *             + kernels per step x 1.7 us              (dependent-launch floor inside a hipGraph on this box,
 *               profiles/r01_ubench6_launch_floor.txt; a step is 1 kernel, or 2 when the sources are split)
 *   %floor    = floor / measured: how close the step is to that bound.  Below N ~ 50 000 a launch cannot fill the
 *               chip for long enough to amortise its latencies; the reference's SIZES[] (bench.c:38) all live there.
 *
 * Several GPUs (SURVEY.md section 8f rank 1 `--gpus`, 8e): one process per GPU, host code in C, no Python, no torch.
 *   --gpus P           fork P ranks BEFORE anything touches HIP; rank r drives device r; the ranks share one
 *                      anonymous MAP_SHARED page (rank_page.h) for barriers, the max-over-ranks time and RCCL's unique id
 *   --transport T      auto (default): rccl -> ipc -> shm.  When a rank of an attempt ends with an error (RCCL that does not come
 *                      up: the library's watchdog leaves with 3, its error convention with abort(); fewer devices than ranks;
 *                      an IPC open the container refuses) or the attempt outlives its share of --budget-s, the parent -- which
 *                      never touches the GPU -- forks a FRESH set of ranks over the next transport and says so on stdout and
 *                      stderr ("transport_fallback"); nothing is retried inside a process that touched HIP.  A VERIFICATION
 *                      failure (--verify: ranks disagree, or differ from one GPU; exit status 5) is not a bring-up failure: the
 *                      next transport is still tried, but the run says "verification_failed" and never exits 0;
 *                      rccl: CreateWorldSharded, per-step in-place ncclAllGather over xGMI inside the library;
 *                      ipc: CreateWorldShardedDirect -- no RCCL: every rank maps its peers' source arrays (hipIpc handles passed
 *                      through the page) and pushes its slice device-to-device into each after every step (on xGMI one copy per
 *                      link), then ONE host barrier per step at the page; the fallback should RCCL not come up;
 *                      shm: CreateWorldShardedWith over the shared page (host-staged data).
 *                      ipc and shm let P ranks share ONE GPU (rank r drives device r mod visible devices), where RCCL refuses
 *                      duplicate devices: the whole multi-process path runs on a one-GPU box
 *   --modes a,b,..     rows per size: plain (kernel, then gather in-stream), overlap (own-shard kernel runs while the
 *                      other shards' positions arrive), graph (the {kernel, all-gather} x K chain captured as a
 *                      hipGraph; rccl only).  Default: all the transport allows
 *   --verify K         before timing, K steps of every mode on a fresh sharded World: every rank must hold the same
 *                      bytes, and rank 0 compares them with an ordinary single-GPU World stepped the same way
 *                      (default 3; with more than one rank 0 is not accepted: at least one step is always checked).  A
 *                      deviation above 1e-5 relative L2 in positions fails the run.
 *                      Without --gpus (and with both backends): K steps of UpdateWorld_GPU against K steps of
 *                      UpdateWorld_CPU per row, relative to what the steps moved; above 1e-4 the run fails
 *   --speedup          rank 0 also times the same K steps on an ordinary single-GPU World (its own device, the other ranks
 *                      wait) and the table gains "1-GPU us" and "speedup" columns: strong scaling from one command
 *   --force-sharded    with --gpus 1: still go through the RCCL path (one-rank communicator)
 *   --wait-timeout S   how long a rank waits for the others at the page before it gives up (default 180; never longer than
 *                      what is left of the attempt's share of the budget)
 *   --budget-s S       total wall-clock budget of a multi-rank run (default 480: below the 600 s a driver allows a command);
 *                      with --transport auto the attempts get 1/2, 1/4, 1/4 of it; an attempt that outlives its share is
 *                      ended by exact pid; 0 = no budget
 *   --selftest-ranks   no GPU: P ranks exercise the page only (barriers, id hand-over, reductions, all-gathers)
 * Rank 0 prints the table: N, ranks, mode, us/step (max over ranks, barrier on both sides of ONE UpdateWorld_GPU(K) call),
 * steps/s, interactions/s, % of P x 157.3 TFLOP/s, and the per-step device time of the kernels and of the all-gathers
 * (nb_hip_last_step_breakdown; 0 for graph rows).  Before the table every rank writes a "# preflight" line to stderr: its
 * device's PCI address, the hipDeviceCanAccessPeer row and (rccl / ipc) one IPC open + close of the next rank's exported
 * word (nb_hip_preflight_*), so that a bring-up that fails says how far it got.  A rank that fails exits non-zero; the parent reaps every child, ends
 * the stragglers (exact pids) and reports -- it never re-executes anything.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <signal.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <galaxy.h>
#include <nbody.h>
#include <nbody_hip.h>

#include "nbody_hip_tuning.h" /* floor-column planning + the --one-wave hook; not the public ABI */
#include "rank_page.h"

/* exit statuses of a rank, by class: 2 = cannot start (devices), 3 = the library's watchdog, 4 = left at the page (a peer
 * failed or a wait timed out), 128 + signal = died (abort() is the library's error convention) -- all BRING-UP / run failures;
 * 5 = the run completed but a --verify check failed: a wrong answer, not a missing one */
#define EXIT_VERIFY_FAILED 5

#define CALIBRATION_N 100000u     /* the last row of the reference's table (bench.c:38): the large-N rate is measured there */
#define LAUNCH_FLOOR_US 1.7       /* per dependent kernel inside a hipGraph (profiles/r01_ubench6_launch_floor.txt) */

typedef void (*UpdateFn)(World *, float, uint32_t);

static double seconds_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* one warm-up call, one timed call; returns seconds per step */
static double time_backend(World *w, UpdateFn update, float dt, uint32_t warmup, uint32_t steps, uint32_t repeats) {
    /* At least one untimed step always runs, also with --warmup 0: the first UpdateWorld_GPU call carries the
     * upload and the first-touch device setup (stream, HBM buffers, page-locking the World's array), which are
     * not part of a step.  A zero-step call cannot stand in: it is a no-op by contract (world.c:113). */
    update(w, dt, warmup > 0 ? warmup : 1);
    double best = 0.0;
    for (uint32_t r = 0; r < repeats; r++) {
        const double t0 = seconds_now();
        update(w, dt, steps);
        const double t1 = seconds_now();
        if (r == 0 || t1 - t0 < best) best = t1 - t0;
    }
    return best / (double)steps;
}

static uint32_t count_massive(const Particle *ps, uint32_t n) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++) m += ps[i].mass > 0;
    return m;
}

/* interactions/s the step kernel sustains on this box once a launch fills the chip: N = 100 000, fastest of 3 timed
 * 100-step calls, the per-kernel launch floor taken out so that the floor formula does not count it twice */
static double measure_large_n_rate(void) {
    Particle *ps = MakeGalaxiesSeeded(CALIBRATION_N, 2, 0x9e3779b97f4a7c15ull);
    const uint32_t m = count_massive(ps, CALIBRATION_N);
    World *w = CreateWorld(ps, CALIBRATION_N);
    const double per_step = time_backend(w, UpdateWorld_GPU, 0.01f, 200, 100, 3);
    DestroyWorld(w);
    free(ps);
    int k = 0, wv = 0, split = 1;
    uint32_t groups = 0;
    nb_hip_plan_launch(CALIBRATION_N, m, 256, &k, &wv, &split, &groups);
    const double kernels = split > 1 && !nb_hip_plan_fused_finish(CALIBRATION_N, m, 256) ? 2.0 : 1.0;
    double busy = per_step - kernels * LAUNCH_FLOOR_US * 1e-6;
    if (busy <= 0.0) busy = per_step;
    return (double)CALIBRATION_N * (double)m / busy;
}

/* libnbody_cpu_best.so (cpu_best.c): informational CPU variants of the stepper, never behind UpdateWorld_CPU */
int nb_cpu_variant_count(void);
const char *nb_cpu_variant_name(int i, int *supported, const char **what);
int nb_cpu_variant_update(const char *isa, Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n);

/* seconds per step of `isa` on a partitioned copy of the world (sources first: what CreateWorld's partition yields) */
static double time_cpu_variant(const char *isa, const Particle *ps, uint32_t n, float dt, uint32_t warmup, uint32_t steps, uint32_t repeats) {
    World *w = CreateWorld(ps, n);
    uint32_t got = 0;
    const Particle *part = GetWorldParticles(w, &got);
    Particle *arr = (Particle *)malloc((size_t)(n ? n : 1) * sizeof(Particle));
    memcpy(arr, part, (size_t)n * sizeof(Particle));
    uint32_t m = 0;
    while (m < n && arr[m].mass > 0) m++;
    DestroyWorld(w);
    nb_cpu_variant_update(isa, arr, n, m, dt, warmup > 0 ? warmup : 1);
    double best = 0.0;
    for (uint32_t r = 0; r < repeats; r++) {
        const double t0 = seconds_now();
        nb_cpu_variant_update(isa, arr, n, m, dt, steps);
        const double t1 = seconds_now();
        if (r == 0 || t1 - t0 < best) best = t1 - t0;
    }
    free(arr);
    return best / (double)steps;
}

/* the fastest variant this CPU runs, by a short calibration on a world of its own generator (libc's rand() stream stays
 * the table's); NULL when the CPU runs none */
static const char *pick_cpu_variant(void) {
    Particle *ps = MakeGalaxiesSeeded(8000, 2, 0x243f6a8885a308d3ull);
    const char *best = NULL;
    double best_s = 0.0;
    for (int i = 0; i < nb_cpu_variant_count(); i++) {
        int ok = 0;
        const char *what = NULL, *isa = nb_cpu_variant_name(i, &ok, &what);
        if (!ok) continue;
        const double s = time_cpu_variant(isa, ps, 8000, 0.01f, 2, 5, 2);
        if (best == NULL || s < best_s) best = isa, best_s = s;
    }
    free(ps);
    return best;
}

static const uint32_t REFERENCE_SIZES[] = {250, 500, 800, 1200, 2000, 4000, 10000, 20000, 50000, 100000};

enum { MODE_PLAIN = 0, MODE_OVERLAP = 1, MODE_GRAPH = 2, MODE_COUNT = 3 };
static const char *const MODE_NAME[MODE_COUNT] = {"plain", "overlap", "graph"};

typedef struct Options {
    bool use_cpu, use_gpu, cpu_best, own_rng, transport_shm, transport_ipc, transport_auto, force_sharded, selftest_ranks, verify_given, speedup, one_wave;
    /* transport_shm: any host-callback transport (shm or ipc: both need the page's exchange area); transport_ipc: the direct one */
    uint32_t sizes[64];
    uint32_t n_sizes, steps, warmup, galaxies, repeats, verify_steps;
    unsigned seed;
    float dt;
    double floor_rate; /* 0: measure it */
    int gpus;
    int modes[8];
    int n_modes;
    double wait_timeout_s, budget_s;
    int selftest_die;  /* --selftest-die R: that rank leaves with status 7 mid-run (the parent's reaping is what is tested) */
    int selftest_hang; /* --selftest-hang R: that rank never reaches the next barrier (the parent's budget is what is tested) */
    bool budget_given;
} Options;

/* Every universe of the table, drawn up front from ONE srand(seed) stream in table order, like the reference
 * (bench.c:42,53) -- and BEFORE this process first touches the GPU: libc's rand() is process-global and the HIP runtime
 * draws from it too (observed: two ranks whose GPU call sequences differed drew different second universes), so
 * interleaving MakeGalaxies with GPU work would make the rows depend on what ran before them. */
static Particle **draw_universes(const Options *o) {
    Particle **u = (Particle **)malloc(sizeof(Particle *) * (o->n_sizes ? o->n_sizes : 1));
    srand(o->seed);
    for (uint32_t s = 0; s < o->n_sizes; s++)
        u[s] = o->own_rng ? MakeGalaxiesSeeded(o->sizes[s], o->galaxies, o->seed + s) : MakeGalaxies(o->sizes[s], o->galaxies);
    return u;
}

/* ---- one process, one GPU: the reference's table ------------------------------------------------------------------ */

/* --verify K with both backends: the same universe stepped K times by UpdateWorld_CPU (the reference AVX path's bits)
 * and by UpdateWorld_GPU, compared relative to what the steps moved: |dpos_gpu - dpos_cpu| / |dpos_cpu| (L2 over all
 * particles).  The stated multi-step tolerance is 1e-4; beyond it the run fails. */
static int verify_backends(const Options *o, const Particle *ps, uint32_t n) {
    World *c = CreateWorld(ps, n), *g = CreateWorld(ps, n);
    Particle *start = (Particle *)malloc((size_t)(n ? n : 1) * sizeof(Particle));
    memcpy(start, GetWorldParticles(c, NULL), (size_t)n * sizeof(Particle));
    UpdateWorld_CPU(c, o->dt, o->verify_steps);
    UpdateWorld_GPU(g, o->dt, o->verify_steps);
    const Particle *pc = GetWorldParticles(c, NULL), *pg = GetWorldParticles(g, NULL);
    double num = 0.0, den = 0.0;
    int statics = 1;
    for (uint32_t i = 0; i < n; i++) {
        const double cx = (double)pc[i].pos.x - (double)start[i].pos.x, cy = (double)pc[i].pos.y - (double)start[i].pos.y;
        const double gx = (double)pg[i].pos.x - (double)start[i].pos.x, gy = (double)pg[i].pos.y - (double)start[i].pos.y;
        num += (gx - cx) * (gx - cx) + (gy - cy) * (gy - cy);
        den += cx * cx + cy * cy;
        statics = statics && pc[i].mass == pg[i].mass && pc[i].radius == pg[i].radius;
    }
    const double rel = den > 0.0 ? sqrt(num / den) : sqrt(num);
    fprintf(stderr, "nbody-bench: verify N=%u steps=%u dt=%g: GPU vs CPU rel_displacement %.3e (tolerance 1e-4); mass/radius equal %s\n", n,
            o->verify_steps, (double)o->dt, rel, statics ? "yes" : "NO");
    free(start);
    DestroyWorld(c);
    DestroyWorld(g);
    return !(rel <= 1e-4) || !statics;
}

static int run_single(const Options *o) {
    int bad = 0;
    Particle **universe = draw_universes(o);
    double floor_rate = o->floor_rate;
    if (o->use_gpu && floor_rate <= 0.0) {
        floor_rate = measure_large_n_rate();
        fprintf(stderr, "nbody-bench: floor rate %.3e interactions/s (measured at N = %u on this box)\n", floor_rate, CALIBRATION_N);
    }

    const char *best_isa = o->cpu_best ? pick_cpu_variant() : NULL;
    if (o->cpu_best)
        fprintf(stderr, "nbody-bench: --cpu-best column = %s (informational: not the reference's bits; the CPU column is the -mavx path that is)\n",
                best_isa ? best_isa : "none of the variants runs on this CPU");

    printf("\t      N");
    if (o->use_cpu) printf("\t    CPU");
    if (best_isa) printf("\t   CPU*");
    if (o->use_gpu) printf("\t    GPU");
    if (o->use_cpu) printf("\t  CPU int/s");
    if (best_isa) printf("\t CPU* int/s");
    if (o->use_gpu) printf("\t  GPU int/s\t GPU %%peak\t  GPU us\tfloor us\t   %%floor");
    printf("\n");

    for (uint32_t s = 0; s < o->n_sizes; s++) {
        const uint32_t n = o->sizes[s];
        Particle *ps = universe[s];
        const double pairs = (double)n * (double)count_massive(ps, n);

        double cpu_s = 0, gpu_s = 0;
        if (o->use_cpu) {
            World *w = CreateWorld(ps, n);
            cpu_s = time_backend(w, UpdateWorld_CPU, o->dt, o->warmup, o->steps, o->repeats);
            DestroyWorld(w);
        }
        if (o->use_gpu) {
            World *w = CreateWorld(ps, n);
            gpu_s = time_backend(w, UpdateWorld_GPU, o->dt, o->warmup, o->steps, o->repeats);
            DestroyWorld(w);
        }
        const double best_s = best_isa ? time_cpu_variant(best_isa, ps, n, o->dt, o->warmup, o->steps, o->repeats) : 0.0;
        printf("\t%7u", n);
        if (o->use_cpu) printf("\t%7ld", (long)(cpu_s * 1e6));
        if (best_isa) printf("\t%7ld", (long)(best_s * 1e6));
        if (o->use_gpu) printf("\t%7ld", (long)(gpu_s * 1e6));
        if (o->use_cpu) printf("\t%11.3e", pairs / cpu_s);
        if (best_isa) printf("\t%11.3e", pairs / best_s);
        /* roofline column: 14 flop per interaction (reference op count, sim_cpu.c:169-188) against the
         * MI355X fp32 vector peak of 157.3 TFLOP/s -- the same convention as bench.py */
        if (o->use_gpu) {
            int k = 0, wv = 0, split = 1;
            uint32_t groups = 0;
            const uint32_t m = count_massive(ps, n);
            /* passes: one launch per <= 3 MiB of (x, y, G*m) sources (step_chain.hip passes_for) */
            const uint32_t passes = m ? (uint32_t)(((uint64_t)m * 12 + (3u << 20) - 1) / (3u << 20)) : 1;
            nb_hip_plan_launch(n, (m + passes - 1) / passes, 256, &k, &wv, &split, &groups);
            /* lane-split steps (small worlds, every knob on auto) are one kernel whatever the classic plan's split says */
            const int lane_split = passes == 1 && nb_hip_plan_launch_lanes(n, m, NULL) > 1;
            /* ... and steps whose last-arriving workgroup finishes its tile in the step kernel have no finish kernel */
            const int fused = passes == 1 && nb_hip_plan_fused_finish(n, m, 256);
            const double kernels = (double)passes * (split > 1 && !lane_split && !fused ? 2.0 : 1.0);
            const double floor_us = pairs / floor_rate * 1e6 + kernels * LAUNCH_FLOOR_US;
            printf("\t%11.3e\t%9.1f\t%8.2f\t%8.2f\t%9.1f", pairs / gpu_s, pairs / gpu_s * 14.0 / 157.3e12 * 100.0, gpu_s * 1e6,
                   floor_us, floor_us / (gpu_s * 1e6) * 100.0);
        }
        printf("\n");
        fflush(stdout);
        /* --verify K wants both paths' results, whichever columns were asked for (--gpu --verify K still steps a CPU World) */
        if (o->use_gpu && o->verify_given && o->verify_steps > 0) bad = verify_backends(o, ps, n) || bad;
        free(ps);
    }
    free(universe);
    return bad;
}

/* ---- P processes, one per GPU --------------------------------------------------------------------------------------- */

static uint64_t fnv1a(const void *data, size_t bytes) {
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < bytes; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

/* --one-wave: one receiver per lane, one wave per workgroup -- the launch shape whose summation order does not depend on
 * how the sources are cut up, so that P shards and one GPU give the same BITS (a test hook, nbody_hip_tuning.h) */
static World *shaped(const Options *o, World *w) {
    if (o->one_wave) {
        nb_hip_tune(GetWorldPipeline(w), "k", 1);
        nb_hip_tune(GetWorldPipeline(w), "w", 1);
    }
    return w;
}

static World *make_sharded_world(const Options *o, NbRankPage *pg, const Particle *ps, uint32_t n) {
    const int rank = nb_rank_page_rank(pg), P = nb_rank_page_nranks(pg);
    if (o->transport_ipc) return shaped(o, CreateWorldShardedDirect(ps, n, rank, P, nb_rank_allgather, pg));
    if (o->transport_shm) return shaped(o, CreateWorldShardedWith(ps, n, rank, P, nb_rank_allgather, pg));
    unsigned char id[NB_HIP_UNIQUE_ID_BYTES];
    memset(id, 0, sizeof id);
    if (rank == 0) nb_hip_comm_unique_id(id); /* one id per communicator: a fresh one for every World */
    nb_rank_share_id(pg, id);
    return shaped(o, CreateWorldSharded(ps, n, rank, P, id));
}

static void set_mode(World *w, int mode) {
    SimPipeline *sim = GetWorldPipeline(w);
    nb_hip_configure(sim, "overlap", mode == MODE_OVERLAP);
    nb_hip_configure(sim, "sharded_graph", mode == MODE_GRAPH);
}

/* K steps of every mode on a fresh sharded World against an ordinary single-GPU World on rank 0; returns 0 when the
 * ranks agree bit for bit and the positions stay within 1e-5 relative L2 of the single-GPU ones */
static int verify_row(const Options *o, NbRankPage *pg, const Particle *ps, uint32_t n) {
    const int rank = nb_rank_page_rank(pg);
    World *w = make_sharded_world(o, pg, ps, n);
    World *one = rank == 0 ? shaped(o, CreateWorld(ps, n)) : NULL;
    int bad = 0;
    for (int mi = 0; mi < o->n_modes; mi++) {
        set_mode(w, o->modes[mi]);
        UpdateWorld_GPU(w, o->dt, o->verify_steps);
        uint32_t got_n = 0;
        const Particle *got = GetWorldParticles(w, &got_n); /* collective: the full array on every rank */
        const int agree = nb_rank_all_equal(pg, fnv1a(got, (size_t)got_n * sizeof(Particle)));
        double rel = 0.0, worst = 0.0;
        int bitwise = 0;
        if (rank == 0) {
            UpdateWorld_GPU(one, o->dt, o->verify_steps);
            const Particle *want = GetWorldParticles(one, NULL);
            double num = 0.0, den = 0.0;
            for (uint32_t i = 0; i < n; i++) {
                const double dx = (double)got[i].pos.x - (double)want[i].pos.x, dy = (double)got[i].pos.y - (double)want[i].pos.y;
                num += dx * dx + dy * dy;
                den += (double)want[i].pos.x * (double)want[i].pos.x + (double)want[i].pos.y * (double)want[i].pos.y;
                const double a = fabs(dx) > fabs(dy) ? fabs(dx) : fabs(dy);
                if (a > worst) worst = a;
            }
            rel = den > 0.0 ? sqrt(num / den) : sqrt(num);
            bitwise = memcmp(got, want, (size_t)n * sizeof(Particle)) == 0;
            fprintf(stderr, "nbody-bench: verify N=%u mode=%s steps=%u: ranks agree %s; vs single GPU: rel_l2_pos %.3e max_abs_pos %.3e bitwise %s\n",
                    n, MODE_NAME[o->modes[mi]], o->verify_steps, agree ? "yes" : "NO", rel, worst, bitwise ? "yes" : "no");
        }
        bad = bad || !agree || !(rel <= 1e-5);
    }
    if (one) DestroyWorld(one);
    DestroyWorld(w); /* every rank, same place: a collective for the direct exchange (one barrier before anyone unmaps) */
    return nb_rank_reduce(pg, (double)bad, 'x') > 0.0;
}

/* what every rank writes down before the first real contact: "# preflight ..." on stderr (one line per rank) */
static void preflight_rank(const Options *o, NbRankPage *pg, int ndev) {
    const int rank = nb_rank_page_rank(pg), P = nb_rank_page_nranks(pg);
    char info[256], peers[128] = "";
    int row[16];
    nb_hip_device_info(info, sizeof info);
    const char *pci = strstr(info, "pci=");
    const int seen = nb_hip_preflight_peers(row, 16);
    for (int q = 0, at = 0; q < seen && q < 16 && at < (int)sizeof peers - 4; q++) at += snprintf(peers + at, sizeof peers - (size_t)at, "%s%d", q ? "," : "", row[q]);
    char ipc[160];
    snprintf(ipc, sizeof ipc, "ipc=not probed (%s)", P == 1 ? "one rank: no peer" : "the shm transport needs none");
    if (P > 1 && (o->transport_ipc || !o->transport_shm)) {
        /* one IPC open + close of the NEXT rank's exported word: handles travel through the page's exchange area */
        unsigned char *all = (unsigned char *)calloc((size_t)P, 72);
        const int erc = nb_hip_preflight_ipc_export(all + (size_t)rank * 72 + 8, 0x6e620000u + (uint32_t)rank);
        memcpy(all + (size_t)rank * 72, &erc, sizeof erc);
        nb_rank_allgather(pg, all, 72, rank, P);
        const int peer = (rank + 1) % P;
        int peer_erc = 0, orc = 0;
        double ms = 0.0;
        memcpy(&peer_erc, all + (size_t)peer * 72, sizeof peer_erc);
        if (erc == 0 && peer_erc == 0) orc = nb_hip_preflight_ipc_open(all + (size_t)peer * 72 + 8, 0x6e620000u + (uint32_t)peer, &ms);
        nb_rank_barrier(pg, "preflight: every rank has closed what it opened");
        nb_hip_preflight_ipc_release();
        if (erc != 0)
            snprintf(ipc, sizeof ipc, "ipc_export=%d (%s)", erc, nb_hip_error_string(erc));
        else if (peer_erc != 0)
            snprintf(ipc, sizeof ipc, "ipc_export=0 ipc_open(rank %d)=skipped: its export failed (%d)", peer, peer_erc);
        else
            snprintf(ipc, sizeof ipc, "ipc_export=0 ipc_open(rank %d)=%d%s%s%s %.2f ms", peer, orc, orc ? " (" : "", orc ? nb_hip_error_string(orc) : "", orc ? ")" : "", ms);
        free(all);
    }
    fprintf(stderr, "# preflight rank %d of %d transport=%s device=%d/%d %s can_access_peer=[%s] %s\n", rank, P,
            o->transport_ipc ? "ipc" : o->transport_shm ? "shm" : "rccl", o->transport_shm ? rank % ndev : rank, ndev, pci ? pci : "pci=?", peers, ipc);
    fflush(stderr);
}

static int run_rank(const Options *o, NbRankPage *pg) {
    const int rank = nb_rank_page_rank(pg), P = nb_rank_page_nranks(pg);
    Particle **universe = draw_universes(o); /* the same stream on every rank, before any GPU call: identical universes */
    const int ndev = nb_hip_device_count();
    if (ndev < 1 || (!o->transport_shm && ndev < P)) {
        fprintf(stderr, "nbody-bench: rank %d: %d HIP device(s) visible, --gpus %d --transport %s needs %d%s\n", rank, ndev, P,
                o->transport_ipc ? "ipc" : o->transport_shm ? "shm" : "rccl", o->transport_shm ? 1 : P, o->transport_shm ? "" : " (one per rank)");
        nb_rank_page_fail(pg);
        return 2;
    }
    nb_hip_set_device(o->transport_shm ? rank % ndev : rank);
    preflight_rank(o, pg, ndev);

    if (rank == 0) {
        printf("\t      N\t  ranks\t   mode\t     GPU us\t   steps/s\t  GPU int/s\t GPU %%peak\tkernel ms\tgather ms%s\n",
               o->speedup ? "\t  1-GPU us\t  speedup" : "");
        fflush(stdout);
    }
    int bad = 0, unverified = 0;
    for (uint32_t s = 0; s < o->n_sizes; s++) {
        const uint32_t n = o->sizes[s];
        Particle *ps = universe[s];
        const double pairs = (double)n * (double)count_massive(ps, n);
        /* belt and braces: the ranks must be about to step the same bytes */
        if (!nb_rank_all_equal(pg, fnv1a(ps, (size_t)n * sizeof(Particle)))) {
            if (rank == 0) fprintf(stderr, "nbody-bench: the ranks drew different universes at N=%u\n", n);
            bad = 1;
        }
        if (o->verify_steps > 0) unverified = verify_row(o, pg, ps, n) || unverified;

        double single_s = 0.0; /* --speedup: the same call on one GPU, timed by rank 0 while the others wait at the barrier */
        if (o->speedup) {
            if (rank == 0) {
                World *one = shaped(o, CreateWorld(ps, n));
                single_s = time_backend(one, UpdateWorld_GPU, o->dt, o->warmup, o->steps, o->repeats);
                DestroyWorld(one);
            }
            nb_rank_barrier(pg, "after the single-GPU reference timing");
        }
        World *w = make_sharded_world(o, pg, ps, n);
        SimPipeline *sim = GetWorldPipeline(w);
        nb_hip_configure(sim, "timing", 1);
        for (int mi = 0; mi < o->n_modes; mi++) {
            const int mode = o->modes[mi];
            set_mode(w, mode);
            /* the first call carries the upload; a captured chain is keyed on its length, so the graph row warms up
             * with the timed length (capture + instantiate stay outside the timed call) */
            UpdateWorld_GPU(w, o->dt, mode == MODE_GRAPH ? o->steps : (o->warmup > 0 ? o->warmup : 1));
            double best = 0.0;
            for (uint32_t r = 0; r < o->repeats; r++) {
                nb_rank_barrier(pg, "before the timed call");
                const double t0 = seconds_now();
                UpdateWorld_GPU(w, o->dt, o->steps); /* blocking: the device is idle on return */
                const double t1 = seconds_now();
                nb_rank_barrier(pg, "after the timed call");
                const double all = nb_rank_reduce(pg, t1 - t0, 'x'); /* max over ranks */
                if (r == 0 || all < best) best = all;
            }
            double k_ms = 0.0, c_ms = 0.0;
            const uint32_t covered = nb_hip_last_step_breakdown(sim, &k_ms, &c_ms);
            k_ms = nb_rank_reduce(pg, covered ? k_ms / covered : 0.0, 'x');
            c_ms = nb_rank_reduce(pg, covered ? c_ms / covered : 0.0, 'x');
            if (s == 0 && mi == 0) {
                /* what the communicator itself says: P real ranks, not P replicas */
                int nr = 0, rk = 0, dev = 0, ver = 0;
                double first_ms = 0.0;
                char lib[256] = {0};
                const int owns = nb_hip_comm_info(sim, &nr, &rk, &dev, &ver, &first_ms, lib, sizeof lib);
                const double owners = nb_rank_reduce(pg, (double)owns, 's');
                const double nr_min = nb_rank_reduce(pg, (double)nr, 'n'), nr_max = nb_rank_reduce(pg, (double)nr, 'x');
                const double rk_sum = nb_rank_reduce(pg, (double)rk, 's');
                if (rank == 0)
                    fprintf(stderr, "nbody-bench: %d ranks, transport %s; ranks_with_communicator=%d %s=%d..%d user_rank_sum=%d "
                            "rccl=%d lib=%s first_gather_ms=%.3f; HIP runtime %d\n", P, o->transport_ipc ? "ipc" : o->transport_shm ? "shm" : "rccl", (int)owners,
                            (int)owners == P ? "ncclCommCount" : "nranks_argument", (int)nr_min, (int)nr_max, (int)rk_sum, ver, lib, first_ms,
                            nb_hip_runtime_version());
                if (!o->transport_shm && ((int)owners != P || (int)nr_min != P || (int)nr_max != P || (int)rk_sum != P * (P - 1) / 2)) bad = 1;
            }
            if (rank == 0) {
                const double per = best / (double)o->steps;
                printf("\t%7u\t%7d\t%7s\t%11.2f\t%10.2f\t%11.3e\t%9.1f\t%9.4f\t%9.4f", n, P, MODE_NAME[mode], per * 1e6, 1.0 / per,
                       pairs / per, pairs / per * 14.0 / (157.3e12 * P) * 100.0, k_ms, c_ms);
                if (o->speedup) printf("\t%10.2f\t%9.2f", single_s * 1e6, single_s / per);
                printf("\n");
                fflush(stdout);
            }
        }
        DestroyWorld(w);
        free(ps);
    }
    free(universe);
    nb_rank_barrier(pg, "end of the table");
    return unverified ? EXIT_VERIFY_FAILED : bad ? 1 : 0;
}

/* the page alone, no GPU: what `pytest -m "not gpu"` can run of the multi-process path */
static int selftest_rank(const Options *o, NbRankPage *pg) {
    const int rank = nb_rank_page_rank(pg), P = nb_rank_page_nranks(pg);
    int bad = 0;
    nb_rank_barrier(pg, "selftest start");
    if (rank == o->selftest_die) _exit(7); /* the others are left waiting at the next barrier */
    if (rank == o->selftest_hang)
        for (;;) pause(); /* a collective that never completes: only the parent's budget ends this */
    for (int round = 0; round < 3; round++) {
        unsigned char id[NB_RANK_ID_BYTES];
        memset(id, 0, sizeof id);
        if (rank == 0)
            for (int i = 0; i < NB_RANK_ID_BYTES; i++) id[i] = (unsigned char)(i * 7 + round);
        nb_rank_share_id(pg, id);
        for (int i = 0; i < NB_RANK_ID_BYTES; i++) bad = bad || id[i] != (unsigned char)(i * 7 + round);
    }
    bad = bad || nb_rank_reduce(pg, (double)(rank + 1), 'x') != (double)P;
    bad = bad || nb_rank_reduce(pg, (double)(rank + 1), 'n') != 1.0;
    bad = bad || nb_rank_reduce(pg, (double)(rank + 1), 's') != (double)(P * (P + 1) / 2);
    bad = bad || !nb_rank_all_equal(pg, 42) || (P > 1 && nb_rank_all_equal(pg, (uint64_t)rank));
    for (uint32_t per = 1; per <= 65536 && !bad; per *= 16) {
        uint32_t *buf = (uint32_t *)malloc((size_t)per * 4 * (size_t)P);
        for (int it = 0; it < 50; it++) {
            memset(buf, 0xff, (size_t)per * 4 * (size_t)P);
            for (uint32_t i = 0; i < per; i++) buf[(size_t)rank * per + i] = (uint32_t)rank * 1000003u + i * 31u + (uint32_t)it;
            nb_rank_allgather(pg, buf, (uint64_t)per * 4, rank, P);
            for (int q = 0; q < P; q++)
                for (uint32_t i = 0; i < per; i++) bad = bad || buf[(size_t)q * per + i] != (uint32_t)q * 1000003u + i * 31u + (uint32_t)it;
        }
        free(buf);
    }
    bad = nb_rank_reduce(pg, (double)bad, 'x') > 0.0;
    if (rank == 0)
        printf("rank page selftest %s: %d ranks, %llu all-gathers per rank\n", bad ? "FAILED" : "ok", P, (unsigned long long)nb_rank_gather_calls(pg));
    return bad;
}

/* fork the ranks (nothing has touched HIP yet), wait for all of them, end the stragglers once one has failed */
static int run_ranks_once(const Options *o, double limit_s, int *timed_out) {
    const int P = o->gpus;
    const double started = seconds_now();
    if (timed_out) *timed_out = 0;
    uint32_t max_n = 0;
    for (uint32_t s = 0; s < o->n_sizes; s++) max_n = o->sizes[s] > max_n ? o->sizes[s] : max_n;
    /* largest exchange: the particle slices of a collective read-back, P x (Mc + Zc) records of 32 bytes with
     * Mc + Zc <= 2 x (N / P + 65) (shard_plan.hip rounds both chunks up to 64) */
    /* ... and never less than the preflight's IPC handles (72 bytes per rank) */
    const size_t exchange = o->selftest_ranks ? (size_t)P * 65536 * 4 : o->transport_shm ? 64 * ((size_t)max_n + 65 * (size_t)P) + 4096 : 8192;
    const double wait_s = limit_s > 0.0 && limit_s < o->wait_timeout_s ? limit_s : o->wait_timeout_s;
    NbRankPage *pg = nb_rank_page_create(P, exchange, wait_s);
    if (!pg) {
        perror("nbody-bench: cannot map the shared rank page");
        return 2;
    }
    /* multi-process GPU work on this pool needs dmabuf IPC (RCCL's P2P handles) */
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    if (o->force_sharded) setenv("NB_HIP_FORCE_SHARDED", "1", 1);
    fflush(stdout);
    fflush(stderr);
    pid_t pids[NB_RANKS_MAX];
    const pid_t parent = getpid();
    for (int r = 0; r < P; r++) {
        pids[r] = fork();
        if (pids[r] < 0) {
            perror("nbody-bench: fork");
            nb_rank_page_fail(pg);
            for (int q = 0; q < r; q++) kill(pids[q], SIGKILL);
            return 2;
        }
        if (pids[r] == 0) {
            /* a rank never outlives the parent that started it (a parent ended from outside must not leave ranks on the GPUs) */
            prctl(PR_SET_PDEATHSIG, SIGKILL);
            if (getppid() != parent) _exit(4);
            nb_rank_page_attach(pg, r);
            const int rc = o->selftest_ranks ? selftest_rank(o, pg) : run_rank(o, pg);
            fflush(stdout);
            fflush(stderr);
            _exit(rc);
        }
    }
    int alive = P, worst = 0;
    double failed_at = 0.0;
    while (alive > 0) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, WNOHANG);
        if (pid > 0) {
            int r = 0;
            while (r < P && pids[r] != pid) r++;
            if (r == P) continue;
            pids[r] = 0;
            alive--;
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
            if (rc != 0) {
                fprintf(stderr, "nbody-bench: rank %d (pid %ld) ended with status %d\n", r, (long)pid, rc);
                if (worst == 0) worst = rc;
                nb_rank_page_fail(pg); /* every rank waiting at the page leaves on its own (exit 4) */
                if (failed_at == 0.0) failed_at = seconds_now();
            }
            continue;
        }
        if (pid < 0 && errno == ECHILD) break; /* nothing left to wait for */
        struct timespec ts = {0, 20 * 1000 * 1000};
        nanosleep(&ts, NULL);
        if (limit_s > 0.0 && failed_at == 0.0 && seconds_now() - started > limit_s) {
            /* the attempt has used up its share of the budget: end exactly the children this process started */
            fprintf(stderr, "nbody-bench: the attempt outlived its %.0f s share of the budget; ending its %d rank(s)\n", limit_s, alive);
            if (timed_out) *timed_out = 1;
            nb_rank_page_fail(pg);
            for (int r = 0; r < P; r++)
                if (pids[r] > 0) kill(pids[r], SIGKILL);
            failed_at = seconds_now() + 1e9;
            if (worst == 0) worst = 4;
        }
        if (failed_at > 0.0 && seconds_now() - failed_at > 15.0) {
            /* a rank stuck inside a collective whose peer is gone: end exactly the children this process started */
            for (int r = 0; r < P; r++)
                if (pids[r] > 0) {
                    fprintf(stderr, "nbody-bench: ending rank %d (pid %ld) 15 s after another rank failed\n", r, (long)pids[r]);
                    kill(pids[r], SIGKILL);
                }
            failed_at = seconds_now() + 1e9; /* once */
        }
    }
    nb_rank_page_destroy(pg);
    return worst;
}

/* --transport auto: rccl -> ipc -> shm, every attempt in FRESH processes (this parent never touches HIP, so nothing is retried
 * inside a process that did), every attempt inside its share of ONE budget.  A verification failure (status 5) is kept
 * apart from a bring-up failure: the chain goes on, but the run never reports success. */
static int run_ranks(const Options *o) {
    if (o->selftest_ranks) return run_ranks_once(o, o->budget_given ? o->budget_s - 1.0 : 0.0, NULL);
    static const char *const NAME[3] = {"rccl", "ipc", "shm"};
    const double t0 = seconds_now();
    const int first = o->transport_auto ? 0 : o->transport_ipc ? 1 : o->transport_shm ? 2 : 0, last = o->transport_auto ? 2 : first;
    /* the library's own bound on a wait for other ranks (default 180 s): shortened while a fallback exists -- unless the
     * user exported a value, which is theirs to keep (and to get back) */
    const int timeout_is_the_users = getenv("NB_HIP_COMM_TIMEOUT_S") != NULL;
    int rc = 0, unverified = 0;
    for (int t = first; t <= last; t++) {
        Options a = *o;
        a.transport_ipc = t == 1;
        a.transport_shm = t >= 1;
        if (t >= 1) { /* a host barrier cannot be captured */
            a.n_modes = 0;
            for (int i = 0; i < o->n_modes; i++)
                if (o->modes[i] != MODE_GRAPH) a.modes[a.n_modes++] = o->modes[i];
            if (a.n_modes == 0) a.modes[a.n_modes++] = MODE_PLAIN;
        }
        double limit = 0.0;
        if (o->budget_s > 0.0) {
            const double left = o->budget_s - (seconds_now() - t0) - 5.0;
            limit = t < last ? left * 0.5 : left;
            if (limit < 5.0) {
                fprintf(stderr, "nbody-bench: the budget of %.0f s is used up before the %s attempt\n", o->budget_s, NAME[t]);
                printf("# budget of %.0f s used up before the %s attempt\n", o->budget_s, NAME[t]);
                return rc ? rc : 4;
            }
        }
        if (!timeout_is_the_users) {
            char buf[32];
            const double bound = t < last ? (limit > 0.0 && limit * 0.5 < 120.0 ? limit * 0.5 : 120.0) : (limit > 0.0 && limit * 0.8 < 180.0 ? limit * 0.8 : 180.0);
            snprintf(buf, sizeof buf, "%d", bound < 5.0 ? 5 : (int)bound);
            setenv("NB_HIP_COMM_TIMEOUT_S", buf, 1);
        }
        int timed_out = 0;
        rc = run_ranks_once(&a, limit, &timed_out);
        if (!timeout_is_the_users) unsetenv("NB_HIP_COMM_TIMEOUT_S"); /* only what this process set */
        if (rc == 0) break;
        if (rc == EXIT_VERIFY_FAILED) {
            unverified = 1;
            fprintf(stderr, "nbody-bench: verification_failed over %s: the ranks disagreed or differed from the single-GPU World (status 5); "
                    "this is a wrong answer, not a bring-up failure\n", NAME[t]);
            printf("# verification_failed transport %s (status 5): the rows above are NOT verified\n", NAME[t]);
        }
        if (t == last) break;
        fprintf(stderr, "nbody-bench: transport_fallback %s -> %s: %s (status %d); starting %d fresh rank processes\n", NAME[t], NAME[t + 1],
                rc == EXIT_VERIFY_FAILED ? "verification failed" : timed_out ? "the attempt outlived its share of the budget" : "a rank of the attempt ended with an error",
                rc, o->gpus);
        printf("# transport_fallback %s -> %s (%s, status %d); what follows is the next attempt: fresh rank processes\n", NAME[t], NAME[t + 1],
               rc == EXIT_VERIFY_FAILED ? "verification_failed" : timed_out ? "timed_out" : "bring_up_failed", rc);
        fflush(stdout);
    }
    if (unverified && rc == 0) {
        fprintf(stderr, "nbody-bench: a later transport delivered, but an earlier one failed its verification: exit status 5\n");
        return EXIT_VERIFY_FAILED;
    }
    return rc;
}

static int parse_modes(Options *o, const char *list) {
    o->n_modes = 0;
    char tmp[128];
    snprintf(tmp, sizeof tmp, "%s", list);
    for (char *tok = strtok(tmp, ","); tok && o->n_modes < 8; tok = strtok(NULL, ",")) {
        int m = -1;
        for (int i = 0; i < MODE_COUNT; i++)
            if (!strcmp(tok, MODE_NAME[i])) m = i;
        if (m < 0) return -1;
        o->modes[o->n_modes++] = m;
    }
    return o->n_modes > 0 ? 0 : -1;
}

int main(int argc, char **argv) {
    Options o;
    memset(&o, 0, sizeof o);
    o.use_cpu = o.use_gpu = true;
    o.steps = 100, o.warmup = 10, o.galaxies = 2, o.repeats = 1, o.verify_steps = 3, o.transport_auto = true;
    o.seed = 11037;
    o.dt = 1.f;
    o.gpus = 1;
    o.wait_timeout_s = 180.0;
    o.budget_s = 480.0;
    o.selftest_die = o.selftest_hang = -1;
    const char *modes = NULL;

    for (int a = 1; a < argc; a++) {
        const char *arg = argv[a];
        const char *val = a + 1 < argc ? argv[a + 1] : NULL;
        if (!strcmp(arg, "--cpu")) {
            o.use_gpu = false;
        } else if (!strcmp(arg, "--gpu")) {
            o.use_cpu = false;
        } else if (!strcmp(arg, "--n") && val && o.n_sizes < 64) {
            o.sizes[o.n_sizes++] = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--steps") && val) {
            o.steps = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--warmup") && val) {
            o.warmup = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--dt") && val) {
            o.dt = strtof(val, NULL), a++;
        } else if (!strcmp(arg, "--galaxies") && val) {
            o.galaxies = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--seed") && val) {
            o.seed = (unsigned)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--repeats") && val) {
            o.repeats = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--floor-rate") && val) {
            o.floor_rate = strtod(val, NULL), a++;
        } else if (!strcmp(arg, "--own-rng")) {
            o.own_rng = true;
        } else if (!strcmp(arg, "--cpu-best")) {
            o.cpu_best = true;
        } else if (!strcmp(arg, "--gpus") && val) {
            o.gpus = atoi(val), a++;
        } else if (!strcmp(arg, "--transport") && val && !strcmp(val, "auto")) {
            o.transport_auto = true, o.transport_ipc = o.transport_shm = false, a++;
        } else if (!strcmp(arg, "--transport") && val && (!strcmp(val, "rccl") || !strcmp(val, "shm") || !strcmp(val, "ipc"))) {
            o.transport_auto = false, o.transport_ipc = !strcmp(val, "ipc"), o.transport_shm = strcmp(val, "rccl") != 0, a++;
        } else if (!strcmp(arg, "--modes") && val) {
            modes = val, a++;
        } else if (!strcmp(arg, "--verify") && val) {
            o.verify_steps = (uint32_t)strtoul(val, NULL, 0), o.verify_given = true, a++;
        } else if (!strcmp(arg, "--wait-timeout") && val) {
            o.wait_timeout_s = strtod(val, NULL), a++;
        } else if (!strcmp(arg, "--budget-s") && val) {
            o.budget_s = strtod(val, NULL), o.budget_given = true, a++;
        } else if (!strcmp(arg, "--force-sharded")) {
            o.force_sharded = true;
        } else if (!strcmp(arg, "--one-wave")) {
            o.one_wave = true;
        } else if (!strcmp(arg, "--speedup")) {
            o.speedup = true;
        } else if (!strcmp(arg, "--selftest-ranks")) {
            o.selftest_ranks = true;
        } else if (!strcmp(arg, "--selftest-die") && val) {
            o.selftest_die = atoi(val), a++;
        } else if (!strcmp(arg, "--selftest-hang") && val) {
            o.selftest_hang = atoi(val), a++;
        } else {
            fprintf(stderr,
                    "usage: %s [--cpu|--gpu] [--n N]... [--steps K] [--warmup W] [--dt DT] [--galaxies G] [--seed S]"
                    " [--own-rng] [--repeats R] [--cpu-best] [--floor-rate INT_PER_S]\n"
                    "       [--gpus P [--transport auto|rccl|ipc|shm] [--modes plain,overlap,graph] [--verify K] [--speedup] [--force-sharded] [--one-wave]"
                    " [--wait-timeout S] [--budget-s S] [--selftest-ranks]]\n",
                    argv[0]);
            return 2;
        }
    }
    if (o.n_sizes == 0) {
        o.n_sizes = sizeof REFERENCE_SIZES / sizeof REFERENCE_SIZES[0];
        memcpy(o.sizes, REFERENCE_SIZES, sizeof REFERENCE_SIZES);
    }
    if (o.steps == 0) o.steps = 1;
    if (o.repeats == 0) o.repeats = 1;
    if (o.gpus < 1 || o.gpus > NB_RANKS_MAX) {
        fprintf(stderr, "nbody-bench: --gpus must be 1..%d\n", NB_RANKS_MAX);
        return 2;
    }
    if (parse_modes(&o, modes ? modes : (o.transport_shm ? "plain,overlap" : "plain,overlap,graph")) != 0) {
        fprintf(stderr, "nbody-bench: --modes takes a comma list of plain, overlap, graph\n");
        return 2;
    }
    if (o.transport_shm)
        for (int i = 0; i < o.n_modes; i++)
            if (o.modes[i] == MODE_GRAPH) {
                fprintf(stderr, "nbody-bench: mode graph needs --transport rccl (a host callback or barrier cannot run inside a captured graph)\n");
                return 2;
            }
    if (o.gpus > 1 && o.verify_steps == 0 && !o.selftest_ranks) {
        /* a multi-rank table without its check is a table of unverified numbers: the check is not optional (one step is enough) */
        fprintf(stderr, "nbody-bench: --verify 0 is not accepted with --gpus %d: every multi-rank row is checked (ranks agree, = one GPU); using --verify 1\n", o.gpus);
        o.verify_steps = 1;
    }
    /* the other ranks wait at the page while rank 0 times the single-GPU reference: K steps of the whole world on one GPU */
    if (o.speedup && o.wait_timeout_s == 180.0) o.wait_timeout_s = 3600.0;
    if (o.speedup && o.budget_s == 480.0) o.budget_s = 0.0; /* a whole-world single-GPU timing per row does not fit a default budget */
    if (o.gpus > 1 || o.force_sharded || o.selftest_ranks) return run_ranks(&o);
    return run_single(&o);
}
