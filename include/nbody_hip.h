/*
 * nbody_hip.h -- C-ABI of the HIP/gfx950 simulation pipeline (libnbody_hip.so).
 *
 * This is the drop-in boundary.  Part 1 is, symbol for symbol, the seam the
 * reference's world layer calls into its Vulkan backend through
 * (reference src/lib/sim_gpu.h:8-42, called from src/lib/world.c:52,69,78,86,115):
 * a maintainer deletes src/lib/sim_gpu.c, src/lib/vulkan_ctx.c and
 * src/shader/particle_cs.glsl, links this library, and world.c is unchanged
 * (INTEGRATION.md shows it; oracle/_ref/libnbody_ref_world.so is exactly that
 * build).  Part 2 adds what a single-queue Vulkan backend had no notion of:
 * device-resident stepping, kernel timing, and the N/P sharded multi-GPU
 * pipeline with its per-step all-gather of source positions over RCCL.
 *
 * Plain C types only; no HIP, RCCL or torch types cross this boundary.
 * Error convention = the reference's (src/lib/util.h:17-29,47-60): any failure
 * prints "file:line [func] ..." to stderr and abort()s.  There is NO CPU
 * fallback: a call that needs the GPU aborts when no gfx950 device answers.
 */
#ifndef NBODY_AMD_NBODY_HIP_H
#define NBODY_AMD_NBODY_HIP_H

#include <stdint.h>
#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Part 1: the reference seam (src/lib/sim_gpu.h)                             */
/* ------------------------------------------------------------------------- */

/* Sizes of a world and its step; replaces reference sim_gpu.h:8-12 (the uniform block). */
typedef struct WorldData {
    uint32_t total_len; /* all particles */
    uint32_t mass_len;  /* particles with mass > 0; they come first */
    float dt;           /* cached step size; PerformSimUpdate's dt wins */
} WorldData;

/* One world's device state + launch machinery; replaces reference sim_gpu.h:15. */
typedef struct SimPipeline SimPipeline;

/*
 * Replaces reference sim_gpu.h:21 (CreateSimPipeline, sim_gpu.c:34-221).
 * Allocates nothing on the GPU yet: device selection, HBM buffers and graphs are
 * made at the first SetSimulationData, so a World used only through
 * UpdateWorld_CPU never touches a GPU (SURVEY.md 8b "init side effects").
 */
SimPipeline *CreateSimPipeline(WorldData data);

/* Replaces reference sim_gpu.h:24 (sim_gpu.c:223-247). NULL is accepted. */
void DestroySimPipeline(SimPipeline *sim);

/*
 * Replaces reference sim_gpu.h:27 (sim_gpu.c:249-251): writes total_len
 * Particles (AoS, partitioned order) holding the device's latest state into ps.
 * Here this is where the D2H copy happens (the reference pays it after every
 * PerformSimUpdate, sim_gpu.c:336-341).
 */
void GetSimulationData(const SimPipeline *sim, Particle *ps);

/*
 * Replaces reference sim_gpu.h:30 (sim_gpu.c:253-256): uploads total_len
 * Particles (AoS, already partitioned: ps[i].mass > 0 exactly for i < mass_len)
 * and splits them into the SoA streams the kernels read.
 */
void SetSimulationData(SimPipeline *sim, const Particle *ps);

/*
 * Replaces reference sim_gpu.h:36 (sim_gpu.c:258-361): n > 0 steps of size dt,
 * blocking.  Simulation data must have been set.  n == 0 is a no-op here
 * (the reference documents n > 0; world.c:113 never passes 0).
 */
void PerformSimUpdate(SimPipeline *sim, uint32_t n, float dt);

/* ------------------------------------------------------------------------- */
/* Part 2: extensions (no reference counterpart)                              */
/* ------------------------------------------------------------------------- */

/* Number of visible HIP devices; 0 when there is none.  Never aborts. */
int nb_hip_device_count(void);

/* Device ordinal this process' pipelines are created on (default: 0, or LOCAL_RANK for sharded ones). */
void nb_hip_set_device(int ordinal);

/* Fills buf (NUL-terminated, at most len bytes) with "name arch CUs clockMHz pci=domain:bus:device"; aborts without a GPU. */
void nb_hip_device_info(char *buf, uint32_t len);

/* PerformSimUpdate without the final host wait: enqueue n steps and return. */
void nb_hip_step_async(SimPipeline *sim, uint32_t n, float dt);

/* Wait for everything enqueued on the pipeline's stream(s). */
void nb_hip_sync(SimPipeline *sim);

/*
 * Device time, in milliseconds, of the force+integrate kernels of the most
 * recent PerformSimUpdate / nb_hip_step_async (HIP events recorded on the
 * launch stream around the step chain; waits for them).  Needs the "timing"
 * knob (nb_hip_configure(sim, "timing", 1)); returns 0 without it.  *launches receives the
 * number of step-kernel launches those events bracket.
 */
double nb_hip_last_step_ms(SimPipeline *sim, uint32_t *launches);

/*
 * Finish-kernel launches inside the interval nb_hip_last_step_ms reports (one per step-kernel launch when the
 * launch shape splits the sources, else 0): the small O(N) kernel that adds the parts and integrates.
 */
uint32_t nb_hip_last_finish_launches(const SimPipeline *sim);

/*
 * Sharded pipelines (plain-launch chains): where the time of the most recent PerformSimUpdate went.  *kernel_ms =
 * sum over its steps of the force+integrate kernels' device time, *comm_ms = sum of the all-gathers' device time
 * (each bracketed by its own HIP event pair on the stream it ran on; in overlap mode the gather runs beside the
 * kernels, so the two do not add up to the wall time).  Returns the number of steps covered (the first 256 of a
 * call); 0 for unsharded pipelines and for chains replayed as a captured hipGraph.
 */
uint32_t nb_hip_last_step_breakdown(SimPipeline *sim, double *kernel_ms, double *comm_ms);

/*
 * What the pipeline's RCCL communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice /
 * ncclGetVersion), the device time of the probe all-gather run at creation, and the file librccl was loaded from.
 * Returns 1 when the pipeline owns a communicator, else 0 (then nranks/rank are the creation arguments).
 * Any out pointer may be NULL.  This is the evidence that N ranks really formed one communicator.
 */
int nb_hip_comm_info(const SimPipeline *sim, int *nranks, int *rank, int *device, int *rccl_version,
                     double *first_gather_ms, char *lib_path, uint32_t len);

/*
 * Bring-up timings of the pipeline's RCCL communicator, taken once at creation: host milliseconds of ncclCommInitRank,
 * device milliseconds of the first (256-byte-per-rank, verified) all-gather including RCCL's lazy channel set-up, and
 * device microseconds of ONE warm 8-byte-per-rank all-gather (mean of 16 issued back to back in-stream): the fixed cost
 * of the per-step gather.  Returns 1 when the pipeline owns a communicator, else 0 (all three read 0 then).
 */
int nb_hip_comm_bringup(const SimPipeline *sim, double *init_ms, double *first_gather_ms, double *small_gather_us);

/*
 * Preflight probes for multi-GPU harnesses (bench.py, nbody-bench): asked BEFORE the first real contact between ranks, so
 * that a failed bring-up can say why.  Unlike the rest of this ABI they REPORT instead of aborting (they still abort when
 * this process has no gfx950 device at all).
 *   nb_hip_preflight_peers       row[q] = hipDeviceCanAccessPeer(this process' device, q): 1 / 0, 1 for the device itself,
 *                                -1 when the query failed, -2 for q >= the visible device count (returned)
 *   nb_hip_preflight_ipc_export  allocates one 4 KiB device word holding `tag` and writes its hipIpcMemHandle_t (64 bytes)
 *                                to handle64; returns the hipError_t (0 = ok)
 *   nb_hip_preflight_ipc_open    maps a PEER's exported handle (hipIpcOpenMemHandle, lazy peer access), reads the word,
 *                                unmaps; returns 0 when the word is expect_tag, the hipError_t of the failing call, or -1
 *                                when the word read is not the peer's; *ms = host milliseconds of open + read + close
 *   nb_hip_preflight_ipc_release frees the exported word (call after every peer has finished its open)
 *   nb_hip_error_string          text of a code returned by the two calls above
 */
int nb_hip_preflight_peers(int *row, int len);
int nb_hip_preflight_ipc_export(void *handle64, uint32_t tag);
int nb_hip_preflight_ipc_open(const void *handle64, uint32_t expect_tag, double *ms);
void nb_hip_preflight_ipc_release(void);
const char *nb_hip_error_string(int hip_error);

/* Cached hipGraph chains of this pipeline (at most 8, least recently used evicted); *dt_uploads = times a new step
 * size was written to device memory (the kernels read dt from there, like the reference's uniform block,
 * sim_gpu.c:268-284, so a changed dt never rebuilds or patches a cached chain). */
uint32_t nb_hip_graph_stats(const SimPipeline *sim, uint32_t *dt_uploads);

/* hipRuntimeGetVersion() of the HIP runtime this process actually bound (0 when it cannot be asked). */
int nb_hip_runtime_version(void);

/*
 * Measurement aid (no reference counterpart): the shader clock the chip holds under the step kernels' instruction mix.
 * Runs a separate probe kernel -- the interaction statement of the step kernels on scalar source operands, two
 * receivers per lane, 1024-thread workgroups filling every SIMD with 8 waves, no memory traffic in the loop -- for about
 * target_ms milliseconds; every wave stamps s_memtime (shader cycles) and s_memrealtime (constant reference clock) around
 * its loop.  *clock_ghz = d(memtime) / d(memrealtime) x the reference rate, median over the waves that spanned the whole
 * loop (a SIMD favours its oldest wave: the others finish early; min / max over all waves beside it);
 * *cycles_per_wave_interaction = the longest wave's shader cycles / 8 waves per SIMD / interactions one wave issued
 * (26 = the floor of this instruction mix: 9 plain fp32 VALU at 2 cycles + one v_rsq_f32 at 8).  The product kernels carry
 * no stamps.  Any out pointer may be NULL.  Returns the number of waves that reported.
 */
int nb_hip_probe_clock(double target_ms, double *clock_ghz, double *clock_ghz_min, double *clock_ghz_max,
                       double *cycles_per_wave_interaction, double *elapsed_ms);

/*
 * Measurement aid: the shader clock WHILE other work runs.  begin launches 8 one-wave workgroups (one per XCD) on their
 * own stream that stamp s_memtime / s_memrealtime every period_ms and sleep in between -- 8 of the chip's 8192 wave slots,
 * a few scalar instructions per period -- until end is called or max_ms has passed (every wave leaves by itself then).
 * end stops them and reports, over all sampled intervals: median / min / max clock in GHz, the median per XCD
 * (per_xcd_ghz8[8], 0 where no wave sat), the mean clock of each tenth of the sampled span as one wave saw it (profile10[10]) and
 * the span in milliseconds.  Intervals in which a counter did not move forward by a sane amount are dropped and counted
 * (*dropped_intervals).  Returns the number of intervals kept.  Any out pointer may be NULL.  One sampler per process.
 * bench.py brackets a repeat of the headline leg with it (roofline.held_clock_ghz): the probe above loads the chip with a
 * denser loop than the step kernel's and therefore reads a lower clock than the step kernel holds.
 */
#define NB_CLOCK_SAMPLER_MAX_MS 20000.0 /* period_ms below 0.05 and max_ms above this are clamped, never refused */
int nb_hip_clock_sampler_begin(double period_ms, double max_ms);
int nb_hip_clock_sampler_end(double *clock_ghz, double *clock_ghz_min, double *clock_ghz_max, double *per_xcd_ghz8,
                             double *profile10, double *span_ms, uint32_t *dropped_intervals);

/*
 * Optional: tell the pipeline which long-lived host array Set/GetSimulationData will be called with (the World's
 * particle array).  It is page-locked (hipHostRegister) when the pipeline first touches the GPU and released in
 * DestroySimPipeline, so the hand-over runs at PCIe speed instead of through pageable memory.  The array must stay
 * allocated until DestroySimPipeline or until this is called again (array = NULL forgets it).  Set/Get with any
 * other pointer keep working unchanged.  In a frame loop (every blocking update followed by a Get into this array,
 * reference src/main.c:157-163,237) the pipeline appends the read-back to the update's own submission.
 */
void nb_hip_note_host_array(SimPipeline *sim, void *array, uint64_t bytes);

/*
 * Run-time knobs a user of the reference harness needs.  key is one of:
 *   "variant"   how the wave-uniform sources reach the VALU: 1 (default) = through the scalar cache as SGPR operands,
 *               0 = wave-private LDS tiles (north star's design; bit-identical results, 8 % slower at N = 2^20: every
 *               bench.py line times both, roofline.alt_lds)
 *   "graph"     how PerformSimUpdate(n > 1) runs its chain: 0 = plain stream launches; 1 = always as a hipGraph, built on
 *               first use and cached per (length, ping-pong phase); 2 (default) = auto: calls shorter than 16 steps are
 *               plain launches; longer calls on small worlds (N x M <= 6e7) replay ONE canonical 32-step chain built when
 *               the data first reaches the device; on larger worlds a chain length runs as plain launches the first time
 *               it is asked for and as a cached hipGraph from the second time on.  The step size is never baked into a
 *               chain: kernels read it from device memory
 *   "timing"    1 = bracket every chain with a HIP event pair so that nb_hip_last_step_ms can answer, 0 (default) =
 *               do not (the two records cost a frame loop 3-7 us per call)
 *   "overlap"   sharded pipelines: 1 = split each step into own-shard / remote-shard kernels with the all-gather in
 *               between on a second stream, 0 = gather then one kernel (default)
 *   "sharded_graph"  sharded pipelines: 1 = capture the {kernel, all-gather} x n chain into a hipGraph and replay it
 *               (non-overlapped step only); 0 = plain stream launches (default)
 *               Both are opt-in by policy: per rank a step is O(N*M/P) of kernel against an O(M)
 *               gather, and every harness times all three modes from one command.
 * The same five can be preset from the environment: NB_HIP_VARIANT, NB_HIP_GRAPH, NB_HIP_OVERLAP, NB_HIP_SHARDED_GRAPH
 * (+ NB_HIP_FORCE_SHARDED = keep the RCCL path for a single rank, NB_HIP_COMM_TIMEOUT_S = bound of every wait on other
 * ranks).  Launch-shape and experiment knobs (receivers per lane, waves per workgroup, source split / passes / granule,
 * lane-split and one-workgroup chains, the fused finish, the frame loop's eager read-back) are NOT part of this ABI: every
 * one is on "auto", auto is what every published number was measured with, and the knobs that lost every measurement
 * live on only as test and tooling hooks (nbody_amd/csrc/nbody_hip_tuning.h: nb_hip_tune; their environment variables
 * exist in TUNING=1 builds only).
 * Returns the previous value; aborts on an unknown key or value.
 */
int nb_hip_configure(SimPipeline *sim, const char *key, int value);

/* What the last step launch actually used (everything is on auto unless a tuning hook moved it): k, w, variant, split, workgroups. */
void nb_hip_launch_shape(const SimPipeline *sim, int *k, int *w, int *variant, int *split, uint32_t *workgroups);

/*
 * The launch-shape arithmetic behind "auto", pure host code (usable without a GPU): for n_recv receivers and
 * n_src sources on a chip with compute_units CUs it returns receivers per lane, waves per workgroup, source
 * split and the resulting workgroup count.  Workgroups of one launch all take the same time, so the model
 * minimises rounds * work-per-workgroup, rounds = ceil(workgroups / resident slots).
 */
void nb_hip_plan_launch(uint32_t n_recv, uint32_t n_src, int compute_units, int *k, int *w, int *split, uint32_t *workgroups);

/* -- sharded (multi-GPU) pipeline: one process per GPU, N/P receivers each -- */

#define NB_HIP_UNIQUE_ID_BYTES 128

/* Rank 0 calls this and ships the 128 bytes to every rank (any transport). Wraps ncclGetUniqueId. */
void nb_hip_comm_unique_id(void *out128);

/*
 * Collective over all ranks.  Each rank passes the same WorldData and later the
 * same full particle array; rank r owns the r-th 1/P slice of the massive range
 * and the r-th 1/P slice of the massless range (nb_hip_shard_plan).
 * SetSimulationData / PerformSimUpdate / GetSimulationData keep their meaning and
 * become collectives: Get returns the FULL array on every rank.
 */
SimPipeline *CreateSimPipelineSharded(WorldData data, int rank, int nranks, const void *unique_id128);

/*
 * The same sharded pipeline over a caller-supplied transport instead of RCCL (MPI, gloo, shared memory ...): an
 * in-place all-gather over HOST memory.  buf holds nranks slots of bytes_per_rank bytes; on entry slot `rank` is
 * filled, on return every slot must hold what the owning rank put there.  The callback runs on the caller's thread,
 * inside PerformSimUpdate / GetSimulationData, once per step (source positions, Mc float2 per rank) and once per Get
 * (particle slices); every rank must reach it the same number of times.  The pipeline stages through one page-locked
 * buffer (D2H own slot, wait, callback, H2D all slots), so this route costs a host round trip per step: it exists for
 * machines without RCCL and for exercising the multi-process path with several ranks on ONE GPU (RCCL refuses two
 * ranks on a device).  Everything else -- shard plan, kernels, mirror / gather layout, overlap mode -- is the RCCL
 * path's; "sharded_graph" is ignored (a host callback cannot run inside a captured graph).
 */
/* NbAllGatherFn: include/nbody.h -- void (*)(void *ctx, void *buf, uint64_t bytes_per_rank, int rank, int nranks) */
SimPipeline *CreateSimPipelineShardedWith(WorldData data, int rank, int nranks, NbAllGatherFn allgather, void *ctx);

/*
 * The same sharded pipeline with the DIRECT exchange: no RCCL, no host staging of the data.  Every rank maps every
 * peer's gathered source array (hipIpcGetMemHandle / hipIpcOpenMemHandle, exchanged once through `control`) and, after
 * each step, copies its own slice device-to-device straight into all of them -- on xGMI, which is point-to-point, each of
 * the P - 1 copies rides its own link: the direct all-gather, where a ring would serialise P - 1 hops per link -- then
 * synchronises its stream and meets the other ranks in ONE host-side barrier per step (an 8-byte all-gather of step
 * counters through `control`; every rank must be at the same step or the call aborts).  `control` is the same in-place
 * host all-gather callback as above; it carries the IPC handles at the first SetSimulationData, a barrier at the end of
 * every SetSimulationData (no peer may push into arrays their owner is still initialising), the per-step barrier, the
 * particle slices of a collective GetSimulationData, and one barrier in DestroySimPipeline (which is therefore a
 * collective for these pipelines: nobody unmaps or frees while a peer may still write).  Costs a host round trip per
 * step like the host transport (no hipGraph capture, no overlap gain) but moves no particle data through the host:
 * a fallback for machines without a working RCCL and a way to run P processes on ONE device at device speed.
 */
SimPipeline *CreateSimPipelineShardedDirect(WorldData data, int rank, int nranks, NbAllGatherFn control, void *ctx);

/* The shard arithmetic, pure host code (usable without a GPU). */
typedef struct NbShardPlan {
    uint32_t mass_chunk;   /* Mc: massive slots per rank = all-gather count (uniform)   */
    uint32_t zero_chunk;   /* Zc: massless slots allocated per rank (uniform, >= any)   */
    uint32_t mass_begin;   /* first global massive index owned: rank * Mc, clamped      */
    uint32_t mass_count;   /* owned massive particles (<= Mc)                           */
    uint32_t zero_begin;   /* first global massless index owned (>= mass_len), clamped  */
    uint32_t zero_count;   /* owned massless particles (<= Zc), dealt to level the      */
                           /* per-rank totals mass_count + zero_count                   */
    uint32_t src_padded;   /* nranks * Mc: length of the gathered source array          */
} NbShardPlan;

NbShardPlan nb_hip_shard_plan(uint32_t total_len, uint32_t mass_len, int rank, int nranks);

/*
 * Local transport (testing aid): all nranks shards live in THIS process on the current device and
 * exchange sources by device copies instead of RCCL.  Same shard plan, same kernels, same mirror /
 * gather layout as the RCCL path, so a single-GPU box can verify the sharded arithmetic end to end.
 * out[] receives nranks pipelines; feed each the same full array with SetSimulationData, advance them
 * together with nb_hip_local_group_step (PerformSimUpdate on a member aborts), read any member with
 * GetSimulationData, destroy every member with DestroySimPipeline.
 */
int nb_hip_local_group_create(WorldData data, int nranks, SimPipeline **out);
void nb_hip_local_group_step(SimPipeline **sims, int nranks, uint32_t n, float dt);

/* Library/ABI version: major*10000 + minor*100 + patch. */
int nb_hip_version(void);

#ifdef __cplusplus
}
#endif

#endif /* NBODY_AMD_NBODY_HIP_H */
