/*
 * world.c -- the include/nbody.h surface: World, its mass partition, and the
 * lazy host<->device coherence between the particle array and the HIP pipeline.
 *
 * Follows the behaviour of the reference's src/lib/world.c:
 *   - CreateWorld copies the caller's particles and partitions them "mass > 0
 *     first" with the reference's two-cursor swap scheme (world.c:32-46), because
 *     that exact permutation is the index order every later read returns
 *     (pinned by reference test/test_particle_sort.c:27-111);
 *   - two dirty flags decide when data crosses PCIe (world.c:18-19,76-89): the
 *     array is uploaded only if the CPU side changed it since the last GPU step,
 *     and downloaded only when GetWorldParticles / UpdateWorld_CPU needs it;
 *   - UpdateWorld_GPU(n == 0) does nothing (world.c:113); UpdateWorld_CPU(n == 0)
 *     still pulls the device state and marks the array dirty (world.c:100,109).
 * Differences: the GPU side is created lazily (no device is touched by a World
 * that only ever steps on the CPU), and DestroyWorld frees the particle array
 * (the reference leaks it, world.c:67-73).
 */
#include "nbody.h"
#include "nbody_hip.h"

#include <stdbool.h>

#include "nb_util.h"
#include "sim_cpu.h"

struct World {
    Particle *particles;  /* partitioned copy of the caller's array */
    uint32_t count;       /* all particles */
    uint32_t massive;     /* particles with mass > 0; they occupy [0, massive) */
    SimPipeline *gpu;     /* HIP pipeline (include/nbody_hip.h) */
    CpuSim *cpu;          /* host-core stepper */
    bool host_is_newer;   /* array changed since the device last saw it */
    bool device_is_newer; /* device stepped since the array was last refreshed */
};

/*
 * In-place unstable partition, massive particles first; returns their count.
 * `lo` hunts upward for a massless slot, `hi` downward for a massive one, and
 * they swap until they meet -- the reference's scheme, kept because its output
 * permutation is part of the observable contract.
 */
static uint32_t partition_by_mass(Particle *p, uint32_t count) {
    uint32_t lo = 0, hi = count;
    for (;;) {
        for (; lo < hi && p[lo].mass > 0; lo++) {
        }
        while (lo < hi) {
            hi--;
            if (!(p[hi].mass <= 0)) break;
        }
        if (lo == hi) return hi;
        const Particle keep = p[lo];
        p[lo] = p[hi];
        p[hi] = keep;
    }
}

/* rank < 0: an ordinary single-GPU World; otherwise the sharded pipeline of include/nbody_hip.h. */
static World *create_world(const Particle *ps, uint32_t size, int rank, int nranks, const void *unique_id128,
                           NbAllGatherFn allgather, void *ctx, bool direct) {
    World *w = NB_NEW(1, World);
    NB_CHECK(w != NULL, "Failed to alloc World");
    w->particles = NB_NEW(size ? size : 1, Particle);
    NB_CHECK(w->particles != NULL, "Failed to alloc %u particles", size);
    if (size) memcpy(w->particles, ps, (size_t)size * sizeof(Particle));

    w->count = size;
    w->massive = partition_by_mass(w->particles, size);
    const WorldData data = {.total_len = size, .mass_len = w->massive, .dt = 0.0f};
    w->gpu = rank < 0     ? CreateSimPipeline(data)
             : direct    ? CreateSimPipelineShardedDirect(data, rank, nranks, allgather, ctx)
             : allgather ? CreateSimPipelineShardedWith(data, rank, nranks, allgather, ctx)
                         : CreateSimPipelineSharded(data, rank, nranks, unique_id128);
    /* this array is what every later Set/GetSimulationData moves: let the pipeline page-lock it when (if) it
     * first touches the GPU.  DestroyWorld destroys the pipeline before freeing the array. */
    nb_hip_note_host_array(w->gpu, w->particles, (uint64_t)size * sizeof(Particle));
    w->cpu = CpuSimCreate(w->massive);
    w->host_is_newer = true;    /* the device has seen nothing yet */
    w->device_is_newer = false;
    return w;
}

World *CreateWorld(const Particle *ps, uint32_t size) { return create_world(ps, size, -1, 1, NULL, NULL, NULL, false); }

/*
 * Extension: one World per process and GPU.  The partition is deterministic, so every rank derives the same
 * mass_len and the same index order from the same input; the pipeline then owns the rank's 1/P of the receivers
 * (nb_hip_shard_plan) and the coherence protocol above is unchanged -- Set/Get/Perform are simply collectives.
 */
World *CreateWorldSharded(const Particle *ps, uint32_t size, int rank, int nranks, const void *unique_id128) {
    NB_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    return create_world(ps, size, rank, nranks, unique_id128, NULL, NULL, false);
}

/* Extension: the same over a caller-supplied host all-gather (several ranks on ONE GPU, machines without RCCL). */
World *CreateWorldShardedWith(const Particle *ps, uint32_t size, int rank, int nranks, NbAllGatherFn allgather, void *ctx) {
    NB_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    NB_CHECK(allgather != NULL, "NULL all-gather callback");
    return create_world(ps, size, rank, nranks, NULL, allgather, ctx, false);
}

/* Extension: the same with the direct device-to-device exchange; `control` carries handles and barriers only. */
World *CreateWorldShardedDirect(const Particle *ps, uint32_t size, int rank, int nranks, NbAllGatherFn control, void *ctx) {
    NB_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    NB_CHECK(control != NULL, "NULL control callback");
    return create_world(ps, size, rank, nranks, NULL, control, ctx, true);
}

SimPipeline *GetWorldPipeline(World *w) { return w ? w->gpu : NULL; }

void DestroyWorld(World *w) {
    if (w == NULL) return;
    DestroySimPipeline(w->gpu);
    CpuSimDestroy(w->cpu);
    free(w->particles);
    free(w);
}

static void push_if_stale(World *w) {
    if (!w->host_is_newer) return;
    SetSimulationData(w->gpu, w->particles);
    w->host_is_newer = false;
}

static void pull_if_stale(World *w) {
    if (!w->device_is_newer) return;
    GetSimulationData(w->gpu, w->particles);
    w->device_is_newer = false;
}

const Particle *GetWorldParticles(World *w, uint32_t *size) {
    pull_if_stale(w);
    if (size != NULL) *size = w->count;
    return w->particles;
}

void UpdateWorld_CPU(World *w, float dt, uint32_t n) {
    pull_if_stale(w);
    for (uint32_t step = 0; step < n; step++) CpuSimStep(w->cpu, w->particles, w->count, w->massive, dt);
    w->host_is_newer = true;
}

void UpdateWorld_GPU(World *w, float dt, uint32_t n) {
    if (n == 0) return;
    push_if_stale(w);
    PerformSimUpdate(w->gpu, n, dt);
    w->device_is_newer = true;
}
