/*
 * nbody.h -- public surface of the MI355X-native direct N-body engine.
 *
 * Drop-in for the reference's include/nbody.h (reference include/nbody.h:8-73):
 * same constant, same POD layouts, same five World functions, so a caller
 * written against the reference (its src/bench.c, src/main.c) recompiles
 * against this header without edits.  What sits behind UpdateWorld_GPU is a
 * HIP/gfx950 pipeline (include/nbody_hip.h) instead of a Vulkan one.
 *
 * Contract notes that the reference leaves implicit:
 *   - CreateWorld copies `ps`; the caller keeps and frees its own array.
 *   - The World stores particles partitioned "mass > 0 first"; the pointer
 *     returned by GetWorldParticles is in that order, not input order.
 *   - Every function aborts on failure; there are no error codes.
 *   - Not thread safe; one caller thread per World.
 */
#ifndef NBODY_AMD_NBODY_H
#define NBODY_AMD_NBODY_H

#include <stdint.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Gravitational constant: |g| = NB_G * mass / dist^2  (reference nbody.h:8). */
#define NB_G 10.0f

/* 2-component float vector (reference nbody.h:11-13). */
typedef struct V2 {
    float x, y;
} V2;

#ifdef __cplusplus
#define V2_FROM(X, Y) (V2{(X), (Y)})
#define V2_ZERO       (V2{0.0f, 0.0f})
#else
#define V2_FROM(X, Y) ((V2){.x = (X), .y = (Y)})
#define V2_ZERO       ((V2){.x = 0.0f, .y = 0.0f})
#endif

/* Inline V2 helpers, same names and meaning as reference nbody.h:22-44. */
static inline V2 AddV2(V2 a, V2 b)      { return V2_FROM(a.x + b.x, a.y + b.y); }
static inline V2 SubV2(V2 a, V2 b)      { return V2_FROM(a.x - b.x, a.y - b.y); }
static inline V2 ScaleV2(V2 v, float f) { return V2_FROM(v.x * f, v.y * f); }
static inline float MagV2(V2 v)         { return hypotf(v.x, v.y); }
static inline float SqMagV2(V2 v)       { return v.x * v.x + v.y * v.y; }

/*
 * One simulated body: 32 bytes, field order fixed (reference nbody.h:47-50).
 * `acc` is an output of the last step; `radius` is the RECEIVER-side
 * softening term added to dist^2 (not squared), see DESIGN.md.
 */
typedef struct Particle {
    V2 pos, vel, acc;
    float mass, radius;
} Particle;

#if defined(__cplusplus)
static_assert(sizeof(Particle) == 32, "Particle must stay 32 bytes");
#elif __STDC_VERSION__ >= 201112L
_Static_assert(sizeof(Particle) == 32, "Particle must stay 32 bytes");
#endif

/* Opaque world with a fixed particle count (reference nbody.h:58). */
typedef struct World World;

/* Build a World from a copy of ps[0..size)  (reference nbody.h:61). */
World *CreateWorld(const Particle *ps, uint32_t size);

/* Tear a World down; NULL is accepted (reference nbody.h:64). */
void DestroyWorld(World *w);

/*
 * Latest particle states, partitioned order; *size receives the count when
 * size != NULL.  The pointer stays valid until DestroyWorld
 * (reference nbody.h:67).
 */
const Particle *GetWorldParticles(World *w, uint32_t *size);

/* n steps of size dt on the host cores (reference nbody.h:70). */
void UpdateWorld_CPU(World *w, float dt, uint32_t n);

/* n steps of size dt on the MI355X; returns when they are done (reference nbody.h:73). */
void UpdateWorld_GPU(World *w, float dt, uint32_t n);

/*
 * Extension (no reference counterpart): the same World over several GPUs, one process per GPU.  Every rank passes
 * the same ps[0..size); rank r of nranks steps the r-th 1/P of the particles on its GPU (set with
 * nb_hip_set_device, include/nbody_hip.h) and the ranks exchange source positions once per step over RCCL.
 * unique_id128 = the 128 bytes rank 0 got from nb_hip_comm_unique_id(), carried to the other ranks by any means.
 * All ranks must then make the same World calls in the same order: UpdateWorld_GPU and GetWorldParticles become
 * collectives (GetWorldParticles returns the full array on every rank), UpdateWorld_CPU steps the full array
 * redundantly on every rank (same bits everywhere).  nranks == 1 gives an ordinary World.
 */
World *CreateWorldSharded(const Particle *ps, uint32_t size, int rank, int nranks, const void *unique_id128);

/*
 * Extension: the same sharded World over a caller-supplied host transport instead of RCCL -- an in-place all-gather
 * over host memory (buf holds nranks slots of bytes_per_rank bytes; slot `rank` is filled on entry, every slot on
 * return; see CreateSimPipelineShardedWith in include/nbody_hip.h).  This is how nbody-bench --gpus P --transport shm
 * steps P real processes on ONE GPU, where RCCL refuses duplicate devices.
 */
typedef void (*NbAllGatherFn)(void *ctx, void *buf, uint64_t bytes_per_rank, int rank, int nranks);
World *CreateWorldShardedWith(const Particle *ps, uint32_t size, int rank, int nranks, NbAllGatherFn allgather, void *ctx);

/*
 * Extension: the sharded World over the direct exchange (CreateSimPipelineShardedDirect, include/nbody_hip.h): each rank
 * pushes its slice device-to-device into every peer's IPC-mapped source array, `control` carries only handles and the
 * per-step barrier.  DestroyWorld is a collective for such Worlds.
 */
World *CreateWorldShardedDirect(const Particle *ps, uint32_t size, int rank, int nranks, NbAllGatherFn control, void *ctx);

/*
 * Extension: the HIP pipeline behind a World (include/nbody_hip.h), for tooling that wants its knobs and timers
 * (nb_hip_configure, nb_hip_last_step_ms, nb_hip_comm_info ...).  Owned by the World; never destroy it.
 */
typedef struct SimPipeline SimPipeline;
SimPipeline *GetWorldPipeline(World *w);

#ifdef __cplusplus
}
#endif

#endif /* NBODY_AMD_NBODY_H */
