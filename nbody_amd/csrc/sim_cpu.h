/*
 * sim_cpu.h -- host-core implementation behind UpdateWorld_CPU.
 *
 * Same role as the reference's src/lib/sim_cpu.h:15-24 (AllocPackArray /
 * PackParticles / PackedUpdate), different shape: one object owns the source
 * snapshot, one call advances the whole world by one step.
 */
#ifndef NB_SIM_CPU_H
#define NB_SIM_CPU_H

#include <stdint.h>
#include "nbody.h"

typedef struct CpuSim CpuSim;

/* Snapshot storage for up to mass_len sources (never NULL; mass_len may be 0). */
CpuSim *CpuSimCreate(uint32_t mass_len);
void CpuSimDestroy(CpuSim *sim);

/*
 * One Jacobi step over arr[0..total_len): sources are arr[0..mass_len) as they
 * were on entry.  Result bits equal the reference's AVX build
 * (src/lib/sim_cpu.c:156-194 driven by src/lib/world.c:101-107).
 */
void CpuSimStep(CpuSim *sim, Particle *arr, uint32_t total_len, uint32_t mass_len, float dt);

#endif /* NB_SIM_CPU_H */
