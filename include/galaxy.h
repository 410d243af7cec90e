/*
 * galaxy.h -- synthetic spiral-galaxy initial conditions.
 *
 * Keeps every tunable the reference exposes in include/galaxy.h:6-61 under the
 * same macro name and value, and the same MakeGalaxies entry point
 * (reference galaxy.h:64), because every benchmark configuration in
 * BASELINE.json is "synthetic galaxy.h initial conditions".
 *
 * Geometry of one galaxy with S particles and core radius Rc:
 *     min_dist = Rc * MIN_PARTICLE_DIST_CR_F
 *     max_dist = Rc * MAX_PARTICLE_DIST_CR_F + sqrt(S) * MAX_PARTICLE_DIST_PC_F
 * Particles sit on MIN_SPIRALS..MAX_SPIRALS arms r(t) = b*t between those two
 * distances.  Galaxy k > 0 is placed MIN..MAX_GALAXY_SEPARATION times the sum
 * of both max_dist away from a random earlier galaxy, rejecting overlaps.
 */
#ifndef NBODY_AMD_GALAXY_H
#define NBODY_AMD_GALAXY_H

#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef PI
#define PI 3.1415927f
#endif

/* spiral arms per galaxy */
#define MIN_SPIRALS 2
#define MAX_SPIRALS 4

/* galaxy cores: radius range and density */
#define GC_MIN_R   200.f
#define GC_MAX_R   600.f
#define GC_DENSITY 30.0f

/* ordinary (massive) particles: radius range and density */
#define NP_MIN_R   1.5f
#define NP_MAX_R   9.5f
#define NP_DENSITY 10.f

/* mass of a sphere of radius R (R is expanded three times) */
#define R_TO_M(R, DENSITY) ((4.f * PI * DENSITY / 3.f) * (R) * (R) * (R))
#define GC_R_TO_M(R)       R_TO_M(R, GC_DENSITY)
#define NP_R_TO_M(R)       R_TO_M(R, NP_DENSITY)
#define MIN_GC_MASS        GC_R_TO_M(GC_MIN_R)

/* every galaxy gets at least this many particles, core included */
#define MIN_PARTICLES_PER_GALAXY 100

/* particle distance limits, see the header comment */
#define MIN_PARTICLE_DIST_CR_F 5.f
#define MAX_PARTICLE_DIST_CR_F 10.f
#define MAX_PARTICLE_DIST_PC_F 300.f

/* core-to-core separation limits, in units of (max_dist_a + max_dist_b) */
#define MIN_GALAXY_SEPARATION 1.4f
#define MAX_GALAXY_SEPARATION 2.0f

/*
 * malloc()s and fills particle_count particles forming galaxy_count galaxies;
 * the caller free()s.  Draws from libc rand(), so srand() selects the
 * universe.  Aborts when particle_count < galaxy_count * MIN_PARTICLES_PER_GALAXY.
 */
Particle *MakeGalaxies(uint32_t particle_count, uint32_t galaxy_count);

/*
 * Extension (not in the reference): the same generator drawing from a built-in
 * xoshiro256** stream seeded with `seed` instead of libc rand(), so that one
 * (particle_count, galaxy_count, seed) names the same draws on any libc.  Does
 * not touch the rand() state.  (The particle formulas still call libm's sqrtf,
 * cosf, sinf and hypotf; a libm that rounds those differently moves a particle
 * by an ulp, not the universe.)
 */
Particle *MakeGalaxiesSeeded(uint32_t particle_count, uint32_t galaxy_count, uint64_t seed);

#ifdef __cplusplus
}
#endif

#endif /* NBODY_AMD_GALAXY_H */
