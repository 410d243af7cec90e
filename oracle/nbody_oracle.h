/*
 * nbody_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's pairwise-gravity + integrator step, used
 * as the parity checker for the HIP path.  Nothing under nbody_amd/ may
 * include, link or call this; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks these functions against
 *   (1) the seven partition known-answers of reference test/test_particle_sort.c:27-111,
 *   (2) tests/golden/ fixtures produced by the reference's own compiled
 *       src/lib/sim_cpu.c (AVX build) through tests/golden/make_golden.py, and
 *   (3) the sha256 digests SURVEY.md section 8c records for the reference's
 *       UpdateWorld_CPU on srand(11037) MakeGalaxies(4096, 2).
 */
#ifndef NBODY_ORACLE_H
#define NBODY_ORACLE_H

#include <stdint.h>
#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/lib/world.c:32-46 -- in-place "mass > 0 first" partition; returns mass_len */
uint32_t orc_partition(Particle *arr, uint32_t size);

/* same algorithm on ints, the shape reference test/test_particle_sort.c:10-25 pins */
uint32_t orc_partition_ints(int *arr, uint32_t size);

/*
 * n steps, bit-exact with the reference's AVX build (SIMD_SET=AVX, -mavx, no FMA):
 * reference src/lib/sim_cpu.c:125-194 driven as src/lib/world.c:99-110 does.
 * Plain C: eight virtual lanes, lane k accumulating sources j = 7-k (mod 8)
 * (sim_cpu.c:32-33), lanes summed 0..7 (sim_cpu.c:146-154).
 */
void orc_step_avx_order(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n);

/*
 * The reference's other SIMD_SET builds (src/lib/CMakeLists.txt:24-33): lanes = 4 is the SSE build
 * (sim_cpu.c:46-68), lanes = 1 the scalar one (sim_cpu.c:70-91), lanes = 8 equals orc_step_avx_order.
 */
void orc_step_lanes(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n, uint32_t lanes);

/* the same arithmetic written with AVX intrinsics + OpenMP: the timed "port" CPU baseline */
void orc_step_avx(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n);

/*
 * Time-only variant for bench.py: one step's force loop for receivers
 * [recv_begin, recv_end) against all mass_len sources, results discarded into
 * a checksum.  Returns seconds.  threads = 0 means omp default.
 */
double orc_time_avx_sample(const Particle *arr, uint32_t mass_len, uint32_t recv_begin,
                           uint32_t recv_end, float dt, int threads, int *threads_used,
                           double *checksum);

/*
 * n steps with sources summed one by one in index order, fp32: the reference's
 * scalar build (sim_cpu.c:70-91) and the order its GLSL kernel uses
 * (src/shader/particle_cs.glsl:35-49).
 */
void orc_step_seq(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n);

/*
 * One step's accelerations from the CURRENT state in float64 ("truth" for the
 * tolerance): acc_xy[2*i], acc_xy[2*i+1]; abs_xy gets sum_j |contribution_j|
 * per component (the scale an fp32 summation error is measured against).
 * Does not modify arr.
 */
void orc_acc_f64(const Particle *arr, uint32_t total_len, uint32_t mass_len,
                 double *acc_xy, double *abs_xy);

/* the same for the receivers idx[0..n_idx) only: acc_xy/abs_xy have 2*n_idx entries (full-size spot checks) */
void orc_acc_f64_subset(const Particle *arr, uint32_t mass_len, const uint32_t *idx, uint32_t n_idx,
                        double *acc_xy, double *abs_xy);

/* the reference-AVX-order fp32 accelerations (bit-exact with orc_step_avx) of the receivers idx only */
void orc_acc_avx_subset(const Particle *arr, uint32_t mass_len, const uint32_t *idx, uint32_t n_idx, float *acc_xy);

/* n steps integrated entirely in float64 state, written back rounded; trend checks only */
void orc_step_f64(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
