/*
 * nbody_hip_tuning.h -- test and tooling hooks of libnbody_hip.so.  NOT part of the C-ABI (include/nbody_hip.h is):
 * nothing here is stable, nothing here is needed to use the library, and every knob below is on "auto" in every
 * published number.  They exist so that the GPU suite can pin a launch shape (bit-equality tests across shardings need
 * one wave per workgroup), so that tools/ can scan shapes, and so that nbody-bench can price its "floor" column.
 *
 * History of why each one is not public (the measurements are under profiles/):
 *   k, w, split, unit, passes   launch shape; "auto" is within 1.2 % of the best explicit neighbour at every size the
 *                               GPU suite asks the hardware about (test_auto_launch_shape_is_near_the_best_...)
 *   lanes, fused_chain          small-world kernels; auto-selected below N ~ 4 000 / N <= 256, frozen since round 3
 *   fused_finish                the last workgroup of a receiver tile finishes it; auto from N x M >= 4e7, -0.4 ... -2.3 us
 *   readback, zero_copy_upload  the GUI frame loop's eager read-back / zero-copy upload heuristics, frozen since round 2
 */
#ifndef NBODY_AMD_NBODY_HIP_TUNING_H
#define NBODY_AMD_NBODY_HIP_TUNING_H

#include "nbody_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * key is one of (0 = auto unless stated):
 *   "k"         receivers per lane: 1 or 2 (4 only in TUNING=1 builds)
 *   "w"         waves (source slices) per workgroup: 1, 4, 8 or 16 (2 only in TUNING=1 builds); w = 1 makes the summation
 *               order independent of the launch geometry
 *   "split"     workgroups per receiver tile, each over 1/split of the sources (a finish kernel adds the parts): 1..16
 *   "unit"      granule of the source slicing: 8, 16, 32 or 64 sources (the LDS route always uses 64)
 *   "passes"    launches per step over consecutive source sub-ranges, chained through acc[]: 1..64 (auto: each pass's
 *               sources fit one XCD's L2)
 *   "lanes"     lane groups per wave over the same 64 / lanes receivers (lane_split_kernel): 2, 4 or 8; 1 = never
 *   "fused_chain"   one-workgroup worlds run a whole n-step call inside one launch: 2 (default) = auto (N <= 256 and
 *               N x M <= 3.6e4, calls of 2+ steps), 1 = whenever the world fits (N <= 512), 0 = never
 *   "fused_finish"  split shapes without their second kernel (the tile's last workgroup adds the parts and integrates):
 *               2 (default) = auto (unsharded, scalar-cache route, N x M >= 4e7, <= 200 000 receivers), 1 = whenever the
 *               shape has a split, 0 = never
 *   "readback"  when the device state reaches the array named by nb_hip_note_host_array: 0 = only when GetSimulationData
 *               asks, 1 = at the end of every blocking PerformSimUpdate, 2 (default) = auto (eager once two updates in a
 *               row were each followed by a Get into the noted array)
 *   "zero_copy_upload"  1 (default) = SetSimulationData from the noted, page-locked array lets the split kernel read the
 *               records over PCIe itself; 0 = DMA copy into device staging, then the kernel
 *   "persist"   TUNING=1 builds only: work items (receiver tile, source part) per workgroup of a persistent launch
 * Returns the previous value; aborts on an unknown key or value.
 */
int nb_hip_tune(SimPipeline *sim, const char *key, int value);

/* 1 when the library was built with make TUNING=1: the launch shapes the cost model never picks (K = 4, W = 2), the
 * persistent-launch experiment kernels ("persist" hook: items per workgroup, 0 / 1 = classic; closed in round 5, slower at
 * every size) and the NB_HIP_* environment presets of the hooks above exist in such builds only. */
int nb_hip_tuning_build(void);

/* Steps of the last update that ran inside one-workgroup chain launches ("fused_chain"). */
uint32_t nb_hip_last_fused_steps(const SimPipeline *sim);

/* Lane groups per wave (1, 2, 4 or 8) and source-slice granule of the last step launch. */
int nb_hip_launch_lanes(const SimPipeline *sim);
int nb_hip_launch_unit(const SimPipeline *sim);

/* What "auto" picks for an unsharded step of that size, pure host code: lane groups per wave (1 = the classic shape
 * nb_hip_plan_launch describes; through w the waves per workgroup), whether the source split runs without the finish
 * kernel, and the source-slice granule. */
int nb_hip_plan_launch_lanes(uint32_t n_recv, uint32_t n_src, int *w);
int nb_hip_plan_fused_finish(uint32_t n_recv, uint32_t n_src, int compute_units);
int nb_hip_plan_launch_unit(uint32_t n_recv, uint32_t n_src, int compute_units);

#ifdef __cplusplus
}
#endif

#endif /* NBODY_AMD_NBODY_HIP_TUNING_H */
