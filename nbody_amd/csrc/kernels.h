// kernels.h -- internal (C++) launch interface between the pipeline and the gfx950 kernels.
// Not part of the C-ABI; include/nbody_hip.h is.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nb {

enum : uint32_t {
    STEP_ACC_IN = 1u,       // start each receiver's sum from acc[] instead of zero
    STEP_NO_FINALIZE = 2u,  // store the sums into acc[] and skip the integrator
};
// (experiment, knob "fused_finish") split steps without the finish kernel: the last-arriving workgroup of a receiver tile
// adds the parts itself -- parts as agent-scope (sc1) stores / loads, one ticket per tile; see step_kernel<..., FUSED>

enum : int {
    VARIANT_LDS = 0,   // wave-private LDS tiles (coalesced float2 loads -> ds_write -> broadcast ds_read_b128)
    VARIANT_SMEM = 1,  // wave-uniform source loads through the scalar cache (s_load_dwordx8/16)
};

// Everything one force(+integrate) launch needs.  Passed by value as the single kernel argument,
// so a hipGraph kernel node is re-pointed by rewriting one struct.
struct StepParams {
    // sources = snapshot of the previous step (Jacobi): positions and premultiplied G*m
    const float2 *src_pos;
    const float *src_gm;
    uint32_t src_begin[2];  // two half-open source ranges, walked back to back;
    uint32_t src_end[2];    // the second is empty except in the overlapped sharded step
    // receivers owned by this launch
    const float2 *pos_in;
    float2 *pos_out;
    float2 *vel;
    float2 *acc;
    const float *radius;
    uint32_t n_recv;      // receivers this launch computes (logical indices 0 .. n_recv)
    // logical receiver i lives in slot i + (i >= recv_split ? recv_gap : 0): a shard's massive slice is padded
    // to the uniform all-gather chunk, and the pad slots must not cost workgroups (one extra workgroup on a
    // 2-round grid is +50 %, profiles/r01_shard_overhead_before_fix.txt).  Unsharded: recv_split = n_recv.
    uint32_t recv_split;
    uint32_t recv_gap;
    // new positions of the receivers in slots [0, n_mirror) are also written here (the shard's slice of
    // the next gathered source array); n_mirror == 0 disables it
    float2 *mirror;
    uint32_t n_mirror;
    // step size, read from device memory -- where the reference keeps it too (the uniform block, sim_gpu.h:8-12,
    // re-uploaded when dt changes, sim_gpu.c:268-284).  A hipGraph chain therefore never needs re-patching for a new dt.
    const float *dt;
    uint32_t flags;
    // source split: gridDim.y = split workgroups share one receiver tile, each over 1/split of the source
    // chunks; with split > 1 the step kernel only stores its sums to parts[part][receiver] and finish_kernel
    // adds the parts in order and integrates.  Fills the chip when there are few receiver tiles and lands the
    // workgroup count near a round boundary (see choose_shape).
    float2 *parts;
    uint32_t split;
    uint32_t *tickets;    // fused finish only: one arrival counter per receiver tile, zero between launches
    // granule of the source slicing: a wave's slice is a whole number of `unit` sources (64, or 32 / 16 / 8 for
    // latency-bound launches whose parts hold fewer 64-source chunks than the workgroup has waves -- with 64 only, a
    // part of 6 chunks keeps 6 of 16 waves busy).  Slices stay 8-aligned, so the scalar loads keep their alignment.
    // Scalar-cache route only: the LDS route stages whole 64-source tiles and always runs with 64 -- the two routes
    // give the same bits whenever the granule is 64.  Two-range (overlapped) steps always use 64.
    uint32_t unit;
};

// A whole n-step chain of a world that fits ONE workgroup, run inside one launch (chain_kernel): positions ping-pong in
// LDS, velocities and radii stay in registers, two workgroup barriers per step and no kernel boundary (1.6-1.8 us each
// on this chip, more than such a step's arithmetic).  Sixteen waves: `tiles` receiver tiles of 64 * K receivers, 16 / tiles
// waves per tile, each over a 1/W slice of the sources in 8-source granules -- the summation order of the per-step
// kernel launched with k = K, w = 16 / tiles, split = 1, unit = 8, so the two paths give the same bits.
struct ChainParams {
    float2 *pos;          // in: state before the chain; out: state after it (updated in place)
    float2 *vel;          // in / out
    float2 *acc;          // out: the last step's sums (Particle.acc is observable)
    const float *radius;
    const float *src_gm;  // G*m of the sources = the first n_src receivers (massive-first order)
    uint32_t n_recv;      // receivers, <= 128 * tiles
    uint32_t n_src;       // sources, <= n_recv
    uint32_t steps;       // steps to run, >= 1
    uint32_t tiles;       // 1, 2 or 4
    const float *dt;      // step size in device memory, as for the per-step kernels
};

constexpr uint32_t CHAIN_MAX_RECV = 512;   // 4 tiles of 128 receivers (K = 2)

struct LaunchShape {
    int k;        // receivers per lane: 1, 2 (4 in tuning builds)
    int w;        // waves per workgroup = source slices: 1, 4, 8, 16 (2 in tuning builds)
    int variant;  // VARIANT_*
    int split;    // workgroups per receiver tile (source parts): 1 .. MAX_SPLIT
    int unit;     // sources per slice granule: 64 (default), 32, 16, 8
    // lane groups per wave (0 / 1: a wave's 64 lanes are 64 * k receivers).  2 or 4: the lanes of a wave split into that
    // many groups over the SAME 64 / lanes receivers, each group walking its own slice of the sources (lane_split_kernel):
    // w * lanes source slices per receiver inside ONE workgroup -- the parallelism a source split buys, without its
    // second kernel.  Latency-bound launches only (k = 1, split = 1, sources staged once in LDS).
    int lanes;
    // experiment ("persist" hook): > 1 = the launch has 1/persist as many workgroups as (tile, part) work items and every
    // workgroup walks `persist` of them (scalar-cache route, classic kernel only); 0 / 1 = one workgroup per item
    int persist;
};

constexpr uint32_t LANE_SPLIT_MAX_SRC = 1u << 18;   // sources a lane-split launch walks (one launch = one source pass)

constexpr int MAX_SPLIT = 16;

// The auto rule for lane-split shapes: lanes (1 = use the classic kernel) and waves per workgroup.
int lane_split_rule(uint32_t n_recv, uint32_t n_src, int *w);

// Resolve "auto" (0) entries of `want` for a launch over n_recv receivers and n_src sources.
LaunchShape choose_shape(LaunchShape want, uint32_t n_recv, uint32_t n_src, int compute_units);

// Kernel entry point and grid for a shape; used both for direct launches and for graph nodes.
const void *step_kernel_fn(LaunchShape s);
const void *step_kernel_fused_fn(LaunchShape s);   // nullptr when the shape has no fused-finish instantiation
dim3 step_grid(LaunchShape s, uint32_t n_recv);
dim3 step_block(LaunchShape s);
size_t step_lds_bytes(LaunchShape s, uint32_t n_src);   // dynamic LDS of the launch (0 except for lane-split shapes)
// second kernel of a split step (split > 1): adds the parts and finishes like the step kernel's epilogue
const void *finish_kernel_fn();
dim3 finish_grid(uint32_t n_recv);
dim3 finish_block();

// the one-workgroup chain: 1024 threads, one block; tiles from chain_tiles(n_recv) (0: the world does not fit)
uint32_t chain_tiles(uint32_t n_recv);
void launch_chain(hipStream_t st, const ChainParams &p);

// AoS <-> SoA converters (reference Particle layout, include/nbody.h).
// split: aos[first .. first+count) -> soa slots [slot0 .. slot0+count)
void launch_split(hipStream_t st, const void *aos, uint32_t first, uint32_t count, float2 *pos, float2 *vel, float2 *acc,
                  float *radius, float *mass, uint32_t slot0);
// fill pad slots so that they are inert sources / harmless receivers
void launch_fill_pad(hipStream_t st, float2 *pos, float2 *vel, float2 *acc, float *radius, float *mass, uint32_t slot0,
                     uint32_t count);
// gm[j] = g * mass[j] for j < count (0 where mass <= 0); g = the host's NB_G, the only place the value is written down
void launch_make_gm(hipStream_t st, const float *mass, float *gm, uint32_t count, float g);
// merge: soa slots [slot0 .. slot0+count) -> aos[first .. first+count)
void launch_merge(hipStream_t st, void *aos, uint32_t first, uint32_t count, const float2 *pos, const float2 *vel,
                  const float2 *acc, const float *radius, const float *mass, uint32_t slot0);
// *dst = value, in stream order (the step-size upload)
void launch_set_scalar(hipStream_t st, float *dst, float value);
// sharded upload: both gathered source arrays + G*m from the AoS world; slots in [mass_len, n_src) become inert pads
void launch_split_sources(hipStream_t st, const void *aos, uint32_t mass_len, uint32_t n_src, float2 *pos0, float2 *pos1,
                          float *gm, float g);

}  // namespace nb
