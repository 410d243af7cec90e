// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the direct N-body step.
//
// Replaces the reference's GLSL compute shader (reference src/shader/particle_cs.glsl:28-55), which
// walks all sources per thread straight from global memory as 32-byte AoS records.  Here:
//
//   * one LANE owns K receivers (register blocking), one WAVE owns 64*K receivers and a 1/W slice of the
//     sources, one WORKGROUP = W waves over the same 64*K receivers; the W partial sums meet in LDS in a
//     fixed order (deterministic results), then the workgroup integrates and stores; a launch may also cut the
//     sources into `split` parts (gridDim.y) whose sums a small finish kernel adds, and slices are whole granules
//     of 64 sources or, for latency-bound launches on the scalar-cache route, of 32 / 16 / 8 (StepParams::unit);
//   * sources reach the VALU wave-uniformly, 12 bytes each (x, y, G*m), by one of two routes:
//       VARIANT_LDS   each wave stages 64-source tiles in its own LDS slab: coalesced float2/float loads
//                     (one source per lane), ds_write_b64 + b32, then broadcast ds_read_b128 (six per 8
//                     sources); double-buffered, no workgroup barrier in the loop;
//       VARIANT_SMEM  the wave reads its slice through the scalar cache (s_load_dwordx8/x16) so sources
//                     arrive in SGPRs and feed the VALU as scalar operands; no LDS, no VGPR staging;
//   * per interaction: 2 sub (dx, dy), 2 fma (dist^2 + receiver radius), v_rsq_f32, 3 mul, 2 fma into the
//     accumulators = 10 plain VALU instructions, none packed, against the reference's 14 counted flops
//     (SURVEY.md 8d keeps 14 as the roofline convention);
//   * sums are two-level (plain over blocks of 256 sources, then Kahan over the block totals), so the fp32 result
//     stays within ~4e-7 * sum|contribution| (rms) of the float64 sum at any N and launch shape, 180x closer than
//     the reference's own AVX sums at N = 2^20;
//   * the integrator keeps the reference's rounding (mul, then add; sim_cpu.c:191-193 /
//     particle_cs.glsl:51-52), the force loop does not (rsq + fma instead of sqrt, div, mul, add):
//     DESIGN.md states the tolerance.
//
// fp32 throughout.  No MFMA: the loop is rsqrt/fma on independent (receiver, source) pairs, not a contraction.
#include "kernels.h"
#include "interaction_asm.h"

#include <hip/hip_runtime.h>

namespace nb {
namespace {

constexpr int WAVE = 64;
constexpr int CHUNK = 64;  // sources per staged tile = one per lane
constexpr int CLOSE_EVERY = 4;  // tiles per summation block: plain sums over 256 sources, Kahan across blocks

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

typedef float f2v __attribute__((ext_vector_type(2)));

template <int K>
struct Receivers {
    f2v p[K];    // position (x, y): one aligned VGPR pair, the operand of v_pk_add_f32
    float r[K];  // radius (softening term)
    f2v a[K];    // running sums (ax, ay) of the current block of 256 sources: operand of v_pk_fma_f32
    f2v s[K];    // sums of the finished blocks ...
    f2v c[K];    // ... and their Kahan compensation
    // Two-level summation: 256 terms per block in a[], then the block totals are added to s[] with
    // compensated (Kahan) summation, 8 adds per block per receiver (0.4 % of a block's 2048 instructions).
    // A plain running sum of 10^5..10^6 same-signed pulls loses 4-5 digits (the reference's 8-lane AVX sums do:
    // profiles/r01_accuracy_vs_f64_and_avx.txt); even a plain sum of chunk totals drifts with the slice length
    // (profiles/r01_accuracy2_shapes_before_kahan.txt).  This way the error no longer depends on N or on the
    // launch shape.  Built without fast-math and with -ffp-contract=off, so the compensation survives.
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < K; k++) a[k] = s[k] = c[k] = f2v{0.0f, 0.0f};
    }
    __device__ __forceinline__ void close_chunk() {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const f2v y = a[k] - c[k];
            const f2v t = s[k] + y;
            const f2v cn = (t - s[k]) - y;
            // a block that added exactly nothing (zero-mass pad sources of a sharded launch) must leave the
            // state untouched, or padded and unpadded launches would differ in the last bit
            const bool lx = a[k].x != 0.0f, ly = a[k].y != 0.0f;
            c[k].x = lx ? cn.x : c[k].x;
            c[k].y = ly ? cn.y : c[k].y;
            s[k].x = lx ? t.x : s[k].x;
            s[k].y = ly ? t.y : s[k].y;
            a[k] = f2v{0.0f, 0.0f};
        }
    }
};

// One source against the K receivers of this lane.  sxy/sg are wave-uniform (SGPRs).
//
// The ten VALU instructions of an interaction are written out as one asm statement per (source, receiver):
//     v_sub_f32   dx  = sx - x
//     v_sub_f32   dy  = sy - y
//     v_fma_f32   q   = dx * dx + radius           softening: + radius of the RECEIVER, not squared
//     v_fmac_f32  q  += dy * dy
//     v_rsq_f32   q   = 1 / sqrt(q)                1 ulp; issued at raised wave priority
//     v_mul_f32   u   = (G*m) * q                  G*m straight from its SGPR
//     v_mul_f32   t   = q * q
//     v_mul_f32   u   = u * t                      G*m / dist^3
//     v_fmac_f32  ax += dx * u
//     v_fmac_f32  ay += dy * u
// = 9 plain instructions (2 issue cycles each) + one quarter-rate transcendental (8) = 26 cycles per wave-interaction,
// the floor of this mix; the kernel runs at 27.2 (profiles/r02_ab_plain_body.txt).
// No packed instruction: round 1 shipped v_pk_add_f32 for (dx, dy) and v_pk_fma_f32 for the accumulation -- two fewer
// instructions, the same nominal cycles -- and measured 28.4-28.9 cycles; the all-plain body is 4.5 % faster on every
// box tried (51.2 -> 48.8 ms per launch at N = 2^20), while unpacking only one of the two is slower than either
// (+1 % and +12 %): packed f32 costs more than its two halves when it sits between plain and transcendental
// instructions (MI355X_MICROARCH.md notes the same beside MFMAs).
// Why asm: (1) left to hipcc, the multiplies become two v_pk_mul_f32 plus a v_mov, and both the schedule and
// the register count of the unrolled loop swing with unrelated edits (55..75 VGPRs; 65 halves the occupancy of a
// 1024-thread workgroup): the same loop measured anywhere from 105 to 151 ms per step at N = 2^20
// (profiles/r01_sweep_auto_split.txt, r01_sweep8_full_asm_nops_alignment.txt).  One fixed sequence on five fixed
// temporaries cannot drift.  (2) A v_rsq_f32 that lands between other waves' plain VALU
// instructions costs ~16 cycles instead of 8 on gfx950; raising the wave priority for just that instruction buys
// 6 % with the packed body and 13 % with this one (profiles/r01_ubench5_setprio_rsq.txt,
// r01_sweep9_nops_and_static_priority.txt, r02_ab_plain_body.txt; static per-wave priorities instead are 3x SLOWER;
// priority 1 instead of 3, or the window opened one instruction earlier: within 0.3 %).  (3) gfx950 needs one wait
// state between a transcendental and the VALU instruction that reads its result, and hipcc cannot pad inside asm:
// the s_setprio 0 that follows the rsq IS that wait state (an earlier version with nothing in between read stale
// values; the parity tests caught it, and tests/test_isa.py now checks the ISA).  The other dependent pairs are
// ordinary VALU read-after-write, which the hardware interlocks.  (4) One statement per interaction: hipcc pads each
// asm boundary with an s_nop (56 + 4 bytes).  Each interaction is a serial dependency chain on purpose: with 8 waves per
// SIMD the other waves fill the gaps, and interleaved or software-pipelined orders measured slower
// (profiles/r01_ubench3_hand_scheduled_bodies.txt).
// The statement is pure (no memory, not volatile); 12 instructions, 56 bytes, all in their short encodings.
// Temporaries (clobbered): dx = v30, dy = v31, t = v32, q = v33, u = v36 (v37 stays on the list: generated tuning
// bodies use it).
// (the statement text itself: interaction_asm.h, shared with the clock probe)

// SRC_IN_SGPR: the source sits in SGPRs (scalar-cache route) or in VGPRs holding a wave-uniform value
// (LDS broadcast reads); the instructions are the same, only the operand class differs.
template <int K, bool SRC_IN_SGPR>
__device__ __forceinline__ void interact(Receivers<K> &R, f2v sxy, float sg) {
    if constexpr (K == 2) {
        if constexpr (SRC_IN_SGPR) {
            asm(NB_INTERACTION2_ASM
                : [ax0] "+v"(R.a[0].x), [ay0] "+v"(R.a[0].y), [ax1] "+v"(R.a[1].x), [ay1] "+v"(R.a[1].y)
                : [sx] "s"(sxy.x), [sy] "s"(sxy.y), [g] "s"(sg), [px0] "v"(R.p[0].x), [py0] "v"(R.p[0].y), [r0] "v"(R.r[0]),
                  [px1] "v"(R.p[1].x), [py1] "v"(R.p[1].y), [r1] "v"(R.r[1])
                : NB_CLOBBERS2);
        } else {
            asm(NB_INTERACTION2_ASM
                : [ax0] "+v"(R.a[0].x), [ay0] "+v"(R.a[0].y), [ax1] "+v"(R.a[1].x), [ay1] "+v"(R.a[1].y)
                : [sx] "v"(sxy.x), [sy] "v"(sxy.y), [g] "v"(sg), [px0] "v"(R.p[0].x), [py0] "v"(R.p[0].y), [r0] "v"(R.r[0]),
                  [px1] "v"(R.p[1].x), [py1] "v"(R.p[1].y), [r1] "v"(R.r[1])
                : NB_CLOBBERS2);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if constexpr (SRC_IN_SGPR) {
            asm(NB_INTERACTION_ASM
                : [ax] "+v"(R.a[k].x), [ay] "+v"(R.a[k].y)
                : [sx] "s"(sxy.x), [sy] "s"(sxy.y), [g] "s"(sg), [px] "v"(R.p[k].x), [py] "v"(R.p[k].y), [r] "v"(R.r[k])
                : NB_CLOBBERS);
        } else {
            asm(NB_INTERACTION_ASM
                : [ax] "+v"(R.a[k].x), [ay] "+v"(R.a[k].y)
                : [sx] "v"(sxy.x), [sy] "v"(sxy.y), [g] "v"(sg), [px] "v"(R.p[k].x), [py] "v"(R.p[k].y), [r] "v"(R.r[k])
                : NB_CLOBBERS);
        }
    }
}

// 8 sources (x,y interleaved in P, G*m in G) against the K receivers: 8*K interaction statements, source-major.
// (Schedule experiments replace this one function from outside the product tree: tools/exp_body_hook.h, force-included
// by tools/build_variants.sh, defines NB_INTERACT8_OVERRIDE and supplies its own.)
#ifndef NB_INTERACT8_OVERRIDE
template <int K, bool SRC_IN_SGPR, typename VP, typename VG>
__device__ __forceinline__ void interact8(Receivers<K> &R, const VP &P, const VG &G) {
#pragma unroll
    for (int u = 0; u < 8; u++) interact<K, SRC_IN_SGPR>(R, f2v{P[2 * u], P[2 * u + 1]}, G[u]);
}
#else
NB_INTERACT8_OVERRIDE
#endif

// Source arrays as the scalar-cache route reads them.  A wave-uniform load becomes a scalar load (s_load_dwordx8/x16) only
// while the compiler can prove that nothing in the kernel has written the memory before it: true by construction in the
// classic launch (all stores sit in the epilogue), not in a persistent one, where the epilogue of work item i precedes the
// loads of item i + 1.  The sources ARE read-only for the whole launch (a step reads src_pos[in] and writes pos[in ^ 1] /
// src_pos[in ^ 1]), which is what the constant address space says: persistent launches read them through it.
typedef const float __attribute__((address_space(4))) *ConstF;
template <typename V>
__device__ __forceinline__ V src_load(const float *ptr) {
    return *reinterpret_cast<const V *>(ptr);
}
template <typename V>
__device__ __forceinline__ V src_load(ConstF ptr) {
    return *(const V __attribute__((address_space(4))) *)ptr;
}
template <bool READ_ONLY_AS>
struct SrcPtr {
    typedef const float *type;
    static __device__ __forceinline__ type of(const void *ptr) { return static_cast<const float *>(ptr); }
};
template <>
struct SrcPtr<true> {
    typedef ConstF type;
    static __device__ __forceinline__ type of(const void *ptr) { return (ConstF)(uintptr_t)ptr; }
};

// Slot of logical receiver i (see StepParams::recv_split).
__device__ __forceinline__ uint32_t receiver_slot(const StepParams &p, uint32_t i) {
    return i + (i >= p.recv_split ? p.recv_gap : 0u);
}

// Map a position v of the concatenated source ranges to an index of src_pos/src_gm.
__device__ __forceinline__ uint32_t source_index(const StepParams &p, uint32_t v, uint32_t n0) {
    return v < n0 ? p.src_begin[0] + v : p.src_begin[1] + (v - n0);
}

// Epilogue of one receiver: optional carried-in sum, store acc, then the reference's integrator.
__device__ __forceinline__ void finish_receiver(const StepParams &p, uint32_t logical, float sx, float sy, float dt) {
    if (logical >= p.n_recv) return;
    const uint32_t i = receiver_slot(p, logical);
    float2 a = make_float2(sx, sy);
    if (p.flags & STEP_ACC_IN) {
        const float2 a0 = p.acc[i];
        a.x = __fadd_rn(a0.x, a.x);
        a.y = __fadd_rn(a0.y, a.y);
    }
    p.acc[i] = a;
    if (p.flags & STEP_NO_FINALIZE) return;
    // semi-implicit Euler with the reference's roundings: vel += acc*dt; pos += vel*dt
    float2 v = p.vel[i];
    v.x = __fadd_rn(v.x, __fmul_rn(a.x, dt));
    v.y = __fadd_rn(v.y, __fmul_rn(a.y, dt));
    float2 q = p.pos_in[i];
    q.x = __fadd_rn(q.x, __fmul_rn(v.x, dt));
    q.y = __fadd_rn(q.y, __fmul_rn(v.y, dt));
    p.vel[i] = v;
    p.pos_out[i] = q;
    if (i < p.n_mirror) p.mirror[i] = q;
}

// Second kernel of a split step: one thread per receiver adds the parts in part order and finishes.
// The kernel is pure latency -- a handful of loads, a handful of adds, three stores -- and what it reads was just
// written by other CUs (the parts) or other XCDs (acc, vel, pos), so every load is a 500-900 cycle trip to the
// Infinity Cache or HBM.  Issued one after the other behind a loop with a run-time trip count they cost `split` + 2 such
// trips in a row (measured: 4.0 us per launch at N = 10 000 / 5 parts, 5.0 us at 50 000 / 7 parts, of a 21 us step:
// profiles/r02_mid_n_pmc.txt); issued all at once, before the first use, they cost one.  Unused slots re-read the last
// part (always a valid address) and are dropped by a select, so the loads need no branch.
__global__ __launch_bounds__(256) void finish_kernel(const StepParams p) {
    const uint32_t logical = blockIdx.x * blockDim.x + threadIdx.x;
    if (logical >= p.n_recv) return;
    const uint32_t i = receiver_slot(p, logical);
    float2 part[MAX_SPLIT];
#pragma unroll
    for (uint32_t s = 0; s < (uint32_t)MAX_SPLIT; s++)
        part[s] = p.parts[(size_t)(s < p.split ? s : p.split - 1) * p.n_recv + logical];
    const bool carry = (p.flags & STEP_ACC_IN) != 0, integrate = (p.flags & STEP_NO_FINALIZE) == 0;
    // clamped-to-valid addresses again: acc / vel / pos_in exist for every slot whatever the flags say
    const float2 a0 = p.acc[i];
    const float2 v0 = p.vel[i];
    const float2 q0 = p.pos_in[i];
    const float dt = *p.dt;
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (uint32_t s = 0; s < (uint32_t)MAX_SPLIT; s++) {
        // same order and roundings as a sequential loop over the parts: 0 + p0 + p1 + ...
        const float nx = __fadd_rn(sx, part[s].x), ny = __fadd_rn(sy, part[s].y);
        sx = s < p.split ? nx : sx;
        sy = s < p.split ? ny : sy;
    }
    float2 a = make_float2(sx, sy);
    if (carry) {
        a.x = __fadd_rn(a0.x, a.x);
        a.y = __fadd_rn(a0.y, a.y);
    }
    p.acc[i] = a;
    if (!integrate) return;
    // semi-implicit Euler with the reference's roundings (finish_receiver): vel += acc*dt; pos += vel*dt
    float2 v = v0, q = q0;
    v.x = __fadd_rn(v.x, __fmul_rn(a.x, dt));
    v.y = __fadd_rn(v.y, __fmul_rn(a.y, dt));
    q.x = __fadd_rn(q.x, __fmul_rn(v.x, dt));
    q.y = __fadd_rn(q.y, __fmul_rn(v.y, dt));
    p.vel[i] = v;
    p.pos_out[i] = q;
    if (i < p.n_mirror) p.mirror[i] = q;
}

// Everything must stay within 64 VGPRs: a 1024-thread workgroup puts 4 waves on every SIMD, so 65 VGPRs (7 waves
// per SIMD) would mean ONE resident workgroup per CU instead of two.  The asm body needs 36 (SMEM, K <= 2) / 62 (LDS);
// the second launch-bound argument (waves per SIMD) makes the limit explicit.  K = 4 then keeps its Kahan state in
// scratch, touched only at block closes outside the inner loop, and runs as fast as K = 2.
//
// PERSIST (experiment, tuning hook "persist"): the launch has FEWER workgroups than (receiver tile, source part) work items
// and every workgroup walks items blockIdx.x, blockIdx.x + gridDim.x, ... -- fewer, longer-lived waves, so that dispatch
// ramp and end-of-kernel write-back are paid by fewer workgroups.  Same bits as the classic launch: an
// item is computed by exactly the code a classic workgroup (tile, part) runs.
template <int K, int W, int VARIANT, bool FUSED = false, bool PERSIST = false>
__global__ __launch_bounds__(WAVE *W, 8) void step_kernel(const StepParams p) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1);
    // wave id as an SGPR value so that everything derived from it stays scalar
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the work item of this workgroup: receiver tile and source part (classic launch: the grid coordinates)
    uint32_t tile_x = blockIdx.x, part_y = blockIdx.y;
    [[maybe_unused]] uint32_t item = blockIdx.x;
    [[maybe_unused]] const uint32_t n_tiles = (p.n_recv + WAVE * K - 1) / (WAVE * K);
    if constexpr (PERSIST) {
        tile_x = item % n_tiles;
        part_y = item / n_tiles;
    }

    // per wave, double-buffered: 64 interleaved (x, y) pairs, then 64 G*m
    __shared__ __attribute__((aligned(16))) float tile[VARIANT == VARIANT_LDS ? W : 1][2][3 * CHUNK];
    __shared__ float2 partial[W > 1 ? W : 1][W > 1 ? WAVE * K : 1];

    // the step size: a scalar load issued first, consumed by the epilogue.  (Fetching the integrating thread's velocity
    // and position here as well, instead of after the force loop, measured 0.1 us per step SLOWER at N = 250 ... 1 000:
    // profiles/r02_ab_early_fetch.txt.)
    const float dt = *p.dt;

#pragma clang loop unroll(disable)
    for (;;) {
        const uint32_t recv_base = tile_x * (WAVE * K);
        Receivers<K> R;
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint32_t i = recv_base + k * WAVE + lane;
            i = i < p.n_recv ? i : p.n_recv - 1;  // tail lanes redo the last receiver; their stores are masked
            i = receiver_slot(p, i);
            const float2 q = p.pos_in[i];
            R.p[k] = f2v{q.x, q.y};
            R.r[k] = p.radius[i];
        }
        R.clear();

        // this wave's slice of the concatenated source ranges, in whole chunks
        const uint32_t n0 = p.src_end[0] - p.src_begin[0];
        const uint32_t n1 = p.src_end[1] - p.src_begin[1];
        const uint32_t total = n0 + n1;
        // this workgroup's part of the sources (all of them unless the step is split), then this wave's slice of it, both
        // in whole granules of p.unit sources (64 = one tile; finer for latency-bound launches, see StepParams::unit)
        const uint32_t unit = p.unit;
        const uint32_t nunits = (total + unit - 1) / unit;
        const uint32_t per_part = (nunits + p.split - 1) / p.split;
        const uint32_t part_lo = min(part_y * per_part, nunits);
        const uint32_t part_hi = min(part_lo + per_part, nunits);
        const uint32_t per_wave = (part_hi - part_lo + W - 1) / W;
        const uint32_t u_lo = min(part_lo + wid * per_wave, part_hi);
        const uint32_t u_hi = min(u_lo + per_wave, part_hi);
        const uint32_t v_lo = u_lo * unit;                // first source of the slice: a multiple of 8
        const uint32_t v_hi = min(u_hi * unit, total);    // one past its last source

        if constexpr (VARIANT == VARIANT_LDS) {
            float(*T)[3 * CHUNK] = tile[wid];
            float2 sp = make_float2(0.f, 0.f);
            float sg = 0.f;
            auto fetch = [&](uint32_t c) {
                const uint32_t v = c * CHUNK + lane;
                const bool live = v < total;
                const uint32_t j = source_index(p, live ? v : total - 1, n0);
                sp = p.src_pos[j];                 // 512 B per wave, coalesced
                sg = live ? p.src_gm[j] : 0.0f;    // pad sources: a real position, zero mass
            };
            // whole 64-source tiles only: this route always runs with the 64-source granule (choose_shape), so the slice
            // is [c_lo, c_hi) tiles and the last one may be ragged (pads: a real position, zero mass)
            const uint32_t c_lo = v_lo / CHUNK, c_hi = (v_hi + CHUNK - 1) / CHUNK;
            if (c_lo < c_hi) fetch(c_lo);
            int buf = 0;
            for (uint32_t c = c_lo; c < c_hi; c++) {
                *reinterpret_cast<float2 *>(&T[buf][2 * lane]) = sp;  // ds_write_b64
                T[buf][2 * CHUNK + lane] = sg;
                if (c + 1 < c_hi) fetch(c + 1);  // next tile's HBM/L2 latency hides under this tile's math
                // LDS executes one wave's accesses in order; this only stops the compiler from reordering
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int jj = 0; jj < CHUNK; jj += 4) {
                    // broadcast ds_read_b128: every lane reads the same 16 bytes.  Four sources per read group (two reads
                    // of positions, one of G*m): 12 staging VGPRs instead of 24, which is what leaves room for the
                    // paired-rsq body with two receivers per lane
                    const v8f P = *reinterpret_cast<const v8f *>(&T[buf][2 * jj]);
                    const v4f G = *reinterpret_cast<const v4f *>(&T[buf][2 * CHUNK + jj]);
#pragma unroll
                    for (int u = 0; u < 4; u++) interact<K, false>(R, f2v{P[2 * u], P[2 * u + 1]}, G[u]);
                }
                if (((c - c_lo) & (CLOSE_EVERY - 1)) == CLOSE_EVERY - 1) R.close_chunk();
                buf ^= 1;
            }
            if ((c_hi - c_lo) & (CLOSE_EVERY - 1)) R.close_chunk();  // a short last block
        } else {
            // scalar-cache route: indices are wave-uniform, the loads become s_load_dwordx8/x16
#pragma unroll
            for (int range = 0; range < 2; range++) {
                // intersection of [v_lo, v_hi) with this range, as indices of the source arrays
                const uint32_t r_lo = range == 0 ? 0u : n0;
                const uint32_t r_hi = range == 0 ? n0 : total;
                const uint32_t a = max(v_lo, r_lo), b = min(v_hi, r_hi);
                if (a >= b) continue;
                uint32_t j = p.src_begin[range] + (a - r_lo);
                const uint32_t j_end = p.src_begin[range] + (b - r_lo);
                // 8 sources per scalar fetch: s_load_dwordx16 (x,y pairs) + s_load_dwordx8 (G*m).  Slices start on
                // multiples of 64 sources from 64-aligned range starts, so j is a multiple of 8 here.
                const typename SrcPtr<PERSIST>::type sp = SrcPtr<PERSIST>::of(p.src_pos), sg = SrcPtr<PERSIST>::of(p.src_gm);
                // every 8 * CLOSE_EVERY groups (256 sources) the block sums are closed, exactly where the LDS variant
                // closes them, so both variants add in the same order; a short last block may end in single sources
                const uint32_t groups = (j_end - j) / 8;
                const uint32_t g0 = (a - v_lo) / 8;  // groups of this slice that lie in the previous range
                if (groups > 0) {
                    // Two register sets, A and B, each fetched while the other one is being consumed (the scalar
                    // cache's latency hides under 8 * K interactions) and each dead before its refill is issued, so
                    // no set is ever copied.  g0 is even (range starts are 64-aligned) and a block ends on an odd
                    // group index, so only the second group of a pair can close one.  The refill address is clamped to
                    // the last group instead of branching around the load.
                    const uint32_t j_last = j + (groups - 1) * 8;
                    v16f PA = src_load<v16f>(sp + 2 * (size_t)j);
                    v8f GA = src_load<v8f>(sg + j);
                    uint32_t g = 0;
                    while (g + 2 <= groups) {
                        // pairs up to the end of the current 32-group block, as one branch-free inner loop
                        const uint32_t to_close = (8u * CLOSE_EVERY - ((g0 + g) & (8u * CLOSE_EVERY - 1))) / 2;
                        const uint32_t pairs = min(to_close, (groups - g) / 2);
                        for (uint32_t i = 0; i < pairs; i++) {
                            // Scalar loads return out of order, so the only wait there is is "all of them"
                            // (lgkmcnt(0)).  The empty asm makes the next fetch's address depend on the set about to
                            // be consumed: the wait lands BEFORE that fetch is issued, where nothing is in flight but
                            // loads that had a whole group's math to land.  (Not volatile: a volatile asm counts as a
                            // memory clobber and would turn the scalar loads into vector loads.)  The scheduling
                            // barriers keep each fetch ahead of the math that hides it.
                            asm("" : "+s"(j) : "s"(PA), "s"(GA));
                            const v16f PB = src_load<v16f>(sp + 2 * (size_t)(j + 8));
                            const v8f GB = src_load<v8f>(sg + j + 8);
                            __builtin_amdgcn_sched_barrier(0);
                            interact8<K, true>(R, PA, GA);
                            __builtin_amdgcn_sched_barrier(0);
                            j = min(j + 16, j_last);
                            asm("" : "+s"(j) : "s"(PB), "s"(GB));
                            PA = src_load<v16f>(sp + 2 * (size_t)j);
                            GA = src_load<v8f>(sg + j);
                            __builtin_amdgcn_sched_barrier(0);
                            interact8<K, true>(R, PB, GB);
                        }
                        g += 2 * pairs;
                        if (pairs == to_close) R.close_chunk();
                    }
                    if (g < groups) interact8<K, true>(R, PA, GA);  // odd count: the last refill fetched it
                    j = j_last + 8;
                }
                for (; j < j_end; j++) interact<K, true>(R, f2v{sp[2 * (size_t)j], sp[2 * (size_t)j + 1]}, sg[j]);
            }
            // a short last block: anything after the last multiple of 256 sources of this slice (same blocks as the
            // LDS variant, whose last tile may be padded)
            if ((v_hi - v_lo) & (CHUNK * CLOSE_EVERY - 1)) R.close_chunk();
        }

        // ---- combine the W slices in wave order, integrate, store -------------------------------------------
        auto finish = [&](uint32_t logical, float sx, float sy) {
            if (p.split > 1) {
                if (logical < p.n_recv) {
                    float2 *slot = &p.parts[(size_t)part_y * p.n_recv + logical];
                    if constexpr (FUSED) {
                        // agent-scope relaxed store (one 8-byte access): written through to the point every XCD's loads of the
                        // same scope read
                        const uint64_t bits = (uint64_t)__float_as_uint(sx) | ((uint64_t)__float_as_uint(sy) << 32);
                        __hip_atomic_store(reinterpret_cast<uint64_t *>(slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        *slot = make_float2(sx, sy);
                    }
                }
            } else {
                finish_receiver(p, logical, sx, sy, dt);
            }
        };

        if constexpr (W == 1) {
#pragma unroll
            for (int k = 0; k < K; k++) finish(recv_base + k * WAVE + lane, R.s[k].x, R.s[k].y);
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) partial[wid][k * WAVE + lane] = make_float2(R.s[k].x, R.s[k].y);
            __syncthreads();
#pragma unroll
            for (uint32_t slot = tid; slot < WAVE * K; slot += WAVE * W) {
                float sx = 0.0f, sy = 0.0f;
#pragma unroll
                for (int s = 0; s < W; s++) {
                    const float2 t = partial[s][slot];
                    sx = __fadd_rn(sx, t.x);
                    sy = __fadd_rn(sy, t.y);
                }
                finish(recv_base + slot, sx, sy);
            }
        }

        if constexpr (FUSED) {
            // EVERY workgroup must reach this tail: there is no early return anywhere above, and none may be added -- a
            // workgroup that left without drawing its ticket would leave its tile unfinished in this launch and the ticket
            // non-zero for the next (the host re-zeroes the tickets at every upload and chain build, step_chain.hip zero_tickets).
            // The last workgroup of this receiver tile to get here adds the parts, in part order like finish_kernel, and
            // integrates.  Every thread's part stores have been acknowledged (vmcnt(0)) before the workgroup takes its ticket;
            // the last arriver therefore finds all parts written, and reads them with the same scope they were written with.
            if (p.split > 1) {
                __shared__ uint32_t is_last;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    const uint32_t t = __hip_atomic_fetch_add(&p.tickets[tile_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    is_last = t == p.split - 1 ? 1u : 0u;
                    if (is_last) __hip_atomic_store(&p.tickets[tile_x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
                }
                __syncthreads();
                if (is_last) {
                    for (uint32_t slot = tid; slot < WAVE * K; slot += WAVE * W) {
                        const uint32_t logical = recv_base + slot;
                        if (logical >= p.n_recv) continue;
                        // like finish_kernel: every part load issued before the first use (unused slots re-read the last part
                        // and are dropped by a select), or the loads would queue up behind each other's round trips
                        uint64_t bits[MAX_SPLIT];
#pragma unroll
                        for (uint32_t s = 0; s < (uint32_t)MAX_SPLIT; s++)
                            bits[s] = __hip_atomic_load(reinterpret_cast<const uint64_t *>(&p.parts[(size_t)(s < p.split ? s : p.split - 1) * p.n_recv + logical]),
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // the integrator's state rides the same round trip (finish_receiver would fetch it behind the sums)
                        const uint32_t i = receiver_slot(p, logical);
                        const float2 a0 = p.acc[i], v0 = p.vel[i], q0 = p.pos_in[i];
                        float sx = 0.0f, sy = 0.0f;
#pragma unroll
                        for (uint32_t s = 0; s < (uint32_t)MAX_SPLIT; s++) {
                            const float nx = __fadd_rn(sx, __uint_as_float((uint32_t)bits[s])), ny = __fadd_rn(sy, __uint_as_float((uint32_t)(bits[s] >> 32)));
                            sx = s < p.split ? nx : sx;
                            sy = s < p.split ? ny : sy;
                        }
                        // same arithmetic, same roundings as finish_receiver / finish_kernel
                        float2 a = make_float2(sx, sy);
                        if (p.flags & STEP_ACC_IN) {
                            a.x = __fadd_rn(a0.x, a.x);
                            a.y = __fadd_rn(a0.y, a.y);
                        }
                        p.acc[i] = a;
                        if (p.flags & STEP_NO_FINALIZE) continue;
                        float2 v = v0, q = q0;
                        v.x = __fadd_rn(v.x, __fmul_rn(a.x, dt));
                        v.y = __fadd_rn(v.y, __fmul_rn(a.y, dt));
                        q.x = __fadd_rn(q.x, __fmul_rn(v.x, dt));
                        q.y = __fadd_rn(q.y, __fmul_rn(v.y, dt));
                        p.vel[i] = v;
                        p.pos_out[i] = q;
                        if (i < p.n_mirror) p.mirror[i] = q;
                    }
                }
            }
        }

        if constexpr (!PERSIST) {
            break;
        } else {
            // the next work item of this workgroup; the LDS reduction buffers are reused, so every wave must be done with them
            // (readfirstlane: the compiler must keep seeing wave-uniform values, or the source loads stop being scalar loads)
            item = __builtin_amdgcn_readfirstlane(item + gridDim.x);
            if (item >= n_tiles * p.split) break;
            tile_x = __builtin_amdgcn_readfirstlane(item % n_tiles);
            part_y = __builtin_amdgcn_readfirstlane(item / n_tiles);
            __syncthreads();
        }
    }
}

// ---- lane-split step: several source slices per receiver inside one wave -------------------------------------------
//
// Between the one-workgroup chain (N <= 256) and launches that fill the chip (N >~ 20 000) a step is bound by latency:
// the kernel boundary plus one wave's dependency chain over its slice of the sources.  The source split shortens the
// chain by giving a receiver tile to several workgroups -- and pays a second dependent kernel (the finish kernel,
// 1.7 us of boundary + ~1 us of its own) to add their sums.  Here the extra slices live INSIDE the wave instead: the 64
// lanes are `H` groups over the same 64 / H receivers, every group walks its own slice, and the W x H partial sums meet
// in LDS like the W of the ordinary kernel.  Lanes of different groups need different sources at the same time, so the
// source cannot be a wave-uniform scalar operand: the workgroup stages the sources in LDS, tile by tile (2 x 64 x W
// sources per tile, coalesced loads, fetched into registers one tile ahead), and every lane reads its own slice of the
// tile -- per-lane data is what LDS is for (cf. "Why the LDS-tile route trails").  K = 1, split = 1, slices in 8-source
// granules, Kahan block closes every 128 sources a lane has added.
template <int W, int H>
__global__ __launch_bounds__(WAVE *W) void lane_split_kernel(const StepParams p) {
    constexpr uint32_t R = WAVE / H;        // receivers per workgroup
    constexpr uint32_t V = W * H;           // source slices per receiver
    constexpr uint32_t T = 2 * WAVE * W;    // sources per staged tile: two per thread (12 KB at W = 8, 24 KB at W = 16)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float2 *sxy = reinterpret_cast<float2 *>(lds_raw);
    float *sgm = reinterpret_cast<float *>(sxy + T);
    float2 *partial = reinterpret_cast<float2 *>(sgm + T);   // [V][R], then [S][R]
    const uint32_t n0 = p.src_end[0] - p.src_begin[0];
    const uint32_t ntiles = (n0 + T - 1) / T;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1);
    const uint32_t wid = tid >> 6;
    const uint32_t r = lane % R, h = lane / R;
    const uint32_t v = wid * H + h;         // this lane's slice of every tile
    const float dt = *p.dt;

    // tile t of the sources, two per thread, fetched into registers one tile ahead (coalesced float2 / float loads);
    // indices past the end are clamped -- the walk below never reads those LDS entries
    float2 q0 = make_float2(0.f, 0.f), q1 = q0;
    float m0 = 0.f, m1 = 0.f;
    auto fetch = [&](uint32_t t) {
        const uint32_t last = p.src_begin[0] + n0 - 1;
        const uint32_t j0 = min(p.src_begin[0] + t * T + tid, last), j1 = min(j0 + WAVE * W, last);
        q0 = p.src_pos[j0];
        q1 = p.src_pos[j1];
        m0 = p.src_gm[j0];
        m1 = p.src_gm[j1];
    };
    if (ntiles > 0) fetch(0);

    Receivers<1> Rv;
    {
        uint32_t i = blockIdx.x * R + r;
        i = i < p.n_recv ? i : p.n_recv - 1;  // tail lanes redo the last receiver; the epilogue masks them
        i = receiver_slot(p, i);
        const float2 q = p.pos_in[i];
        Rv.p[0] = f2v{q.x, q.y};
        Rv.r[0] = p.radius[i];
    }
    Rv.clear();
    // The integrating threads (first wave, lanes < R: lane == r there, so Rv.p IS their receiver's position) fetch the
    // velocity now: behind the reduction it would be one more memory round trip (~0.3 us of a ~3 us step) on the
    // critical path of a launch that is nothing but latency.
    const uint32_t my_logical = blockIdx.x * R + lane;
    const bool integrates = wid == 0 && lane < R && my_logical < p.n_recv && p.flags == 0;
    float2 vel0 = make_float2(0.f, 0.f);
    if (integrates) vel0 = p.vel[receiver_slot(p, my_logical)];
    uint32_t open_groups = 0;   // groups of four added since the last block close

    for (uint32_t t = 0; t < ntiles; t++) {
        if (t > 0) __syncthreads();   // every lane is done reading the previous tile
        sxy[tid] = q0;
        sxy[tid + WAVE * W] = q1;
        sgm[tid] = m0;
        sgm[tid + WAVE * W] = m1;
        if (t + 1 < ntiles) fetch(t + 1);   // lands while this tile is being walked
        __syncthreads();

        // this lane's slice of the tile, in whole 8-source granules; empty slices have v_hi == v_lo (clamped: no wrap-around)
        const uint32_t n_t = min(T, n0 - t * T);
        const uint32_t nunits = (n_t + 7u) / 8u;
        const uint32_t per = (nunits + V - 1) / V;
        const uint32_t u_lo = min(v * per, nunits);
        const uint32_t u_hi = min(u_lo + per, nunits);
        const uint32_t v_lo = u_lo * 8u;
        const uint32_t v_hi = max(min(u_hi * 8u, n_t), v_lo);

        // Four sources per group: two 16-byte reads of positions, one of G*m (lanes of one lane group read the same
        // address, the groups different ones).  The next group's reads are issued before this group's arithmetic: a
        // lane's slice is a serial chain, and an LDS round trip per four interactions would otherwise sit on it.
        uint32_t j = v_lo;
        const uint32_t groups = (v_hi - v_lo) / 4u;
        v4f P01 = {0.f, 0.f, 0.f, 0.f}, P23 = {0.f, 0.f, 0.f, 0.f}, G4 = {0.f, 0.f, 0.f, 0.f};
        if (groups > 0) {
            P01 = *reinterpret_cast<const v4f *>(&sxy[j]);
            P23 = *reinterpret_cast<const v4f *>(&sxy[j + 2]);
            G4 = *reinterpret_cast<const v4f *>(&sgm[j]);
        }
        for (uint32_t g = 0; g < groups; g++) {
            const v4f A01 = P01, A23 = P23, AG = G4;
            const uint32_t jn = g + 1 < groups ? j + 4 : j;   // the last group re-reads itself instead of branching
            P01 = *reinterpret_cast<const v4f *>(&sxy[jn]);
            P23 = *reinterpret_cast<const v4f *>(&sxy[jn + 2]);
            G4 = *reinterpret_cast<const v4f *>(&sgm[jn]);
            interact<1, false>(Rv, f2v{A01[0], A01[1]}, AG[0]);
            interact<1, false>(Rv, f2v{A01[2], A01[3]}, AG[1]);
            interact<1, false>(Rv, f2v{A23[0], A23[1]}, AG[2]);
            interact<1, false>(Rv, f2v{A23[2], A23[3]}, AG[3]);
            j += 4;
            if (++open_groups == 32) {   // 128 sources: half the classic kernel's block (K = 1 chains round more often)
                Rv.close_chunk();
                open_groups = 0;
            }
        }
        for (; j < v_hi; j++) {
            const float2 a0 = sxy[j];
            interact<1, false>(Rv, f2v{a0.x, a0.y}, sgm[j]);
        }
    }
    Rv.close_chunk();   // whatever is still open (a no-op on exact zeros)

    // W x H partial sums per receiver meet in LDS.  Two levels, fixed order: thread (c, r) of the first wave adds the
    // slices [c * V / S, (c + 1) * V / S) of receiver r, then thread r adds those S sums -- 64 dependent adds by 16
    // threads would be the longest serial chain of a short launch.  Plain adds, like the classic kernel's sum over its W
    // slices and split parts (compensating them measured +0.15 us per step at N = 500 ... 2 000 and bought no accuracy:
    // the error sits in the lanes' own block sums, which is why those close every 128 sources here).
    partial[v * R + r] = make_float2(Rv.s[0].x, Rv.s[0].y);
    __syncthreads();
    constexpr uint32_t S = WAVE / R;        // = H second-level terms per receiver, computed by the first wave's 64 lanes
    constexpr uint32_t PER = V / S;         // = W slices per first-level sum
    if (wid == 0) {
        const uint32_t c = lane / R, rr = lane % R;
        float sx = 0.0f, sy = 0.0f;
#pragma unroll
        for (uint32_t s2 = 0; s2 < PER; s2++) {
            const float2 t = partial[(c * PER + s2) * R + rr];
            sx = __fadd_rn(sx, t.x);
            sy = __fadd_rn(sy, t.y);
        }
        // same wave: LDS executes one wave's accesses in order; the fences only stop the compiler from reordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        partial[V * R + lane] = make_float2(sx, sy);   // second-level buffer behind the first: [S][R]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < R) {
            float ax = 0.0f, ay = 0.0f;
#pragma unroll
            for (uint32_t c2 = 0; c2 < S; c2++) {
                const float2 t = partial[V * R + c2 * R + lane];
                ax = __fadd_rn(ax, t.x);
                ay = __fadd_rn(ay, t.y);
            }
            if (integrates) {
                // finish_receiver with everything it reads already in registers: acc, then the reference's integrator
                // roundings (vel += acc*dt; pos += vel*dt; sim_cpu.c:191-193)
                const uint32_t i = receiver_slot(p, my_logical);
                p.acc[i] = make_float2(ax, ay);
                float2 vv = vel0;
                vv.x = __fadd_rn(vv.x, __fmul_rn(ax, dt));
                vv.y = __fadd_rn(vv.y, __fmul_rn(ay, dt));
                float2 q = make_float2(Rv.p[0].x, Rv.p[0].y);
                q.x = __fadd_rn(q.x, __fmul_rn(vv.x, dt));
                q.y = __fadd_rn(q.y, __fmul_rn(vv.y, dt));
                p.vel[i] = vv;
                p.pos_out[i] = q;
                if (i < p.n_mirror) p.mirror[i] = q;
            } else {
                finish_receiver(p, my_logical, ax, ay, dt);   // chained passes (flags): the general epilogue
            }
        }
    }
}

// ---- the one-workgroup chain -------------------------------------------------------------------------------------
//
// Worlds of a few hundred particles are bound by the kernel boundary, not by arithmetic: at N = 250 a step is 30 000
// interactions -- 1.3 us on ONE compute unit -- while a dependent launch costs 1.6-1.8 us before its kernel has
// loaded anything (profiles/r01_ubench6_launch_floor.txt).  So such a world runs its whole n-step chain inside one
// launch of one 1024-thread workgroup: positions ping-pong between two LDS arrays, G*m sits in LDS, radii and
// velocities stay in registers, and a step is {force loop from LDS, partial sums to LDS, barrier, sum + integrate,
// barrier}.  Nothing crosses the chip per step.  (More than one workgroup would need an in-kernel all-gather of the
// positions per step: 2.4 us for 8 KB between 32 CUs, MI355X_MICROARCH.md price list "allgather" -- dearer than the
// kernel boundary it would replace, which is why the path stops at one workgroup.)
//
// Bit-compatible with the per-step kernel by construction: same interaction statements, same slicing arithmetic
// (granule 8, W = 16 / tiles slices per receiver tile), same block closes, same reduction order, same integrator
// roundings -- tests/test_gpu_parity.py holds it to plain launches of k = 2, w = 16 / tiles, split = 1, unit = 8.
constexpr uint32_t CHAIN_K = 2;

__global__ __launch_bounds__(1024) void chain_kernel(const ChainParams p) {
    constexpr int K = CHAIN_K;
    __shared__ __attribute__((aligned(16))) float spos[2][2 * CHAIN_MAX_RECV];  // (x, y) interleaved, ping-pong
    __shared__ __attribute__((aligned(16))) float sgm[CHAIN_MAX_RECV];
    __shared__ float2 partial[16][WAVE * K];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (WAVE - 1);
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t W = 16u / p.tiles;              // waves (= source slices) per receiver tile
    const uint32_t tile = wid / W, slice = wid % W;
    const uint32_t tile_base = tile * (WAVE * K);
    const float dt = *p.dt;

    // ---- load: every position and G*m into LDS, this lane's radii and (integrating threads) velocity into registers
    for (uint32_t i = tid; i < p.n_recv; i += 1024) {
        const float2 q = p.pos[i];
        spos[0][2 * i] = q.x;
        spos[0][2 * i + 1] = q.y;
        if (i < p.n_src) sgm[i] = p.src_gm[i];
    }
    Receivers<K> R;
    uint32_t ridx[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t i = tile_base + k * WAVE + lane;
        i = i < p.n_recv ? i : p.n_recv - 1;  // tail lanes redo the last receiver; they never store
        ridx[k] = i;
        R.r[k] = p.radius[i];
    }
    // the thread that integrates receiver slot `local` of its tile (the per-step kernel's `slot = tid` thread)
    const uint32_t local = tid - tile * W * WAVE;
    const uint32_t mine = tile_base + local;
    const bool integrates = local < WAVE * K && mine < p.n_recv;
    float2 v = make_float2(0.f, 0.f), a = make_float2(0.f, 0.f);
    if (integrates) v = p.vel[mine];

    // this wave's slice of the sources: whole 8-source granules, exactly StepParams::unit = 8 with split = 1
    const uint32_t nunits = (p.n_src + 7u) / 8u;
    const uint32_t per_wave = (nunits + W - 1) / W;
    const uint32_t u_lo = min(slice * per_wave, nunits);
    const uint32_t u_hi = min(u_lo + per_wave, nunits);
    const uint32_t v_lo = u_lo * 8u;
    // an empty slice past a ragged last granule would have v_hi < v_lo (7 granules over 16 waves, 50 sources: wave 7
    // starts at 56): clamp, so that the unsigned lengths below are 0 there and the loops cannot run away
    const uint32_t v_hi = max(min(u_hi * 8u, p.n_src), v_lo);
    __syncthreads();

    int cur = 0;
    for (uint32_t step = 0; step < p.steps; step++) {
        const float *S = spos[cur];
#pragma unroll
        for (int k = 0; k < K; k++) R.p[k] = f2v{S[2 * ridx[k]], S[2 * ridx[k] + 1]};
        R.clear();
        uint32_t j = v_lo;
        const uint32_t groups = (v_hi - v_lo) / 8u;
        for (uint32_t g = 0; g < groups; g++, j += 8) {
            const v16f P = *reinterpret_cast<const v16f *>(&S[2 * j]);   // broadcast reads: every lane the same address
            const v8f G = *reinterpret_cast<const v8f *>(&sgm[j]);
            interact8<K, false>(R, P, G);
            if ((g & (8u * CLOSE_EVERY - 1)) == 8u * CLOSE_EVERY - 1) R.close_chunk();   // every 256 sources of the slice
        }
        for (; j < v_hi; j++) interact<K, false>(R, f2v{S[2 * j], S[2 * j + 1]}, sgm[j]);
        if ((v_hi - v_lo) & (CHUNK * CLOSE_EVERY - 1)) R.close_chunk();                   // a short last block
#pragma unroll
        for (int k = 0; k < K; k++) partial[wid][k * WAVE + lane] = make_float2(R.s[k].x, R.s[k].y);
        __syncthreads();
        if (integrates) {
            float sx = 0.0f, sy = 0.0f;
            for (uint32_t s = 0; s < W; s++) {   // the tile's slices in wave order: deterministic, = the per-step kernel
                const float2 t = partial[tile * W + s][local];
                sx = __fadd_rn(sx, t.x);
                sy = __fadd_rn(sy, t.y);
            }
            a = make_float2(sx, sy);
            // semi-implicit Euler with the reference's roundings (finish_receiver): vel += acc*dt; pos += vel*dt
            v.x = __fadd_rn(v.x, __fmul_rn(a.x, dt));
            v.y = __fadd_rn(v.y, __fmul_rn(a.y, dt));
            float qx = S[2 * mine], qy = S[2 * mine + 1];
            qx = __fadd_rn(qx, __fmul_rn(v.x, dt));
            qy = __fadd_rn(qy, __fmul_rn(v.y, dt));
            spos[cur ^ 1][2 * mine] = qx;
            spos[cur ^ 1][2 * mine + 1] = qy;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (integrates) {
        p.pos[mine] = make_float2(spos[cur][2 * mine], spos[cur][2 * mine + 1]);
        p.vel[mine] = v;
        p.acc[mine] = a;
    }
}

// ---- AoS <-> SoA ----------------------------------------------------------------------------------------

struct alignas(16) ParticleRec {  // == Particle (include/nbody.h): pos vel | acc mass radius
    float4 a, b;
};

__global__ void split_kernel(const ParticleRec *aos, uint32_t first, uint32_t count, float2 *pos, float2 *vel, float2 *acc,
                             float *radius, float *mass, uint32_t slot0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const ParticleRec r = aos[first + i];
    pos[slot0 + i] = make_float2(r.a.x, r.a.y);
    vel[slot0 + i] = make_float2(r.a.z, r.a.w);
    acc[slot0 + i] = make_float2(r.b.x, r.b.y);
    mass[slot0 + i] = r.b.z;
    radius[slot0 + i] = r.b.w;
}

__global__ void merge_kernel(ParticleRec *aos, uint32_t first, uint32_t count, const float2 *pos, const float2 *vel,
                             const float2 *acc, const float *radius, const float *mass, uint32_t slot0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float2 q = pos[slot0 + i], v = vel[slot0 + i], a = acc[slot0 + i];
    ParticleRec r;
    r.a = make_float4(q.x, q.y, v.x, v.y);
    r.b = make_float4(a.x, a.y, mass[slot0 + i], radius[slot0 + i]);
    aos[first + i] = r;
}

__global__ void fill_pad_kernel(float2 *pos, float2 *vel, float2 *acc, float *radius, float *mass, uint32_t slot0,
                                uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // far away, finite, massless: contributes exactly 0 as a source and stays finite as a receiver
    pos[slot0 + i] = make_float2(1.0e15f, 1.0e15f);
    vel[slot0 + i] = make_float2(0.f, 0.f);
    acc[slot0 + i] = make_float2(0.f, 0.f);
    radius[slot0 + i] = 1.0f;
    mass[slot0 + i] = 0.0f;
}

__global__ void set_scalar_kernel(float *dst, float value) { *dst = value; }

// `g` is the host's NB_G (include/nbody.h), handed in at launch like the reference's specialisation constant
// (sim_gpu.c:54-72, particle_cs.glsl:26): the device code holds no copy of the value.
__global__ void make_gm_kernel(const float *mass, float *gm, uint32_t count, float g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float m = mass[i];
    gm[i] = m > 0.0f ? __fmul_rn(m, g) : 0.0f;  // rounded as the reference's `gm = m * g` (sim_cpu.c:179)
}

// Sharded upload: the gathered source arrays (both ping-pong buffers) and the static G*m straight from the AoS
// world every rank holds.  Slots past mass_len are pads: far away, finite, massless (exact zero contribution).
__global__ void split_sources_kernel(const ParticleRec *aos, uint32_t mass_len, uint32_t n_src, float2 *pos0, float2 *pos1,
                                     float *gm, float big_g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src) return;
    float2 q = make_float2(1.0e15f, 1.0e15f);
    float g = 0.0f;
    if (i < mass_len) {
        const ParticleRec r = aos[i];
        q = make_float2(r.a.x, r.a.y);
        g = r.b.z > 0.0f ? __fmul_rn(r.b.z, big_g) : 0.0f;  // as make_gm_kernel
    }
    pos0[i] = q;
    pos1[i] = q;
    gm[i] = g;
}

inline dim3 grid1d(uint32_t count) { return dim3((count + 255u) / 256u); }

template <int VARIANT>
const void *pick(int k, int w) {
#define NB_CASE(KK, WW) \
    if (k == KK && w == WW) return reinterpret_cast<const void *>(&step_kernel<KK, WW, VARIANT>);
    // W = 4, 8, 16 are what choose_shape picks from; W = 1 is the shape whose summation order does not depend on
    // how the sources are cut up (one wave walks them all), which the sharded-vs-single bit-equality tests rely on.
    NB_CASE(1, 1) NB_CASE(1, 4) NB_CASE(1, 8) NB_CASE(1, 16)
    NB_CASE(2, 1) NB_CASE(2, 4) NB_CASE(2, 8) NB_CASE(2, 16)
#ifdef NB_TUNING_SHAPES
    // never auto-selected (profiles/r01_sweep4_shapes_by_n.txt): built only for shape scans (make TUNING=1)
    NB_CASE(1, 2) NB_CASE(2, 2)
    NB_CASE(4, 1) NB_CASE(4, 2) NB_CASE(4, 4) NB_CASE(4, 8) NB_CASE(4, 16)
#endif
#undef NB_CASE
    return nullptr;
}

const void *pick_fused(int k, int w) {
#define NB_CASE(KK, WW) \
    if (k == KK && w == WW) return reinterpret_cast<const void *>(&step_kernel<KK, WW, VARIANT_SMEM, true>);
    NB_CASE(1, 4) NB_CASE(1, 8) NB_CASE(1, 16) NB_CASE(2, 4) NB_CASE(2, 8) NB_CASE(2, 16)
#undef NB_CASE
    return nullptr;
}

// persistent launches (experiment, "persist" hook): scalar-cache route, with and without the fused finish
// (built only with make TUNING=1: the experiment lost at every size, profiles/r05_persist_probe.txt, and twelve more
// instantiations of the step kernel are not worth carrying in the library that ships)
template <bool FUSED>
const void *pick_persist(int k, int w) {
#ifdef NB_TUNING_SHAPES
#define NB_CASE(KK, WW) \
    if (k == KK && w == WW) return reinterpret_cast<const void *>(&step_kernel<KK, WW, VARIANT_SMEM, FUSED, true>);
    NB_CASE(1, 4) NB_CASE(1, 8) NB_CASE(1, 16) NB_CASE(2, 4) NB_CASE(2, 8) NB_CASE(2, 16)
#undef NB_CASE
#else
    (void)k;
    (void)w;
#endif
    return nullptr;
}

const void *pick_lane_split(int w, int h) {
#define NB_CASE(WW, HH) \
    if (w == WW && h == HH) return reinterpret_cast<const void *>(&lane_split_kernel<WW, HH>);
    NB_CASE(4, 2) NB_CASE(8, 2) NB_CASE(16, 2) NB_CASE(4, 4) NB_CASE(8, 4) NB_CASE(16, 4) NB_CASE(8, 8) NB_CASE(16, 8)
#undef NB_CASE
    return nullptr;
}

}  // namespace

// NB_HASH_OFF -- host-side launch-shape arithmetic: not part of the kernel-source hash bench.py ties PMC figures to
// Launches that do not even fill the chip once with K = 2 / W = 16 workgroups (fewer than 65 536 receivers on 256
// CUs) are priced in microseconds by a model fitted to exhaustive (K, W, split, unit) scans at N = 250 ... 50 000
// (tools/sweep_shapes.py; profiles/r02_sweep_shapes_units.txt holds the latest scan, 3 360 timed shapes).  There a wave is
// latency-bound, not issue-bound: alone on its SIMD it needs LAT us per 64-source chunk and receiver set (a serial
// dependency chain), and only beyond ~2 waves per SIMD does the chunk time grow with occupancy -- at THR us per wave,
// worse (factor A) the emptier the SIMD, and worse again for K = 1 (K1A).  Every wave also costs UFIX chunks of fixed
// work (launch, receiver loads, the LDS reduction), which is what stops the split from growing without bound, and a
// split adds the finish kernel.  Cutting the sources into more parts and finer granules shortens every wave's chain, so
// small launches want shapes the big-launch model would never pay for: N = 250 runs 3.2 us per step with 16 waves of 8
// sources instead of 4.6 us with two waves of 64, N = 2 000 5.4 instead of 6.7, N = 4 000 6.8 with 8 parts of 256-thread
// workgroups.  Mean regret of the model's pick against the scan's best: 1.7 % (worst 4.4 %).
static double small_launch_cost_us(uint32_t n_recv, uint32_t n_src, int k, int w, int sp, int unit, int cus) {
    constexpr double LAT = 1.575, THR = 0.7545, K1 = 0.987, K1A = 0.178, A = 0.179, MIX = 0.874;
    constexpr double FINISH = 2.08, W4 = 1.035, UFIX = 0.15, BASE = 2.27;
    // the longest wave slice of a workgroup, in 64-source chunks (a fraction of one when the slice granule is finer)
    const uint32_t granules = (n_src + unit - 1) / unit;
    const uint64_t groups = ((uint64_t)n_recv + WAVE * k - 1) / (WAVE * k) * (uint64_t)sp;
    const uint64_t capacity = (uint64_t)cus * (32 / w);
    const uint64_t full = groups / capacity, left = groups % capacity;
    const uint32_t part_granules = (granules + sp - 1) / sp;
    const uint32_t wave_granules = (part_granules + w - 1) / w;
    const double units = (double)k * (wave_granules ? wave_granules : 1) * (double)unit / (double)CHUNK + UFIX;
    auto chunk_time = [&](double occ) {  // us per chunk and receiver set with `occ` waves on every SIMD
        const double f = k == 1 ? K1 + K1A * (8.0 - occ) / 8.0 : 1.0;
        const double busy = occ * THR * f * (1.0 + A * (8.0 - occ) / 8.0);
        return LAT > busy ? LAT : busy;
    };
    double t = (double)full * units * chunk_time(8.0);
    if (left) {
        // the busiest CU holds ceil(left / CUs) workgroups of w waves on its 4 SIMDs
        double occ = (w / 4.0) * (double)((left + cus - 1) / cus);
        if (occ > 8.0) occ = 8.0;
        const double lock = units * chunk_time(occ);                                    // runs after the full rounds
        const double fluid = units * chunk_time(8.0) * (double)left / (double)capacity;  // packs in behind them
        t += full ? MIX * lock + (1.0 - MIX) * fluid : lock;
    }
    if (sp > 1) t += FINISH;
    if (w == 4) t *= W4;
    return t + BASE;  // what every step pays whatever its shape (dispatch, kernel boundary): keeps ties ties
}

// Lane-split shapes ("lanes" = 0, auto), from a scan of lanes x w over N = 300 ... 10 000 (tools/lane_probe.py,
// profiles/r03_lane_split_scan.txt; us per step, best lane-split shape vs the best classic shape the model above picks):
//   N = 500: 3.11 vs 3.86   800: 3.25 vs 4.25   1 200: 3.70 vs 4.98   2 000: 3.93 vs 5.41   4 000: 5.88 vs 6.93
//   5 000: 8.62 vs 8.42     8 000: 13.3 vs 12.7   10 000: 20.6 vs 15.8
// i.e. 15-28 % faster while a step is latency (N x M <~ 9e6), slower once it is throughput: every workgroup stages ALL
// the sources in LDS and the per-lane LDS reads cost issue slots a wave-uniform scalar operand does not.  Which (lanes, w)
// wins moves with the size; neighbours are within 2-3 % of each other.
int lane_split_rule(uint32_t n_recv, uint32_t n_src, int *w) {
    const double pairs = (double)n_recv * (double)n_src;
    *w = 16;
    if (n_src == 0 || n_recv == 0 || n_src > LANE_SPLIT_MAX_SRC || pairs > 9.0e6) return 1;
    if (pairs <= 1.5e5) {
        *w = 8;
        return 4;
    }
    if (pairs <= 2.5e6) {
        *w = 8;
        return 8;
    }
    return 4;
}

LaunchShape choose_shape(LaunchShape want, uint32_t n_recv, uint32_t n_src, int compute_units) {
    // Workgroups of one launch all take the same time, so a launch costs
    //     (rounds + tail) * (work per workgroup),   rounds = ceil(workgroups / resident capacity),
    // and one workgroup past a round boundary costs a whole round (1025 workgroups on 512 slots run 1.5x as
    // long as 1024; profiles/r01_shard_overhead_before_fix.txt).  Work per workgroup = K receivers per lane x
    // the 64-source chunks one wave walks.  Pick the cheapest (K, W, split); ties go to the larger K, larger W,
    // smaller split.  More, shorter workgroups also shrink the launch's ramp-up/ragged-end share.  K = 4 is left out: 71 VGPRs, lower
    // occupancy, never faster (profiles/r01_sweep4_shapes_by_n.txt).
    if (compute_units <= 0) compute_units = 256;
    if (want.lanes == 0 && want.k == 0 && want.w == 0 && want.split == 0 && want.unit == 0 && want.persist <= 1 && want.variant == VARIANT_SMEM) {
        // everything on auto (an explicit LDS-tile route or shape knob asks for the classic kernel)
        int w = 16;
        const int lanes = lane_split_rule(n_recv, n_src, &w);
        if (lanes > 1) {
            want.lanes = lanes;
            want.w = w;
        }
    }
    if (want.lanes > 1) {
        // one receiver per lane, no source split, 8-source granules, LDS-staged sources
        LaunchShape sh = want;
        sh.k = 1;
        sh.w = (want.w == 4 || want.w == 8 || want.w == 16) ? want.w : 16;
        if (sh.lanes == 8 && sh.w == 4) sh.w = 8;   // eight groups: instantiated for 8 and 16 waves
        if (sh.lanes != 2 && sh.lanes != 4 && sh.lanes != 8) sh.lanes = 4;
        sh.split = 1;
        sh.unit = 8;
        sh.variant = VARIANT_LDS;
        sh.persist = 0;
        return sh;
    }
    if (want.variant == VARIANT_LDS) want.unit = CHUNK;  // the LDS route stages whole 64-source tiles, whatever was asked
    const uint32_t chunks = (n_src + CHUNK - 1) / CHUNK;
    const bool small = ((uint64_t)n_recv + 2 * WAVE - 1) / (2 * WAVE) < (uint64_t)compute_units * 2;
    LaunchShape best = want;
    best.lanes = 1;
    double best_cost = -1.0;
    for (int k = 2; k >= 1; k--) {
        if (want.k != 0 && want.k != k) continue;
        for (int w = 16; w >= 4; w /= 2) {
            if (want.w != 0 && want.w != w) continue;
            for (int sp = 1; sp <= MAX_SPLIT; sp++) {
                if (want.split != 0 && want.split != sp) continue;
                // small launches: 1024-thread workgroups exactly while the whole launch is a handful of unsplit tiles
                // (16 waves per tile beat 8 there: 4.2 vs 4.6 us at N = 800); beyond that 256- and 512-thread
                // workgroups pack better, and the model overrates W = 16
                const bool few_unsplit_tiles = sp == 1 && ((uint64_t)n_recv + WAVE * k - 1) / (WAVE * k) <= 24;
                // ... with at least ~16 sources for every wave: 8 waves when there are no more than 128 sources
                // (N = 250: 2.9 us with 8 waves of 16 sources, 3.2 with 16 waves of 8)
                const int tiny_w = n_src <= 128 ? 8 : 16;
                if (small && want.w == 0 && (few_unsplit_tiles ? w != tiny_w : w == 16)) continue;
                // slice granule: 64 unless the launch is latency-bound; a finer one only has to win where a part holds
                // fewer chunks than the workgroup has waves, and ties keep the coarser granule (64 first)
                for (int unit = CHUNK; unit >= 8; unit /= 2) {
                    if (want.unit != 0 && want.unit != unit) continue;
                    if (!small && want.unit == 0 && unit != CHUNK) continue;
                    double cost;
                    if (small) {
                        cost = small_launch_cost_us(n_recv, n_src, k, w, sp, unit, compute_units);
                    } else {
                        const uint64_t groups = ((uint64_t)n_recv + WAVE * k - 1) / (WAVE * k) * (uint64_t)sp;
                        const uint64_t capacity = (uint64_t)compute_units * (32 / w);  // 8 waves per SIMD at <= 64 VGPRs
                        const uint64_t rounds = (groups + capacity - 1) / capacity;
                        const uint32_t part_chunks = (chunks + sp - 1) / sp;
                        const uint32_t wave_chunks = (part_chunks + w - 1) / w;
                        // + 1 chunk-equivalent per workgroup for prologue/epilogue; + TAIL rounds per launch for ramp-up
                        // and the ragged end (measured: 2-round launches run 4.5 % over, 16-round ones 0.1 % over:
                        // profiles/r01_shard_overhead_split.txt); a split adds the finish kernel and the parts traffic
                        constexpr double TAIL = 0.13;
                        cost = ((double)rounds + TAIL) * ((double)k * (wave_chunks ? wave_chunks : 1) + 1.0);
                        if (sp > 1) cost += 3.0 + 0.02 * sp;
                        if (w < 16) cost *= 1.01;
                        // K = 1 per interaction at large N: 1.6 % slower than K = 2 with the plain body (48.6 vs 47.8 ms
                        // per launch at N = 2^20; it was 25 % with the packed body, whose single statement ran alone)
                        if (k == 1) cost *= 1.02;
                    }
                    if (best_cost < 0.0 || cost < best_cost * 0.999) {
                        best_cost = cost;
                        best.k = k;
                        best.w = w;
                        best.split = sp;
                        best.unit = unit;
                    }
                }
            }
        }
    }
    if (best_cost < 0.0) {  // explicit w = 1 (or a tuning-build shape): honour the request as given
        best.k = want.k ? want.k : 2;
        best.w = want.w ? want.w : 16;
        best.split = want.split ? want.split : 1;
        best.unit = want.unit ? want.unit : CHUNK;
    }
    return best;
}

// NB_HASH_ON
bool shape_is_persistent(LaunchShape s) { return s.persist > 1 && s.lanes <= 1 && s.variant == VARIANT_SMEM && s.w >= 4 && s.k <= 2; }

const void *step_kernel_fn(LaunchShape s) {
    if (s.lanes > 1) return pick_lane_split(s.w, s.lanes);
    if (shape_is_persistent(s)) return pick_persist<false>(s.k, s.w);
    return s.variant == VARIANT_SMEM ? pick<VARIANT_SMEM>(s.k, s.w) : pick<VARIANT_LDS>(s.k, s.w);
}

const void *step_kernel_fused_fn(LaunchShape s) {
    if (s.lanes > 1 || s.variant != VARIANT_SMEM) return nullptr;
    if (shape_is_persistent(s)) return pick_persist<true>(s.k, s.w);
    return pick_fused(s.k, s.w);
}

dim3 step_grid(LaunchShape s, uint32_t n_recv) {
    if (s.lanes > 1) return dim3((n_recv + WAVE / s.lanes - 1) / (WAVE / s.lanes), 1);
    const uint32_t tiles = (n_recv + WAVE * s.k - 1) / (WAVE * s.k), parts = s.split > 1 ? s.split : 1;
    if (shape_is_persistent(s)) {
        // `persist` work items (tile, part) per workgroup, walked with a stride of the grid size
        const uint64_t items = (uint64_t)tiles * parts;
        return dim3((uint32_t)((items + (uint32_t)s.persist - 1) / (uint32_t)s.persist), 1);
    }
    return dim3(tiles, parts);
}

size_t step_lds_bytes(LaunchShape s, uint32_t n_src) {
    if (s.lanes <= 1) return 0;
    (void)n_src;   // the sources pass through one tile of 2 * 64 * w entries, whatever their number
    return (size_t)2 * WAVE * s.w * 12 + ((size_t)s.w + 1) * WAVE * sizeof(float2);   // tile (x, y, G*m) + [w * lanes][64 / lanes] partial sums + 64 second-level sums
}
const void *finish_kernel_fn() { return reinterpret_cast<const void *>(&finish_kernel); }
dim3 finish_grid(uint32_t n_recv) { return dim3((n_recv + 255u) / 256u); }
dim3 finish_block() { return dim3(256); }
dim3 step_block(LaunchShape s) { return dim3(WAVE * s.w); }

uint32_t chain_tiles(uint32_t n_recv) {
    for (uint32_t t = 1; t <= 4; t *= 2)
        if (n_recv <= t * WAVE * CHAIN_K) return t;
    return 0;
}

void launch_chain(hipStream_t st, const ChainParams &p) {
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(1024), 0, st, p);
}

void launch_split(hipStream_t st, const void *aos, uint32_t first, uint32_t count, float2 *pos, float2 *vel, float2 *acc,
                  float *radius, float *mass, uint32_t slot0) {
    if (count == 0) return;
    hipLaunchKernelGGL(split_kernel, grid1d(count), dim3(256), 0, st, static_cast<const ParticleRec *>(aos), first, count,
                       pos, vel, acc, radius, mass, slot0);
}

void launch_fill_pad(hipStream_t st, float2 *pos, float2 *vel, float2 *acc, float *radius, float *mass, uint32_t slot0,
                     uint32_t count) {
    if (count == 0) return;
    hipLaunchKernelGGL(fill_pad_kernel, grid1d(count), dim3(256), 0, st, pos, vel, acc, radius, mass, slot0, count);
}

void launch_set_scalar(hipStream_t st, float *dst, float value) {
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, dst, value);
}

void launch_make_gm(hipStream_t st, const float *mass, float *gm, uint32_t count, float g) {
    if (count == 0) return;
    hipLaunchKernelGGL(make_gm_kernel, grid1d(count), dim3(256), 0, st, mass, gm, count, g);
}

void launch_merge(hipStream_t st, void *aos, uint32_t first, uint32_t count, const float2 *pos, const float2 *vel,
                  const float2 *acc, const float *radius, const float *mass, uint32_t slot0) {
    if (count == 0) return;
    hipLaunchKernelGGL(merge_kernel, grid1d(count), dim3(256), 0, st, static_cast<ParticleRec *>(aos), first, count, pos,
                       vel, acc, radius, mass, slot0);
}

void launch_split_sources(hipStream_t st, const void *aos, uint32_t mass_len, uint32_t n_src, float2 *pos0, float2 *pos1,
                          float *gm, float g) {
    if (n_src == 0) return;
    hipLaunchKernelGGL(split_sources_kernel, grid1d(n_src), dim3(256), 0, st, static_cast<const ParticleRec *>(aos), mass_len,
                       n_src, pos0, pos1, gm, g);
}

}  // namespace nb
