// clock_probe.hip -- which shader clock does the chip hold under THIS instruction mix, and how many cycles does one
// wave-interaction take when nothing but the interaction body is issued?  (include/nbody_hip.h nb_hip_probe_clock)
//
// A measurement aid, not a step kernel: the product kernels carry no time stamps (MI355X_MICROARCH.md "DVFS give-back"
// item 6: stamps live in a separate diagnostic kernel).  The probe fills the chip exactly like the N = 2^20 step launch
// does -- 1024-thread workgroups, 8 waves per SIMD, two receivers per lane, the paired-rsq statement of
// interaction_asm.h on scalar (SGPR) source operands, eight sources per group -- and loops over the same eight sources
// `iters` times, so the loop is the step kernel's inner loop minus its scalar loads.  Every wave stamps s_memtime (shader
// cycles) and s_memrealtime (the constant 100 MHz reference) once before and once after the loop:
//     held clock              = d(memtime) / d(memrealtime) x reference rate
//     cycles per wave-interaction = d(memtime) / (waves per SIMD) / (interactions the wave issued)
// A caller that runs it right after a timed leg reads the clock the chip held for that leg's kind of work.
#include "interaction_asm.h"
#include "pipeline_internal.h"

#include <algorithm>

namespace {

constexpr int PROBE_WAVES = 16;           // waves per workgroup (1024 threads), as the N = 2^20 shape
constexpr int PROBE_WAVES_PER_SIMD = 8;   // 2 workgroups per CU x 16 waves / 4 SIMDs

struct ProbeStamp {
    uint64_t cycles;    // s_memtime ticks across the loop
    uint64_t ref;       // s_memrealtime ticks across the same interval
};

__global__ __launch_bounds__(64 * PROBE_WAVES, 8) void clock_probe_kernel(const float *__restrict__ src, uint32_t iters,
                                                                            ProbeStamp *__restrict__ stamps, float *__restrict__ sink) {
    const uint32_t tid = threadIdx.x;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // eight sources in SGPRs: (x, y) x 8, then G*m x 8 (wave-uniform address -> s_load_dwordx16 / x8)
    float sx[8], sy[8], sg[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        sx[u] = src[2 * u];
        sy[u] = src[2 * u + 1];
        sg[u] = src[16 + u];
    }
    // two receivers per lane, somewhere else than the sources
    float px0 = 1000.0f + (float)tid, py0 = -500.0f + (float)blockIdx.x, r0 = 2.0f;
    float px1 = -3000.0f - (float)tid, py1 = 700.0f + (float)blockIdx.x, r1 = 3.0f;
    float ax0 = 0.f, ay0 = 0.f, ax1 = 0.f, ay1 = 0.f;

    __builtin_amdgcn_sched_barrier(0);
    uint64_t c0 = __builtin_amdgcn_s_memtime();
    uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): both stamps (and the source loads) are back
    __builtin_amdgcn_sched_barrier(0);
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm(NB_INTERACTION2_ASM
                : [ax0] "+v"(ax0), [ay0] "+v"(ay0), [ax1] "+v"(ax1), [ay1] "+v"(ay1)
                : [sx] "s"(sx[u]), [sy] "s"(sy[u]), [g] "s"(sg[u]), [px0] "v"(px0), [py0] "v"(py0), [r0] "v"(r0), [px1] "v"(px1),
                  [py1] "v"(py1), [r1] "v"(r1)
                : NB_CLOBBERS2);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the stamps must not be read before the last accumulation has issued: make them depend on it
    asm volatile("" ::"v"(ax0), "v"(ay0), "v"(ax1), "v"(ay1));
    uint64_t c1 = __builtin_amdgcn_s_memtime();
    uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if ((tid & 63u) == 0) {
        ProbeStamp s;
        s.cycles = c1 - c0;
        s.ref = t1 - t0;
        stamps[blockIdx.x * PROBE_WAVES + wid] = s;
    }
    // keep the sums alive
    if (ax0 + ay0 + ax1 + ay1 == 12345.678f) sink[tid] = ax0;
}

}  // namespace

extern "C" int nb_hip_probe_clock(double target_ms, double *clock_ghz, double *clock_ghz_min, double *clock_ghz_max,
                                  double *cycles_per_wave_interaction, double *elapsed_ms) {
    using namespace nbi;
    use_device();
    NB_ASSERT(target_ms > 0.0 && target_ms <= 2000.0, "probe length %g ms (0 < length <= 2000)", target_ms);
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, g_dev.ordinal) != hipSuccess || wall_khz <= 0) {
        (void)hipGetLastError();
        wall_khz = 100000;   // s_memrealtime: 100 MHz on gfx9 (MI355X_MICROARCH.md)
    }
    const int groups = g_dev.compute_units * 2;   // exactly the resident capacity: every wave lives for the whole launch
    const size_t waves = (size_t)groups * PROBE_WAVES;
    // one loop trip = 8 sources x 2 receivers = 16 wave-interactions per wave, 8 waves per SIMD, ~27 cycles each at ~2.3 GHz
    const double trip_us = 16.0 * PROBE_WAVES_PER_SIMD * 27.0 / 2300.0;
    const uint32_t iters = (uint32_t)std::max(64.0, target_ms * 1000.0 / trip_us);

    float host_src[24];
    for (int u = 0; u < 8; u++) {
        host_src[2 * u] = 10.0f * (float)u;
        host_src[2 * u + 1] = -7.0f * (float)u;
        host_src[16 + u] = 1.0e4f + (float)u;
    }
    float *src = dev_alloc<float>(24);
    float *sink = dev_alloc<float>(64 * PROBE_WAVES);
    ProbeStamp *stamps = dev_alloc<ProbeStamp>(waves);
    hipStream_t st;
    hipEvent_t e0, e1;
    ASSERT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "probe stream");
    ASSERT_HIP(hipEventCreate(&e0), "event");
    ASSERT_HIP(hipEventCreate(&e1), "event");
    ASSERT_HIP(hipMemcpyAsync(src, host_src, sizeof host_src, hipMemcpyHostToDevice, st), "probe sources");
    ASSERT_HIP(hipEventRecord(e0, st), "record");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(groups), dim3(64 * PROBE_WAVES), 0, st, src, iters, stamps, sink);
    ASSERT_HIP(hipGetLastError(), "clock probe launch (%d workgroups, %u trips)", groups, iters);
    ASSERT_HIP(hipEventRecord(e1, st), "record");
    std::vector<ProbeStamp> host(waves);
    ASSERT_HIP(hipMemcpyAsync(host.data(), stamps, waves * sizeof(ProbeStamp), hipMemcpyDeviceToHost, st), "probe stamps");
    ASSERT_HIP(hipStreamSynchronize(st), "probe sync");
    float ms = 0.0f;
    ASSERT_HIP(hipEventElapsedTime(&ms, e0, e1), "elapsed");
    ASSERT_HIP(hipEventDestroy(e0), "event");
    ASSERT_HIP(hipEventDestroy(e1), "event");
    ASSERT_HIP(hipStreamDestroy(st), "probe stream");
    dev_free(src);
    dev_free(sink);
    dev_free(stamps);

    std::vector<double> ghz, cyc;
    ghz.reserve(waves);
    cyc.reserve(waves);
    for (const ProbeStamp &s : host) {
        if (s.ref == 0) continue;
        ghz.push_back((double)s.cycles / (double)s.ref * (double)wall_khz * 1.0e-6);
        cyc.push_back((double)s.cycles / PROBE_WAVES_PER_SIMD / ((double)iters * 16.0));
    }
    NB_ASSERT(!ghz.empty(), "clock probe: no wave reported a stamp");
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    if (clock_ghz) *clock_ghz = ghz[ghz.size() / 2];
    if (clock_ghz_min) *clock_ghz_min = ghz.front();
    if (clock_ghz_max) *clock_ghz_max = ghz.back();
    if (cycles_per_wave_interaction) *cycles_per_wave_interaction = cyc[cyc.size() / 2];
    if (elapsed_ms) *elapsed_ms = (double)ms;
    return (int)ghz.size();
}
