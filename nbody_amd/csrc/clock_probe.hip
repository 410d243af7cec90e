// clock_probe.hip -- which shader clock does the chip hold under THIS instruction mix, and how many cycles does one
// wave-interaction take when nothing but the interaction body is issued?  (include/nbody_hip.h nb_hip_probe_clock)
//
// A measurement aid, not a step kernel: the product kernels carry no time stamps (MI355X_MICROARCH.md "DVFS give-back"
// item 6: stamps live in a separate diagnostic kernel).  The probe fills the chip exactly like the N = 2^20 step launch
// does -- 1024-thread workgroups, 8 waves per SIMD, two receivers per lane, the paired-rsq statement of
// interaction_asm.h on scalar (SGPR) source operands, eight sources per group -- and loops over the same eight sources
// `iters` times, so the loop is the step kernel's inner loop minus its scalar loads.  Every wave stamps s_memtime (shader
// cycles) and s_memrealtime (the constant 100 MHz reference) once before and once after the loop:
//     held clock              = d(memtime) / d(memrealtime) x reference rate      (waves that spanned the whole loop)
//     cycles per wave-interaction = longest d(memtime) / (waves per SIMD) / (interactions one wave issued)
// The clock it reads is the clock the chip holds FOR THIS LOOP, which is denser than the step kernel's (no scalar loads,
// no epilogue) and therefore a little lower; the sampler below reads the clock under the step kernel itself.
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

// ---- the clock sampler: the shader clock WHILE other kernels run ------------------------------------------------------
//
// The probe above loads the chip with its own loop, and a denser loop than the step kernel's (no scalar loads, no
// epilogue) makes the chip hold a LOWER clock than the step kernel does (measured: 2.23 GHz against 2.30).  What a timed leg
// needs is the clock the chip held during that leg.  So: a handful of one-wave workgroups (the dispatcher deals
// consecutive workgroups to consecutive XCDs: one per XCD) that do nothing but stamp s_memtime / s_memrealtime once per
// period and sleep in between, launched on their own stream BEFORE the leg and told to leave after it through a word of
// page-locked host memory.  They hold 8 of the chip's 8192 wave slots and issue a few scalar instructions per period.  Every
// wave leaves by itself after max_samples periods (a bounded kernel: no wave can outlive the stop word by more than one
// period, nor the bound without it).
struct SamplerPair {
    uint64_t cycles, ref;
};

__global__ __launch_bounds__(64) void clock_sampler_kernel(SamplerPair *__restrict__ out, uint32_t *__restrict__ count, uint32_t *__restrict__ xcc,
                                                           const uint32_t *stop, uint32_t max_samples, uint32_t period_ticks) {
    const bool writer = threadIdx.x == 0;
    uint32_t id = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    uint32_t i = 0;
    while (i < max_samples) {
        const uint64_t c = __builtin_amdgcn_s_memtime();
        const uint64_t t = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (writer) {
            SamplerPair v;
            v.cycles = c;
            v.ref = t;
            out[(size_t)blockIdx.x * max_samples + i] = v;
        }
        i++;
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
        uint64_t now = t;
        while (now - t < period_ticks) {
            __builtin_amdgcn_s_sleep(127);
            now = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    }
    if (writer) {
        count[blockIdx.x] = i;
        xcc[blockIdx.x] = id & 0xfu;
    }
}

constexpr int SAMPLER_WAVES = 8;

struct Sampler {
    bool running = false;
    hipStream_t stream = nullptr;
    SamplerPair *out = nullptr;
    uint32_t *count = nullptr, *xcc = nullptr;
    uint32_t *stop = nullptr;   // page-locked host word, read by the waves with system scope
    uint32_t max_samples = 0;
    int wall_khz = 100000;
} g_sampler;

}  // namespace

extern "C" int nb_hip_clock_sampler_begin(double period_ms, double max_ms) {
    using namespace nbi;
    use_device();
    Sampler &S = g_sampler;
    NB_ASSERT(!S.running, "clock sampler already running");
    // a measurement aid must not take the process down over its own bound: out-of-range requests are clamped (the waves
    // leave by themselves after max_ms, so a leg longer than the bound is simply sampled over its first 20 s)
    if (!(period_ms >= 0.05)) period_ms = 0.05;
    if (!(max_ms <= NB_CLOCK_SAMPLER_MAX_MS)) max_ms = NB_CLOCK_SAMPLER_MAX_MS;
    if (!(max_ms >= period_ms)) max_ms = period_ms;
    if (hipDeviceGetAttribute(&S.wall_khz, hipDeviceAttributeWallClockRate, g_dev.ordinal) != hipSuccess || S.wall_khz <= 0) {
        (void)hipGetLastError();
        S.wall_khz = 100000;
    }
    S.max_samples = (uint32_t)(max_ms / period_ms) + 2;
    S.out = dev_alloc<SamplerPair>((size_t)SAMPLER_WAVES * S.max_samples);
    S.count = dev_alloc<uint32_t>(SAMPLER_WAVES);
    S.xcc = dev_alloc<uint32_t>(SAMPLER_WAVES);
    ASSERT_HIP(hipHostMalloc(reinterpret_cast<void **>(&S.stop), sizeof(uint32_t), hipHostMallocDefault), "sampler stop word");
    *S.stop = 0u;
    ASSERT_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking), "sampler stream");
    ASSERT_HIP(hipMemsetAsync(S.count, 0, SAMPLER_WAVES * sizeof(uint32_t), S.stream), "sampler counts");
    const uint32_t period_ticks = (uint32_t)(period_ms * (double)S.wall_khz);
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(SAMPLER_WAVES), dim3(64), 0, S.stream, S.out, S.count, S.xcc, S.stop, S.max_samples, period_ticks);
    ASSERT_HIP(hipGetLastError(), "clock sampler launch");
    S.running = true;
    return SAMPLER_WAVES;
}

extern "C" int nb_hip_clock_sampler_end(double *clock_ghz, double *clock_ghz_min, double *clock_ghz_max, double *per_xcd_ghz8,
                                        double *profile10, double *span_ms, uint32_t *dropped_intervals) {
    using namespace nbi;
    use_device();
    Sampler &S = g_sampler;
    NB_ASSERT(S.running, "clock sampler not running");
    __atomic_store_n(S.stop, 1u, __ATOMIC_RELEASE);
    ASSERT_HIP(hipStreamSynchronize(S.stream), "sampler sync");
    const size_t stride = S.max_samples;
    const double khz = (double)S.wall_khz;   // ticks of s_memrealtime per millisecond
    std::vector<SamplerPair> host((size_t)SAMPLER_WAVES * stride);
    uint32_t count[SAMPLER_WAVES], xcc[SAMPLER_WAVES];
    ASSERT_HIP(hipMemcpy(host.data(), S.out, host.size() * sizeof(SamplerPair), hipMemcpyDeviceToHost), "sampler stamps");
    ASSERT_HIP(hipMemcpy(count, S.count, sizeof count, hipMemcpyDeviceToHost), "sampler counts");
    ASSERT_HIP(hipMemcpy(xcc, S.xcc, sizeof xcc, hipMemcpyDeviceToHost), "sampler xcc ids");
    ASSERT_HIP(hipStreamDestroy(S.stream), "sampler stream");
    dev_free(S.out);
    dev_free(S.count);
    dev_free(S.xcc);
    ASSERT_HIP(hipHostFree(S.stop), "sampler stop word");
    S = Sampler();

    std::vector<double> all, by_xcd[8];
    double span = 0.0;
    uint32_t dropped = 0;
    if (per_xcd_ghz8)
        for (int x = 0; x < 8; x++) per_xcd_ghz8[x] = 0.0;
    std::vector<double> first_wave;   // interval clocks of the longest-lived wave, in time order, for the profile
    for (int w = 0; w < SAMPLER_WAVES; w++) {
        const SamplerPair *p = host.data() + (size_t)w * stride;
        std::vector<double> mine;
        for (uint32_t i = 1; i < count[w] && i < stride; i++) {
            // an interval counts when both counters moved forward by a sane amount (under 5 GHz, under a second): a stamp
            // pair that straddles a counter hiccup is dropped and counted, not averaged in
            const uint64_t dr = p[i].ref - p[i - 1].ref, dc = p[i].cycles - p[i - 1].cycles;
            if (dr == 0 || dr > (uint64_t)(1000.0 * khz) || dc == 0 || (double)dc > 50.0 * (double)dr) {
                dropped++;
                continue;
            }
            mine.push_back((double)dc / (double)dr * khz * 1.0e-6);   // cycles per ms -> GHz
        }
        if (count[w] >= 2) span = std::max(span, (double)(p[count[w] - 1].ref - p[0].ref) / khz);
        all.insert(all.end(), mine.begin(), mine.end());
        auto &bucket = by_xcd[xcc[w] & 7u];
        bucket.insert(bucket.end(), mine.begin(), mine.end());
        if (mine.size() > first_wave.size()) first_wave = mine;
    }
    if (dropped_intervals) *dropped_intervals = dropped;
    if (all.empty()) {
        if (clock_ghz) *clock_ghz = 0.0;
        if (clock_ghz_min) *clock_ghz_min = 0.0;
        if (clock_ghz_max) *clock_ghz_max = 0.0;
        if (span_ms) *span_ms = span;
        return 0;
    }
    std::sort(all.begin(), all.end());
    if (clock_ghz) *clock_ghz = all[all.size() / 2];
    if (clock_ghz_min) *clock_ghz_min = all.front();
    if (clock_ghz_max) *clock_ghz_max = all.back();
    if (per_xcd_ghz8)
        for (int x = 0; x < 8; x++)
            if (!by_xcd[x].empty()) {
                std::sort(by_xcd[x].begin(), by_xcd[x].end());
                per_xcd_ghz8[x] = by_xcd[x][by_xcd[x].size() / 2];
            }
    if (profile10)
        for (int q = 0; q < 10; q++) {
            // mean clock of the q-th tenth of the sampled span, as one wave saw it
            const size_t lo = first_wave.size() * (size_t)q / 10, hi = std::max(lo + 1, first_wave.size() * (size_t)(q + 1) / 10);
            double sum = 0.0;
            size_t n = 0;
            for (size_t i = lo; i < hi && i < first_wave.size(); i++, n++) sum += first_wave[i];
            profile10[q] = n ? sum / (double)n : 0.0;
        }
    if (span_ms) *span_ms = span;
    return (int)all.size();
}

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
        host_src[2 * u] = 11.0f * (float)u;
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

    // The SIMD's arbiter favours its oldest wave, so the eight waves of a SIMD do not finish together: the oldest leaves
    // after about an eighth of the launch, the youngest spans all of it (measured: median lifetime 0.56 of the launch).
    // The waves that lived (nearly) as long as the longest one saw the whole loop: their clock is the launch's clock, and
    // the longest lifetime is the SIMDs' busy time, in which 8 waves issued iters x 16 wave-interactions each.
    uint64_t longest = 0;
    for (const ProbeStamp &s : host) longest = std::max(longest, s.cycles);
    std::vector<double> ghz, ghz_all;
    for (const ProbeStamp &s : host) {
        if (s.ref == 0) continue;
        const double g = (double)s.cycles / (double)s.ref * (double)wall_khz * 1.0e-6;
        ghz_all.push_back(g);
        if ((double)s.cycles >= 0.9 * (double)longest) ghz.push_back(g);
    }
    NB_ASSERT(!ghz.empty(), "clock probe: no wave reported a stamp");
    std::sort(ghz.begin(), ghz.end());
    std::sort(ghz_all.begin(), ghz_all.end());
    if (clock_ghz) *clock_ghz = ghz[ghz.size() / 2];
    if (clock_ghz_min) *clock_ghz_min = ghz_all.front();
    if (clock_ghz_max) *clock_ghz_max = ghz_all.back();
    if (cycles_per_wave_interaction) *cycles_per_wave_interaction = (double)longest / PROBE_WAVES_PER_SIMD / ((double)iters * 16.0);
    if (elapsed_ms) *elapsed_ms = (double)ms;
    return (int)ghz_all.size();
}
