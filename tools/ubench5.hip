// ubench5.hip -- does raising the wave priority around a run of v_rsq_f32 remove the transcendental<->plain
// VALU switch penalty (other waves of the SIMD stop interleaving plain VALU between the rsqs)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)
#define FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n\t"
#define F4 FMA(0) FMA(1) FMA(2) FMA(3)
#define F8 F4 FMA(0) FMA(1) FMA(2) FMA(3)
#define R4 RSQ(4) RSQ(5) RSQ(6) RSQ(7)
#define HI "s_setprio 3\n\t"
#define LO "s_setprio 0\n\t"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c)

template <int PAT>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, r0 = seed + 4, r1 = seed + 5, r2 = seed + 6, r3 = seed + 7;
    float b = seed * 1.0001f, c = seed * 0.5f;
    for (int it = 0; it < iters; it++) {
        if (PAT == 0) asm volatile(F8 F8 F8 F8 R4 R4 OPS);                      // 32 fma, 8 rsq
        if (PAT == 1) asm volatile(F8 F8 F8 F8 HI R4 R4 LO OPS);                // same, prio 3 around the rsq run
        if (PAT == 2) asm volatile(F8 F8 R4 F8 F8 R4 OPS);                      // (16 fma, 4 rsq) x 2
        if (PAT == 3) asm volatile(F8 F8 HI R4 LO F8 F8 HI R4 LO OPS);          // same with prio
        if (PAT == 4) asm volatile(F8 F8 F8 F8 F8 F8 F8 F8 R4 R4 R4 R4 OPS);    // 64 fma, 16 rsq (body x2)
        if (PAT == 5) asm volatile(F8 F8 F8 F8 F8 F8 F8 F8 HI R4 R4 R4 R4 LO OPS);
        if (PAT == 6) asm volatile(HI F8 F8 F8 F8 LO R4 R4 OPS);                // prio around the fma run instead
        if (PAT == 7) asm volatile(F4 RSQ(4) F4 RSQ(5) F4 RSQ(6) F4 RSQ(7) F4 RSQ(4) F4 RSQ(5) F4 RSQ(6) F4 RSQ(7) OPS);
        if (PAT == 8) asm volatile(F4 HI RSQ(4) LO F4 HI RSQ(5) LO F4 HI RSQ(6) LO F4 HI RSQ(7) LO F4 HI RSQ(4) LO F4 HI RSQ(5) LO F4 HI RSQ(6) LO F4 HI RSQ(7) LO OPS);
    }
    float s = a0 + a1 + a2 + a3 + r0 + r1 + r2 + r3;
    if (s == 12345.678f) out[0] = s;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, cus = prop.multiProcessorCount;
    const char *names[] = {"32fma,8rsq", "32fma,[8rsq]prio", "(16fma,4rsq)x2", "(16fma,[4rsq]prio)x2", "64fma,16rsq", "64fma,[16rsq]prio",
                           "[32fma]prio,8rsq", "(4fma,1rsq)x8", "(4fma,[1rsq]prio)x8"};
    const double bodies[] = {1, 1, 1, 1, 2, 2, 1, 1, 1};
    void (*fn[])(float *, int, float) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>};
    for (int pat = 0; pat < 9; pat++)
        for (int wps = 4; wps <= 8; wps *= 2) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(fn[pat], grid, block, 0, 0, out, 100, 1.5f); CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(fn[pat], grid, block, 0, 0, out, iters, 1.5f); CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            printf("%-22s waves/SIMD %d  %8.3f ms  %7.1f ns-at-2.4GHz-cycles per (32 fma + 8 rsq) body (nominal 128)\n", names[pat], wps, best,
                   best * 1e-3 * 2.4e9 / ((double)iters * wps * bodies[pat]));
        }
    return 0;
}
