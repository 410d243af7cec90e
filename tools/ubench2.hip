// ubench2.hip -- how v_rsq_f32 (quarter rate) mixes with full-rate VALU on gfx950:
// does grouping the transcendentals amortise the switch cost, and do transcendental and plain VALU
// instructions of DIFFERENT waves on one SIMD overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)

// 32 fma + 8 rsq per body in every pattern; only the order differs.
#define FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n\t"
#define F4 FMA(0) FMA(1) FMA(2) FMA(3)
#define F8 F4 FMA(0) FMA(1) FMA(2) FMA(3)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c)

template <int PAT>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, r0 = seed + 4, r1 = seed + 5, r2 = seed + 6, r3 = seed + 7;
    float b = seed * 1.0001f, c = seed * 0.5f;
    const int wave = threadIdx.x >> 6;
    (void)wave;
    for (int it = 0; it < iters; it++) {
        if (PAT == 0) {  // (4 fma, 1 rsq) x 8
            asm volatile(F4 RSQ(4) F4 RSQ(5) F4 RSQ(6) F4 RSQ(7) F4 RSQ(4) F4 RSQ(5) F4 RSQ(6) F4 RSQ(7) OPS);
        } else if (PAT == 1) {  // (8 fma, 2 rsq) x 4
            asm volatile(F8 RSQ(4) RSQ(5) F8 RSQ(6) RSQ(7) F8 RSQ(4) RSQ(5) F8 RSQ(6) RSQ(7) OPS);
        } else if (PAT == 2) {  // (16 fma, 4 rsq) x 2
            asm volatile(F8 F8 RSQ(4) RSQ(5) RSQ(6) RSQ(7) F8 F8 RSQ(4) RSQ(5) RSQ(6) RSQ(7) OPS);
        } else if (PAT == 3) {  // 32 fma, 8 rsq
            asm volatile(F8 F8 F8 F8 RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) OPS);
        } else if (PAT == 4) {  // 32 fma only
            asm volatile(F8 F8 F8 F8 OPS);
        } else if (PAT == 5) {  // 8 rsq only
            asm volatile(RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) OPS);
        } else if (PAT == 6) {  // wave-specialised: even blocks do 64 fma, odd blocks do 16 rsq (same total per pair)
            if (blockIdx.x & 1) {
                asm volatile(RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) OPS);
            } else {
                asm volatile(F8 F8 F8 F8 F8 F8 F8 F8 OPS);
            }
        } else if (PAT == 7) {  // (2 fma, rsq, 2 fma) x 8 : rsq in the middle of short runs
            asm volatile(FMA(0) FMA(1) RSQ(4) FMA(2) FMA(3) FMA(0) FMA(1) RSQ(5) FMA(2) FMA(3) FMA(0) FMA(1) RSQ(6) FMA(2) FMA(3) FMA(0) FMA(1) RSQ(7) FMA(2) FMA(3)
                         FMA(0) FMA(1) RSQ(4) FMA(2) FMA(3) FMA(0) FMA(1) RSQ(5) FMA(2) FMA(3) FMA(0) FMA(1) RSQ(6) FMA(2) FMA(3) FMA(0) FMA(1) RSQ(7) FMA(2) FMA(3) OPS);
        } else if (PAT == 8) {  // 32 fma + 8 rsq where each rsq result feeds the next fma (dependency right after)
            asm volatile(F4 RSQ(4) "v_fma_f32 %0, %4, %9, %0\n\t" FMA(1) FMA(2) FMA(3) RSQ(5) "v_fma_f32 %1, %5, %9, %1\n\t" FMA(0) FMA(2) FMA(3) RSQ(6) "v_fma_f32 %2, %6, %9, %2\n\t" FMA(0) FMA(1) FMA(3) RSQ(7) "v_fma_f32 %3, %7, %9, %3\n\t" FMA(0) FMA(1) FMA(2)
                         F4 RSQ(4) "v_fma_f32 %0, %4, %9, %0\n\t" FMA(1) FMA(2) FMA(3) RSQ(5) "v_fma_f32 %1, %5, %9, %1\n\t" FMA(0) FMA(2) FMA(3) RSQ(6) "v_fma_f32 %2, %6, %9, %2\n\t" FMA(0) FMA(1) FMA(3) RSQ(7) "v_fma_f32 %3, %7, %9, %3\n\t" FMA(0) FMA(1) FMA(2) OPS);
        }
    }
    float s = a0 + a1 + a2 + a3 + r0 + r1 + r2 + r3;
    if (s == 12345.678f) out[0] = s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    float *out;
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000, cus = prop.multiProcessorCount;
    const char *names[] = {"(4fma,1rsq)x8", "(8fma,2rsq)x4", "(16fma,4rsq)x2", "32fma,8rsq", "32fma only", "8rsq only",
                           "waves split: 64fma | 16rsq", "(2fma,rsq,2fma)x8", "(4fma,rsq,dep fma..)x8 (36 fma)"};
    void (*fn[])(float *, int, float) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>};
    for (int pat = 0; pat < 9; pat++)
        for (int wps = 2; wps <= 8; wps *= 2) {
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(fn[pat], grid, block, 0, 0, out, 100, 1.5f);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(fn[pat], grid, block, 0, 0, out, iters, 1.5f);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            // cycles (at 2.4 GHz) each SIMD spends per body-iteration per wave
            double cyc = best * 1e-3 * 2.4e9 / ((double)iters * wps);
            printf("%-34s waves/SIMD %d  %8.3f ms  %7.1f cyc per body per wave (sum of parts: 32x2.38+8x8.17=141.5)\n", names[pat], wps, best, cyc);
        }
    return 0;
}
