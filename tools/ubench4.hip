// ubench4.hip -- true per-SIMD cycle costs (s_memtime, independent of DVFS) and the clock each pattern holds.
// One workgroup of 256 threads per CU x wps; every wave stamps s_memtime / s_memrealtime around its loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)

#define FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define PKF(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n\t"
#define F4 FMA(0) FMA(1) FMA(2) FMA(3)
#define F8 F4 FMA(0) FMA(1) FMA(2) FMA(3)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c)

template <int PAT>
__global__ __launch_bounds__(256) void k(unsigned long long *stamps, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, r0 = seed + 4, r1 = seed + 5, r2 = seed + 6, r3 = seed + 7;
    float b = seed * 1.0001f, c = seed * 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {r0, r1}, p3 = {r2, r3}, pb = {b, b}, pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (PAT == 0) asm volatile(F8 F8 F8 F8 OPS);                                            // 32 fma
        if (PAT == 1) asm volatile(RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) OPS);  // 8 rsq
        if (PAT == 2) asm volatile(F4 RSQ(4) F4 RSQ(5) F4 RSQ(6) F4 RSQ(7) F4 RSQ(4) F4 RSQ(5) F4 RSQ(6) F4 RSQ(7) OPS);  // (4 fma,1 rsq)x8
        if (PAT == 3) asm volatile(F8 F8 F8 F8 RSQ(4) RSQ(5) RSQ(6) RSQ(7) RSQ(4) RSQ(5) RSQ(6) RSQ(7) OPS);             // 32 fma, 8 rsq
        if (PAT == 4) asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                                   "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                                   "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                                   "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));  // 16 pk_fma
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + r0 + r1 + r2 + r3 + p0.x + p1.x + p2.x + p3.x;
    if (s == 12345.678f) stamps[0] = 1;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = w1 - w0;
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 20000;
    unsigned long long *d;
    CK(hipMalloc(&d, sizeof(unsigned long long) * 2 * cus * 8 * 4));
    const char *names[] = {"32 fma", "8 rsq", "(4fma,1rsq)x8", "32fma,8rsq", "16 pk_fma"};
    const double instr[] = {32, 8, 40, 40, 16};
    void (*fn[])(unsigned long long *, int, float) = {k<0>, k<1>, k<2>, k<3>, k<4>};
    for (int pat = 0; pat < 5; pat++)
        for (int wps = 4; wps <= 8; wps *= 2) {
            const int waves = cus * wps * 4;
            hipLaunchKernelGGL(fn[pat], dim3(cus * wps), dim3(256), 0, 0, d, 200, 1.5f);
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(fn[pat], dim3(cus * wps), dim3(256), 0, 0, d, iters, 1.5f);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(2 * waves);
            CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
            std::vector<double> cyc(waves), clk(waves);
            for (int w = 0; w < waves; w++) { cyc[w] = (double)h[2 * w]; clk[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10e-9) / 1e9; }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double med = cyc[waves / 2];
            // wps waves share a SIMD for the whole run: SIMD cycles per wave-body = elapsed / (iters * wps)
            printf("%-16s waves/SIMD %d  median wave time %.0f cyc  -> %.2f SIMD cycles per instruction, shader clock %.3f GHz (median)\n",
                   names[pat], wps, med, med / (iters * (double)wps * instr[pat]), clk[waves / 2]);
        }
    return 0;
}
