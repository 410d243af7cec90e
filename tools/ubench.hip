// ubench.hip -- VALU issue-rate microbenchmarks for gfx950 (MI355X).
// Answers, before the step kernel is designed around them:
//   * how many cycles a wave64 v_fma_f32 / v_pk_fma_f32 / v_rsq_f32 / v_sub with an SGPR operand costs,
//     at 1, 2, 4, 8 waves per SIMD;
//   * what the 9-VALU + 1-rsq interaction body can sustain.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8], b = seed * 1.0001f, c = seed * 0.5f;
    float2v p[8], pb = {b, b}, pc = {c, c};
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = float2v{a[i], a[i] + 1.f}; }
    float sg = __builtin_amdgcn_readfirstlane(seed * 3.0f);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == 0) {  // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 1) {  // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
                REP8(X)
#undef X
            } else if (KIND == 2) {  // v_rsq_f32
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (KIND == 3) {  // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                REP8(X)
#undef X
            } else if (KIND == 4) {  // v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                REP8(X)
#undef X
            } else if (KIND == 5) {  // v_sub_f32 with SGPR operand
#define X(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sg));
                REP8(X)
#undef X
            } else if (KIND == 6) {  // interaction body: 2 sub(sgpr) 2 fma 1 rsq 3 mul 2 fma, one per accumulator pair
#define X(i) { float dx, dy, d2, inv, i2, f; \
                asm volatile("v_sub_f32 %0, %6, %7\n\tv_sub_f32 %1, %6, %8\n\tv_fma_f32 %2, %0, %0, %9\n\tv_fma_f32 %2, %1, %1, %2\n\t" \
                             "v_rsq_f32 %3, %2\n\tv_mul_f32 %4, %3, %3\n\tv_mul_f32 %5, %6, %3\n\tv_mul_f32 %5, %5, %4\n\t" \
                             : "=&v"(dx), "=&v"(dy), "=&v"(d2), "=&v"(inv), "=&v"(i2), "=&v"(f) : "s"(sg), "v"(b), "v"(c), "v"(a[i & 3])); \
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i].x) : "v"(dx), "v"(f)); \
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i].y) : "v"(dy), "v"(f)); }
                REP8(X)
#undef X
            } else if (KIND == 7) {  // v_mul_f32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (KIND == 8) {  // fma + rsq interleaved 4:1 to see whether rsq overlaps other VALU
#define X(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_rsq_f32 %4, %4" : "+v"(p[i].x), "+v"(p[i].y) : "v"(b), "v"(c), "v"(a[i]));
                REP8(X)
#undef X
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;  // keep live
}

struct Kind { const char *name; int instr_per_rep; int slots; };

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  arch %s  CUs %d  clock %d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    float *out;
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000;
    // instructions per inner u-step (8 reps): KIND 6 has 10 instr per rep, KIND 8 has 5
    Kind kinds[] = {{"v_fma_f32", 1, 0}, {"v_pk_fma_f32", 1, 0}, {"v_rsq_f32", 1, 0}, {"v_pk_mul_f32", 1, 0}, {"v_pk_add_f32", 1, 0},
                    {"v_sub_f32 sgpr", 1, 0}, {"interaction(10 instr)", 10, 0}, {"v_mul_f32", 1, 0}, {"4fma+1rsq", 5, 0}};
    void (*fn[])(float *, int, float) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>};
    const int cus = prop.multiProcessorCount;
    for (int kind = 0; kind < 9; kind++) {
        for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads per CU
            dim3 grid(cus * wps), block(256);
            hipLaunchKernelGGL(fn[kind], grid, block, 0, 0, out, 100, 1.5f);  // warm
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(fn[kind], grid, block, 0, 0, out, iters, 1.5f);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double winstr = (double)iters * 4 * 8 * kinds[kind].instr_per_rep;            // wave-instructions per wave
            double total_wave_instr = winstr * (double)cus * wps * 4;                      // 4 waves per block
            double per_simd_per_s = total_wave_instr / (cus * 4.0) / (best * 1e-3);        // wave-instr per SIMD per second
            double cyc_at_2400 = 2.4e9 / per_simd_per_s;                                    // cycles per wave-instr per SIMD at 2.4 GHz
            printf("%-24s waves/SIMD %d  %8.3f ms  %.3e lane-instr/s  %.2f cyc/wave-instr/SIMD@2.4GHz\n", kinds[kind].name, wps, best,
                   total_wave_instr * 64 / (best * 1e-3), cyc_at_2400);
        }
    }
    return 0;
}
