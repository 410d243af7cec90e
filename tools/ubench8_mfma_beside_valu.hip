// ubench8_mfma_beside_valu.hip -- feasibility probe for the one idea that could beat the 26-cycle instruction mix:
// let the matrix pipe compute d^2 = |p_j - p_i|^2 + r_i for tile pairs far enough apart that the expanded form
// (|p_i|^2 + r_i) + |p_j|^2 - 2 p_i . p_j does not cancel (a rank-4 product: two v_mfma_f32_32x32x2_f32 per 1024 pairs),
// leaving the VALU 2 sub + rsq + 3 mul + 2 fmac = 22 cycles per wave-interaction instead of 26.  The question here is only
// the hardware one: does an f32 MFMA issued beside the interaction body cost the VALU anything?
//   mode 0  the shipped single-receiver body                              (26 cycles nominal)
//   mode 1  the body without its two FMAs, q read from a register          (22 cycles nominal, no MFMA: the ceiling)
//   mode 2  mode 1 + two v_mfma_f32_32x32x2_f32 per 16 interactions, whose 16 result registers ARE the q's
//   mode 3  mode 0 + the same two MFMAs per 16 interactions (pure interference measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));

#define FULL_BODY                                 \
    "v_sub_f32 v30, %[sx], %[px]\n\t"             \
    "v_sub_f32 v31, %[sy], %[py]\n\t"             \
    "v_fma_f32 v33, v30, v30, %[r]\n\t"           \
    "v_fmac_f32 v33, v31, v31\n\t"                \
    "s_setprio 3\n\t"                             \
    "v_rsq_f32 v33, v33\n\t"                      \
    "s_setprio 0\n\t"                             \
    "v_mul_f32 v36, %[g], v33\n\t"                \
    "v_mul_f32 v32, v33, v33\n\t"                 \
    "v_mul_f32 v36, v36, v32\n\t"                 \
    "v_fmac_f32 %[ax], v30, v36\n\t"              \
    "v_fmac_f32 %[ay], v31, v36"
#define SHORT_BODY                                \
    "v_sub_f32 v30, %[sx], %[px]\n\t"             \
    "v_sub_f32 v31, %[sy], %[py]\n\t"             \
    "s_setprio 3\n\t"                             \
    "v_rsq_f32 v33, %[q]\n\t"                     \
    "s_setprio 0\n\t"                             \
    "v_mul_f32 v36, %[g], v33\n\t"                \
    "v_mul_f32 v32, v33, v33\n\t"                 \
    "v_mul_f32 v36, v36, v32\n\t"                 \
    "v_fmac_f32 %[ax], v30, v36\n\t"              \
    "v_fmac_f32 %[ay], v31, v36"
#define CLOB "v30", "v31", "v32", "v33", "v36"

template <int MODE>
__global__ __launch_bounds__(1024, 8) void k(float *out, int iters, float sxs, float sys, float sgs) {
    const float t = (float)threadIdx.x;
    float px = t, py = t * 0.5f, r = 1.0f + t, ax = 0, ay = 0;
    v16f q;
    for (int i = 0; i < 16; i++) q[i] = 1.0f + t + (float)i;
    float ma = t * 1e-3f, mb = 2.0f - t * 1e-3f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 2 || MODE == 3) {
            // k = 4 as two k = 2 instructions chained through the accumulator: 128 matrix-pipe cycles per SIMD
            v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc, 0, 0, 0);
            if (MODE == 2) q = acc;
            else ax += acc[0] * 1e-30f;   // keep it alive
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0 || MODE == 3)
                asm(FULL_BODY : [ax] "+v"(ax), [ay] "+v"(ay) : [sx] "s"(sxs), [sy] "s"(sys), [g] "s"(sgs), [px] "v"(px), [py] "v"(py), [r] "v"(r) : CLOB);
            else
                asm(SHORT_BODY : [ax] "+v"(ax), [ay] "+v"(ay) : [sx] "s"(sxs), [sy] "s"(sys), [g] "s"(sgs), [px] "v"(px), [py] "v"(py), [q] "v"(q[u]) : CLOB);
        }
        sxs += 1e-6f; ma += 1e-7f;
    }
    const float s = ax + ay;
    if (s == 12345.678f) out[0] = s;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 8000, cus = prop.multiProcessorCount;
    const char *names[] = {"shipped body (26 nominal)", "body without its 2 FMAs (22 nominal)", "22-cycle body + 2 MFMA 32x32x2 f32 per 16 (q from the matrix pipe)",
                           "shipped body + the same 2 MFMAs (interference only)"};
    void (*fn[])(float *, int, float, float, float) = {k<0>, k<1>, k<2>, k<3>};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 4; mode++)
            for (int wg = 2; wg >= 1; wg--) {
                dim3 grid(cus * wg), block(1024);
                hipLaunchKernelGGL(fn[mode], grid, block, 0, 0, out, 200, 1.5f, 2.5f, 3.5f); CK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int r = 0; r < 3; r++) {
                    CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(fn[mode], grid, block, 0, 0, out, iters, 1.5f, 2.5f, 3.5f); CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                }
                const double wi = (double)(4 * wg) * iters * 16.0;   // wave-interactions per SIMD
                printf("%-72s waves/SIMD %d  %8.3f ms  %6.2f cycles-at-2.4GHz per wave-interaction\n", names[mode], 4 * wg, best, best * 1e-3 * 2.4e9 / wi);
            }
    return 0;
}
