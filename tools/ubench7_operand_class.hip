// ubench7_operand_class.hip -- what does the step kernel's interaction body cost per wave-interaction when the source
// (x, y, G*m) reaches the VALU as (a) SGPR operands (the scalar-cache route), (b) plain VGPR operands (the LDS route's
// broadcast ds_read_b128 results), (c) DPP row_newbcast operands (each lane holds one of 16 sources; the instruction
// itself picks lane n of the row -- no broadcast load, no per-source register writes)?  No memory traffic at all:
// this isolates the operand class.  K = 2 paired-rsq body of kernels.hip (NB_INTERACTION2_ASM), 16 sources per
// iteration, 8 and 4 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); abort(); } } while (0)

// SX/SY/SG: operand text of the source; SUB/MUL: mnemonic (plain or _dpp); CTL: DPP control suffix
#define BODY(SUB, MUL, SX, SY, SG, CTL)                         \
    SUB " v30, " SX ", %[px0]" CTL "\n\t"                       \
    SUB " v31, " SY ", %[py0]" CTL "\n\t"                       \
    "v_fma_f32 v33, v30, v30, %[r0]\n\t"                        \
    "v_fmac_f32 v33, v31, v31\n\t"                              \
    SUB " v38, " SX ", %[px1]" CTL "\n\t"                       \
    SUB " v39, " SY ", %[py1]" CTL "\n\t"                       \
    "v_fma_f32 v40, v38, v38, %[r1]\n\t"                        \
    "v_fmac_f32 v40, v39, v39\n\t"                              \
    "s_setprio 3\n\t"                                           \
    "v_rsq_f32 v33, v33\n\t"                                    \
    "v_rsq_f32 v40, v40\n\t"                                    \
    "s_setprio 0\n\t"                                           \
    MUL " v36, " SG ", v33" CTL "\n\t"                          \
    "v_mul_f32 v32, v33, v33\n\t"                               \
    "v_mul_f32 v36, v36, v32\n\t"                               \
    "v_fmac_f32 %[ax0], v30, v36\n\t"                           \
    "v_fmac_f32 %[ay0], v31, v36\n\t"                           \
    MUL " v36, " SG ", v40" CTL "\n\t"                          \
    "v_mul_f32 v32, v40, v40\n\t"                               \
    "v_mul_f32 v36, v36, v32\n\t"                               \
    "v_fmac_f32 %[ax1], v38, v36\n\t"                           \
    "v_fmac_f32 %[ay1], v39, v36"
#define CLOB "v30", "v31", "v32", "v33", "v36", "v38", "v39", "v40"
#define OUTS [ax0] "+v"(ax0), [ay0] "+v"(ay0), [ax1] "+v"(ax1), [ay1] "+v"(ay1)
#define RECV [px0] "v"(px0), [py0] "v"(py0), [r0] "v"(r0), [px1] "v"(px1), [py1] "v"(py1), [r1] "v"(r1)
#define DPP(n) " row_newbcast:" #n " row_mask:0xf bank_mask:0xf"

template <int MODE>
__global__ __launch_bounds__(1024, 8) void k(float *out, int iters, float sxs, float sys, float sgs) {
    const float t = (float)threadIdx.x;
    float px0 = t, py0 = t * 0.5f, r0 = 1.0f + t, px1 = t + 7.0f, py1 = t * 0.25f, r1 = 2.0f + t;
    float ax0 = 0, ay0 = 0, ax1 = 0, ay1 = 0;
    __shared__ __attribute__((aligned(16))) float tile[16][192];
    if (MODE == 3) {
        for (int i = threadIdx.x & 63; i < 192; i += 64) tile[threadIdx.x >> 6][i] = sxs + (float)i;
        __syncthreads();
    }
    float vx = sxs + t * 1e-3f, vy = sys + t * 2e-3f, vg = sgs + t * 3e-3f;   // per-lane sources (modes 1, 2)
    for (int it = 0; it < iters; it++) {
#define ONE_S asm(BODY("v_sub_f32", "v_mul_f32", "%[sx]", "%[sy]", "%[sg]", "") : OUTS : [sx] "s"(sxs), [sy] "s"(sys), [sg] "s"(sgs), RECV : CLOB);
#define ONE_V asm(BODY("v_sub_f32", "v_mul_f32", "%[sx]", "%[sy]", "%[sg]", "") : OUTS : [sx] "v"(vx), [sy] "v"(vy), [sg] "v"(vg), RECV : CLOB);
#define ONE_D(n) asm(BODY("v_sub_f32_dpp", "v_mul_f32_dpp", "%[sx]", "%[sy]", "%[sg]", DPP(n)) : OUTS : [sx] "v"(vx), [sy] "v"(vy), [sg] "v"(vg), RECV : CLOB);
        if (MODE == 0) { ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S ONE_S }
        if (MODE == 1) { ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V ONE_V }
        if (MODE == 2) { ONE_D(0) ONE_D(1) ONE_D(2) ONE_D(3) ONE_D(4) ONE_D(5) ONE_D(6) ONE_D(7) ONE_D(8) ONE_D(9) ONE_D(10) ONE_D(11) ONE_D(12) ONE_D(13) ONE_D(14) ONE_D(15) }
        if (MODE == 3) {
            // the LDS route's inner loop: per 4 sources three broadcast ds_read_b128 (8 position floats, 4 G*m), then the
            // 4 x 2 interactions on those VGPRs -- what do the reads' 12 wave-wide register writes per 8 interactions cost?
            typedef float v8f __attribute__((ext_vector_type(8)));
            typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int jj = 0; jj < 16; jj += 4) {
                const v8f P = *reinterpret_cast<const v8f *>(&tile[threadIdx.x >> 6][2 * (jj + 16 * (it & 3))]);
                const v4f G = *reinterpret_cast<const v4f *>(&tile[threadIdx.x >> 6][128 + jj + 16 * (it & 3)]);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    asm(BODY("v_sub_f32", "v_mul_f32", "%[sx]", "%[sy]", "%[sg]", "") : OUTS : [sx] "v"(P[2 * u]), [sy] "v"(P[2 * u + 1]), [sg] "v"(G[u]), RECV : CLOB);
            }
        }
        // keep the loop from being hoisted: the sources drift
        sxs += 1e-6f; vx += 1e-6f;
    }
    const float s = ax0 + ay0 + ax1 + ay1;
    if (s == 12345.678f) out[0] = s;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, cus = prop.multiProcessorCount;
    const char *names[] = {"SGPR operands (scalar-cache route)", "VGPR operands, no loads", "DPP row_newbcast operands",
                           "VGPR operands + 3 ds_read_b128 / 4 src"};
    void (*fn[])(float *, int, float, float, float) = {k<0>, k<1>, k<2>, k<3>};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 4; mode++)
            for (int wg = 2; wg >= 1; wg--) {   // 1024-thread workgroups per CU: 2 -> 8 waves per SIMD, 1 -> 4
                dim3 grid(cus * wg), block(1024);
                hipLaunchKernelGGL(fn[mode], grid, block, 0, 0, out, 200, 1.5f, 2.5f, 3.5f); CK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int r = 0; r < 3; r++) {
                    CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(fn[mode], grid, block, 0, 0, out, iters, 1.5f, 2.5f, 3.5f); CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
                }
                // wave-interactions per SIMD = waves per SIMD x iters x 16 sources x 2 receivers
                const double wi = (double)(4 * wg) * iters * 32.0;
                printf("%-40s waves/SIMD %d  %8.3f ms  %6.2f cycles-at-2.4GHz per wave-interaction (floor of the mix: 26)\n", names[mode], 4 * wg, best,
                       best * 1e-3 * 2.4e9 / wi);
            }
    return 0;
}
