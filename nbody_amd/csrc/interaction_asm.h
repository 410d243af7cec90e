// interaction_asm.h -- the gfx950 instruction sequence of ONE pairwise interaction, as text for an asm statement.
//
// Shared by the product kernels (kernels.hip, where the comments explain every choice) and the clock probe
// (clock_probe.hip), which must issue exactly the instruction mix the step kernels issue.  Operands: %[sx] %[sy] %[g] the
// source (SGPRs or wave-uniform VGPRs), %[px] %[py] %[r] the receiver, %[ax] %[ay] its running sums.
#pragma once

#define NB_INTERACTION_ASM                                  \
    "v_sub_f32 v30, %[sx], %[px]\n\t"                       \
    "v_sub_f32 v31, %[sy], %[py]\n\t"                       \
    "v_fma_f32 v33, v30, v30, %[r]\n\t"                     \
    "v_fmac_f32 v33, v31, v31\n\t"                          \
    "s_setprio 3\n\t"                                       \
    "v_rsq_f32 v33, v33\n\t"                                \
    "s_setprio 0\n\t"                                       \
    "v_mul_f32 v36, %[g], v33\n\t"                          \
    "v_mul_f32 v32, v33, v33\n\t"                           \
    "v_mul_f32 v36, v36, v32\n\t"                           \
    "v_fmac_f32 %[ax], v30, v36\n\t"                        \
    "v_fmac_f32 %[ay], v31, v36"
#define NB_CLOBBERS "v30", "v31", "v32", "v33", "v36", "v37"

// K = 2: the two receivers of a lane against one source as ONE statement -- both heads, both v_rsq_f32 back to back
// inside one priority window, then the tails in receiver order (the sums see the same operands in the same order as
// two single statements).  1 % faster than two single statements (48.0 vs 48.5 ms per launch at N = 2^20); the heads
// alone interleaved: 0.7 %; four or eight rsq per window: 0.8 % / 0.5 % (profiles/r02_ab_plain_body.txt, second part).
// With round 1's packed body the paired form was 5-7 % SLOWER: transcendentals only like company once nothing packed
// sits next to them.  Extra temporaries: dx', dy', q' = v38, v39, v40.
#define NB_INTERACTION2_ASM                                 \
    "v_sub_f32 v30, %[sx], %[px0]\n\t"                      \
    "v_sub_f32 v31, %[sy], %[py0]\n\t"                      \
    "v_fma_f32 v33, v30, v30, %[r0]\n\t"                    \
    "v_fmac_f32 v33, v31, v31\n\t"                          \
    "v_sub_f32 v38, %[sx], %[px1]\n\t"                      \
    "v_sub_f32 v39, %[sy], %[py1]\n\t"                      \
    "v_fma_f32 v40, v38, v38, %[r1]\n\t"                    \
    "v_fmac_f32 v40, v39, v39\n\t"                          \
    "s_setprio 3\n\t"                                       \
    "v_rsq_f32 v33, v33\n\t"                                \
    "v_rsq_f32 v40, v40\n\t"                                \
    "s_setprio 0\n\t"                                       \
    "v_mul_f32 v36, %[g], v33\n\t"                          \
    "v_mul_f32 v32, v33, v33\n\t"                           \
    "v_mul_f32 v36, v36, v32\n\t"                           \
    "v_fmac_f32 %[ax0], v30, v36\n\t"                       \
    "v_fmac_f32 %[ay0], v31, v36\n\t"                       \
    "v_mul_f32 v36, %[g], v40\n\t"                          \
    "v_mul_f32 v32, v40, v40\n\t"                           \
    "v_mul_f32 v36, v36, v32\n\t"                           \
    "v_fmac_f32 %[ax1], v38, v36\n\t"                       \
    "v_fmac_f32 %[ay1], v39, v36"
#define NB_CLOBBERS2 "v30", "v31", "v32", "v33", "v36", "v37", "v38", "v39", "v40"
