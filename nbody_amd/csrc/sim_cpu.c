/*
 * sim_cpu.c -- the AVX + OpenMP step behind UpdateWorld_CPU.
 *
 * What it must reproduce: per receiver the
 * reference sums sources in 8 interleaved AVX lanes -- element e of its
 * accumulator takes sources j with j mod 8 == 7 - e (src/lib/sim_cpu.c:32-33) --
 * then adds elements 0..7 onto 0 (sim_cpu.c:146-154), with sqrt and div and no
 * FMA (built -mavx only).  Here the snapshot is plain SoA in natural order
 * (lane e = source 8g + e), so the final horizontal sum runs lanes 7..0; every
 * intermediate is the same IEEE operation on the same operands, hence the same bits.
 *
 * Receivers are processed two at a time against each loaded source vector,
 * which halves the load traffic of the reference's one-receiver loop without
 * touching the per-receiver arithmetic.
 *
 * SIMD_SET, like the reference's CMake option (CMakeLists.txt:4, src/lib/CMakeLists.txt:24-33): the default
 * build is AVX (8 lanes); -DNB_SIMD_SSE gives the reference's SSE build (4 lanes), -DNB_SIMD_NONE its scalar
 * build (1 lane = sources in index order).  Each is bit-exact with the corresponding reference build.
 * -DNB_SIMD_F64 (SIMD_SET=f64, no reference counterpart) is the "truth" mode: every term
 * and the sum in float64 from the same fp32 inputs, sources in index order, the sum rounded once to the fp32 acc; the
 * integrator stays the reference's fp32 mul-then-add.  It is the tie-breaker the fp32 tolerance is stated against.
 *
 * Informational variants (SURVEY.md section 8d "best CPU" row; NOT bit-exact with any reference build, never behind
 * UpdateWorld_CPU): -DNB_SIMD_AVX2FMA (8 lanes, built -mavx2 -mfma with contraction allowed: what -march=native makes
 * of the reference's AVX source), -DNB_SIMD_AVX512 (16 lanes, -mavx512f -mfma), and with -DNB_CPU_RSQRT on top of either
 * the sqrt + div pair replaced by the hardware reciprocal-sqrt estimate plus one Newton step (what a CPU-tuned build
 * would do).  They are compiled with -DNB_CPU_SUFFIX=<name>, which renames the three entry points, into
 * libnbody_cpu_best.so (cpu_best.c picks by __builtin_cpu_supports); bench.py times them beside the -mavx reference row.
 *
 * Build: -mavx -ffp-contract=off (no FMA contraction), see csrc/Makefile.
 */
#ifdef NB_CPU_SUFFIX
#define NB_CPU_CAT2(a, b) a##_##b
#define NB_CPU_CAT(a, b) NB_CPU_CAT2(a, b)
#define CpuSimCreate NB_CPU_CAT(CpuSimCreate, NB_CPU_SUFFIX)
#define CpuSimDestroy NB_CPU_CAT(CpuSimDestroy, NB_CPU_SUFFIX)
#define CpuSimStep NB_CPU_CAT(CpuSimStep, NB_CPU_SUFFIX)
#endif
#include "sim_cpu.h"
#include "nb_util.h"

#if defined(NB_SIMD_SSE)
#include <xmmintrin.h>
#define V 4u /* floats per vector */
typedef __m128 vf;
#define vf_set1 _mm_set1_ps
#define vf_zero _mm_setzero_ps
#define vf_load _mm_load_ps
#define vf_storeu _mm_storeu_ps
#define vf_add _mm_add_ps
#define vf_sub _mm_sub_ps
#define vf_mul _mm_mul_ps
#define vf_div _mm_div_ps
#define vf_sqrt _mm_sqrt_ps
#elif defined(NB_SIMD_F64)
#include <math.h>
#define V 1u
typedef double vf;
#define vf_set1(x) ((double)(x))
#define vf_zero() 0.0
#define vf_load(p) ((double)*(p))
#define vf_storeu(p, x) (*(p) = (float)(x))
#define vf_add(a, b) ((a) + (b))
#define vf_sub(a, b) ((a) - (b))
#define vf_mul(a, b) ((a) * (b))
#define vf_div(a, b) ((a) / (b))
#define vf_sqrt(a) sqrt(a)
#elif defined(NB_SIMD_NONE)
#include <math.h>
#define V 1u
typedef float vf;
#define vf_set1(x) (x)
#define vf_zero() 0.0f
#define vf_load(p) (*(p))
#define vf_storeu(p, x) (*(p) = (x))
#define vf_add(a, b) ((a) + (b))
#define vf_sub(a, b) ((a) - (b))
#define vf_mul(a, b) ((a) * (b))
#define vf_div(a, b) ((a) / (b))
#define vf_sqrt(a) sqrtf(a)
#elif defined(NB_SIMD_AVX512)
#include <immintrin.h>
#define V 16u
typedef __m512 vf;
#define vf_set1 _mm512_set1_ps
#define vf_zero _mm512_setzero_ps
#define vf_load _mm512_load_ps
#define vf_storeu _mm512_storeu_ps
#define vf_add _mm512_add_ps
#define vf_sub _mm512_sub_ps
#define vf_mul _mm512_mul_ps
#define vf_div _mm512_div_ps
#define vf_sqrt _mm512_sqrt_ps
#define vf_rsqrt _mm512_rsqrt14_ps /* relative error <= 2^-14; one Newton step brings it to ~6e-9 */
#else /* AVX (8 lanes): the reference's default build and the parity target; NB_SIMD_AVX2FMA = the same source, other flags */
#include <immintrin.h>
#define V 8u
typedef __m256 vf;
#define vf_set1 _mm256_set1_ps
#define vf_zero _mm256_setzero_ps
#define vf_load _mm256_load_ps
#define vf_storeu _mm256_storeu_ps
#define vf_add _mm256_add_ps
#define vf_sub _mm256_sub_ps
#define vf_mul _mm256_mul_ps
#define vf_div _mm256_div_ps
#define vf_sqrt _mm256_sqrt_ps
#define vf_rsqrt _mm256_rsqrt_ps /* relative error <= 1.5 * 2^-12; one Newton step brings it to ~2e-7 */
#endif

struct CpuSim {
    float *sx, *sy, *sm; /* snapshot, 64-byte aligned, zero-padded to a multiple of V */
    uint32_t capacity;   /* padded element count */
};

CpuSim *CpuSimCreate(uint32_t mass_len) {
    CpuSim *sim = NB_NEW(1, CpuSim);
    NB_CHECK(sim != NULL, "Failed to alloc CpuSim");
    sim->capacity = (mass_len + V - 1u) / V * V;
    size_t bytes = ((size_t)(sim->capacity ? sim->capacity : V) * sizeof(float) + 63u) / 64u * 64u;
    sim->sx = (float *)aligned_alloc(64, bytes);
    sim->sy = (float *)aligned_alloc(64, bytes);
    sim->sm = (float *)aligned_alloc(64, bytes);
    NB_CHECK(sim->sx && sim->sy && sim->sm, "Failed to alloc snapshot for %u sources", mass_len);
    return sim;
}

void CpuSimDestroy(CpuSim *sim) {
    if (sim == NULL) return;
    free(sim->sx);
    free(sim->sy);
    free(sim->sm);
    free(sim);
}

/* lanes V-1..0 added onto 0: the reference's element order 0..V-1 in our lane numbering */
static inline float hsum_ref_order(vf v) {
    float lane[V];
    vf_storeu(lane, v);
    float s = 0.0f;
    for (int e = (int)V - 1; e >= 0; e--) s += lane[e];
    return s;
}

static inline void euler(Particle *p, float ax, float ay, float dt) {
    p->acc = V2_FROM(ax, ay);
    p->vel = AddV2(p->vel, ScaleV2(p->acc, dt));
    p->pos = AddV2(p->pos, ScaleV2(p->vel, dt));
}

#ifdef NB_CPU_RSQRT
/* informational only: y ~ 1/sqrt(r2) by the hardware estimate, one Newton step y (1.5 - 0.5 r2 y^2), f = gm y^3 */
#define PAIR_TERM(X, Y, R, AX, AY)                                                               \
    do {                                                                                         \
        vf dx = vf_sub(px, X);                                                                   \
        vf dy = vf_sub(py, Y);                                                                   \
        vf r2 = vf_add(vf_add(vf_mul(dx, dx), vf_mul(dy, dy)), R);                               \
        vf y = vf_rsqrt(r2);                                                                     \
        y = vf_mul(y, vf_sub(vf_set1(1.5f), vf_mul(vf_mul(vf_set1(0.5f), r2), vf_mul(y, y))));   \
        vf f = vf_mul(gm, vf_mul(y, vf_mul(y, y)));                                              \
        AX = vf_add(AX, vf_mul(dx, f));                                                          \
        AY = vf_add(AY, vf_mul(dy, f));                                                          \
    } while (0)
#else
#define PAIR_TERM(X, Y, R, AX, AY)                          \
    do {                                                    \
        vf dx = vf_sub(px, X);                              \
        vf dy = vf_sub(py, Y);                              \
        vf d2 = vf_add(vf_mul(dx, dx), vf_mul(dy, dy));     \
        vf r2 = vf_add(d2, R);                              \
        vf r3 = vf_mul(vf_sqrt(r2), r2);                    \
        vf f = vf_div(gm, r3);                              \
        AX = vf_add(AX, vf_mul(dx, f));                     \
        AY = vf_add(AY, vf_mul(dy, f));                     \
    } while (0)
#endif

void CpuSimStep(CpuSim *sim, Particle *arr, uint32_t total_len, uint32_t mass_len, float dt) {
    const uint32_t padded = (mass_len + V - 1u) / V * V;
    NB_CHECK(padded <= sim->capacity, "snapshot holds %u sources, asked for %u", sim->capacity, mass_len);
    float *sx = sim->sx, *sy = sim->sy, *sm = sim->sm;

    /* below ~2*10^5 interactions a parallel region costs more than it saves (and far more on a box whose
     * OpenMP default oversubscribes its CPU quota); results do not depend on the thread count */
    const int go_parallel = (double)total_len * (double)(mass_len ? mass_len : 1) >= 2.0e5;

    /* snapshot = Jacobi semantics: every receiver sees the pre-step sources */
#pragma omp parallel for schedule(static, 1024) if (go_parallel)
    for (uint32_t j = 0; j < padded; j++) {
        const int live = j < mass_len;
        sx[j] = live ? arr[j].pos.x : 0.0f;
        sy[j] = live ? arr[j].pos.y : 0.0f;
        sm[j] = live ? arr[j].mass : 0.0f;
    }

    const vf g = vf_set1(NB_G);
    const uint32_t pairs = total_len / 2u;
#pragma omp parallel for schedule(static, 16) if (go_parallel)
    for (uint32_t q = 0; q < pairs; q++) {
        Particle *a = &arr[2u * q], *b = a + 1;
        const vf xa = vf_set1(a->pos.x), ya = vf_set1(a->pos.y), ra = vf_set1(a->radius);
        const vf xb = vf_set1(b->pos.x), yb = vf_set1(b->pos.y), rb = vf_set1(b->radius);
        vf axa = vf_zero(), aya = axa, axb = axa, ayb = axa;
        for (uint32_t j = 0; j < padded; j += V) {
            const vf px = vf_load(sx + j), py = vf_load(sy + j);
            const vf gm = vf_mul(vf_load(sm + j), g);
            PAIR_TERM(xa, ya, ra, axa, aya);
            PAIR_TERM(xb, yb, rb, axb, ayb);
        }
        euler(a, hsum_ref_order(axa), hsum_ref_order(aya), dt);
        euler(b, hsum_ref_order(axb), hsum_ref_order(ayb), dt);
    }
    if (total_len & 1u) {
        Particle *a = &arr[total_len - 1u];
        const vf xa = vf_set1(a->pos.x), ya = vf_set1(a->pos.y), ra = vf_set1(a->radius);
        vf axa = vf_zero(), aya = axa;
        for (uint32_t j = 0; j < padded; j += V) {
            const vf px = vf_load(sx + j), py = vf_load(sy + j);
            const vf gm = vf_mul(vf_load(sm + j), g);
            PAIR_TERM(xa, ya, ra, axa, aya);
        }
        euler(a, hsum_ref_order(axa), hsum_ref_order(aya), dt);
    }
}
