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
 * Build: -mavx -ffp-contract=off (no FMA contraction), see csrc/Makefile.
 */
#include "sim_cpu.h"
#include "nb_util.h"

#include <immintrin.h>

#define V 8u /* floats per __m256 */

struct CpuSim {
    float *sx, *sy, *sm; /* snapshot, 32-byte aligned, zero-padded to a multiple of V */
    uint32_t capacity;   /* padded element count */
};

CpuSim *CpuSimCreate(uint32_t mass_len) {
    CpuSim *sim = NB_NEW(1, CpuSim);
    NB_CHECK(sim != NULL, "Failed to alloc CpuSim");
    sim->capacity = (mass_len + V - 1u) / V * V;
    size_t bytes = (size_t)(sim->capacity ? sim->capacity : V) * sizeof(float);
    sim->sx = (float *)aligned_alloc(32, bytes);
    sim->sy = (float *)aligned_alloc(32, bytes);
    sim->sm = (float *)aligned_alloc(32, bytes);
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

/* lanes 7..0 added onto 0: the reference's element order 0..7 in our lane numbering */
static inline float hsum_ref_order(__m256 v) {
    float lane[V];
    _mm256_storeu_ps(lane, v);
    float s = 0.0f;
    for (int e = (int)V - 1; e >= 0; e--) s += lane[e];
    return s;
}

static inline void euler(Particle *p, float ax, float ay, float dt) {
    p->acc = V2_FROM(ax, ay);
    p->vel = AddV2(p->vel, ScaleV2(p->acc, dt));
    p->pos = AddV2(p->pos, ScaleV2(p->vel, dt));
}

#define PAIR_TERM(X, Y, R, AX, AY)                                   \
    do {                                                             \
        __m256 dx = _mm256_sub_ps(px, X);                            \
        __m256 dy = _mm256_sub_ps(py, Y);                            \
        __m256 d2 = _mm256_add_ps(_mm256_mul_ps(dx, dx),             \
                                  _mm256_mul_ps(dy, dy));            \
        __m256 r2 = _mm256_add_ps(d2, R);                            \
        __m256 r3 = _mm256_mul_ps(_mm256_sqrt_ps(r2), r2);           \
        __m256 f = _mm256_div_ps(gm, r3);                            \
        AX = _mm256_add_ps(AX, _mm256_mul_ps(dx, f));                \
        AY = _mm256_add_ps(AY, _mm256_mul_ps(dy, f));                \
    } while (0)

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

    const __m256 g = _mm256_set1_ps(NB_G);
    const uint32_t pairs = total_len / 2u;
#pragma omp parallel for schedule(static, 16) if (go_parallel)
    for (uint32_t q = 0; q < pairs; q++) {
        Particle *a = &arr[2u * q], *b = a + 1;
        const __m256 xa = _mm256_set1_ps(a->pos.x), ya = _mm256_set1_ps(a->pos.y), ra = _mm256_set1_ps(a->radius);
        const __m256 xb = _mm256_set1_ps(b->pos.x), yb = _mm256_set1_ps(b->pos.y), rb = _mm256_set1_ps(b->radius);
        __m256 axa = _mm256_setzero_ps(), aya = axa, axb = axa, ayb = axa;
        for (uint32_t j = 0; j < padded; j += V) {
            const __m256 px = _mm256_load_ps(sx + j), py = _mm256_load_ps(sy + j);
            const __m256 gm = _mm256_mul_ps(_mm256_load_ps(sm + j), g);
            PAIR_TERM(xa, ya, ra, axa, aya);
            PAIR_TERM(xb, yb, rb, axb, ayb);
        }
        euler(a, hsum_ref_order(axa), hsum_ref_order(aya), dt);
        euler(b, hsum_ref_order(axb), hsum_ref_order(ayb), dt);
    }
    if (total_len & 1u) {
        Particle *a = &arr[total_len - 1u];
        const __m256 xa = _mm256_set1_ps(a->pos.x), ya = _mm256_set1_ps(a->pos.y), ra = _mm256_set1_ps(a->radius);
        __m256 axa = _mm256_setzero_ps(), aya = axa;
        for (uint32_t j = 0; j < padded; j += V) {
            const __m256 px = _mm256_load_ps(sx + j), py = _mm256_load_ps(sy + j);
            const __m256 gm = _mm256_mul_ps(_mm256_load_ps(sm + j), g);
            PAIR_TERM(xa, ya, ra, axa, aya);
        }
        euler(a, hsum_ref_order(axa), hsum_ref_order(aya), dt);
    }
}
