/*
 * cpu_best.c -- libnbody_cpu_best.so: the informational "best CPU" row of the measurement (the
 * reference's whole SIMD matrix is src/lib/CMakeLists.txt:24-33, its CPU kernel src/lib/sim_cpu.c:24-44,156-194).
 *
 * UpdateWorld_CPU stays the -mavx build whose bits equal the reference's AVX build; that row is the stated baseline.
 * This library holds the SAME source (sim_cpu.c) built for what the host cores can actually do -- FMA contraction,
 * 16 lanes, reciprocal-sqrt estimate + Newton -- none of it bit-exact with any reference build, all of it within the
 * stated fp32 tolerance of the float64 sum (tests/test_world_cpu.py).  A variant is offered only when
 * __builtin_cpu_supports says this CPU runs it.  Tooling for bench.py's cpu_baseline.best_cpu; nothing in the product
 * path links or loads it.
 */
#include <stdint.h>
#include <string.h>

#include <nbody.h>

typedef struct CpuSim CpuSim;
#define DECLARE(S)                                     \
    CpuSim *CpuSimCreate_##S(uint32_t mass_len);       \
    void CpuSimDestroy_##S(CpuSim *sim);               \
    void CpuSimStep_##S(CpuSim *sim, Particle *arr, uint32_t total_len, uint32_t mass_len, float dt);
DECLARE(avx2fma)
DECLARE(avx2fma_rsqrt)
DECLARE(avx512)
DECLARE(avx512_rsqrt)

typedef struct Variant {
    const char *isa, *what;
    CpuSim *(*create)(uint32_t);
    void (*destroy)(CpuSim *);
    void (*step)(CpuSim *, Particle *, uint32_t, uint32_t, float);
} Variant;

#define ROW(S, WHAT) {#S, WHAT, CpuSimCreate_##S, CpuSimDestroy_##S, CpuSimStep_##S}
static const Variant VARIANTS[] = {
    ROW(avx2fma, "8 lanes, -mavx2 -mfma, mul+add contracted to FMA, sqrt + div"),
    ROW(avx2fma_rsqrt, "8 lanes, -mavx2 -mfma, vrsqrtps + one Newton step instead of sqrt + div"),
    ROW(avx512, "16 lanes, -mavx512f -mfma, mul+add contracted to FMA, sqrt + div"),
    ROW(avx512_rsqrt, "16 lanes, -mavx512f -mfma, vrsqrt14ps + one Newton step instead of sqrt + div"),
};
enum { N_VARIANTS = sizeof VARIANTS / sizeof VARIANTS[0] };

static int runs_here(int i) {
    __builtin_cpu_init();
    if (i < 2) return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma");
}

int nb_cpu_variant_count(void) { return N_VARIANTS; }

/* name of variant i ("avx2fma", ...), or NULL; *supported = whether this CPU runs it; *what = one line of description */
const char *nb_cpu_variant_name(int i, int *supported, const char **what) {
    if (i < 0 || i >= N_VARIANTS) return NULL;
    if (supported) *supported = runs_here(i);
    if (what) *what = VARIANTS[i].what;
    return VARIANTS[i].isa;
}

/* n Jacobi steps of arr[0..total_len) (partitioned: sources first) with variant `isa`, OpenMP threads as set by the caller.
 * Returns 0, or -1 for an unknown or unsupported variant (nothing is run then). */
int nb_cpu_variant_update(const char *isa, Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n) {
    for (int i = 0; i < N_VARIANTS; i++) {
        if (strcmp(isa, VARIANTS[i].isa) != 0) continue;
        if (!runs_here(i)) return -1;
        CpuSim *sim = VARIANTS[i].create(mass_len);
        for (uint32_t s = 0; s < n; s++) VARIANTS[i].step(sim, arr, total_len, mass_len, dt);
        VARIANTS[i].destroy(sim);
        return 0;
    }
    return -1;
}
