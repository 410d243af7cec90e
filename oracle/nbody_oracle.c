/*
 * nbody_oracle.c -- TEST INFRASTRUCTURE ONLY (see nbody_oracle.h).
 *
 * CPU restatement of the reference hot path.  Build with
 *     gcc -O2 -mavx -ffp-contract=off -fopenmp
 * (the reference builds its AVX variant with -mavx and no FMA,
 * reference src/lib/CMakeLists.txt:24-30; contraction would change result bits).
 *
 * Parity status: PINNED against the compiled reference (tests/golden/) and the
 * SURVEY.md section 8c digests; see tests/test_oracle.py.
 */
#include "nbody_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES 8u

/* ------------------------------------------------------------------ */
/* partition: reference src/lib/world.c:32-46                          */
/* ------------------------------------------------------------------ */

uint32_t orc_partition(Particle *arr, uint32_t size) {
    uint32_t lo = 0, hi = size;
    for (;;) {
        /* lo walks up to the first massless body */
        while (lo < hi && arr[lo].mass > 0) lo++;
        /* hi walks down to the last massive body (pre-decrement, as the reference) */
        while (lo < hi) {
            hi--;
            if (!(arr[hi].mass <= 0)) break;
        }
        if (lo == hi) break;
        Particle t = arr[lo];
        arr[lo] = arr[hi];
        arr[hi] = t;
    }
    return hi;
}

uint32_t orc_partition_ints(int *arr, uint32_t size) {
    uint32_t lo = 0, hi = size;
    for (;;) {
        while (lo < hi && arr[lo] != 0) lo++;
        while (lo < hi) {
            hi--;
            if (!(arr[hi] == 0)) break;
        }
        if (lo == hi) break;
        int t = arr[lo];
        arr[lo] = arr[hi];
        arr[hi] = t;
    }
    return hi;
}

/* ------------------------------------------------------------------ */
/* source snapshot: reference sim_cpu.c:125-143 (PackParticles)        */
/* zero-padded to a multiple of 8 like the reference's tail pack       */
/* ------------------------------------------------------------------ */

typedef struct Snapshot {
    float *x, *y, *m;
    uint32_t padded;
} Snapshot;

static Snapshot snapshot_alloc(uint32_t mass_len) {
    Snapshot s;
    s.padded = (mass_len + LANES - 1u) / LANES * LANES;
    size_t bytes = (size_t)(s.padded ? s.padded : LANES) * sizeof(float);
    s.x = (float *)aligned_alloc(32, bytes);
    s.y = (float *)aligned_alloc(32, bytes);
    s.m = (float *)aligned_alloc(32, bytes);
    if (!s.x || !s.y || !s.m) abort();
    return s;
}

static void snapshot_free(Snapshot *s) {
    free(s->x);
    free(s->y);
    free(s->m);
}

static void snapshot_fill(Snapshot *s, const Particle *arr, uint32_t mass_len) {
    for (uint32_t j = 0; j < mass_len; j++) {
        s->x[j] = arr[j].pos.x;
        s->y[j] = arr[j].pos.y;
        s->m[j] = arr[j].mass;
    }
    for (uint32_t j = mass_len; j < s->padded; j++) {
        s->x[j] = 0.0f;
        s->y[j] = 0.0f;
        s->m[j] = 0.0f;
    }
}

/* semi-implicit Euler, reference sim_cpu.c:191-193: mul then add, per component */
static inline void integrate(Particle *p, float ax, float ay, float dt) {
    p->acc.x = ax;
    p->acc.y = ay;
    p->vel.x = p->vel.x + ax * dt;
    p->vel.y = p->vel.y + ay * dt;
    p->pos.x = p->pos.x + p->vel.x * dt;
    p->pos.y = p->pos.y + p->vel.y * dt;
}

/* ------------------------------------------------------------------ */
/* AVX summation order in plain C: reference sim_cpu.c:156-189         */
/* ------------------------------------------------------------------ */

static void receiver_avx_order(Particle *p, const Snapshot *s, float dt) {
    const float g = NB_G;
    const float x = p->pos.x, y = p->pos.y, r = p->radius;
    float ax[LANES], ay[LANES];
    for (uint32_t e = 0; e < LANES; e++) ax[e] = ay[e] = 0.0f;

    for (uint32_t base = 0; base < s->padded; base += LANES) {
        for (uint32_t k = 0; k < LANES; k++) {
            /* _mm256_set_ps(P[0],...,P[7]) puts P[k] in element 7-k (sim_cpu.c:32-33) */
            const uint32_t e = LANES - 1u - k;
            const uint32_t j = base + k;
            float dx = s->x[j] - x;
            float dy = s->y[j] - y;
            float xx = dx * dx;
            float yy = dy * dy;
            float dist_sq = xx + yy;
            float r2 = dist_sq + r;
            float r1 = sqrtf(r2);
            float gm = s->m[j] * g;
            float r3 = r1 * r2;
            float f = gm / r3;
            float cx = dx * f;
            float cy = dy * f;
            ax[e] = ax[e] + cx;
            ay[e] = ay[e] + cy;
        }
    }
    /* simd_sum, sim_cpu.c:146-154: elements added 0..7 onto 0 */
    float sx = 0.0f, sy = 0.0f;
    for (uint32_t e = 0; e < LANES; e++) sx += ax[e];
    for (uint32_t e = 0; e < LANES; e++) sy += ay[e];
    integrate(p, sx, sy, dt);
}

/*
 * The reference's three SIMD_SET builds differ only in the pack width L (sim_cpu.c:27,56,77: 8, 4, 1):
 * element e of the accumulator takes sources j with j mod L == L-1-e (SIMD_SET_ARR puts P[0] in the top
 * element), the tail pack is zero-filled, and simd_sum adds elements 0..L-1.  L = 1 is the scalar build.
 */
static void receiver_lanes(Particle *p, const float *sx, const float *sy, const float *sm, uint32_t padded,
                           uint32_t L, float dt) {
    const float g = NB_G;
    const float x = p->pos.x, y = p->pos.y, r = p->radius;
    float ax[LANES], ay[LANES];
    for (uint32_t e = 0; e < L; e++) ax[e] = ay[e] = 0.0f;
    for (uint32_t base = 0; base < padded; base += L) {
        for (uint32_t k = 0; k < L; k++) {
            const uint32_t e = L - 1u - k, j = base + k;
            float dx = sx[j] - x;
            float dy = sy[j] - y;
            float xx = dx * dx;
            float yy = dy * dy;
            float r2 = (xx + yy) + r;
            float r1 = sqrtf(r2);
            float gm = sm[j] * g;
            float r3 = r1 * r2;
            float f = gm / r3;
            float cx = dx * f;
            float cy = dy * f;
            ax[e] = ax[e] + cx;
            ay[e] = ay[e] + cy;
        }
    }
    float tx = 0.0f, ty = 0.0f;
    for (uint32_t e = 0; e < L; e++) tx += ax[e];
    for (uint32_t e = 0; e < L; e++) ty += ay[e];
    integrate(p, tx, ty, dt);
}

void orc_step_lanes(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n, uint32_t lanes) {
    if (lanes != 1 && lanes != 4 && lanes != 8) abort();
    const uint32_t padded = (mass_len + lanes - 1u) / lanes * lanes;
    float *sx = (float *)malloc((size_t)(padded + 1) * sizeof(float));
    float *sy = (float *)malloc((size_t)(padded + 1) * sizeof(float));
    float *sm = (float *)malloc((size_t)(padded + 1) * sizeof(float));
    if (!sx || !sy || !sm) abort();
    for (uint32_t it = 0; it < n; it++) {
        for (uint32_t j = 0; j < padded; j++) {
            const int live = j < mass_len;
            sx[j] = live ? arr[j].pos.x : 0.0f;
            sy[j] = live ? arr[j].pos.y : 0.0f;
            sm[j] = live ? arr[j].mass : 0.0f;
        }
#pragma omp parallel for schedule(static, 20)
        for (uint32_t i = 0; i < total_len; i++) receiver_lanes(&arr[i], sx, sy, sm, padded, lanes, dt);
    }
    free(sx);
    free(sy);
    free(sm);
}

void orc_step_avx_order(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n) {
    Snapshot s = snapshot_alloc(mass_len);
    for (uint32_t it = 0; it < n; it++) {
        snapshot_fill(&s, arr, mass_len);
#pragma omp parallel for schedule(static, 20)
        for (uint32_t i = 0; i < total_len; i++) receiver_avx_order(&arr[i], &s, dt);
    }
    snapshot_free(&s);
}

/* ------------------------------------------------------------------ */
/* the same with AVX intrinsics (natural lane order, reversed hsum)     */
/* ------------------------------------------------------------------ */

static inline void force_avx(const Snapshot *s, float x, float y, float r, float *out_ax, float *out_ay) {
    const __m256 g = _mm256_set1_ps(NB_G);
    const __m256 vx = _mm256_set1_ps(x), vy = _mm256_set1_ps(y), vr = _mm256_set1_ps(r);
    __m256 ax = _mm256_setzero_ps(), ay = _mm256_setzero_ps();
    for (uint32_t base = 0; base < s->padded; base += LANES) {
        __m256 dx = _mm256_sub_ps(_mm256_load_ps(s->x + base), vx);
        __m256 dy = _mm256_sub_ps(_mm256_load_ps(s->y + base), vy);
        __m256 d2 = _mm256_add_ps(_mm256_mul_ps(dx, dx), _mm256_mul_ps(dy, dy));
        __m256 r2 = _mm256_add_ps(d2, vr);
        __m256 r1 = _mm256_sqrt_ps(r2);
        __m256 gm = _mm256_mul_ps(_mm256_load_ps(s->m + base), g);
        __m256 r3 = _mm256_mul_ps(r1, r2);
        __m256 f = _mm256_div_ps(gm, r3);
        ax = _mm256_add_ps(ax, _mm256_mul_ps(dx, f));
        ay = _mm256_add_ps(ay, _mm256_mul_ps(dy, f));
    }
    /* our lane e holds the reference's element 7-e, so add lanes 7..0 */
    float fx[LANES], fy[LANES];
    _mm256_storeu_ps(fx, ax);
    _mm256_storeu_ps(fy, ay);
    float sx = 0.0f, sy = 0.0f;
    for (int e = (int)LANES - 1; e >= 0; e--) sx += fx[e];
    for (int e = (int)LANES - 1; e >= 0; e--) sy += fy[e];
    *out_ax = sx;
    *out_ay = sy;
}

void orc_step_avx(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n) {
    Snapshot s = snapshot_alloc(mass_len);
    for (uint32_t it = 0; it < n; it++) {
        snapshot_fill(&s, arr, mass_len);
#pragma omp parallel for schedule(static, 20)
        for (uint32_t i = 0; i < total_len; i++) {
            float ax, ay;
            force_avx(&s, arr[i].pos.x, arr[i].pos.y, arr[i].radius, &ax, &ay);
            integrate(&arr[i], ax, ay, dt);
        }
    }
    snapshot_free(&s);
}

static double wall_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_time_avx_sample(const Particle *arr, uint32_t mass_len, uint32_t recv_begin,
                           uint32_t recv_end, float dt, int threads, int *threads_used,
                           double *checksum) {
    Snapshot s = snapshot_alloc(mass_len);
    snapshot_fill(&s, arr, mass_len);
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#else
    (void)threads;
#endif
    double sum = 0.0;
    double t0 = wall_seconds();
#pragma omp parallel for schedule(static, 20) reduction(+ : sum)
    for (uint32_t i = recv_begin; i < recv_end; i++) {
        Particle p = arr[i];
        float ax, ay;
        force_avx(&s, p.pos.x, p.pos.y, p.radius, &ax, &ay);
        integrate(&p, ax, ay, dt);
        sum += (double)p.pos.x + (double)p.pos.y;
    }
    double t1 = wall_seconds();
    snapshot_free(&s);
    if (threads_used) *threads_used = used;
    if (checksum) *checksum = sum;
    return t1 - t0;
}

/* ------------------------------------------------------------------ */
/* sequential-j fp32: reference scalar build / GLSL loop order          */
/* ------------------------------------------------------------------ */

void orc_step_seq(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n) {
    Snapshot s = snapshot_alloc(mass_len);
    const float g = NB_G;
    for (uint32_t it = 0; it < n; it++) {
        snapshot_fill(&s, arr, mass_len);
#pragma omp parallel for schedule(static, 20)
        for (uint32_t i = 0; i < total_len; i++) {
            const float x = arr[i].pos.x, y = arr[i].pos.y, r = arr[i].radius;
            float ax = 0.0f, ay = 0.0f;
            for (uint32_t j = 0; j < mass_len; j++) {
                float dx = s.x[j] - x;
                float dy = s.y[j] - y;
                float xx = dx * dx;
                float yy = dy * dy;
                float r2 = (xx + yy) + r;
                float r1 = sqrtf(r2);
                float gm = s.m[j] * g;
                float r3 = r1 * r2;
                float f = gm / r3;
                ax = ax + dx * f;
                ay = ay + dy * f;
            }
            integrate(&arr[i], ax, ay, dt);
        }
    }
    snapshot_free(&s);
}

/* ------------------------------------------------------------------ */
/* float64 truth                                                        */
/* ------------------------------------------------------------------ */

void orc_acc_f64(const Particle *arr, uint32_t total_len, uint32_t mass_len,
                 double *acc_xy, double *abs_xy) {
#pragma omp parallel for schedule(static, 20)
    for (uint32_t i = 0; i < total_len; i++) {
        const double x = arr[i].pos.x, y = arr[i].pos.y, r = arr[i].radius;
        double ax = 0, ay = 0, bx = 0, by = 0;
        for (uint32_t j = 0; j < mass_len; j++) {
            double dx = (double)arr[j].pos.x - x;
            double dy = (double)arr[j].pos.y - y;
            double r2 = dx * dx + dy * dy + r;
            double f = ((double)arr[j].mass * (double)NB_G) / (sqrt(r2) * r2);
            ax += dx * f;
            ay += dy * f;
            bx += fabs(dx * f);
            by += fabs(dy * f);
        }
        acc_xy[2 * i] = ax;
        acc_xy[2 * i + 1] = ay;
        if (abs_xy) {
            abs_xy[2 * i] = bx;
            abs_xy[2 * i + 1] = by;
        }
    }
}

void orc_acc_f64_subset(const Particle *arr, uint32_t mass_len, const uint32_t *idx, uint32_t n_idx,
                        double *acc_xy, double *abs_xy) {
#pragma omp parallel for schedule(dynamic, 4)
    for (uint32_t q = 0; q < n_idx; q++) {
        const uint32_t i = idx[q];
        const double x = arr[i].pos.x, y = arr[i].pos.y, r = arr[i].radius;
        double ax = 0, ay = 0, bx = 0, by = 0;
        for (uint32_t j = 0; j < mass_len; j++) {
            double dx = (double)arr[j].pos.x - x;
            double dy = (double)arr[j].pos.y - y;
            double r2 = dx * dx + dy * dy + r;
            double f = ((double)arr[j].mass * (double)NB_G) / (sqrt(r2) * r2);
            ax += dx * f;
            ay += dy * f;
            bx += fabs(dx * f);
            by += fabs(dy * f);
        }
        acc_xy[2 * q] = ax;
        acc_xy[2 * q + 1] = ay;
        abs_xy[2 * q] = bx;
        abs_xy[2 * q + 1] = by;
    }
}

void orc_acc_avx_subset(const Particle *arr, uint32_t mass_len, const uint32_t *idx, uint32_t n_idx, float *acc_xy) {
    Snapshot s = snapshot_alloc(mass_len);
    snapshot_fill(&s, arr, mass_len);
#pragma omp parallel for schedule(dynamic, 4)
    for (uint32_t q = 0; q < n_idx; q++) {
        const Particle *p = &arr[idx[q]];
        force_avx(&s, p->pos.x, p->pos.y, p->radius, &acc_xy[2 * q], &acc_xy[2 * q + 1]);
    }
    snapshot_free(&s);
}

void orc_step_f64(Particle *arr, uint32_t total_len, uint32_t mass_len, float dt, uint32_t n) {
    if (n == 0) return;
    double *st = (double *)malloc((size_t)total_len * 4 * sizeof(double)); /* x y vx vy */
    double *acc = (double *)malloc((size_t)total_len * 2 * sizeof(double));
    double *sx = (double *)malloc((size_t)(mass_len + 1) * 2 * sizeof(double));
    if (!st || !acc || !sx) abort();
    for (uint32_t i = 0; i < total_len; i++) {
        st[4 * i + 0] = arr[i].pos.x;
        st[4 * i + 1] = arr[i].pos.y;
        st[4 * i + 2] = arr[i].vel.x;
        st[4 * i + 3] = arr[i].vel.y;
    }
    const double h = (double)dt;
    for (uint32_t it = 0; it < n; it++) {
        for (uint32_t j = 0; j < mass_len; j++) {
            sx[2 * j] = st[4 * j];
            sx[2 * j + 1] = st[4 * j + 1];
        }
#pragma omp parallel for schedule(static, 20)
        for (uint32_t i = 0; i < total_len; i++) {
            const double x = st[4 * i], y = st[4 * i + 1], r = arr[i].radius;
            double ax = 0, ay = 0;
            for (uint32_t j = 0; j < mass_len; j++) {
                double dx = sx[2 * j] - x, dy = sx[2 * j + 1] - y;
                double r2 = dx * dx + dy * dy + r;
                double f = ((double)arr[j].mass * (double)NB_G) / (sqrt(r2) * r2);
                ax += dx * f;
                ay += dy * f;
            }
            acc[2 * i] = ax;
            acc[2 * i + 1] = ay;
        }
        for (uint32_t i = 0; i < total_len; i++) {
            st[4 * i + 2] += acc[2 * i] * h;
            st[4 * i + 3] += acc[2 * i + 1] * h;
            st[4 * i + 0] += st[4 * i + 2] * h;
            st[4 * i + 1] += st[4 * i + 3] * h;
        }
    }
    for (uint32_t i = 0; i < total_len; i++) {
        arr[i].pos.x = (float)st[4 * i + 0];
        arr[i].pos.y = (float)st[4 * i + 1];
        arr[i].vel.x = (float)st[4 * i + 2];
        arr[i].vel.y = (float)st[4 * i + 3];
        arr[i].acc.x = (float)acc[2 * i];
        arr[i].acc.y = (float)acc[2 * i + 1];
    }
    free(st);
    free(acc);
    free(sx);
}
