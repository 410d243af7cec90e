/*
 * bench_main.c -- nbody-bench for the MI355X build.
 *
 * With no size flags it reproduces the reference harness (src/bench.c:21-74):
 * srand(11037) once, then for each N of its table MakeGalaxies(N, 2), one World
 * per backend, 10 warm-up steps and 100 timed steps at dt = 1 in a single
 * UpdateWorld_* call, microseconds per step printed per backend.  `--cpu` /
 * `--gpu` restrict the backends exactly as there (bench.c:44-48).
 *
 * Extras the BASELINE.json configurations need:
 *   --n N          one size instead of the table (repeatable)
 *   --steps K      timed steps            (default 100)
 *   --warmup W     untimed steps          (default 10; 0 still runs one: it carries the upload)
 *   --dt DT        step size              (default 1.0, the reference's)
 *   --galaxies G   galaxies per universe  (default 2)
 *   --seed S       srand seed             (default 11037)
 *   --own-rng      draw the universes from MakeGalaxiesSeeded(seed + row) instead of libc rand()
 *   --repeats R    time the K-step call R times on the same world and report the fastest (default 1 = the
 *                  reference's single timed call; a row of a few hundred microseconds is at the mercy of one OS hiccup)
 * and more columns: interactions/s = N * mass_len * steps / time per backend, and for the GPU the share of the
 * fp32 roofline that is (14 flop per interaction, 157.3 TFLOP/s), then what the chip allows at THIS size:
 *   GPU floor = N * mass_len / R interactions/s       (R = the rate THIS run measures on THIS box at N = 100 000 before
 *               the table starts -- a calibration world drawn from MakeGalaxiesSeeded, so libc's rand() stream and with
 *               it the table's universes stay the reference's; boxes of the pool differ by up to 8 % on this kernel, so a
 *               constant would silently shift %floor; printed on stderr; --floor-rate R overrides it)
 *             + kernels per step x 1.7 us              (dependent-launch floor inside a hipGraph on this box,
 *               profiles/r01_ubench6_launch_floor.txt; a step is 1 kernel, or 2 when the sources are split)
 *   %floor    = floor / measured: how close the step is to that bound.  Below N ~ 50 000 a launch cannot fill the
 *               chip for long enough to amortise its latencies; the reference's SIZES[] (bench.c:38) all live there.
 * Several GPUs are driven one process per GPU (bench.py under torch.distributed.run, include/nbody_hip.h part 2),
 * not from this single-process harness.
 */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <galaxy.h>
#include <nbody.h>
#include <nbody_hip.h>

#define CALIBRATION_N 100000u     /* the last row of the reference's table (bench.c:38): the large-N rate is measured there */
#define LAUNCH_FLOOR_US 1.7       /* per dependent kernel inside a hipGraph (profiles/r01_ubench6_launch_floor.txt) */

typedef void (*UpdateFn)(World *, float, uint32_t);

static double seconds_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* one warm-up call, one timed call; returns seconds per step */
static double time_backend(World *w, UpdateFn update, float dt, uint32_t warmup, uint32_t steps, uint32_t repeats) {
    /* At least one untimed step always runs, also with --warmup 0: the first UpdateWorld_GPU call carries the
     * upload and the first-touch device setup (stream, HBM buffers, page-locking the World's array), which are
     * not part of a step.  A zero-step call cannot stand in: it is a no-op by contract (world.c:113). */
    update(w, dt, warmup > 0 ? warmup : 1);
    double best = 0.0;
    for (uint32_t r = 0; r < repeats; r++) {
        const double t0 = seconds_now();
        update(w, dt, steps);
        const double t1 = seconds_now();
        if (r == 0 || t1 - t0 < best) best = t1 - t0;
    }
    return best / (double)steps;
}

static uint32_t count_massive(const Particle *ps, uint32_t n) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++) m += ps[i].mass > 0;
    return m;
}

/* interactions/s the step kernel sustains on this box once a launch fills the chip: N = 100 000, fastest of 3 timed
 * 100-step calls, the per-kernel launch floor taken out so that the floor formula does not count it twice */
static double measure_large_n_rate(void) {
    Particle *ps = MakeGalaxiesSeeded(CALIBRATION_N, 2, 0x9e3779b97f4a7c15ull);
    const uint32_t m = count_massive(ps, CALIBRATION_N);
    World *w = CreateWorld(ps, CALIBRATION_N);
    const double per_step = time_backend(w, UpdateWorld_GPU, 0.01f, 200, 100, 3);
    DestroyWorld(w);
    free(ps);
    int k = 0, wv = 0, split = 1;
    uint32_t groups = 0;
    nb_hip_plan_launch(CALIBRATION_N, m, 256, &k, &wv, &split, &groups);
    const double kernels = split > 1 ? 2.0 : 1.0;
    double busy = per_step - kernels * LAUNCH_FLOOR_US * 1e-6;
    if (busy <= 0.0) busy = per_step;
    return (double)CALIBRATION_N * (double)m / busy;
}

static const uint32_t REFERENCE_SIZES[] = {250, 500, 800, 1200, 2000, 4000, 10000, 20000, 50000, 100000};

int main(int argc, char **argv) {
    bool use_cpu = true, use_gpu = true;
    uint32_t sizes[64];
    uint32_t n_sizes = 0;
    uint32_t steps = 100, warmup = 10, galaxies = 2, repeats = 1;
    unsigned seed = 11037;
    bool own_rng = false;
    float dt = 1.f;
    double floor_rate = 0.0; /* 0: measure it */

    for (int a = 1; a < argc; a++) {
        const char *arg = argv[a];
        const char *val = a + 1 < argc ? argv[a + 1] : NULL;
        if (!strcmp(arg, "--cpu")) {
            use_gpu = false;
        } else if (!strcmp(arg, "--gpu")) {
            use_cpu = false;
        } else if (!strcmp(arg, "--n") && val && n_sizes < 64) {
            sizes[n_sizes++] = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--steps") && val) {
            steps = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--warmup") && val) {
            warmup = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--dt") && val) {
            dt = strtof(val, NULL), a++;
        } else if (!strcmp(arg, "--galaxies") && val) {
            galaxies = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--seed") && val) {
            seed = (unsigned)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--repeats") && val) {
            repeats = (uint32_t)strtoul(val, NULL, 0), a++;
        } else if (!strcmp(arg, "--floor-rate") && val) {
            floor_rate = strtod(val, NULL), a++;
        } else if (!strcmp(arg, "--own-rng")) {
            own_rng = true;
        } else {
            fprintf(stderr,
                    "usage: %s [--cpu|--gpu] [--n N]... [--steps K] [--warmup W] [--dt DT] [--galaxies G] [--seed S]"
                    " [--own-rng] [--repeats R] [--floor-rate INT_PER_S]\n",
                    argv[0]);
            return 2;
        }
    }
    if (n_sizes == 0) {
        n_sizes = sizeof REFERENCE_SIZES / sizeof REFERENCE_SIZES[0];
        memcpy(sizes, REFERENCE_SIZES, sizeof REFERENCE_SIZES);
    }
    if (steps == 0) steps = 1;
    if (repeats == 0) repeats = 1;

    if (use_gpu && floor_rate <= 0.0) {
        floor_rate = measure_large_n_rate();
        fprintf(stderr, "nbody-bench: floor rate %.3e interactions/s (measured at N = %u on this box)\n", floor_rate, CALIBRATION_N);
    }

    srand(seed); /* one seed for the whole table, as the reference */

    printf("\t      N");
    if (use_cpu) printf("\t    CPU");
    if (use_gpu) printf("\t    GPU");
    if (use_cpu) printf("\t  CPU int/s");
    if (use_gpu) printf("\t  GPU int/s\t GPU %%peak\t  GPU us\tfloor us\t   %%floor");
    printf("\n");

    for (uint32_t s = 0; s < n_sizes; s++) {
        const uint32_t n = sizes[s];
        Particle *ps = own_rng ? MakeGalaxiesSeeded(n, galaxies, seed + s) : MakeGalaxies(n, galaxies);
        const double pairs = (double)n * (double)count_massive(ps, n);

        double cpu_s = 0, gpu_s = 0;
        if (use_cpu) {
            World *w = CreateWorld(ps, n);
            cpu_s = time_backend(w, UpdateWorld_CPU, dt, warmup, steps, repeats);
            DestroyWorld(w);
        }
        if (use_gpu) {
            World *w = CreateWorld(ps, n);
            gpu_s = time_backend(w, UpdateWorld_GPU, dt, warmup, steps, repeats);
            DestroyWorld(w);
        }
        printf("\t%7u", n);
        if (use_cpu) printf("\t%7ld", (long)(cpu_s * 1e6));
        if (use_gpu) printf("\t%7ld", (long)(gpu_s * 1e6));
        if (use_cpu) printf("\t%11.3e", pairs / cpu_s);
        /* roofline column: 14 flop per interaction (reference op count, sim_cpu.c:169-188) against the
         * MI355X fp32 vector peak of 157.3 TFLOP/s -- the same convention as bench.py */
        if (use_gpu) {
            int k = 0, wv = 0, split = 1;
            uint32_t groups = 0;
            const uint32_t m = count_massive(ps, n);
            /* passes: one launch per <= 3 MiB of (x, y, G*m) sources (step_chain.hip passes_for) */
            const uint32_t passes = m ? (uint32_t)(((uint64_t)m * 12 + (3u << 20) - 1) / (3u << 20)) : 1;
            nb_hip_plan_launch(n, (m + passes - 1) / passes, 256, &k, &wv, &split, &groups);
            /* lane-split steps (small worlds, every knob on auto) are one kernel whatever the classic plan's split says */
            const int lane_split = passes == 1 && nb_hip_plan_launch_lanes(n, m, NULL) > 1;
            const double kernels = (double)passes * (split > 1 && !lane_split ? 2.0 : 1.0);
            const double floor_us = pairs / floor_rate * 1e6 + kernels * LAUNCH_FLOOR_US;
            printf("\t%11.3e\t%9.1f\t%8.2f\t%8.2f\t%9.1f", pairs / gpu_s, pairs / gpu_s * 14.0 / 157.3e12 * 100.0, gpu_s * 1e6,
                   floor_us, floor_us / (gpu_s * 1e6) * 100.0);
        }
        printf("\n");
        fflush(stdout);
        free(ps);
    }
    return 0;
}
