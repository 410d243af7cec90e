/*
 * rank_page.c -- the shared page of a multi-process run (rank_page.h): barriers, id hand-over, reductions and the
 * shared-memory all-gather, in C11 atomics over one anonymous MAP_SHARED mapping made before fork().
 * No HIP, no RCCL, no torch: this is all the "runtime" the ranks of nbody-bench --gpus P share on the host.
 */
#define _GNU_SOURCE
#include "rank_page.h"

#include <errno.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

/* what lives in the mapping (shared by all ranks) */
typedef struct Shared {
    atomic_uint arrived;     /* ranks that reached the current barrier */
    atomic_uint generation;  /* bumps when a barrier opens */
    atomic_uint failed;      /* set once by whoever cannot go on */
    atomic_uint id_seq;      /* ids published so far */
    unsigned char id[NB_RANK_ID_BYTES];
    double slot[NB_RANKS_MAX];
    uint64_t word[NB_RANKS_MAX];
    uint64_t exchange_bytes;
} Shared;

/* process-local handle (copied by fork; `rank` differs per child) */
struct NbRankPage {
    Shared *sh;
    unsigned char *exchange;
    size_t map_bytes;
    int rank, nranks;
    double timeout_s;
    uint32_t ids_seen;
    uint64_t gathers;
};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

NbRankPage *nb_rank_page_create(int nranks, size_t exchange_bytes, double wait_timeout_s) {
    if (nranks < 1 || nranks > NB_RANKS_MAX) {
        errno = EINVAL;
        return NULL;
    }
    const size_t page = 4096;
    const size_t head = (sizeof(Shared) + page - 1) / page * page;
    const size_t total = head + (exchange_bytes + page - 1) / page * page;
    void *m = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return NULL;
    NbRankPage *pg = (NbRankPage *)calloc(1, sizeof *pg);
    if (!pg) {
        munmap(m, total);
        return NULL;
    }
    pg->sh = (Shared *)m;  /* fresh anonymous pages are zero: counters start at 0 */
    pg->sh->exchange_bytes = exchange_bytes;
    pg->exchange = (unsigned char *)m + head;
    pg->map_bytes = total;
    pg->rank = -1;
    pg->nranks = nranks;
    pg->timeout_s = wait_timeout_s > 0 ? wait_timeout_s : 180.0;
    return pg;
}

void nb_rank_page_attach(NbRankPage *pg, int rank) { pg->rank = rank; }
int nb_rank_page_rank(const NbRankPage *pg) { return pg->rank; }
int nb_rank_page_nranks(const NbRankPage *pg) { return pg->nranks; }
void nb_rank_page_fail(NbRankPage *pg) { atomic_store(&pg->sh->failed, 1u); }
int nb_rank_page_failed(const NbRankPage *pg) { return (int)atomic_load(&pg->sh->failed); }
uint64_t nb_rank_gather_calls(const NbRankPage *pg) { return pg->gathers; }

void nb_rank_page_destroy(NbRankPage *pg) {
    if (!pg) return;
    munmap(pg->sh, pg->map_bytes);
    free(pg);
}

static void leave(NbRankPage *pg, const char *why, const char *what) {
    nb_rank_page_fail(pg);
    fprintf(stderr, "%s:%d [rank %d of %d] %s while waiting at \"%s\"; leaving (exit 4)\n", __FILE__, __LINE__, pg->rank, pg->nranks,
            why, what ? what : "?");
    fflush(stderr);
    _exit(4);
}

/* spin briefly (the common case: ranks arrive within microseconds of each other), then yield, then sleep */
static void wait_until(NbRankPage *pg, atomic_uint *var, unsigned not_equal_to, const char *what) {
    unsigned spins = 0;
    double t0 = 0.0;
    while (atomic_load_explicit(var, memory_order_acquire) == not_equal_to) {
        if (atomic_load_explicit(&pg->sh->failed, memory_order_relaxed)) leave(pg, "another rank failed", what);
        spins++;
        if (spins < 4096) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        } else if (spins < 8192) {
            sched_yield();
        } else {
            if (t0 == 0.0) t0 = now_s();
            struct timespec ts = {0, 200000};  /* 0.2 ms */
            nanosleep(&ts, NULL);
            if ((spins & 255u) == 0 && now_s() - t0 > pg->timeout_s) leave(pg, "timed out", what);
        }
    }
}

void nb_rank_barrier(NbRankPage *pg, const char *what) {
    Shared *sh = pg->sh;
    if (atomic_load(&sh->failed)) leave(pg, "another rank failed", what);
    const unsigned gen = atomic_load_explicit(&sh->generation, memory_order_acquire);
    if (atomic_fetch_add_explicit(&sh->arrived, 1u, memory_order_acq_rel) + 1u == (unsigned)pg->nranks) {
        atomic_store_explicit(&sh->arrived, 0u, memory_order_relaxed);
        atomic_fetch_add_explicit(&sh->generation, 1u, memory_order_release);
    } else {
        wait_until(pg, &sh->generation, gen, what);
    }
}

void nb_rank_share_id(NbRankPage *pg, void *id128) {
    Shared *sh = pg->sh;
    /* the previous id must have been read by everyone before rank 0 overwrites it */
    nb_rank_barrier(pg, "before sharing a unique id");
    if (pg->rank == 0) {
        memcpy(sh->id, id128, NB_RANK_ID_BYTES);
        atomic_store_explicit(&sh->id_seq, pg->ids_seen + 1u, memory_order_release);
    } else {
        wait_until(pg, &sh->id_seq, pg->ids_seen, "the RCCL unique id from rank 0");
        memcpy(id128, sh->id, NB_RANK_ID_BYTES);
    }
    pg->ids_seen++;
}

double nb_rank_reduce(NbRankPage *pg, double v, char op) {
    Shared *sh = pg->sh;
    sh->slot[pg->rank] = v;
    nb_rank_barrier(pg, "reduce (values in)");
    double r = sh->slot[0];
    for (int q = 1; q < pg->nranks; q++) {
        const double x = sh->slot[q];
        r = op == 'x' ? (x > r ? x : r) : op == 'n' ? (x < r ? x : r) : r + x;
    }
    nb_rank_barrier(pg, "reduce (values read)");
    return r;
}

int nb_rank_all_equal(NbRankPage *pg, uint64_t v) {
    Shared *sh = pg->sh;
    sh->word[pg->rank] = v;
    nb_rank_barrier(pg, "compare (values in)");
    int same = 1;
    for (int q = 0; q < pg->nranks; q++) same = same && sh->word[q] == sh->word[0];
    nb_rank_barrier(pg, "compare (values read)");
    return same;
}

void nb_rank_allgather(void *ctx, void *buf, uint64_t bytes_per_rank, int rank, int nranks) {
    NbRankPage *pg = (NbRankPage *)ctx;
    if (rank != pg->rank || nranks != pg->nranks || bytes_per_rank * (uint64_t)nranks > pg->sh->exchange_bytes) {
        fprintf(stderr, "%s:%d [rank %d of %d] all-gather of %llu bytes x %d as rank %d does not fit this page (%llu bytes)\n", __FILE__,
                __LINE__, pg->rank, pg->nranks, (unsigned long long)bytes_per_rank, nranks, rank,
                (unsigned long long)pg->sh->exchange_bytes);
        leave(pg, "bad all-gather request", "all-gather");
    }
    unsigned char *mine = (unsigned char *)buf + (size_t)rank * bytes_per_rank;
    memcpy(pg->exchange + (size_t)rank * bytes_per_rank, mine, bytes_per_rank);
    nb_rank_barrier(pg, "all-gather (slots in)");
    if (rank > 0) memcpy(buf, pg->exchange, (size_t)rank * bytes_per_rank);
    if (rank + 1 < nranks)
        memcpy(mine + bytes_per_rank, pg->exchange + (size_t)(rank + 1) * bytes_per_rank, (size_t)(nranks - 1 - rank) * bytes_per_rank);
    nb_rank_barrier(pg, "all-gather (slots read)");  /* nobody refills the area before everyone has read it */
    pg->gathers++;
}
