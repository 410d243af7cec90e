/*
 * rank_page.h -- one shared page between the processes of a multi-GPU run: rendezvous without any runtime.
 *
 * The reference is single-process (src/bench.c); north_star asks for one process per GPU with host code in C.  The
 * parent maps one anonymous MAP_SHARED region BEFORE it forks and before anything touches HIP; every child (one per
 * rank) inherits the mapping.  Through it the ranks
 *   - meet at barriers (central counter + generation, C11 atomics),
 *   - pass RCCL's 128-byte unique id from rank 0 to the others (nb_hip_comm_unique_id, include/nbody_hip.h),
 *   - reduce a few doubles (max / min / sum over ranks: the contract's max-over-ranks time),
 *   - and, for `--transport shm`, run the caller-supplied host all-gather of CreateWorldShardedWith
 *     (include/nbody.h) through an exchange area behind the page, so that P real processes can step one World on
 *     ONE GPU where RCCL refuses duplicate devices.
 * A rank that cannot go on marks the page failed; every wait notices within a millisecond, prints where it stood and
 * leaves with exit code 4 -- a fresh exit, never a re-exec (the process has initialised the GPU).
 */
#ifndef NB_RANK_PAGE_H
#define NB_RANK_PAGE_H

#include <stddef.h>
#include <stdint.h>

#define NB_RANKS_MAX 64
#define NB_RANK_ID_BYTES 128

typedef struct NbRankPage NbRankPage;

/* Parent, before fork(): maps the page plus exchange_bytes of all-gather area.  NULL on failure (errno set). */
NbRankPage *nb_rank_page_create(int nranks, size_t exchange_bytes, double wait_timeout_s);

/* Child, first thing after fork(): which rank this process is. */
void nb_rank_page_attach(NbRankPage *pg, int rank);

int nb_rank_page_rank(const NbRankPage *pg);
int nb_rank_page_nranks(const NbRankPage *pg);

/* All ranks meet here.  Leaves the process (exit 4) when the page was marked failed or the wait timed out. */
void nb_rank_barrier(NbRankPage *pg, const char *what);

/* Rank 0 publishes the next 128-byte id; the others block until that id (the seq-th of the run) is there. */
void nb_rank_share_id(NbRankPage *pg, void *id128);

/* All-reduce of one double: op = 'x' max, 'n' min, 's' sum.  Collective (two barriers). */
double nb_rank_reduce(NbRankPage *pg, double v, char op);

/* 1 when every rank passed the same 64-bit value. Collective. */
int nb_rank_all_equal(NbRankPage *pg, uint64_t v);

/* NbAllGatherFn (include/nbody.h): ctx = the NbRankPage.  Counts its calls (nb_rank_gather_calls). */
void nb_rank_allgather(void *ctx, void *buf, uint64_t bytes_per_rank, int rank, int nranks);
uint64_t nb_rank_gather_calls(const NbRankPage *pg);

/* Parent or child: mark the run failed so that every waiting rank leaves. */
void nb_rank_page_fail(NbRankPage *pg);
int nb_rank_page_failed(const NbRankPage *pg);

/* Unmaps the region and frees the handle (parent, after the last child is reaped). */
void nb_rank_page_destroy(NbRankPage *pg);

#endif /* NB_RANK_PAGE_H */
