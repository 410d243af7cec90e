// shard_plan.hip -- who owns what in a sharded run (the reference is single-device).
//
// Pure host arithmetic, no GPU call: rank r of P owns the r-th uniform chunk of the massive range [0, mass_len) --
// uniform so that every rank contributes the same count to the per-step all-gather and a gathered index equals the
// global massive index -- plus a slice of the massless range dealt out so that every rank computes the same number of
// receivers (every receiver costs all the sources, whatever its own mass).
#include "pipeline_internal.h"

extern "C" NbShardPlan nb_hip_shard_plan(uint32_t total_len, uint32_t mass_len, int rank, int nranks) {
    NB_ASSERT(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
    NB_ASSERT(mass_len <= total_len, "mass_len %u > total_len %u", mass_len, total_len);
    NbShardPlan p;
    const uint32_t P = (uint32_t)nranks;
    const uint32_t Z = total_len - mass_len;
    // Massive slices: uniform, wave-aligned chunks.  The last ranks may own fewer (or no) real sources.
    const uint32_t Mc = mass_len ? nbi::round_up((mass_len + P - 1) / P, 64) : 0;
    auto mass_of = [&](uint32_t q) -> uint32_t {
        const uint64_t b = (uint64_t)q * Mc;
        return b < mass_len ? (mass_len - (uint32_t)b < Mc ? mass_len - (uint32_t)b : Mc) : 0;
    };
    // Massless slices, "water filling": find the lowest level L with sum_q max(0, L - mass_q) >= Z, give rank q
    // max(0, L - mass_q), and take the surplus back one by one from the highest ranks that got any.  An equal total
    // per rank keeps the workgroup count on a round boundary (choose_shape): one workgroup past it costs a whole
    // round (profiles/r01_shard_overhead_before_fix.txt).
    uint64_t lo = 0, hi = (uint64_t)total_len + 1;
    while (lo < hi) {
        const uint64_t L = (lo + hi) / 2;
        uint64_t got = 0;
        for (uint32_t q = 0; q < P; q++) got += L > mass_of(q) ? L - mass_of(q) : 0;
        if (got >= Z)
            hi = L;
        else
            lo = L + 1;
    }
    const uint64_t level = lo;
    uint64_t surplus = 0;
    for (uint32_t q = 0; q < P; q++) surplus += level > mass_of(q) ? level - mass_of(q) : 0;
    surplus -= Z;
    uint32_t zero_begin = mass_len, zero_max = 0, my_zero_begin = mass_len, my_zero = 0;
    // surplus < number of ranks at the level: rank q gives one back if it is among the last `surplus` takers
    uint32_t takers = 0;
    for (uint32_t q = 0; q < P; q++) takers += level > mass_of(q);
    uint32_t seen = 0;
    for (uint32_t q = 0; q < P; q++) {
        uint32_t z = level > mass_of(q) ? (uint32_t)(level - mass_of(q)) : 0;
        if (level > mass_of(q)) {
            if (seen >= takers - (uint32_t)surplus) z -= 1;
            seen++;
        }
        if (q == (uint32_t)rank) {
            my_zero_begin = zero_begin;
            my_zero = z;
        }
        zero_begin += z;
        zero_max = z > zero_max ? z : zero_max;
    }
    p.mass_chunk = Mc;
    p.zero_chunk = nbi::round_up(zero_max, 64);
    const uint64_t mb = (uint64_t)rank * Mc;
    p.mass_begin = mb < mass_len ? (uint32_t)mb : mass_len;
    p.mass_count = mass_of((uint32_t)rank);
    p.zero_begin = my_zero_begin;
    p.zero_count = my_zero;
    p.src_padded = P * Mc;
    return p;
}
