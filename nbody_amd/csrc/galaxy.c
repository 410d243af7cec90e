/*
 * galaxy.c -- MakeGalaxies: synthetic spiral galaxies (include/galaxy.h).
 *
 * Restates the generator of the reference's src/lib/galaxy.c:31-221 so that,
 * under the same libc rand() stream, it produces the same particles bit for bit
 * (tests/test_world_cpu.py::test_make_galaxies_bit_exact_with_compiled_reference checks it
 * against the reference's compiled galaxy.c and against tests/golden/ic_*.bin).  That requires drawing random numbers in the
 * same order and rounding each expression the same way; the phases are:
 *
 *   1. sizes      galaxy g < G-1 takes rand % (1 + remaining) extra particles on
 *                 top of MIN_PARTICLES_PER_GALAXY, the last takes the rest      (galaxy.c:43-65)
 *   2. cores      radius U[GC_MIN_R, GC_MAX_R), mass from GC_DENSITY, reach     (galaxy.c:67-79)
 *   3. placement  ring around a random earlier galaxy, retried on overlap       (galaxy.c:81-118)
 *   4. drift      each pair of cores gets opposite tangential kicks             (galaxy.c:120-142)
 *   5. arms       particles on 2..4 Archimedean arms, outer ones massless,
 *                 circular-orbit speed around the core                          (galaxy.c:144-217)
 */
#include "galaxy.h"

#include <math.h>
#include <stdbool.h>

#include "nb_util.h"

typedef struct Galaxy {
    uint32_t first; /* index of the core in the output array */
    uint32_t count; /* particles including the core */
    float inner;    /* no particle closer to the core than this */
    float outer;    /* nominal extent */
} Galaxy;

/*
 * Source of the raw 31-bit draws.  Default: libc rand(), which is what makes the output match the
 * reference's under srand().  MakeGalaxiesSeeded swaps in a generator of our own (below) so that a
 * universe can be regenerated on a box with a different libc; everything downstream of the draw is shared.
 */
static uint64_t own_state[4];

static uint64_t rotl64(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }

/* xoshiro256** (Blackman & Vigna), top 31 bits -> same range as glibc's rand(), RAND_MAX = 2^31 - 1 */
static int own_rand(void) {
    uint64_t *s = own_state;
    const uint64_t out = rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return (int)(out >> 33);
}

/* splitmix64 expands the seed into the four state words (never all zero) */
static void own_seed(uint64_t seed) {
    for (int i = 0; i < 4; i++) {
        uint64_t z = (seed += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        own_state[i] = z ^ (z >> 31);
    }
}

#define OWN_RAND_MAX 2147483647 /* glibc's RAND_MAX, so both sources cover the same range */

static int (*raw_draw)(void) = rand;
static int raw_max = RAND_MAX;

/* uniform float in [lo, hi): evaluated in double, one draw (reference galaxy.c:18-20) */
static float draw_float(double lo, double hi) { return (float)(lo + (hi - lo) * raw_draw() / raw_max); }

/* uniform integer in [lo, hi), one draw (reference galaxy.c:23-25) */
static uint32_t draw_uint(uint32_t lo, uint32_t hi) { return lo + ((uint32_t)raw_draw() % (hi - lo)); }

/* one draw, low bit (reference galaxy.c:27-29) */
static bool draw_bool(void) { return raw_draw() & 1; }

static float sign_draw(void) { return draw_bool() ? -1.f : 1.f; }

static void split_sizes(Galaxy *gs, uint32_t galaxies, uint32_t particles) {
    uint32_t spare = particles - galaxies * MIN_PARTICLES_PER_GALAXY;
    uint32_t next = 0;
    for (uint32_t g = 0; g < galaxies; g++) {
        uint32_t extra = spare;
        if (g + 1 < galaxies) {
            extra = draw_uint(0, 1 + spare);
            spare -= extra;
        }
        gs[g].first = next;
        gs[g].count = MIN_PARTICLES_PER_GALAXY + extra;
        next += gs[g].count;
    }
}

static void make_cores(Galaxy *gs, uint32_t galaxies, Particle *out) {
    for (uint32_t g = 0; g < galaxies; g++) {
        const float rc = draw_float(GC_MIN_R, GC_MAX_R);
        const float root = sqrtf((float)gs[g].count);
        gs[g].inner = rc * MIN_PARTICLE_DIST_CR_F;
        gs[g].outer = rc * MAX_PARTICLE_DIST_CR_F + root * MAX_PARTICLE_DIST_PC_F;
        Particle core = {0};
        core.radius = rc;
        core.mass = GC_R_TO_M(rc);
        out[gs[g].first] = core;
    }
}

static void place_cores(const Galaxy *gs, uint32_t galaxies, Particle *out) {
    /* galaxy 0 stays at the origin */
    for (uint32_t g = 1; g < galaxies; g++) {
        Particle *core = &out[gs[g].first];
        for (bool clash = true; clash;) {
            const uint32_t anchor = draw_uint(0, g);
            const Particle *anchor_core = &out[gs[anchor].first];
            const float reach = gs[g].outer + gs[anchor].outer;
            const float near = MIN_GALAXY_SEPARATION * reach;
            const float far = MAX_GALAXY_SEPARATION * reach;
            /* uniform over the annulus area: sqrt of a uniform squared radius */
            const float dist = sqrtf(draw_float(near * near, far * far));
            const float angle = draw_float(0, 2 * PI);
            core->pos.x = anchor_core->pos.x + dist * cosf(angle);
            core->pos.y = anchor_core->pos.y + dist * sinf(angle);

            clash = false;
            for (uint32_t o = 0; o < g && !clash; o++) {
                if (o == anchor) continue;
                const float keep_out = MIN_GALAXY_SEPARATION * (gs[g].outer + gs[o].outer);
                const float gap_sq = SqMagV2(SubV2(core->pos, out[gs[o].first].pos));
                clash = gap_sq < keep_out * keep_out;
            }
        }
    }
}

static void kick_cores(const Galaxy *gs, uint32_t galaxies, Particle *out) {
    for (uint32_t g = 1; g < galaxies; g++) {
        Particle *a = &out[gs[g].first];
        for (uint32_t o = 0; o < g; o++) {
            Particle *b = &out[gs[o].first];
            const V2 ab = SubV2(b->pos, a->pos);
            const float dist = MagV2(ab);
            const V2 dir = ScaleV2(ab, 1.f / dist);
            /* a fraction of the two-body orbital speed, perpendicular to the line of centres */
            const float va = 0.3f * sqrtf(NB_G * b->mass / dist);
            const float vb = 0.3f * sqrtf(NB_G * a->mass / dist);
            a->vel = AddV2(a->vel, ScaleV2(V2_FROM(dir.y, -dir.x), va));
            b->vel = AddV2(b->vel, ScaleV2(V2_FROM(-dir.y, dir.x), vb));
        }
    }
}

static void fill_arms(const Galaxy *gal, Particle *out) {
    const Particle core = out[gal->first];
    const float span = gal->outer - gal->inner;

    float arm_phase[MAX_SPIRALS];
    const float phase0 = draw_float(0, 2 * PI);
    const uint32_t arms = draw_uint(MIN_SPIRALS, 1 + MAX_SPIRALS);
    const float arm_gap = 2 * PI / (float)arms;
    for (uint32_t a = 0; a < arms; a++) arm_phase[a] = phase0 + (float)a * arm_gap;

    /* r(t) = pitch * t reaches `outer` at t = 2*pi and `inner` at t_min */
    const float t_max = 2 * PI;
    const float pitch = gal->outer / t_max;
    const float t_min = gal->inner / pitch;

    for (uint32_t k = 1; k < gal->count; k++) {
        Particle p = {0};
        const float t = draw_float(t_min, t_max);
        const float r = pitch * t;
        /* squared uniform jitter keeps most particles close to the arm */
        const float jitter_t = draw_float(0, 0.6f * sqrtf(arm_gap));
        const float jitter_r = draw_float(0, 0.6f * sqrtf(fminf(pitch, r - gal->inner)));
        const float dist = r + sign_draw() * (jitter_r * jitter_r);
        const float ang = t + sign_draw() * (jitter_t * jitter_t);

        const float phase = arm_phase[draw_uint(0, arms)];
        const float dx = dist * cosf(ang + phase);
        const float dy = dist * sinf(ang + phase);
        p.pos.x = core.pos.x + dx;
        p.pos.y = core.pos.y + dy;

        /* the further out, the likelier a massless tracer */
        if (draw_float(0.f, 1.f) < (dist - gal->inner) / span) {
            p.radius = 0.5f;
            p.mass = 0.f;
        } else {
            p.radius = draw_float(NP_MIN_R, NP_MAX_R);
            p.mass = NP_R_TO_M(p.radius);
        }

        const float speed = sqrtf(NB_G * core.mass / dist);
        p.vel.x = core.vel.x + speed * (dy / dist);
        p.vel.y = core.vel.y + speed * (-dx / dist);
        out[gal->first + k] = p;
    }
}

Particle *MakeGalaxies(uint32_t particle_count, uint32_t galaxy_count) {
    NB_CHECK(particle_count >= galaxy_count * MIN_PARTICLES_PER_GALAXY,
             "Need at least %u particles to make %u galaxies, called with %u",
             galaxy_count * MIN_PARTICLES_PER_GALAXY, galaxy_count, particle_count);

    Particle *out = NB_NEW(particle_count ? particle_count : 1, Particle);
    NB_CHECK(out != NULL, "Failed to alloc %u particles", particle_count);
    Galaxy *gs = NB_NEW(galaxy_count ? galaxy_count : 1, Galaxy);
    NB_CHECK(gs != NULL, "Failed to alloc %u galaxies", galaxy_count);

    split_sizes(gs, galaxy_count, particle_count);
    make_cores(gs, galaxy_count, out);
    place_cores(gs, galaxy_count, out);
    kick_cores(gs, galaxy_count, out);
    for (uint32_t g = 0; g < galaxy_count; g++) fill_arms(&gs[g], out);

    free(gs);
    return out;
}

Particle *MakeGalaxiesSeeded(uint32_t particle_count, uint32_t galaxy_count, uint64_t seed) {
    own_seed(seed);
    raw_draw = own_rand;
    raw_max = OWN_RAND_MAX;
    Particle *out = MakeGalaxies(particle_count, galaxy_count);
    raw_draw = rand;
    raw_max = RAND_MAX;
    return out;
}
