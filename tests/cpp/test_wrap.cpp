// Host check of the branch-free wrap / minimum-image closed forms (groan_rs_amd/csrc/gr_math.h) against the reference's loops
// (Vector3D::wrap_coordinate src/structures/vector3d.rs:398-417, min_image :575-592), bit for bit, on the values where
// rounding decides: tiny negatives (the loop's `w += L` rounds to exactly L and stays there -- the closed upper end),
// exact multiples of L, neighbours of 0 / L / L/2, and random values within a few boxes.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include "../../groan_rs_amd/csrc/gr_math.h"

static float loop_wrap(float w, float L) { while (w > L) w -= L; while (w < 0.0f) w += L; return w; }
static float loop_minimg(float d, float L) { const float h = L / 2.0f; while (d > h) d -= L; while (d < -h) d += L; return d; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    std::mt19937_64 rng(7);
    const float Ls[] = { 1.0f, 3.0f, 6.5f, 7.25f, 24.18f, 17.097843f, 0.37f, 100.0f };
    long n = 0, bad = 0;
    for (float L : Ls) {
        std::vector<float> ts;
        for (int m = -2; m <= 3; ++m) {                       // exact multiples and their neighbours
            float t = (float)m * L;
            ts.push_back(t);
            float up = t, dn = t;
            for (int k = 0; k < 4; ++k) { up = std::nextafterf(up, INFINITY); dn = std::nextafterf(dn, -INFINITY); ts.push_back(up); ts.push_back(dn); }
        }
        for (float e : { 1e-9f, 1e-8f, 3e-7f, 1e-6f, 1e-5f, 1e-12f, 1e-30f, 1e-45f }) { ts.push_back(-e); ts.push_back(e); ts.push_back(L - e); ts.push_back(L + e); ts.push_back(-L - e); ts.push_back(-L + e); }
        std::uniform_real_distribution<float> u(-2.0f * L, 3.0f * L), far(-15.0f * L, 16.0f * L), beyond(-300.0f * L, 300.0f * L);
        for (int k = 0; k < 200000; ++k) ts.push_back(u(rng));
        for (int k = 0; k < 100000; ++k) ts.push_back(far(rng));
        for (int k = 0; k < 20000; ++k) ts.push_back(beyond(rng));
        for (int m = -16; m <= 17; ++m) { ts.push_back((float)m * L); ts.push_back(std::nextafterf((float)m * L, INFINITY)); ts.push_back(std::nextafterf((float)m * L, -INFINITY)); ts.push_back(((float)m + 0.5f) * L); }
        for (float t : ts) {
            // bit-identity with the loops up to GR_LOOP_TURNS turns (one turn: closed form; more: the loop itself); farther away the
            // closed form is an ulp-scale approximation of the loop's repeated rounding
            const float want = loop_wrap(t, L), got = gr_wrap_coordinate(t, L);
            const bool one_turn = std::fabs(t) <= 15.0f * L;
            ++n;
            const float tol = 1.2e-7f * std::fabs(t) * (std::fabs(t) / L + 2.0f);   // the loop's own drift: an ulp of t per turn (and either end of the cell)
            if (one_turn ? bits(want) != bits(got) : (std::fabs(want - got) > tol && std::fabs(std::fabs(want - got) - L) > tol)) { if (bad++ < 10) printf("wrap L=%g t=%.9g: loop %.9g closed %.9g\n", L, t, want, got); }
            ++n;   // floor_mod (vector_to, vector3d.rs:561-569): conditional subtractions == the two fmodf calls, bit for bit
            if (bits(gr_floor_mod(t, L)) != bits(gr_floor_mod_ref(t, L)) && !(gr_floor_mod(t, L) == 0.0f && gr_floor_mod_ref(t, L) == 0.0f)) { if (bad++ < 30) printf("floor_mod L=%g t=%.9g: %.9g vs %.9g\n", L, t, gr_floor_mod(t, L), gr_floor_mod_ref(t, L)); }
            const float wm = loop_minimg(t, L), gm = gr_min_image(t, L);
            const bool one = std::fabs(t) <= 15.0f * L;
            ++n;
            if (one ? bits(wm) != bits(gm) : (std::fabs(wm - gm) > tol && std::fabs(std::fabs(wm - gm) - L) > tol)) { if (bad++ < 20) printf("min_image L=%g d=%.9g: loop %.9g closed %.9g\n", L, t, wm, gm); }
        }
    }
    printf("%ld comparisons, %ld mismatches\n", n, bad);
    return bad ? 1 : 0;
}
