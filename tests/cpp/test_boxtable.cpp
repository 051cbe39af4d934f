// Host check of the minimum-image table gr_box_setup builds (groan_rs_amd/csrc/gr_math.h) -- the reference has no such table
// (simbox.rs:230-236 rejects non-orthogonal boxes), so what is pinned here is the definition: for every brick-reduced vector the
// best entry of the table gives THE minimum image, found here by an fp64 search over 7 x 7 x 7 lattice vectors.
//  * layout: when cand_pairs is set, the entry count is even and every odd entry is its even neighbour + the first box vector,
//    or a pad that can never win; the real entries come in the order of the plain (k, j, i) enumeration (ties go to the first
//    entry on every path, so the order is part of the contract);
//  * completeness: random and corner vectors of the brick, gr_min_image_vec and the length-only search of the packed kernels
//    against the brute force (length to 2e-6 relative; the length-only form + its stated cancellation error);
//  * cells: the benchmark's ([24, 23, 22; 75, 80, 70] degrees), rhombic dodecahedron, truncated octahedron, random
//    GROMACS-reduced cells (|bx| <= ax/2, |cx| <= ax/2, |cy| <= by/2), cells skewed to those limits.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <array>
#include "../../groan_rs_amd/csrc/gr_math.h"

static void lengths_angles(const double l[3], const double a_deg[3], float box9[9]) {   // simbox.rs:96-123
    const double d2r = M_PI / 180.0, al = a_deg[0] * d2r, be = a_deg[1] * d2r, ga = a_deg[2] * d2r;
    const double ax = l[0], bx = l[1] * cos(ga), by = l[1] * sin(ga), cx = l[2] * cos(be), cy = l[2] * (cos(al) - cos(be) * cos(ga)) / sin(ga);
    const double cz = sqrt(l[2] * l[2] - cx * cx - cy * cy);
    const float v[9] = { (float)ax, (float)by, (float)cz, 0, 0, (float)bx, 0, (float)cx, (float)cy };
    memcpy(box9, v, sizeof v);
}

static long checks = 0, bad = 0;
static void fail(const char *what, const float *b9) { if (bad++ < 20) printf("FAIL %s: box %g %g %g | %g %g %g\n", what, b9[0], b9[1], b9[2], b9[5], b9[7], b9[8]); }

static void check_box(const float box9[9], std::mt19937_64 &rng) {
    GrBox b;
    if (!gr_box_setup(box9, &b) || b.ortho || b.ncand > GR_MAX_CAND) return;
    // --- layout
    std::vector<int> real;
    for (int m = 0; m < b.ncand; ++m) if (b.cand_t2[m] < 1e29f) real.push_back(m);
    if (b.cand_pairs) {
        ++checks;
        if (b.ncand & 1) fail("odd table", box9);
        for (int m = 1; m < b.ncand; m += 2) {
            if (b.cand_t2[m] >= 1e29f) continue;                                  // a pad
            const bool ok = b.cand_t2[m - 1] < 1e29f && fabsf(b.cand[m][0] - (b.cand[m - 1][0] + b.ax)) <= 1e-5f * b.ax &&
                            b.cand[m][1] == b.cand[m - 1][1] && b.cand[m][2] == b.cand[m - 1][2];
            if (!ok) fail("odd entry is not its neighbour + a", box9);
        }
        for (int m = 0; m < b.ncand; m += 2) if (b.cand_t2[m] >= 1e29f) fail("pad on an even slot", box9);
    }
    // the plain enumeration (the criterion of gr_box_setup, restated): same entries, same order
    {
        std::vector<std::array<float, 3>> plain;
        for (int k = -9; k <= 9; ++k) for (int j = -9; j <= 9; ++j) for (int i = -9; i <= 9; ++i) {
            if (!i && !j && !k) continue;
            if (k < 0 || (k == 0 && (j < 0 || (j == 0 && i < 0)))) continue;
            const double tx = (double)i * b.ax + (double)j * b.bx + (double)k * b.cx, ty = (double)j * b.by + (double)k * b.cy, tz = (double)k * b.cz;
            const double t2 = tx * tx + ty * ty + tz * tz;
            if (fabs(tx) * b.ax + fabs(ty) * b.by + fabs(tz) * b.cz > t2 * (1.0 + 1e-6)) plain.push_back({ (float)tx, (float)ty, (float)tz });
        }
        ++checks;
        if (plain.size() != real.size()) fail("entry count", box9);
        else for (size_t q = 0; q < plain.size(); ++q)
            if (plain[q][0] != b.cand[real[q]][0] || plain[q][1] != b.cand[real[q]][1] || plain[q][2] != b.cand[real[q]][2]) { fail("entry order", box9); break; }
    }
    // --- completeness: brick-reduced vectors (random, and pushed into the corners where the far entries matter)
    std::uniform_real_distribution<double> u(-0.5, 0.5);
    for (int t = 0; t < 4000; ++t) {
        double fx = u(rng), fy = u(rng), fz = u(rng);
        if (t % 4 == 1) { fx = copysign(0.5 - 1e-3 * fabs(fx), fx); fy = copysign(0.5 - 1e-3 * fabs(fy), fy); fz = copysign(0.5 - 1e-3 * fabs(fz), fz); }
        if (t % 4 == 2) { fx = copysign(0.5 - 0.1 * fabs(fx), fx); fz = copysign(0.5 - 0.1 * fabs(fz), fz); }
        float dx = (float)(fx * b.ax), dy = (float)(fy * b.by), dz = (float)(fz * b.cz);     // inside the brick
        double best = 1e300;
        for (int k = -3; k <= 3; ++k) for (int j = -3; j <= 3; ++j) for (int i = -3; i <= 3; ++i) {
            const double x = (double)dx - (i * (double)b.ax + j * (double)b.bx + k * (double)b.cx), y = (double)dy - (j * (double)b.by + k * (double)b.cy), z = (double)dz - k * (double)b.cz;
            best = fmin(best, x * x + y * y + z * z);
        }
        float vx = dx, vy = dy, vz = dz;
        gr_min_image_vec(vx, vy, vz, b);
        const double got = (double)vx * vx + (double)vy * vy + (double)vz * vz;
        ++checks;
        if (fabs(sqrt(got) - sqrt(best)) > 2e-6 * (1.0 + sqrt(best))) { fail("minimum image", box9); printf("   d = %g %g %g: |v| = %.9g, brute force %.9g\n", dx, dy, dz, sqrt(got), sqrt(best)); }
        const float r2 = gr_tric_refine_r2<GR_MAX_CAND>(dx, dy, dz, b);                  // the length-only search of the packed kernels
        // |v|^2 = |d|^2 + gain in f32: where the brick-reduced d is not the minimum image (|v| >= half the shortest box length) the
        // sum cancels, and an ulp of |d|^2 shows in |v| as ~1.2e-7 |d|^2 / (2 |v|): 2e-6 nm in the benchmark's cell, 1e-5 in a flat one
        const double d2 = (double)dx * dx + (double)dy * dy + (double)dz * dz;
        if (fabs(sqrt((double)r2) - sqrt(best)) > 2e-6 * (1.0 + sqrt(best)) + 2.5e-7 * d2 / fmax(sqrt(best), 1e-3)) { fail("minimum image (length only)", box9); printf("   d = %.9g %.9g %.9g: |v| = %.9g, brute force %.9g\n", dx, dy, dz, sqrt((double)r2), sqrt(best)); }
    }
}

int main() {
    std::mt19937_64 rng(11);
    float b9[9];
    const double cells[][6] = { { 24, 23, 22, 75, 80, 70 }, { 24.18, 24.18, 24.18, 60, 60, 90 }, { 24, 24, 24, 70.53, 109.47, 70.53 },
                                { 7, 6.5, 6, 75, 80, 70 }, { 6, 6, 6, 60, 60, 90 }, { 5, 6, 7, 80, 85, 75 }, { 4, 5, 6, 70, 80, 75 }, { 10, 10, 10, 60, 60, 60 },
                                { 10, 10, 10, 90, 90, 60 }, { 10, 10, 10, 109.47, 109.47, 109.47 }, { 3, 9, 27, 88, 95, 100 } };
    int paired = 0, total = 0;
    for (auto &c : cells) { lengths_angles(c, c + 3, b9); check_box(b9, rng); GrBox b; gr_box_setup(b9, &b); paired += b.cand_pairs; ++total; }
    std::uniform_real_distribution<double> len(2.0, 30.0), s(-0.5, 0.5);
    for (int t = 0; t < 400; ++t) {                         // GROMACS-reduced cells, a quarter of them on the limits
        const double ax = len(rng), by = len(rng), cz = len(rng);
        double fbx = s(rng), fcx = s(rng), fcy = s(rng);
        if (t % 4 == 0) { fbx = copysign(0.5, fbx); fcx = copysign(0.5, fcx); fcy = copysign(0.5, fcy); }
        const float v[9] = { (float)ax, (float)by, (float)cz, 0, 0, (float)(fbx * ax), 0, (float)(fcx * ax), (float)(fcy * by) };
        check_box(v, rng);
        GrBox b;
        if (gr_box_setup(v, &b) && !b.ortho && b.ncand <= GR_MAX_CAND) { paired += b.cand_pairs; ++total; }
    }
    printf("%ld checks, %ld failures; %d of %d cells have the paired layout\n", checks, bad, paired, total);
    printf(bad ? "FAILED\n" : "PASS\n");
    return bad ? 1 : 0;
}
