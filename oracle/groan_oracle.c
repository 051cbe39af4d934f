/*
 * groan_oracle.c -- CPU parity oracle (TEST INFRASTRUCTURE ONLY; see groan_oracle.h).
 *
 * A literal plain-C restatement of the groan_rs v0.11.3 CPU hot path.  Every function
 * cites the reference file:line it follows (paths relative to the reference root).
 * f32 arithmetic, sequential sums in the reference's order, loop-based wrap / min_image,
 * floor_mod via fmodf, libm sinf/cosf/atan2f.  Compile with -ffp-contract=off and
 * without -ffast-math (oracle/Makefile) so no FMA contraction changes the rounding.
 *
 * Parity: PINNED for orthorhombic boxes by the reference's known-answer tests
 * (tests/test_oracle_golden.py).  Triclinic branches are an extension with no reference
 * arithmetic (PARITY UNPINNED; validated against an fp64 brute-force image search).
 */
#define _GNU_SOURCE
#include "groan_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_strict_ortho = 0;
void go_set_strict_orthogonal(int on) { g_strict_ortho = on; }

/* Accumulator width of the long sums (centres, covariance, RMSD sum, sum of weights).
 * 0 (default) = f32 sequential, the reference's arithmetic (iterators.rs:1430-1434, rmsd.rs:567-570,592-598).
 * 1 = the same per-atom f32 terms summed in double.  A sequential f32 sum of S terms carries a rounding
 * error of up to ~S*2^-24 relative, which exceeds the 1e-5 nm parity bar beyond a few thousand atoms; the
 * GPU sums in fp64, so large-S parity tests compare against mode 1 and bound mode 0 against mode 1. */
static int g_acc64 = 0;
void go_set_accumulate_f64(int on) { g_acc64 = on; }
#define ACC(name) float name = 0.0f; double name##_d = 0.0
#define ADD(name, v) do { float v_ = (v); name += v_; name##_d += (double)v_; } while (0)
#define DIVF(num, den) (g_acc64 ? (float)(num##_d / den##_d) : (num / den))

/* box9 accessors (src/structures/simbox.rs:13-26) */
#define V1X(b) ((b)[0])
#define V2Y(b) ((b)[1])
#define V3Z(b) ((b)[2])
#define V2X(b) ((b)[5])
#define V3X(b) ((b)[7])
#define V3Y(b) ((b)[8])

static inline const float *POS(const void *base, size_t stride, uint64_t i) {
    return (const float *)((const char *)base + (size_t)i * stride);
}
static inline float *POSM(void *base, size_t stride, uint64_t i) {
    return (float *)((char *)base + (size_t)i * stride);
}
static inline float MASS(const void *base, size_t stride, uint64_t i) {
    return *(const float *)((const char *)base + (size_t)i * stride);
}

/* src/structures/simbox.rs:185-188 */
int go_box_is_orthogonal(const float b[9]) { return V2X(b) == 0.0f && V3X(b) == 0.0f && V3Y(b) == 0.0f; }

/* simbox_check, src/structures/simbox.rs:230-236 (+ the triclinic extension switch) */
static int box_check(const float *b) {
    if (b == NULL) return GO_E_NO_BOX;
    if (!go_box_is_orthogonal(b) && g_strict_ortho) return GO_E_NOT_ORTHOGONAL;
    /* the reference panics on a zero box length (vector3d.rs:402-404,576-578) */
    if (V1X(b) == 0.0f || V2Y(b) == 0.0f || V3Z(b) == 0.0f) return GO_E_ZERO_BOX;
    return GO_OK;
}

/* src/structures/simbox.rs:96-123 */
void go_box_from_lengths_angles(const float len[3], const float ang[3], float b[9]) {
    const float PI_F = 3.14159265358979323846f;
    memset(b, 0, 9 * sizeof(float));
    V1X(b) = len[0];
    if (ang[0] == 90.0f && ang[1] == 90.0f && ang[2] == 90.0f) {
        V2Y(b) = len[1];
        V3Z(b) = len[2];
    } else {
        float alpha = ang[0] * PI_F / 180.0f;
        float beta = ang[1] * PI_F / 180.0f;
        float gamma = ang[2] * PI_F / 180.0f;
        V2X(b) = len[1] * cosf(gamma);
        V2Y(b) = len[1] * sinf(gamma);
        V3X(b) = len[2] * cosf(beta);
        V3Y(b) = len[2] * (cosf(alpha) - cosf(beta) * cosf(gamma)) / sinf(gamma);
        V3Z(b) = sqrtf(len[2] * len[2] - V3X(b) * V3X(b) - V3Y(b) * V3Y(b));
    }
}

/* src/system/mod.rs:298-308.  Triclinic extension: the same formula -- half the box diagonal -- i.e. the
 * centre of the rectangular ("brick") unit cell 0<=x<=v1x, 0<=y<=v2y, 0<=z<=v3z that go_wrap maps into
 * (GROMACS put_atoms_in_box convention), so "shift the group to the box centre, then wrap" keeps a compact
 * group whole exactly as in the orthorhombic case. */
void go_box_center(const float b[9], float out[3]) {
    out[0] = V1X(b) / 2.0f;
    out[1] = V2Y(b) / 2.0f;
    out[2] = V3Z(b) / 2.0f;
}

/* src/structures/vector3d.rs:28-30 ; Rust f32 `%` is C fmodf */
float go_floor_mod(float x, float y) { return fmodf(fmodf(x, y) + y, y); }

/* src/structures/vector3d.rs:398-417 */
float go_wrap_coordinate(float coor, float box_len) {
    float wrapped = coor;
    while (wrapped > box_len) wrapped -= box_len;
    while (wrapped < 0.0f) wrapped += box_len;
    return wrapped;
}

/* src/structures/vector3d.rs:575-592 */
float go_min_image(float dx, float box_len) {
    float half_box = box_len / 2.0f;
    float new_dx = dx;
    while (new_dx > half_box) new_dx -= box_len;
    while (new_dx < -half_box) new_dx += box_len;
    return new_dx;
}

/* ------------------------------------------------------------------------------------------
 * Triclinic extension helpers (no reference arithmetic).  Box vectors
 *   a = (v1x,0,0)  b = (v2x,v2y,0)  c = (v3x,v3y,v3z)
 * Candidate lattice translations that can shorten a vector already reduced to the brick
 * |d.z|<=cz/2, |d.y|<=by/2, |d.x|<=ax/2:  t = i a + j b + k c with
 *   |t.x| ax + |t.y| by + |t.z| cz > |t|^2     (otherwise |d+t| >= |d| for every d in the brick).
 * ------------------------------------------------------------------------------------------ */
#define TRIC_MAX 1024   /* candidates kept (both signs); cells flat enough to need more are not used by any test */
typedef struct { int n; float t[TRIC_MAX][3]; } tric_cand;

/* every candidate satisfies |t|^2 < |t.x| ax + |t.y| by + |t.z| cz <= |t| D (D = the brick's diagonal), so |t| < D: the loops
 * visit exactly the (k, j, i) that can reach that ball (rounds 1-3 visited -2 .. 2 whatever the cell: too few for flat cells) */
static void tric_candidates(const float *b, tric_cand *c) {
    c->n = 0;
    const double ax = V1X(b), by = V2Y(b), cz = V3Z(b);
    const double D = sqrt(ax * ax + by * by + cz * cz);
    const int kmax = (int)floor(D / cz);
    for (int k = -kmax; k <= kmax; ++k) {
        const double tz = (double)k * cz, cyk = (double)k * V3Y(b), cxk = (double)k * V3X(b);
        const int jlo = (int)ceil((-D - cyk) / by), jhi = (int)floor((D - cyk) / by);
        for (int j = jlo; j <= jhi; ++j) {
            const double ty = (double)j * by + cyk, x0 = (double)j * V2X(b) + cxk;
            const int ilo = (int)ceil((-D - x0) / ax), ihi = (int)floor((D - x0) / ax);
            for (int i = ilo; i <= ihi; ++i) {
                if (!i && !j && !k) continue;
                const double tx = (double)i * ax + x0;
                double lhs = fabs(tx) * ax + fabs(ty) * by + fabs(tz) * cz;
                double t2 = tx * tx + ty * ty + tz * tz;
                if (lhs > t2 * (1.0 + 1e-6) && c->n < TRIC_MAX) {
                    c->t[c->n][0] = (float)tx;
                    c->t[c->n][1] = (float)ty;
                    c->t[c->n][2] = (float)tz;
                    c->n++;
                }
            }
        }
    }
}

/* pick the shortest of d and d + t over the candidate set (strictly shorter only) */
static void tric_refine(float d[3], const tric_cand *c) {
    float best2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    float bx = d[0], by = d[1], bz = d[2];
    for (int m = 0; m < c->n; ++m) {
        float x = d[0] + c->t[m][0], y = d[1] + c->t[m][1], z = d[2] + c->t[m][2];
        float r2 = x * x + y * y + z * z;
        if (r2 < best2) { best2 = r2; bx = x; by = y; bz = z; }
    }
    d[0] = bx; d[1] = by; d[2] = bz;
}

/* src/structures/vector3d.rs:380-384 ; triclinic: same loops applied along c, then b, then a */
void go_wrap(float p[3], const float b[9]) {
    if (go_box_is_orthogonal(b)) {
        p[0] = go_wrap_coordinate(p[0], V1X(b));
        p[1] = go_wrap_coordinate(p[1], V2Y(b));
        p[2] = go_wrap_coordinate(p[2], V3Z(b));
        return;
    }
    while (p[2] > V3Z(b)) { p[0] -= V3X(b); p[1] -= V3Y(b); p[2] -= V3Z(b); }
    while (p[2] < 0.0f)   { p[0] += V3X(b); p[1] += V3Y(b); p[2] += V3Z(b); }
    while (p[1] > V2Y(b)) { p[0] -= V2X(b); p[1] -= V2Y(b); }
    while (p[1] < 0.0f)   { p[0] += V2X(b); p[1] += V2Y(b); }
    p[0] = go_wrap_coordinate(p[0], V1X(b));
}

/* min-image displacement d (in place).  Orthorhombic: per-axis go_min_image (vector3d.rs:458-486).
 * Triclinic: brick reduction along c, b, a with the same loops, then candidate refinement. */
static void min_image_vec(float d[3], const float *b, const tric_cand *cand) {
    if (go_box_is_orthogonal(b)) {
        d[0] = go_min_image(d[0], V1X(b));
        d[1] = go_min_image(d[1], V2Y(b));
        d[2] = go_min_image(d[2], V3Z(b));
        return;
    }
    float hz = V3Z(b) / 2.0f, hy = V2Y(b) / 2.0f;
    while (d[2] > hz)  { d[0] -= V3X(b); d[1] -= V3Y(b); d[2] -= V3Z(b); }
    while (d[2] < -hz) { d[0] += V3X(b); d[1] += V3Y(b); d[2] += V3Z(b); }
    while (d[1] > hy)  { d[0] -= V2X(b); d[1] -= V2Y(b); }
    while (d[1] < -hy) { d[0] += V2X(b); d[1] += V2Y(b); }
    d[0] = go_min_image(d[0], V1X(b));
    tric_refine(d, cand);
}

/* nalgebra Vector3::magnitude: sqrt((x*x + y*y) + z*z) */
static inline float mag3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

static float distance_c(const float a[3], const float p[3], int dim, const float *b, const tric_cand *cand) {
    if (dim == GO_DIM_NONE) return 0.0f;
    if (go_box_is_orthogonal(b)) {
        /* src/structures/vector3d.rs:458-486, literally */
        float dx, dy, dz;
        switch (dim) {
        case GO_DIM_X: return go_min_image(a[0] - p[0], V1X(b));
        case GO_DIM_Y: return go_min_image(a[1] - p[1], V2Y(b));
        case GO_DIM_Z: return go_min_image(a[2] - p[2], V3Z(b));
        case GO_DIM_XY:
            dx = go_min_image(a[0] - p[0], V1X(b)); dy = go_min_image(a[1] - p[1], V2Y(b));
            return mag3(dx, dy, 0.0f);
        case GO_DIM_XZ:
            dx = go_min_image(a[0] - p[0], V1X(b)); dz = go_min_image(a[2] - p[2], V3Z(b));
            return mag3(dx, 0.0f, dz);
        case GO_DIM_YZ:
            dy = go_min_image(a[1] - p[1], V2Y(b)); dz = go_min_image(a[2] - p[2], V3Z(b));
            return mag3(0.0f, dy, dz);
        default:
            dx = go_min_image(a[0] - p[0], V1X(b)); dy = go_min_image(a[1] - p[1], V2Y(b));
            dz = go_min_image(a[2] - p[2], V3Z(b));
            return mag3(dx, dy, dz);
        }
    }
    /* triclinic extension: components of the 3-D minimum-image vector */
    float d[3] = { a[0] - p[0], a[1] - p[1], a[2] - p[2] };
    min_image_vec(d, b, cand);
    switch (dim) {
    case GO_DIM_X: return d[0];
    case GO_DIM_Y: return d[1];
    case GO_DIM_Z: return d[2];
    case GO_DIM_XY: return mag3(d[0], d[1], 0.0f);
    case GO_DIM_XZ: return mag3(d[0], 0.0f, d[2]);
    case GO_DIM_YZ: return mag3(0.0f, d[1], d[2]);
    default: return mag3(d[0], d[1], d[2]);
    }
}

float go_distance(const float a[3], const float p[3], int dim, const float b[9]) {
    tric_cand cand; cand.n = 0;
    if (!go_box_is_orthogonal(b)) tric_candidates(b, &cand);
    return distance_c(a, p, dim, b, &cand);
}

/* src/structures/vector3d.rs:522-533 */
float go_distance_naive(const float a[3], const float p[3], int dim) {
    switch (dim) {
    case GO_DIM_NONE: return 0.0f;
    case GO_DIM_X: return a[0] - p[0];
    case GO_DIM_Y: return a[1] - p[1];
    case GO_DIM_Z: return a[2] - p[2];
    case GO_DIM_XY: { float x = a[0] - p[0], y = a[1] - p[1]; return sqrtf(x * x + y * y); }
    case GO_DIM_XZ: { float x = a[0] - p[0], z = a[2] - p[2]; return sqrtf(x * x + z * z); }
    case GO_DIM_YZ: { float y = a[1] - p[1], z = a[2] - p[2]; return sqrtf(y * y + z * z); }
    default: return mag3(a[0] - p[0], a[1] - p[1], a[2] - p[2]);
    }
}

/* src/structures/vector3d.rs:561-569 ; triclinic: floor_mod reduction along c, b, a + refinement */
static void vector_to_c(const float from[3], const float to[3], const float *b, const tric_cand *cand, float out[3]) {
    if (go_box_is_orthogonal(b)) {
        float hx = V1X(b) / 2.0f, hy = V2Y(b) / 2.0f, hz = V3Z(b) / 2.0f;
        out[0] = go_floor_mod(to[0] - from[0] + hx, V1X(b)) - hx;
        out[1] = go_floor_mod(to[1] - from[1] + hy, V2Y(b)) - hy;
        out[2] = go_floor_mod(to[2] - from[2] + hz, V3Z(b)) - hz;
        return;
    }
    float d[3] = { to[0] - from[0], to[1] - from[1], to[2] - from[2] };
    float hx = V1X(b) / 2.0f, hy = V2Y(b) / 2.0f, hz = V3Z(b) / 2.0f;
    float nz = go_floor_mod(d[2] + hz, V3Z(b)) - hz;
    float kc = rintf((nz - d[2]) / V3Z(b));
    d[2] = nz; d[1] += kc * V3Y(b); d[0] += kc * V3X(b);
    float ny = go_floor_mod(d[1] + hy, V2Y(b)) - hy;
    float kb = rintf((ny - d[1]) / V2Y(b));
    d[1] = ny; d[0] += kb * V2X(b);
    d[0] = go_floor_mod(d[0] + hx, V1X(b)) - hx;
    tric_refine(d, cand);
    out[0] = d[0]; out[1] = d[1]; out[2] = d[2];
}

void go_vector_to(const float from[3], const float to[3], const float b[9], float out[3]) {
    tric_cand cand; cand.n = 0;
    if (!go_box_is_orthogonal(b)) tric_candidates(b, &cand);
    vector_to_c(from, to, b, &cand, out);
}

/* ============================ AtomContainer ============================ */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* src/structures/container.rs:51-104 -- including its quirks: the first index is never range
 * checked, and an out-of-range index closes the current block at n_atoms-1 and stops. */
size_t go_container_from_indices(const uint64_t *indices, size_t n, uint64_t n_atoms,
                                 uint64_t *os, uint64_t *oe) {
    if (n == 0) return 0;
    uint64_t *v = (uint64_t *)malloc(n * sizeof(uint64_t));
    memcpy(v, indices, n * sizeof(uint64_t));
    qsort(v, n, sizeof(uint64_t), cmp_u64);
    size_t nb = 0;
    uint64_t start = v[0], end = v[0];
    for (size_t i = 1; i < n; ++i) {
        uint64_t index = v[i];
        if (index >= n_atoms) { end = n_atoms - 1; break; }
        if (index == end) continue;
        if (index == end + 1) {
            end = index;
        } else {
            os[nb] = start; oe[nb] = end; nb++;
            start = index; end = index;
        }
    }
    os[nb] = start; oe[nb] = end; nb++;
    free(v);
    return nb;
}

typedef struct { uint64_t s, e; } blk;
static int cmp_blk(const void *a, const void *b) {
    const blk *x = (const blk *)a, *y = (const blk *)b;
    if (x->s != y->s) return (x->s > y->s) - (x->s < y->s);
    return (x->e > y->e) - (x->e < y->e);
}

/* src/structures/container.rs:167-215 (from_blocks) */
static size_t from_blocks(blk *bl, size_t n, uint64_t *os, uint64_t *oe) {
    if (n == 0) return 0;
    qsort(bl, n, sizeof(blk), cmp_blk);
    size_t nb = 0;
    uint64_t cur_s = UINT64_MAX, cur_e = 0;
    for (size_t i = 0; i < n; ++i) {
        uint64_t s = bl[i].s, e = bl[i].e;
        if (s > cur_e + 1 || (cur_e == 0 && cur_s != 0)) {
            if (cur_s != UINT64_MAX) { os[nb] = cur_s; oe[nb] = cur_e; nb++; }
            cur_s = s; cur_e = e;
        } else if (e > cur_e) {
            cur_e = e;
        }
    }
    if (cur_s != UINT64_MAX) { os[nb] = cur_s; oe[nb] = cur_e; nb++; }
    return nb;
}

/* src/structures/container.rs:122-153 */
size_t go_container_from_ranges(const uint64_t *start, const uint64_t *end, size_t n, uint64_t n_atoms,
                                uint64_t *os, uint64_t *oe) {
    if (n_atoms == 0 || n == 0) return 0;
    blk *bl = (blk *)malloc(n * sizeof(blk));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        uint64_t s = start[i];
        uint64_t e = end[i] < n_atoms ? end[i] : n_atoms - 1;
        if (s > e) continue;
        bl[m].s = s; bl[m].e = e; m++;
    }
    size_t nb = from_blocks(bl, m, os, oe);
    free(bl);
    return nb;
}

/* src/structures/container.rs:268-276 */
size_t go_container_union(const uint64_t *s1, const uint64_t *e1, size_t n1,
                          const uint64_t *s2, const uint64_t *e2, size_t n2,
                          uint64_t *os, uint64_t *oe) {
    if (n1 + n2 == 0) return 0;
    blk *bl = (blk *)malloc((n1 + n2) * sizeof(blk));
    for (size_t i = 0; i < n1; ++i) { bl[i].s = s1[i]; bl[i].e = e1[i]; }
    for (size_t i = 0; i < n2; ++i) { bl[n1 + i].s = s2[i]; bl[n1 + i].e = e2[i]; }
    size_t nb = from_blocks(bl, n1 + n2, os, oe);
    free(bl);
    return nb;
}

/* src/structures/container.rs:161-165 */
uint64_t go_container_n_atoms(const uint64_t *s, const uint64_t *e, size_t n) {
    uint64_t t = 0;
    for (size_t i = 0; i < n; ++i) t += e[i] - s[i] + 1;
    return t;
}

/* iteration order of next_index, src/structures/container.rs:381-411 */
size_t go_container_expand(const uint64_t *s, const uint64_t *e, size_t n, uint64_t *out) {
    size_t m = 0;
    size_t cur_block = 0;
    uint64_t cur_atom = 0;
    while (cur_block < n) {
        if (cur_atom < s[cur_block]) {
            cur_atom = s[cur_block] + 1;
            out[m++] = s[cur_block];
            continue;
        }
        if (cur_atom <= e[cur_block]) {
            out[m++] = cur_atom;
            cur_atom += 1;
            continue;
        }
        cur_block += 1;
    }
    return m;
}

/* src/structures/container.rs:241-258 */
int go_container_isin(const uint64_t *s, const uint64_t *e, size_t n, uint64_t index) {
    for (size_t i = 0; i < n; ++i) {
        if (index < s[i]) return 0;
        if (index <= e[i]) return 1;
    }
    return 0;
}

/* src/structures/container.rs:278-291 */
size_t go_container_intersection(const uint64_t *s1, const uint64_t *e1, size_t n1,
                                 const uint64_t *s2, const uint64_t *e2, size_t n2,
                                 uint64_t *os, uint64_t *oe) {
    uint64_t na = go_container_n_atoms(s1, e1, n1);
    if (na == 0) return 0;
    uint64_t *all = (uint64_t *)malloc(na * sizeof(uint64_t));
    go_container_expand(s1, e1, n1, all);
    size_t m = 0;
    uint64_t mx = 0;
    for (uint64_t i = 0; i < na; ++i)
        if (go_container_isin(s2, e2, n2, all[i])) { all[m++] = all[i]; if (all[m - 1] > mx) mx = all[m - 1]; }
    size_t nb = 0;
    if (m > 0) nb = go_container_from_indices(all, m, mx + 1, os, oe);
    free(all);
    return nb;
}

/* ============================ centres ============================ */

/* src/structures/iterators.rs:886-903 (mass==NULL) and :946-967 */
int go_center_naive(const void *pos, size_t ps, const void *mass, size_t ms,
                    const uint64_t *idx, size_t n, float out[3], uint64_t *err) {
    ACC(tx); ACC(ty); ACC(tz); ACC(sum);
    size_t n_atoms = 0;
    for (size_t k = 0; k < n; ++k) {
        const float *p = POS(pos, ps, idx[k]);
        if (isnan(p[0])) { if (err) *err = idx[k]; return GO_E_NO_POSITION; }
        if (mass) {
            float m = MASS(mass, ms, idx[k]);
            if (isnan(m)) { if (err) *err = idx[k]; return GO_E_NO_MASS; }
            ADD(tx, p[0] * m); ADD(ty, p[1] * m); ADD(tz, p[2] * m);
            ADD(sum, m);
        } else {
            ADD(tx, p[0]); ADD(ty, p[1]); ADD(tz, p[2]);
            n_atoms += 1;
        }
    }
    if (!mass) { sum = (float)n_atoms; sum_d = (double)n_atoms; }
    out[0] = DIVF(tx, sum); out[1] = DIVF(ty, sum); out[2] = DIVF(tz, sum);
    return GO_OK;
}

/* src/structures/iterators.rs:1152-1191 (mass==NULL -> 1.0) / :1314-1357, with
 * center_atom_contribution + from_circle_to_line, src/auxiliary.rs:59-99.
 * Triclinic extension: the same sums on the fractional ("u") coordinates of the wrapped position. */
int go_estimate_center(const void *pos, size_t ps, const void *mass, size_t ms,
                       const uint64_t *idx, size_t n, const float *b, float out[3], uint64_t *err) {
    int st = box_check(b);
    if (st != GO_OK) return st;
    const float PI_F = 3.14159265358979323846f;
    const float PI_X2 = PI_F * 2.0f;                                  /* auxiliary.rs:15 */
    float sc[3] = { PI_X2 / V1X(b), PI_X2 / V2Y(b), PI_X2 / V3Z(b) }; /* iterators.rs:1154 */
    int ortho = go_box_is_orthogonal(b);
    float xi[3] = { 0, 0, 0 }, zeta[3] = { 0, 0, 0 };
    double xi_d[3] = { 0, 0, 0 }, zeta_d[3] = { 0, 0, 0 };
    int empty = 1;
    for (size_t k = 0; k < n; ++k) {
        float m = 1.0f;
        if (mass) {
            m = MASS(mass, ms, idx[k]);
            if (isnan(m)) { if (err) *err = idx[k]; return GO_E_NO_MASS; } /* mass checked first, :1324-1326 */
        }
        const float *p = POS(pos, ps, idx[k]);
        if (isnan(p[0])) { if (err) *err = idx[k]; return GO_E_NO_POSITION; }
        float w[3] = { p[0], p[1], p[2] };
        go_wrap(w, b);
        if (!ortho) {
            float sc_ = w[2] / V3Z(b);
            float uy = w[1] - sc_ * V3Y(b);
            float ux = w[0] - (uy / V2Y(b)) * V2X(b) - sc_ * V3X(b);
            w[0] = ux; w[1] = uy;
        }
        float th[3] = { w[0] * sc[0], w[1] * sc[1], w[2] * sc[2] };
        for (int a = 0; a < 3; ++a) {
            float c_ = m * cosf(th[a]), s_ = m * sinf(th[a]);
            xi[a] += c_; zeta[a] += s_; xi_d[a] += (double)c_; zeta_d[a] += (double)s_;
        }
        empty = 0;
    }
    if (empty) { out[0] = out[1] = out[2] = NAN; return GO_OK; }
    if (g_acc64) for (int a = 0; a < 3; ++a) { xi[a] = (float)xi_d[a]; zeta[a] = (float)zeta_d[a]; }
    float t[3];
    for (int a = 0; a < 3; ++a) t[a] = (atan2f(-zeta[a], -xi[a]) + PI_F) / sc[a];
    if (ortho) { out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; }
    else {
        float s_c = t[2] / V3Z(b), s_b = t[1] / V2Y(b);
        out[2] = t[2];
        out[1] = t[1] + s_c * V3Y(b);
        out[0] = t[0] + s_b * V2X(b) + s_c * V3X(b);
    }
    return GO_OK;
}

/* src/structures/iterators.rs:1237-1266 (mass==NULL) / :1404-1438 */
int go_get_center(const void *pos, size_t ps, const void *mass, size_t ms,
                  const uint64_t *idx, size_t n, const float *b, float out[3], uint64_t *err) {
    float c[3];
    int st = go_estimate_center(pos, ps, NULL, 0, idx, n, b, c, err); /* always unweighted */
    if (st != GO_OK) return st;
    tric_cand cand; cand.n = 0;
    if (!go_box_is_orthogonal(b)) tric_candidates(b, &cand);
    ACC(tx); ACC(ty); ACC(tz); ACC(sum);
    size_t n_atoms = 0;
    for (size_t k = 0; k < n; ++k) {
        const float *p = POS(pos, ps, idx[k]);
        if (isnan(p[0])) { if (err) *err = idx[k]; return GO_E_NO_POSITION; }
        float m = 1.0f;
        if (mass) {
            m = MASS(mass, ms, idx[k]);
            if (isnan(m)) { if (err) *err = idx[k]; return GO_E_NO_MASS; }
        }
        float v[3];
        vector_to_c(c, p, b, &cand, v);
        float np[3] = { c[0] + v[0], c[1] + v[1], c[2] + v[2] };
        if (mass) { ADD(tx, np[0] * m); ADD(ty, np[1] * m); ADD(tz, np[2] * m); ADD(sum, m); }
        else { ADD(tx, np[0]); ADD(ty, np[1]); ADD(tz, np[2]); n_atoms += 1; }
    }
    if (!mass) { sum = (float)n_atoms; sum_d = (double)n_atoms; }
    out[0] = DIVF(tx, sum); out[1] = DIVF(ty, sum); out[2] = DIVF(tz, sum);
    return GO_OK;
}

/* ============================ distances ============================ */
/* src/system/analysis.rs:401-427 */
int go_group_all_distances(const void *pos, size_t ps, const uint64_t *idx1, size_t n1,
                           const uint64_t *idx2, size_t n2, int dim, const float *b,
                           float *out, uint64_t *err) {
    int st = box_check(b);
    if (st != GO_OK) return st;
    tric_cand cand; cand.n = 0;
    if (!go_box_is_orthogonal(b)) tric_candidates(b, &cand);
    for (size_t i = 0; i < n1; ++i) {
        const float *a = POS(pos, ps, idx1[i]);
        for (size_t j = 0; j < n2; ++j) {
            const float *p = POS(pos, ps, idx2[j]);
            /* atom.rs:780-790: self checked first, then the other atom */
            if (isnan(a[0])) { if (err) *err = idx1[i]; return GO_E_NO_POSITION; }
            if (isnan(p[0])) { if (err) *err = idx2[j]; return GO_E_NO_POSITION; }
            out[i * n2 + j] = distance_c(a, p, dim, b, &cand);
        }
    }
    return GO_OK;
}

/* ============================ translate / wrap / centre ============================ */
/* src/structures/iterators.rs:1520-1525 + src/structures/atom.rs:498-511 */
int go_translate(void *pos, size_t ps, const uint64_t *idx, size_t n, const float v[3],
                 const float *b, uint64_t *err) {
    int st = box_check(b);
    if (st != GO_OK) return st;
    for (size_t k = 0; k < n; ++k) {
        float *p = POSM(pos, ps, idx[k]);
        if (isnan(p[0])) { if (err) *err = idx[k]; return GO_E_NO_POSITION; }
        p[0] += v[0]; p[1] += v[1]; p[2] += v[2];
        go_wrap(p, b);
    }
    return GO_OK;
}

/* src/structures/iterators.rs:1548-1553 + src/structures/atom.rs:535-545 */
int go_wrap_atoms(void *pos, size_t ps, const uint64_t *idx, size_t n, const float *b, uint64_t *err) {
    int st = box_check(b);
    if (st != GO_OK) return st;
    for (size_t k = 0; k < n; ++k) {
        float *p = POSM(pos, ps, idx[k]);
        if (isnan(p[0])) { if (err) *err = idx[k]; return GO_E_NO_POSITION; }
        go_wrap(p, b);
    }
    return GO_OK;
}

/* src/system/utility.rs:109-127 (atoms_center) and :167-185 (atoms_center_mass) */
int go_atoms_center(void *pos, size_t ps, const void *mass, size_t ms,
                    const uint64_t *ref_idx, size_t n_ref, const uint64_t *all_idx, size_t n_all,
                    int dim, int weighted, const float *b, uint64_t *err) {
    if (n_ref == 0) return GO_E_EMPTY_GROUP; /* analysis.rs:52-55 */
    float c[3];
    int st = go_estimate_center(pos, ps, weighted ? mass : NULL, ms, ref_idx, n_ref, b, c, err);
    if (st != GO_OK) return st;
    float bc[3];
    go_box_center(b, bc);
    float shift[3] = { bc[0] - c[0], bc[1] - c[1], bc[2] - c[2] };
    /* Vector3D::filter, vector3d.rs:610-622 */
    int isx = (dim == GO_DIM_X || dim == GO_DIM_XY || dim == GO_DIM_XZ || dim == GO_DIM_XYZ);
    int isy = (dim == GO_DIM_Y || dim == GO_DIM_XY || dim == GO_DIM_YZ || dim == GO_DIM_XYZ);
    int isz = (dim == GO_DIM_Z || dim == GO_DIM_XZ || dim == GO_DIM_YZ || dim == GO_DIM_XYZ);
    if (!isx) shift[0] = 0.0f;
    if (!isy) shift[1] = 0.0f;
    if (!isz) shift[2] = 0.0f;
    return go_translate(pos, ps, all_idx, n_all, shift, b, err);
}

/* ============================ Kabsch ============================ */

/* symmetric 3x3 Jacobi eigen-decomposition (double): A = V diag(w) V^T, columns of V */
static void jacobi_eig3(double A[3][3], double V[3][3], double w[3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-60 || off <= 1e-32 * dg) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0][0]; w[1] = A[1][1]; w[2] = A[2][2];
}

/* R = U D V^T, D = diag(1,1,sign det(U V^T)), for the SVD H = U S V^T (src/system/rmsd.rs:573-583).
 * R is the SVD-implementation-independent polar-like factor, so it is computed here from the
 * eigenvectors of H^T H in double:  u_k = H v_k / s_k (k = 1,2), and with u_3' = u_1 x u_2
 *   R = u_1 v_1^T + u_2 v_2^T + det(V) u_3' v_3^T
 * which equals U D V^T for either sign of the true u_3 and stays defined for rank-2 H. */
void go_kabsch_rotation(const float Hf[9], float R[9]) {
    double H[3][3], HtH[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[i][j] = Hf[3 * i + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += H[k][i] * H[k][j];
            HtH[i][j] = s;
        }
    jacobi_eig3(HtH, V, w);
    int o[3] = { 0, 1, 2 };
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (w[o[j]] > w[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
    double v[3][3]; /* v[k] = k-th right singular vector */
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) v[k][i] = V[i][o[k]];
    double u[3][3];
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 3; ++i) u[k][i] = H[i][0] * v[k][0] + H[i][1] * v[k][1] + H[i][2] * v[k][2];
    }
    double n0 = sqrt(u[0][0] * u[0][0] + u[0][1] * u[0][1] + u[0][2] * u[0][2]);
    if (n0 < 1e-300) { /* H == 0: identity */
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
        return;
    }
    for (int i = 0; i < 3; ++i) u[0][i] /= n0;
    double d01 = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
    for (int i = 0; i < 3; ++i) u[1][i] -= d01 * u[0][i];
    double n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    if (n1 < 1e-12 * n0) { /* rank 1: any unit vector orthogonal to u_0 */
        int m = fabs(u[0][0]) < fabs(u[0][1]) ? (fabs(u[0][0]) < fabs(u[0][2]) ? 0 : 2) : (fabs(u[0][1]) < fabs(u[0][2]) ? 1 : 2);
        double e[3] = { 0, 0, 0 }; e[m] = 1.0;
        double d = e[0] * u[0][0] + e[1] * u[0][1] + e[2] * u[0][2];
        for (int i = 0; i < 3; ++i) u[1][i] = e[i] - d * u[0][i];
        n1 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
    }
    for (int i = 0; i < 3; ++i) u[1][i] /= n1;
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    double detV = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1])
                - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0])
                + v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    double sg = detV < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double r = u[0][i] * v[0][j] + u[1][i] * v[1][j] + sg * u[2][i] * v[2][j];
            R[3 * j + i] = (float)r; /* column-major */
        }
}

/* src/system/rmsd.rs:547-603 */
void go_kabsch_rmsd(const float *p, const float *q, const float *w, size_t n,
                    const float cp[3], const float cq[3], float sum_w,
                    float R[9], float t[3], float *rmsd) {
    float *pc = (float *)malloc(3 * n * sizeof(float) + 4); /* :563 */
    float *qc = (float *)malloc(3 * n * sizeof(float) + 4); /* :564 */
    for (size_t i = 0; i < n; ++i) {
        pc[3 * i] = p[3 * i] - cp[0]; pc[3 * i + 1] = p[3 * i + 1] - cp[1]; pc[3 * i + 2] = p[3 * i + 2] - cp[2];
    }
    for (size_t i = 0; i < n; ++i) {
        qc[3 * i] = q[3 * i] - cq[0]; qc[3 * i + 1] = q[3 * i + 1] - cq[1]; qc[3 * i + 2] = q[3 * i + 2] - cq[2];
    }
    float H[9] = { 0 }; /* row-major h[a][b] += p_a q_b, :567-570 */
    double H_d[9] = { 0 };
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) { float v_ = pc[3 * i + a] * qc[3 * i + b]; H[3 * a + b] += v_; H_d[3 * a + b] += (double)v_; }
    if (g_acc64) for (int a = 0; a < 9; ++a) H[a] = (float)H_d[a];
    go_kabsch_rotation(H, R);
    /* p_rotated = R^T p_c (:586-589); rmsd = sqrt(sum w |p_rot - q_c|^2 / sum_w) (:592-599) */
    float *pr = (float *)malloc(3 * n * sizeof(float) + 4);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) /* (R^T)_{a k} = R_{k a} = R[3*a + k] (column-major) */
            pr[3 * i + a] = R[3 * a + 0] * pc[3 * i] + R[3 * a + 1] * pc[3 * i + 1] + R[3 * a + 2] * pc[3 * i + 2];
    ACC(sum);
    for (size_t i = 0; i < n; ++i) {
        float dx = pr[3 * i] - qc[3 * i], dy = pr[3 * i + 1] - qc[3 * i + 1], dz = pr[3 * i + 2] - qc[3 * i + 2];
        ADD(sum, w[i] * (dx * dx + dy * dy + dz * dz));
    }
    *rmsd = g_acc64 ? (float)sqrt(sum_d / (double)sum_w) : sqrtf(sum / sum_w);
    t[0] = cq[0] - cp[0]; t[1] = cq[1] - cp[1]; t[2] = cq[2] - cp[2];
    free(pc); free(qc); free(pr);
}

/* src/system/rmsd.rs:425-446 (+ :464-492, :496-505) */
int go_rmsd_extract(const void *pos, size_t ps, const void *mass, size_t ms,
                    const uint64_t *idx, size_t n, const float *b,
                    float *coords, float bc[3], uint64_t *err) {
    /* get_box_center: box must exist and be orthogonal (:430) -- or triclinic with the extension */
    if (b == NULL) return GO_E_NO_BOX;
    if (!go_box_is_orthogonal(b) && g_strict_ortho) return GO_E_NOT_ORTHOGONAL;
    go_box_center(b, bc);
    /* group_get_com (analysis.rs:258-274): empty group first */
    if (n == 0) return GO_E_EMPTY_GROUP;
    float com[3];
    int st = go_get_center(pos, ps, mass, ms, idx, n, b, com, err);
    if (st != GO_OK) return st;
    float shift[3] = { bc[0] - com[0], bc[1] - com[1], bc[2] - com[2] };
    for (size_t k = 0; k < n; ++k) {
        const float *p = POS(pos, ps, idx[k]);
        float y[3] = { p[0] + shift[0], p[1] + shift[1], p[2] + shift[2] };
        go_wrap(y, b);
        coords[3 * k] = y[0]; coords[3 * k + 1] = y[1]; coords[3 * k + 2] = y[2];
    }
    return GO_OK;
}

/* src/system/rmsd.rs:141-166 */
int go_calc_rmsd(const void *rpos, size_t rps, const void *rmass, size_t rms,
                 const uint64_t *ridx, size_t nr, const float *rbox,
                 const void *cpos, size_t cps, const void *cmass, size_t cms,
                 const uint64_t *cidx, size_t nc, const float *cbox,
                 float R[9], float *rmsd, uint64_t *err, uint64_t counts[2]) {
    float *rc = (float *)malloc(3 * (nr + 1) * sizeof(float));
    float *cc = (float *)malloc(3 * (nc + 1) * sizeof(float));
    float rbc[3], cbc[3];
    int st = go_rmsd_extract(rpos, rps, rmass, rms, ridx, nr, rbox, rc, rbc, err);
    if (st == GO_OK) st = go_rmsd_extract(cpos, cps, cmass, cms, cidx, nc, cbox, cc, cbc, err);
    if (st == GO_OK && nr != nc) { /* :405-422 */
        if (counts) { counts[0] = nr; counts[1] = nc; }
        st = GO_E_INCONSISTENT_GROUP;
    }
    if (st == GO_OK) {
        float *w = (float *)malloc((nr + 1) * sizeof(float)); /* extract_masses(reference) :154 */
        ACC(sum_w);
        for (size_t k = 0; k < nr; ++k) { w[k] = MASS(rmass, rms, ridx[k]); ADD(sum_w, w[k]); }
        if (g_acc64) sum_w = (float)sum_w_d;
        float t[3];
        go_kabsch_rmsd(rc, cc, w, nr, rbc, cbc, sum_w, R, t, rmsd);
        free(w);
    }
    free(rc); free(cc);
    return st;
}

/* src/system/rmsd.rs:508-528 with atom.rs:498-528,894-903 */
int go_fit_structure(void *pos, size_t ps, const void *mass, size_t ms,
                     const uint64_t *gidx, size_t ng, const uint64_t *all_idx, size_t n_all,
                     const float *b, const float ref_com[3], const float R[9]) {
    float bc[3], com[3];
    go_box_center(b, bc);
    uint64_t e;
    int st = go_get_center(pos, ps, mass, ms, gidx, ng, b, com, &e);
    if (st != GO_OK) return st;
    float inv_bc[3] = { -bc[0], -bc[1], -bc[2] };
    float shift[3] = { bc[0] - com[0], bc[1] - com[1], bc[2] - com[2] };
    for (size_t k = 0; k < n_all; ++k) {
        float *p = POSM(pos, ps, all_idx[k]);
        if (isnan(p[0])) return GO_E_NO_POSITION; /* the reference would panic on unwrap() */
        p[0] += shift[0]; p[1] += shift[1]; p[2] += shift[2];
        go_wrap(p, b);
        p[0] += inv_bc[0]; p[1] += inv_bc[1]; p[2] += inv_bc[2];
        /* rotation_matrix * pos, nalgebra column-major gemv: col0*x + col1*y + col2*z */
        float x = p[0], y = p[1], z = p[2];
        float rx = R[0] * x + R[3] * y + R[6] * z;
        float ry = R[1] * x + R[4] * y + R[7] * z;
        float rz = R[2] * x + R[5] * y + R[8] * z;
        p[0] = rx + ref_com[0]; p[1] = ry + ref_com[1]; p[2] = rz + ref_com[2];
    }
    return GO_OK;
}

/* src/system/rmsd.rs:131-139 */
int go_calc_rmsd_and_fit(const void *rpos, size_t rps, const void *rmass, size_t rms,
                         const uint64_t *ridx, size_t nr, const float *rbox,
                         void *cpos, size_t cps, const void *cmass, size_t cms,
                         const uint64_t *cidx, size_t nc, const uint64_t *all_idx, size_t n_all,
                         const float *cbox, float *rmsd, uint64_t *err, uint64_t counts[2]) {
    float R[9];
    int st = go_calc_rmsd(rpos, rps, rmass, rms, ridx, nr, rbox, cpos, cps, cmass, cms, cidx, nc, cbox,
                          R, rmsd, err, counts);
    if (st != GO_OK) return st;
    float ref_com[3];
    uint64_t e;
    st = go_get_center(rpos, rps, rmass, rms, ridx, nr, rbox, ref_com, &e); /* reference.group_get_com :134 */
    if (st != GO_OK) return st;
    return go_fit_structure(cpos, cps, cmass, cms, cidx, nc, all_idx, n_all, cbox, ref_com, R);
}

/* ============================ CPU baseline ============================ */
#define GO_ATOM_BYTES 232
#define GO_ATOM_MASS_OFF 96   /* Option<f32> payload */
#define GO_ATOM_POS_OFF 164   /* Option<Vector3D> payload */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* per-frame body of RMSDConverterAnalyzer::convert_analyze (src/system/rmsd.rs:238-251): the
 * reference-side data (shifted+wrapped coordinates, box centre, masses, sum of masses, group COM)
 * is cached once (RMSDConverterAnalyzer::new, :186-203); per frame: extract_data_from_system,
 * number_of_positions_consistent, kabsch_rmsd, fit_structure. */
static int frame_rmsd_fit_cached(const float *rc, const float rbc[3], const float *w, float sum_w,
                                 const float ref_com[3], size_t n_sel,
                                 void *cpos, size_t cps, const void *cmass, size_t cms,
                                 const uint64_t *sel, const uint64_t *all, size_t n_all,
                                 const float *box9, float *rmsd) {
    float *cc = (float *)malloc(3 * (n_sel + 1) * sizeof(float));
    float cbc[3], R[9], t[3];
    uint64_t err;
    int st = go_rmsd_extract(cpos, cps, cmass, cms, sel, n_sel, box9, cc, cbc, &err);
    if (st == GO_OK) {
        go_kabsch_rmsd(rc, cc, w, n_sel, rbc, cbc, sum_w, R, t, rmsd);
        st = go_fit_structure(cpos, cps, cmass, cms, sel, n_sel, all, n_all, box9, ref_com, R);
    }
    free(cc);
    return st;
}

double go_baseline_rmsd_fit(float *frames, size_t n_frames, size_t n_atoms,
                            const float *ref_xyz, const float *masses,
                            const float *box9, int n_threads, int layout, float *rmsd_out) {
    if (n_threads < 1) n_threads = 1;
    uint64_t *all = (uint64_t *)malloc(n_atoms * sizeof(uint64_t));
    for (size_t i = 0; i < n_atoms; ++i) all[i] = i;
    /* RMSDConverterAnalyzer::new -- once, untimed */
    float *rc = (float *)malloc(3 * (n_atoms + 1) * sizeof(float));
    float rbc[3], ref_com[3];
    ACC(sum_w);
    uint64_t err0;
    if (go_rmsd_extract(ref_xyz, 12, masses, 4, all, n_atoms, box9, rc, rbc, &err0) != GO_OK) { free(all); free(rc); return -1.0; }
    for (size_t i = 0; i < n_atoms; ++i) ADD(sum_w, masses[i]);
    if (g_acc64) sum_w = (float)sum_w_d;
    go_get_center(ref_xyz, 12, masses, 4, all, n_atoms, box9, ref_com, &err0);
    double worst = 0.0;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
        int nt = omp_get_num_threads();
#else
        int tid = 0, nt = 1;
#endif
        double mine = 0.0;
        /* one System clone per worker (parallel.rs:236) */
        char *atoms = NULL;
        if (layout == 0) {
            atoms = (char *)calloc(n_atoms, GO_ATOM_BYTES);
            for (size_t i = 0; i < n_atoms; ++i)
                *(float *)(atoms + i * GO_ATOM_BYTES + GO_ATOM_MASS_OFF) = masses[i];
        }
#ifdef _OPENMP
#pragma omp barrier
#endif
        /* round-robin frames: parallel.rs:424-448 */
        for (size_t f = (size_t)tid; f < n_frames; f += (size_t)nt) {
            float *fr = frames + f * n_atoms * 3;
            float rmsd = NAN;
            if (layout == 0) {
                /* update_system (frame ingest, untimed on both sides): scatter into the AoS records */
                for (size_t i = 0; i < n_atoms; ++i) {
                    float *p = (float *)(atoms + i * GO_ATOM_BYTES + GO_ATOM_POS_OFF);
                    p[0] = fr[3 * i]; p[1] = fr[3 * i + 1]; p[2] = fr[3 * i + 2];
                }
                double t0 = now_s();
                frame_rmsd_fit_cached(rc, rbc, masses, sum_w, ref_com, n_atoms,
                                      atoms + GO_ATOM_POS_OFF, GO_ATOM_BYTES, atoms + GO_ATOM_MASS_OFF, GO_ATOM_BYTES,
                                      all, all, n_atoms, box9, &rmsd);
                mine += now_s() - t0;
                for (size_t i = 0; i < n_atoms; ++i) {
                    const float *p = (const float *)(atoms + i * GO_ATOM_BYTES + GO_ATOM_POS_OFF);
                    fr[3 * i] = p[0]; fr[3 * i + 1] = p[1]; fr[3 * i + 2] = p[2];
                }
            } else {
                double t0 = now_s();
                frame_rmsd_fit_cached(rc, rbc, masses, sum_w, ref_com, n_atoms,
                                      fr, 12, masses, 4, all, all, n_atoms, box9, &rmsd);
                mine += now_s() - t0;
            }
            if (rmsd_out) rmsd_out[f] = rmsd;
        }
        free(atoms);
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (mine > worst) worst = mine; }
    }
    free(all); free(rc);
    return worst;
}

/* ================================================================== geometry selection */
/* src/structures/shape.rs:343-378 */
int go_shape_prism_init(go_shape *s, const float b1[3], const float b2[3], const float b3[3], float height) {
    static const int orient[3] = { GO_DIM_X, GO_DIM_Y, GO_DIM_Z }, plane[3] = { GO_DIM_YZ, GO_DIM_XZ, GO_DIM_XY };
    int found = -1;
    for (int a = 0; a < 3; ++a)
        if (b1[a] == b2[a] && b2[a] == b3[a]) {
            if (found >= 0) return 2; /* "can not be constructed" */
            found = a;
        }
    if (found < 0) return 1; /* "does not lie in xy, xz, nor yz plane" */
    memset(s, 0, sizeof *s);
    s->kind = GO_SHAPE_TRIANGULAR_PRISM;
    for (int a = 0; a < 3; ++a) { s->position[a] = b1[a]; s->base2[a] = b2[a]; s->base3[a] = b3[a]; }
    s->size[0] = height;
    s->orientation = orient[found]; s->plane = plane[found];
    return 0;
}

/* TriangularPrism::sign :408-428 */
static float prism_sign(const float p1[3], const float p2[3], const float p3[3], int plane) {
    int u = 0, v = 1;
    if (plane == GO_DIM_XZ) { u = 0; v = 2; }
    else if (plane == GO_DIM_YZ) { u = 1; v = 2; }
    return (p1[u] - p3[u]) * (p2[v] - p3[v]) - (p2[u] - p3[u]) * (p1[v] - p3[v]);
}

static float box_len(const float *b, int dim) { return dim == GO_DIM_X ? V1X(b) : (dim == GO_DIM_Y ? V2Y(b) : V3Z(b)); }

/* Non-orthogonal boxes (extension, same definition as groan_rs_amd/csrc/gr_shape.h: SOME lattice image of the point lies inside
 * the shape taken as a plain body anchored at its position; images = those the body can reach from the brick-reduced difference).  Operation by
 * operation the device code's arithmetic (this file is built with -ffp-contract=off). */
static int shape_free_inside(const go_shape *s, float ex, float ey, float ez) {
    switch (s->kind) {
    case GO_SHAPE_RECTANGULAR:
        return ex >= 0.0f && ex <= s->size[0] && ey >= 0.0f && ey <= s->size[1] && ez >= 0.0f && ez <= s->size[2];
    case GO_SHAPE_CYLINDER: {
        const float along = s->orientation == GO_DIM_X ? ex : (s->orientation == GO_DIM_Y ? ey : ez);
        if (!(along >= 0.0f && along <= s->size[1])) return 0;
        float u, v;
        if (s->orientation == GO_DIM_X) { u = ey; v = ez; } else if (s->orientation == GO_DIM_Y) { u = ex; v = ez; } else { u = ex; v = ey; }
        const float uu = u * u, vv = v * v;
        /* gr_mag3_exact(x, y, z) = sqrt((xx + yy) + zz) with one of the three terms an exact 0 */
        const float sum = s->orientation == GO_DIM_X ? ((0.0f + uu) + vv) : (s->orientation == GO_DIM_Y ? ((uu + 0.0f) + vv) : ((uu + vv) + 0.0f));
        return sqrtf(sum) <= s->size[0];
    }
    case GO_SHAPE_TRIANGULAR_PRISM: {
        const float along = s->orientation == GO_DIM_X ? ex : (s->orientation == GO_DIM_Y ? ey : ez);
        if (!(along >= 0.0f && along < s->size[0])) return 0;
        const float img[3] = { s->position[0] + ex, s->position[1] + ey, s->position[2] + ez };
        float d1 = prism_sign(img, s->position, s->base2, s->plane);
        float d2 = prism_sign(img, s->base2, s->base3, s->plane);
        float d3 = prism_sign(img, s->base3, s->position, s->plane);
        int has_neg = (d1 < 0.0f) || (d2 < 0.0f) || (d3 < 0.0f);
        int has_pos = (d1 > 0.0f) || (d2 > 0.0f) || (d3 > 0.0f);
        return !(has_neg && has_pos);
    }
    }
    return 0;
}
static int shape_inside_tric(const go_shape *s, const float pt[3], const float *b) {
    float dx = pt[0] - s->position[0], dy = pt[1] - s->position[1], dz = pt[2] - s->position[2];
    float k = rintf(dz / V3Z(b));
    dx = dx - k * V3X(b); dy = dy - k * V3Y(b); dz = dz - k * V3Z(b);
    k = rintf(dy / V2Y(b));
    dx = dx - k * V2X(b); dy = dy - k * V2Y(b);
    k = rintf(dx / V1X(b));
    dx = dx - k * V1X(b);
    /* the images the body can reach: |t| <= reach + half the brick's diagonal (gr_shape.h, gr_shape_inside_tric) */
    double reach;
    if (s->kind == GO_SHAPE_RECTANGULAR) reach = sqrt((double)s->size[0] * s->size[0] + (double)s->size[1] * s->size[1] + (double)s->size[2] * s->size[2]);
    else if (s->kind == GO_SHAPE_CYLINDER) reach = sqrt((double)s->size[0] * s->size[0] + (double)s->size[1] * s->size[1]);
    else {
        double e2 = 0.0, e3 = 0.0;
        for (int a = 0; a < 3; ++a) {
            e2 += ((double)s->base2[a] - s->position[a]) * ((double)s->base2[a] - s->position[a]);
            e3 += ((double)s->base3[a] - s->position[a]) * ((double)s->base3[a] - s->position[a]);
        }
        reach = (double)s->size[0] + sqrt(e2 > e3 ? e2 : e3);
    }
    const double T = (reach + 0.5 * sqrt((double)V1X(b) * V1X(b) + (double)V2Y(b) * V2Y(b) + (double)V3Z(b) * V3Z(b))) * (1.0 + 1e-6);
    const double ia = 1.0 / V1X(b), ib = 1.0 / V2Y(b), ic = 1.0 / V3Z(b);
    const int kmax = (int)floor(T * ic);
    for (int kc = -kmax; kc <= kmax; ++kc) {
        const double cyk = (double)kc * V3Y(b), cxk = (double)kc * V3X(b);
        const int jlo = (int)ceil((-T - cyk) * ib), jhi = (int)floor((T - cyk) * ib);
        for (int kb = jlo; kb <= jhi; ++kb) {
            const double x0 = (double)kb * V2X(b) + cxk;
            const int ilo = (int)ceil((-T - x0) * ia), ihi = (int)floor((T - x0) * ia);
            for (int ka = ilo; ka <= ihi; ++ka) {
                const float tx = ((float)ka * V1X(b) + (float)kb * V2X(b)) + (float)kc * V3X(b);
                const float ty = (float)kb * V2Y(b) + (float)kc * V3Y(b);
                const float tz = (float)kc * V3Z(b);
                if (shape_free_inside(s, dx + tx, dy + ty, dz + tz)) return 1;
            }
        }
    }
    return 0;
}

int go_shape_inside(const go_shape *s, const float pt[3], const float *b) {
    if ((V2X(b) != 0.0f || V3X(b) != 0.0f || V3Y(b) != 0.0f) && s->kind != GO_SHAPE_SPHERE) return shape_inside_tric(s, pt, b);
    switch (s->kind) {
    case GO_SHAPE_SPHERE: /* :114-116 */
        return go_distance(pt, s->position, GO_DIM_XYZ, b) < s->size[0];
    case GO_SHAPE_RECTANGULAR: { /* :169-184 */
        float dx = go_distance(pt, s->position, GO_DIM_X, b); if (dx < 0.0f) dx += V1X(b);
        float dy = go_distance(pt, s->position, GO_DIM_Y, b); if (dy < 0.0f) dy += V2Y(b);
        float dz = go_distance(pt, s->position, GO_DIM_Z, b); if (dz < 0.0f) dz += V3Z(b);
        return dx <= s->size[0] && dy <= s->size[1] && dz <= s->size[2];
    }
    case GO_SHAPE_CYLINDER: { /* :256-275 */
        float da = go_distance(pt, s->position, s->orientation, b);
        if (da < 0.0f) da += box_len(b, s->orientation);
        if (da > s->size[1] || go_distance(pt, s->position, s->plane, b) > s->size[0]) return 0;
        return 1;
    }
    case GO_SHAPE_TRIANGULAR_PRISM: { /* :435-460 */
        float d = go_distance(pt, s->position, s->orientation, b);
        if (d < 0.0f) d += box_len(b, s->orientation);
        if (d >= s->size[0]) return 0;
        float d1 = prism_sign(pt, s->position, s->base2, s->plane);
        float d2 = prism_sign(pt, s->base2, s->base3, s->plane);
        float d3 = prism_sign(pt, s->base3, s->position, s->plane);
        int has_neg = (d1 < 0.0f) || (d2 < 0.0f) || (d3 < 0.0f);
        int has_pos = (d1 > 0.0f) || (d2 > 0.0f) || (d3 > 0.0f);
        return !(has_neg && has_pos);
    }
    }
    return 0;
}

int go_shape_inside_naive(const go_shape *s, const float pt[3]) {
    switch (s->kind) {
    case GO_SHAPE_SPHERE: /* :473-475 */
        return go_distance_naive(pt, s->position, GO_DIM_XYZ) < s->size[0];
    case GO_SHAPE_CYLINDER: { /* :482-487 */
        float d = go_distance_naive(pt, s->position, s->orientation);
        return d >= 0.0f && d < s->size[1] && go_distance_naive(pt, s->position, s->plane) < s->size[0];
    }
    case GO_SHAPE_RECTANGULAR: { /* :494-500 */
        float dx = go_distance_naive(pt, s->position, GO_DIM_X), dy = go_distance_naive(pt, s->position, GO_DIM_Y),
              dz = go_distance_naive(pt, s->position, GO_DIM_Z);
        return dx >= 0.0f && dx <= s->size[0] && dy >= 0.0f && dy <= s->size[1] && dz >= 0.0f && dz <= s->size[2];
    }
    }
    return -1;
}

/* src/structures/group.rs:119-175 */
size_t go_group_from_geometries(const void *pos, size_t ps, const uint64_t *idx, size_t n, const float *b,
                                const go_shape *shapes, size_t ns, int naive, uint64_t *out) {
    size_t cnt = 0;
    for (size_t k = 0; k < n; ++k) {
        const float *p = POS(pos, ps, idx[k]);
        if (isnan(p[0])) continue; /* atoms that have no positions are not inside the shape */
        int inside = 1;
        for (size_t q = 0; q < ns && inside; ++q)
            inside = naive ? (go_shape_inside_naive(&shapes[q], p) == 1) : go_shape_inside(&shapes[q], p, b);
        if (inside) out[cnt++] = idx[k];
    }
    return cnt;
}

/* ================================================================== cut-off pair search (brute force) */
size_t go_pairs_within(const void *pos, size_t ps, const uint64_t *idx1, size_t n1, const uint64_t *idx2, size_t n2,
                       const float *b, float cutoff, size_t max_pairs, uint64_t *out_i, uint64_t *out_j, float *out_d) {
    size_t cnt = 0;
    for (size_t a = 0; a < n1; ++a) {
        const float *pi = POS(pos, ps, idx1[a]);
        for (size_t c = 0; c < n2; ++c) {
            if (idx2[c] == idx1[a]) continue; /* hbonds.rs:250 */
            const float *pj = POS(pos, ps, idx2[c]);
            const float d = go_distance(pj, pi, GO_DIM_XYZ, b); /* acceptor.distance(donor) :261 */
            if (d > cutoff) continue;                           /* :262-264 */
            if (cnt < max_pairs) { out_i[cnt] = idx1[a]; out_j[cnt] = idx2[c]; out_d[cnt] = d; }
            ++cnt;
        }
    }
    return cnt;
}
