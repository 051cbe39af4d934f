// gr_math.h -- scalar / 3-vector PBC arithmetic shared by the HIP kernels and the host side of
// libgroan_hip.so.  Everything here is f32 like the reference (src/lib.rs:357-358).
//
// Reference semantics restated (file:line relative to the groan_rs root):
//   wrap_coordinate  src/structures/vector3d.rs:398-417   result in [0, L]  (closed upper end)
//   min_image        src/structures/vector3d.rs:575-592   result in [-L/2, L/2]
//   floor_mod        src/structures/vector3d.rs:28-30
//   vector_to        src/structures/vector3d.rs:561-569
//   distance(dim)    src/structures/vector3d.rs:458-486
//   box centre       src/system/mod.rs:298-308
// Non-orthogonal boxes are an extension (the reference rejects them, simbox.rs:230-236): the same
// operations applied along c, then b, then a, plus a search over the lattice translations that
// can still shorten a brick-reduced vector (GrBox::cand).  With zero off-diagonals every function
// reduces to the orthorhombic arithmetic exactly.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GR_HD __host__ __device__ __forceinline__
#else
#define GR_HD inline
#endif

#define GR_MAX_CAND 16   // half-set: one of each +-t pair

// fminf / fmaxf without the canonicalising `v_max_f32 x, x, x` the compiler puts in front of every IEEE minnum / maxnum
// whose operand is a loop-carried value: one instruction instead of two or three.  v_min_f32 / v_max_f32 return the
// other operand when one is NaN, like fminf / fmaxf.  (Device only; the host build of this header keeps libm.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float gr_fminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float gr_fmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
GR_HD float gr_fminf(float a, float b) { return fminf(a, b); }
GR_HD float gr_fmaxf(float a, float b) { return fmaxf(a, b); }
#endif

// Per-frame simulation box, prepared on the host (gr_box_setup) and read by the kernels.
struct GrBox {
    float ax, by, cz;      // v1x v2y v3z
    float bx, cx, cy;      // v2x v3x v3y
    float bcx, bcy, bcz;   // box centre
    float iax, iby, icz;   // reciprocals of the diagonal (k-estimates only; every use is followed by an exact correction)
    float r_ws;            // half the shortest non-zero lattice vector: |d| < r_ws => d is its own minimum image
    int ortho;             // v2x == v3x == v3y == 0
    int ncand;             // number of entries in cand
    int valid;             // 0: the frame has no box
    int cand_pairs;        // 1: the table is laid out in pairs -- entry 2m + 1 is entry 2m + the first box vector (or a never-winning pad)
    float cand[GR_MAX_CAND][3];   // one of each +-t pair; unused entries and pads are 0 with cand_t2 = 1e30 (never win)
    float cand_t2[GR_MAX_CAND];   // |t|^2
};

// Closed forms of the reference's loops, branch-free.
//   wrap:      `while w > L: w -= L; while w < 0: w += L` (vector3d.rs:398-417).  For t > 0 the result lies in
//              (0, L] (the upper end stays closed: t == L is left alone), for t <= 0 in [0, L] -- L itself when a tiny
//              negative t plus L rounds to L (tests/cpp/test_wrap.cpp walks these values against the loops).
//              k = floor(t/L), and k - 1 when that lands a positive t exactly on 0.
//   min_image: `while d > L/2: d -= L; while d < -L/2: d += L` (vector3d.rs:575-592): result in [-L/2, L/2].
// k is first estimated with the reciprocal of L and then corrected by comparing the actual remainder, so the
// returned k always puts the result in range whatever the rounding of the estimate.  For |k| <= 1 the result
// t - k L is bit-identical to the reference's loop.  Beyond one turn the loop rounds once per turn: gr_wrap_value /
// gr_minimg_value then RUN the loop for up to GR_LOOP_TURNS turns (bit-identical again) and keep the closed form only for
// coordinates farther away than that (an ulp from the reference's value) -- a far-away, infinite or NaN coordinate cannot
// stall a wavefront.
GR_HD float gr_wrap_k(float t, float L, float iL) {
    float k = floorf(t * iL);
    float r = fmaf(-k, L, t);
    k += (r < 0.0f) ? -1.0f : 0.0f;
    k += (r > L) ? 1.0f : 0.0f;
    // r == L: either t is an exact multiple of L (a negative one must come down to 0, a positive one stays on L), or a
    // tiny negative t whose `w += L` ROUNDS to L -- the reference's loop then stops on L (the closed upper end)
    if (r == L && t < 0.0f && fmaf(-(k + 1.0f), L, t) == 0.0f) k += 1.0f;
    r = fmaf(-k, L, t);
    k -= (r == 0.0f && t > 0.0f) ? 1.0f : 0.0f;
    return k;
}
#define GR_LOOP_TURNS 16
GR_HD float gr_wrap_value(float t, float L, float iL) {
    const float k = gr_wrap_k(t, L, iL);
    if (fabsf(k) <= 1.0f || !(fabsf(k) <= (float)GR_LOOP_TURNS)) return fmaf(-k, L, t);
    float w = t;
    for (int it = 0; it < GR_LOOP_TURNS + 2 && w > L; ++it) w -= L;
    for (int it = 0; it < GR_LOOP_TURNS + 2 && w < 0.0f; ++it) w += L;
    return w;
}
GR_HD float gr_wrap_coordinate(float coor, float L) { return gr_wrap_value(coor, L, 1.0f / L); }

GR_HD float gr_minimg_k(float d, float L, float iL, float h) {
    float k = rintf(d * iL);
    float r = fmaf(-k, L, d);
    k += (r > h) ? 1.0f : 0.0f;
    k += (r < -h) ? -1.0f : 0.0f;
    return k;
}
GR_HD float gr_minimg_value(float d, float L, float iL, float h) {
    const float k = gr_minimg_k(d, L, iL, h);
    if (fabsf(k) <= 1.0f || !(fabsf(k) <= (float)GR_LOOP_TURNS)) return fmaf(-k, L, d);
    float w = d;
    for (int it = 0; it < GR_LOOP_TURNS + 2 && w > h; ++it) w -= L;
    for (int it = 0; it < GR_LOOP_TURNS + 2 && w < -h; ++it) w += L;
    return w;
}
GR_HD float gr_min_image(float dx, float L) {
    // d exactly +-h must stay (the loops use strict comparisons): rint(+-0.5) = 0 keeps it
    return gr_minimg_value(dx, L, 1.0f / L, L / 2.0f);
}

// auxiliary floor_mod = ((x % y) + y) % y.  For |x| < 2y both remainders are a conditional, exact subtraction (fmodf of a
// value in [y, 2y) is value - y), so the result is the same bits as the two fmodf calls at a fraction of their cost; anything
// farther away takes the library calls.  (tests/cpp/test_wrap.cpp compares the two bit for bit.)
GR_HD float gr_floor_mod_ref(float x, float y) { return fmodf(fmodf(x, y) + y, y); }
GR_HD float gr_floor_mod(float x, float y) {
    if (!(fabsf(x) < 2.0f * y)) return gr_floor_mod_ref(x, y);
    float s = x;
    if (x >= y) s = x - y; else if (x <= -y) s = x + y;   // x % y (sign of x, exact)
    float u = s + y;                                       // rounds once, like the reference's addition
    if (u >= y) u -= y;                                    // (...) % y: exact
    if (u >= y) u -= y;                                    // s + y rounded up to 2y
    return u;
}

// A vector shorter than r_ws (half the shortest lattice vector) is its own unique minimum image
// (|d + t| >= |t| - |d| > |d| for every lattice vector t), so the table is searched only beyond it.
// |d +- t|^2 - |d|^2 = |t|^2 +- 2 d.t, and the table holds one of each +-t pair, so the better sign of a pair costs
// one dot product: gain = |t|^2 - 2|d.t|.
// (rolled over the real entry count: the search is rarely taken and must not bloat its callers)
GR_HD void gr_tric_refine(float &dx, float &dy, float &dz, const GrBox &b) {
    const float r2 = dx * dx + dy * dy + dz * dz;
    if (r2 < b.r_ws * b.r_ws) return;
    float best = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
#pragma unroll 1
    for (int m = 0; m < b.ncand; ++m) {
        const float tx = b.cand[m][0], ty = b.cand[m][1], tz = b.cand[m][2];
        const float dt = fmaf(tx, dx, fmaf(ty, dy, tz * dz));
        const float g = fmaf(-2.0f, fabsf(dt), b.cand_t2[m]);
        const bool win = g < best;
        const float sg = dt < 0.0f ? 1.0f : -1.0f;   // add t when d.t < 0, subtract it otherwise
        best = win ? g : best;
        sx = win ? sg * tx : sx; sy = win ? sg * ty : sy; sz = win ? sg * tz : sz;
    }
    dx += sx; dy += sy; dz += sz;
}

// squared length of the minimum image of an already brick-reduced d: no vector needed, 5 ops per +-pair, fixed trip
// count (padding entries never win) and fully unrolled -- with the box a by-value kernel argument (k_pairdist) the
// table entries are SGPR operands
template <int NC = GR_MAX_CAND>
GR_HD float gr_tric_refine_r2(float dx, float dy, float dz, const GrBox &b) {
    const float r2 = dx * dx + dy * dy + dz * dz;
    float best = 0.0f;
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        const float dt = fmaf(b.cand[m][0], dx, fmaf(b.cand[m][1], dy, b.cand[m][2] * dz));
        best = gr_fminf(best, fmaf(-2.0f, fabsf(dt), b.cand_t2[m]));
    }
    return fmaxf(r2 + best, 0.0f);
}

// wrap a position into the unit cell
GR_HD void gr_wrap(float &x, float &y, float &z, const GrBox &b) {
    if (b.ortho) {
        x = gr_wrap_value(x, b.ax, b.iax);
        y = gr_wrap_value(y, b.by, b.iby);
        z = gr_wrap_value(z, b.cz, b.icz);
        return;
    }
    // along c, then b, then a
    float k = gr_wrap_k(z, b.cz, b.icz);
    x = fmaf(-k, b.cx, x); y = fmaf(-k, b.cy, y); z = fmaf(-k, b.cz, z);
    k = gr_wrap_k(y, b.by, b.iby);
    x = fmaf(-k, b.bx, x); y = fmaf(-k, b.by, y);
    k = gr_wrap_k(x, b.ax, b.iax);
    x = fmaf(-k, b.ax, x);
}

// minimum-image displacement (in place)
GR_HD void gr_min_image_vec(float &dx, float &dy, float &dz, const GrBox &b) {
    if (b.ortho) {
        dx = gr_minimg_value(dx, b.ax, b.iax, b.ax / 2.0f);
        dy = gr_minimg_value(dy, b.by, b.iby, b.by / 2.0f);
        dz = gr_minimg_value(dz, b.cz, b.icz, b.cz / 2.0f);
        return;
    }
    float k = gr_minimg_k(dz, b.cz, b.icz, b.cz / 2.0f);
    dx = fmaf(-k, b.cx, dx); dy = fmaf(-k, b.cy, dy); dz = fmaf(-k, b.cz, dz);
    k = gr_minimg_k(dy, b.by, b.iby, b.by / 2.0f);
    dx = fmaf(-k, b.bx, dx); dy = fmaf(-k, b.by, dy);
    k = gr_minimg_k(dx, b.ax, b.iax, b.ax / 2.0f);
    dx = fmaf(-k, b.ax, dx);
    gr_tric_refine(dx, dy, dz, b);
}

// shortest vector from `from` to `to` (vector_to)
GR_HD void gr_vector_to(float fx, float fy, float fz, float tx, float ty, float tz, const GrBox &b,
                        float &ox, float &oy, float &oz) {
    const float hx = b.ax / 2.0f, hy = b.by / 2.0f, hz = b.cz / 2.0f;
    if (b.ortho) {
        ox = gr_floor_mod(tx - fx + hx, b.ax) - hx;
        oy = gr_floor_mod(ty - fy + hy, b.by) - hy;
        oz = gr_floor_mod(tz - fz + hz, b.cz) - hz;
        return;
    }
    float dx = tx - fx, dy = ty - fy, dz = tz - fz;
    float nz = gr_floor_mod(dz + hz, b.cz) - hz;
    float kc = rintf((nz - dz) / b.cz);
    dz = nz; dy += kc * b.cy; dx += kc * b.cx;
    float ny = gr_floor_mod(dy + hy, b.by) - hy;
    float kb = rintf((ny - dy) / b.by);
    dy = ny; dx += kb * b.bx;
    dx = gr_floor_mod(dx + hx, b.ax) - hx;
    gr_tric_refine(dx, dy, dz, b);
    ox = dx; oy = dy; oz = dz;
}

// Dimension: 0 None 1 X 2 Y 3 Z 4 XY 5 XZ 6 YZ 7 XYZ (src/structures/dimension.rs:13-23)
// device: the hardware square root (1 ulp) instead of the correctly rounded sequence -- 1e-7 nm at 1 nm
GR_HD float gr_mag3(float x, float y, float z) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x * x + y * y + z * z);
#else
    return sqrtf(x * x + y * y + z * z);
#endif
}

// the reference's own arithmetic, bit for bit: every product and sum rounded on its own (Rust never contracts to FMA) and a
// correctly rounded square root -- for the places where a distance is COMPARED with a threshold (shapes, cut-off pairs) and
// one ulp decides membership
GR_HD float gr_mag3_exact(float x, float y, float z) {
#pragma clang fp contract(off)
    const float xx = x * x, yy = y * y, zz = z * z;
    const float s = (xx + yy) + zz;
    return sqrtf(s);
}
template <bool EXACT> GR_HD float gr_mag3_sel(float x, float y, float z) { return EXACT ? gr_mag3_exact(x, y, z) : gr_mag3(x, y, z); }

// NC = unrolled image-table length for the magnitude-only path (>= b.ncand; the table is padded with never-winning entries)
// EXACT = gr_mag3_exact instead of the fast magnitude
template <int NC = GR_MAX_CAND, bool EXACT = false>
GR_HD float gr_distance(float ax_, float ay_, float az_, float px, float py, float pz, int dim, const GrBox &b) {
    if (dim == 0) return 0.0f;
    float dx = ax_ - px, dy = ay_ - py, dz = az_ - pz;
    if (b.ortho) {
        // only the requested components are min-imaged, exactly as vector3d.rs:458-486
        const float mx = gr_minimg_value(dx, b.ax, b.iax, b.ax / 2.0f);
        const float my = gr_minimg_value(dy, b.by, b.iby, b.by / 2.0f);
        const float mz = gr_minimg_value(dz, b.cz, b.icz, b.cz / 2.0f);
        switch (dim) {
        case 1: return mx;
        case 2: return my;
        case 3: return mz;
        case 4: return gr_mag3_sel<EXACT>(mx, my, 0.0f);
        case 5: return gr_mag3_sel<EXACT>(mx, 0.0f, mz);
        case 6: return gr_mag3_sel<EXACT>(0.0f, my, mz);
        default: return gr_mag3_sel<EXACT>(mx, my, mz);
        }
    }
    if (dim == 7 && !EXACT) {   // magnitude only: brick reduction + gain search, the image vector itself is never formed
        float k = gr_minimg_k(dz, b.cz, b.icz, b.cz / 2.0f);
        dx = fmaf(-k, b.cx, dx); dy = fmaf(-k, b.cy, dy); dz = fmaf(-k, b.cz, dz);
        k = gr_minimg_k(dy, b.by, b.iby, b.by / 2.0f);
        dx = fmaf(-k, b.bx, dx); dy = fmaf(-k, b.by, dy);
        k = gr_minimg_k(dx, b.ax, b.iax, b.ax / 2.0f);
        dx = fmaf(-k, b.ax, dx);
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_sqrtf(gr_tric_refine_r2<NC>(dx, dy, dz, b));
#else
        return sqrtf(gr_tric_refine_r2<NC>(dx, dy, dz, b));
#endif
    }
    gr_min_image_vec(dx, dy, dz, b);
    switch (dim) {
    case 1: return dx;
    case 2: return dy;
    case 3: return dz;
    case 4: return gr_mag3_sel<EXACT>(dx, dy, 0.0f);
    case 5: return gr_mag3_sel<EXACT>(dx, 0.0f, dz);
    case 6: return gr_mag3_sel<EXACT>(0.0f, dy, dz);
    default: return gr_mag3_sel<EXACT>(dx, dy, dz);
    }
}

// Host-side preparation of a GrBox from the gro-order box9 (simbox.rs:13-26). Returns 0 when the
// diagonal is not strictly positive / not finite (the reference panics or never terminates there).
#define GR_BOX_WALK_MAX 2.0e5
inline int gr_box_setup(const float *box9, GrBox *b) {
    b->valid = 0; b->ncand = 0; b->cand_pairs = 0; b->iax = b->iby = b->icz = 0;
    if (!box9) { b->ax = b->by = b->cz = b->bx = b->cx = b->cy = 0; b->ortho = 1; b->bcx = b->bcy = b->bcz = 0; b->r_ws = 0; return 1; }
    b->ax = box9[0]; b->by = box9[1]; b->cz = box9[2];
    b->bx = box9[5]; b->cx = box9[7]; b->cy = box9[8];
    b->ortho = (b->bx == 0.0f && b->cx == 0.0f && b->cy == 0.0f);
    b->valid = 1;
    // box centre = half the diagonal (system/mod.rs:298-308); for a triclinic box that is the centre of the
    // rectangular unit cell gr_wrap maps into, so shift-to-centre + wrap keeps a compact group whole
    b->bcx = b->ax / 2.0f; b->bcy = b->by / 2.0f; b->bcz = b->cz / 2.0f;
    if (!(b->ax > 0.0f) || !(b->by > 0.0f) || !(b->cz > 0.0f) || !isfinite(b->ax) || !isfinite(b->by) || !isfinite(b->cz)) {
        b->r_ws = 0; return 0;
    }
    b->iax = 1.0f / b->ax; b->iby = 1.0f / b->by; b->icz = 1.0f / b->cz;
    double tmin2 = 1e300;
    for (int q = 0; q < GR_MAX_CAND; ++q) { b->cand[q][0] = b->cand[q][1] = b->cand[q][2] = 0.0f; b->cand_t2[q] = 1.0e30f; }
    int overflow = 0;
    // Every lattice vector t that can shorten SOME vector of the brick satisfies |t|^2 < |tx| ax + |ty| by + |tz| cz <= |t| D with
    // D the brick's diagonal, so |t| < D: the enumeration below covers exactly the (k, j, i) that can reach that ball -- a flat
    // cell (cz much shorter than the skew of c) needs k = 3, 4, ... where a compact one needs |i|, |j|, |k| <= 1.  (Rounds 1-3
    // enumerated -2 .. 2 whatever the cell: flat cells got an incomplete table and silently longer "minimum" images;
    // tests/cpp/test_boxtable.cpp.)  One of each +-t pair: k >= 0, then j >= 0, then i > 0; order k, j, i ascending.
    const double D = sqrt((double)b->ax * b->ax + (double)b->by * b->by + (double)b->cz * b->cz);
    int kmax = 0;
    if (b->ortho) {
        // an orthorhombic cell has no table and its shortest lattice vector is its shortest edge: no enumeration (a 100 x 0.1 x 0.1 nm
        // cell would walk millions of lattice points here on every set_box)
        const double m = fmin((double)b->ax, fmin((double)b->by, (double)b->cz));
        tmin2 = m * m;
        kmax = -1;
    } else {
        // the walk is bounded: a cell whose brick diagonal spans more than GR_BOX_WALK_MAX lattice points along its axes is far beyond
        // what the 16-pair table can serve -- refused like any other over-skewed cell instead of enumerated (and no double -> int
        // conversion of an unbounded quotient)
        const double nk = D / b->cz + 1.0, nj = 2.0 * D / b->by + 1.0, ni = 2.0 * D / b->ax + 1.0;
        if (!(nk * nj * ni <= GR_BOX_WALK_MAX)) { overflow = 1; kmax = -1; tmin2 = 0.0; }
        else kmax = (int)floor(D / b->cz);
    }
    for (int k = 0; k <= kmax && !overflow; ++k) {
        const double tz = (double)k * b->cz, cyk = (double)k * b->cy, cxk = (double)k * b->cx;
        const int jlo = k == 0 ? 0 : (int)ceil((-D - cyk) / b->by), jhi = (int)floor((D - cyk) / b->by);
        for (int j = jlo; j <= jhi && !overflow; ++j) {
            const double ty = (double)j * b->by + cyk, x0 = (double)j * b->bx + cxk;
            const int ilo = (k == 0 && j == 0) ? 1 : (int)ceil((-D - x0) / b->ax), ihi = (int)floor((D - x0) / b->ax);
            for (int i = ilo; i <= ihi; ++i) {
                const double tx = (double)i * b->ax + x0;
                const double t2 = tx * tx + ty * ty + tz * tz;
                if (t2 < tmin2) tmin2 = t2;
                const double lhs = fabs(tx) * b->ax + fabs(ty) * b->by + fabs(tz) * b->cz;
                if (lhs > t2 * (1.0 + 1e-6)) {
                    if (b->ncand < GR_MAX_CAND) {
                        b->cand[b->ncand][0] = (float)tx; b->cand[b->ncand][1] = (float)ty; b->cand[b->ncand][2] = (float)tz;
                        b->cand_t2[b->ncand] = (float)t2;
                        b->ncand++;
                    } else { overflow = 1; break; }
                }
            }
        }
    }
    if (overflow) b->ncand = GR_MAX_CAND + 1;   // too skewed for the table -> GR_E_UNSUPPORTED_BOX
    else if (b->ncand) {
        // Pair layout for the packed searches (gr_kernels.h): after the x step of the brick reduction the entries of one (j, k)
        // come as neighbours i, i + 1 -- t and t + a -- and d.(t + a) = d.t + ax dx is one addition instead of three FMAs.  The
        // order of the entries is kept (ties go to the first entry on every path); an entry without its neighbour gets a pad that
        // never wins.  When the pads do not fit the table stays as it is and the searches take all dot products.
        float pc[GR_MAX_CAND][3], pt2[GR_MAX_CAND];
        int pn = 0, fits = 1;
        for (int m = 0; m < b->ncand && fits; ++m) {
            const bool partner = (pn & 1) && b->cand_t2[m] < 1.0e29f && pt2[pn - 1] < 1.0e29f &&
                                 b->cand[m][1] == pc[pn - 1][1] && b->cand[m][2] == pc[pn - 1][2] &&
                                 fabsf(b->cand[m][0] - (pc[pn - 1][0] + b->ax)) <= 1e-5f * b->ax;
            if ((pn & 1) && !partner) { if (pn < GR_MAX_CAND) { pc[pn][0] = pc[pn][1] = pc[pn][2] = 0.0f; pt2[pn] = 1.0e30f; pn++; } else fits = 0; }
            if (pn < GR_MAX_CAND) { pc[pn][0] = b->cand[m][0]; pc[pn][1] = b->cand[m][1]; pc[pn][2] = b->cand[m][2]; pt2[pn] = b->cand_t2[m]; pn++; } else fits = 0;
        }
        if (fits && (pn & 1)) { if (pn < GR_MAX_CAND) { pc[pn][0] = pc[pn][1] = pc[pn][2] = 0.0f; pt2[pn] = 1.0e30f; pn++; } else fits = 0; }
        if (fits) {
            for (int m = 0; m < pn; ++m) { b->cand[m][0] = pc[m][0]; b->cand[m][1] = pc[m][1]; b->cand[m][2] = pc[m][2]; b->cand_t2[m] = pt2[m]; }
            b->ncand = pn; b->cand_pairs = 1;
        }
    }
    b->r_ws = (float)(0.5 * sqrt(tmin2));
    return 1;
}
