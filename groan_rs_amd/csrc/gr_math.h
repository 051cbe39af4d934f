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

#define GR_MAX_CAND 32

// Per-frame simulation box, prepared on the host (gr_box_setup) and read by the kernels.
struct GrBox {
    float ax, by, cz;      // v1x v2y v3z
    float bx, cx, cy;      // v2x v3x v3y
    float bcx, bcy, bcz;   // box centre
    float r_ws;            // half the shortest non-zero lattice vector: |d| < r_ws => d is its own minimum image
    int ortho;             // v2x == v3x == v3y == 0
    int ncand;             // number of entries in cand
    int valid;             // 0: the frame has no box
    int pad;
    float cand[GR_MAX_CAND][3];
};

// Closed forms of the reference's loops (`while w > L: w -= L; while w < 0: w += L`, vector3d.rs:398-417):
//   t > L : k = ceil(t/L) - 1   (result in (0, L], the upper end stays closed exactly as in the reference)
//   t < 0 : k = floor(t/L)      (result in [0, L))
// followed by one conditional correction each way for the case where the f32 quotient rounded across an
// integer.  For |k| <= 1 this is bit-identical to the loops; beyond that it differs from the repeated f32
// subtraction by an ulp.  No loop: a far-away coordinate cannot stall or diverge a wavefront.
GR_HD float gr_wrap_k(float t, float L) {
    float k = 0.0f;
    if (t > L) k = ceilf(t / L) - 1.0f;
    else if (t < 0.0f) k = floorf(t / L);
    return k;
}
GR_HD float gr_wrap_coordinate(float coor, float L) {
    const float k = gr_wrap_k(coor, L);
    float w = (k == 1.0f) ? coor - L : ((k == -1.0f) ? coor + L : coor - k * L);
    if (w > L) w -= L;
    if (w < 0.0f) w += L;
    return w;
}

// `while d > L/2: d -= L; while d < -L/2: d += L` (vector3d.rs:575-592): result in [-L/2, L/2]
GR_HD float gr_minimg_k(float d, float L, float h) {
    float k = 0.0f;
    if (d > h) k = ceilf((d - h) / L);
    else if (d < -h) k = -ceilf((-h - d) / L);
    return k;
}
GR_HD float gr_min_image(float dx, float L) {
    const float h = L / 2.0f;
    const float k = gr_minimg_k(dx, L, h);
    float d = (k == 1.0f) ? dx - L : ((k == -1.0f) ? dx + L : dx - k * L);
    if (d > h) d -= L;
    if (d < -h) d += L;
    return d;
}

GR_HD float gr_floor_mod(float x, float y) { return fmodf(fmodf(x, y) + y, y); }

// A vector shorter than r_ws (half the shortest lattice vector) is its own unique minimum image
// (|d + t| >= |t| - |d| > |d| for every lattice vector t), so the table is searched only beyond it.
GR_HD void gr_tric_refine(float &dx, float &dy, float &dz, const GrBox &b) {
    float best = dx * dx + dy * dy + dz * dz;
    if (best < b.r_ws * b.r_ws) return;
    float ox = dx, oy = dy, oz = dz;
    for (int m = 0; m < b.ncand; ++m) {
        float x = ox + b.cand[m][0], y = oy + b.cand[m][1], z = oz + b.cand[m][2];
        float r2 = x * x + y * y + z * z;
        if (r2 < best) { best = r2; dx = x; dy = y; dz = z; }
    }
}

// wrap a position into the unit cell
GR_HD void gr_wrap(float &x, float &y, float &z, const GrBox &b) {
    if (b.ortho) {
        x = gr_wrap_coordinate(x, b.ax);
        y = gr_wrap_coordinate(y, b.by);
        z = gr_wrap_coordinate(z, b.cz);
        return;
    }
    // along c, then b, then a
    float k = gr_wrap_k(z, b.cz);
    x -= k * b.cx; y -= k * b.cy; z -= k * b.cz;
    if (z > b.cz) { x -= b.cx; y -= b.cy; z -= b.cz; }
    if (z < 0.0f) { x += b.cx; y += b.cy; z += b.cz; }
    k = gr_wrap_k(y, b.by);
    x -= k * b.bx; y -= k * b.by;
    if (y > b.by) { x -= b.bx; y -= b.by; }
    if (y < 0.0f) { x += b.bx; y += b.by; }
    x = gr_wrap_coordinate(x, b.ax);
}

// minimum-image displacement (in place)
GR_HD void gr_min_image_vec(float &dx, float &dy, float &dz, const GrBox &b) {
    if (b.ortho) {
        dx = gr_min_image(dx, b.ax);
        dy = gr_min_image(dy, b.by);
        dz = gr_min_image(dz, b.cz);
        return;
    }
    const float hz = b.cz / 2.0f, hy = b.by / 2.0f;
    float k = gr_minimg_k(dz, b.cz, hz);
    dx -= k * b.cx; dy -= k * b.cy; dz -= k * b.cz;
    if (dz > hz) { dx -= b.cx; dy -= b.cy; dz -= b.cz; }
    if (dz < -hz) { dx += b.cx; dy += b.cy; dz += b.cz; }
    k = gr_minimg_k(dy, b.by, hy);
    dx -= k * b.bx; dy -= k * b.by;
    if (dy > hy) { dx -= b.bx; dy -= b.by; }
    if (dy < -hy) { dx += b.bx; dy += b.by; }
    dx = gr_min_image(dx, b.ax);
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
GR_HD float gr_mag3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

GR_HD float gr_distance(float ax_, float ay_, float az_, float px, float py, float pz, int dim, const GrBox &b) {
    if (dim == 0) return 0.0f;
    float dx = ax_ - px, dy = ay_ - py, dz = az_ - pz;
    if (b.ortho) {
        // only the requested components are min-imaged, exactly as vector3d.rs:458-486
        switch (dim) {
        case 1: return gr_min_image(dx, b.ax);
        case 2: return gr_min_image(dy, b.by);
        case 3: return gr_min_image(dz, b.cz);
        case 4: return gr_mag3(gr_min_image(dx, b.ax), gr_min_image(dy, b.by), 0.0f);
        case 5: return gr_mag3(gr_min_image(dx, b.ax), 0.0f, gr_min_image(dz, b.cz));
        case 6: return gr_mag3(0.0f, gr_min_image(dy, b.by), gr_min_image(dz, b.cz));
        default: return gr_mag3(gr_min_image(dx, b.ax), gr_min_image(dy, b.by), gr_min_image(dz, b.cz));
        }
    }
    gr_min_image_vec(dx, dy, dz, b);
    switch (dim) {
    case 1: return dx;
    case 2: return dy;
    case 3: return dz;
    case 4: return gr_mag3(dx, dy, 0.0f);
    case 5: return gr_mag3(dx, 0.0f, dz);
    case 6: return gr_mag3(0.0f, dy, dz);
    default: return gr_mag3(dx, dy, dz);
    }
}

// Host-side preparation of a GrBox from the gro-order box9 (simbox.rs:13-26). Returns 0 when the
// diagonal is not strictly positive / not finite (the reference panics or never terminates there).
inline int gr_box_setup(const float *box9, GrBox *b) {
    b->valid = 0; b->ncand = 0; b->pad = 0;
    if (!box9) { b->ax = b->by = b->cz = b->bx = b->cx = b->cy = 0; b->ortho = 1; b->bcx = b->bcy = b->bcz = 0; b->r_ws = 0; return 1; }
    b->ax = box9[0]; b->by = box9[1]; b->cz = box9[2];
    b->bx = box9[5]; b->cx = box9[7]; b->cy = box9[8];
    b->ortho = (b->bx == 0.0f && b->cx == 0.0f && b->cy == 0.0f);
    b->valid = 1;
    if (b->ortho) { b->bcx = b->ax / 2.0f; b->bcy = b->by / 2.0f; b->bcz = b->cz / 2.0f; }
    else { b->bcx = (b->ax + b->bx + b->cx) / 2.0f; b->bcy = (b->by + b->cy) / 2.0f; b->bcz = b->cz / 2.0f; }
    if (!(b->ax > 0.0f) || !(b->by > 0.0f) || !(b->cz > 0.0f) || !isfinite(b->ax) || !isfinite(b->by) || !isfinite(b->cz)) {
        b->r_ws = 0; return 0;
    }
    double tmin2 = 1e300;
    for (int k = -2; k <= 2; ++k)
        for (int j = -2; j <= 2; ++j)
            for (int i = -2; i <= 2; ++i) {
                if (!i && !j && !k) continue;
                double tx = (double)i * b->ax + (double)j * b->bx + (double)k * b->cx;
                double ty = (double)j * b->by + (double)k * b->cy;
                double tz = (double)k * b->cz;
                double t2 = tx * tx + ty * ty + tz * tz;
                if (t2 < tmin2) tmin2 = t2;
                if (b->ortho) continue;
                double lhs = fabs(tx) * b->ax + fabs(ty) * b->by + fabs(tz) * b->cz;
                if (lhs > t2 * (1.0 + 1e-6) && b->ncand < GR_MAX_CAND) {
                    b->cand[b->ncand][0] = (float)tx; b->cand[b->ncand][1] = (float)ty; b->cand[b->ncand][2] = (float)tz;
                    b->ncand++;
                }
            }
    b->r_ws = (float)(0.5 * sqrt(tmin2));
    return 1;
}
