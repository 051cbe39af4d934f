// gr_shape.h -- geometry-selection predicates (host + device).
// Reference: Shape::inside for Sphere / Rectangular / Cylinder / TriangularPrism (src/structures/shape.rs:110-185,
// 252-276, 431-461), the PBC-free NaiveShape variants (:466-505) and Group::apply_geometries
// (src/structures/group.rs:119-175: an atom without a position is inside nothing; with several shapes it must be inside
// all of them).  The predicates are compositions of gr_distance (1-D distances are signed), so their truth values are
// the reference's wherever gr_distance is bit-compatible (orthorhombic boxes, which is all the reference accepts here:
// src/system/groups.rs:104-110).  Non-orthogonal boxes: gr_shape_inside_tric below.
#pragma once
#include "gr_math.h"
#if defined(__HIPCC__)
#include "gr_layout.h"
#endif

enum { GR_SH_SPHERE = 1, GR_SH_RECTANGULAR = 2, GR_SH_CYLINDER = 3, GR_SH_PRISM = 4 };
#define GR_MAX_SHAPES 8

struct GrShapeDev {
    int kind;
    float px, py, pz;          // sphere centre / box origin / centre of the cylinder base / base1 of the prism
    float a, b, c;             // sphere: radius ; rectangular: x y z ; cylinder: radius height ; prism: height
    float b2x, b2y, b2z, b3x, b3y, b3z;
    int orientation, plane;    // GR_DIM_* ordinals (1 X, 2 Y, 3 Z, 4 XY, 5 XZ, 6 YZ)
};
struct GrShapeSet { int n, naive; GrShapeDev s[GR_MAX_SHAPES]; };

GR_HD float gr_box_len(const GrBox &b, int dim) { return dim == 1 ? b.ax : (dim == 2 ? b.by : b.cz); }

// vector3d.rs:522-533 -- compared with a radius: the reference's own arithmetic (no FMA contraction, correctly rounded
// square root), an atom ON the surface must fall on the reference's side
GR_HD float gr_distance_naive(float ax_, float ay_, float az_, float px, float py, float pz, int dim) {
    const float dx = ax_ - px, dy = ay_ - py, dz = az_ - pz;
    switch (dim) {
    case 0: return 0.0f;
    case 1: return dx;
    case 2: return dy;
    case 3: return dz;
    case 4: return gr_mag3_exact(dx, dy, 0.0f);
    case 5: return gr_mag3_exact(dx, 0.0f, dz);
    case 6: return gr_mag3_exact(0.0f, dy, dz);
    default: return gr_mag3_exact(dx, dy, dz);
    }
}

// TriangularPrism::sign (shape.rs:408-428)
// every product and the difference rounded on its own, as the reference computes them: a point on a face must get the same sign
GR_HD float gr_prism_sign(float u1, float v1, float u2, float v2, float u3, float v3) {
#pragma clang fp contract(off)
    const float a = (u1 - u3) * (v2 - v3), b = (u2 - u3) * (v1 - v3);
    return a - b;
}

// Non-orthogonal boxes (an extension: the reference needs an orthogonal box here, groups.rs:108-110).  Definition: a point is
// inside a shape under periodic boundary conditions iff SOME lattice image of the point lies inside the shape taken as a plain
// (non-periodic) body anchored at its position -- which is what the orthorhombic arithmetic of the reference computes for its
// boxes, where "some image" can be found per axis (`if d < 0 { d += L }`).  In a triclinic cell the lattice vectors couple the
// axes, so the images are enumerated: the difference to the anchor is reduced into the brick about it (k = rint(d / L) along
// c, b, a) and the images i a + j b + k c around it that the body can reach are tested (see gr_shape_inside_tric).
// The boundary conventions are the PBC variants' (Rectangular and Cylinder closed, the prism's height half-open).  Every
// operation is rounded on its own, in this order, on the device and in the oracle alike.
GR_HD bool gr_shape_free_inside(const GrShapeDev &s, float ex, float ey, float ez) {
#pragma clang fp contract(off)
    switch (s.kind) {
    case GR_SH_RECTANGULAR:
        return ex >= 0.0f && ex <= s.a && ey >= 0.0f && ey <= s.b && ez >= 0.0f && ez <= s.c;
    case GR_SH_CYLINDER: {
        const float along = s.orientation == 1 ? ex : (s.orientation == 2 ? ey : ez);
        if (!(along >= 0.0f && along <= s.b)) return false;
        const float planar = s.orientation == 1 ? gr_mag3_exact(0.0f, ey, ez) : (s.orientation == 2 ? gr_mag3_exact(ex, 0.0f, ez) : gr_mag3_exact(ex, ey, 0.0f));
        return planar <= s.a;
    }
    case GR_SH_PRISM: {
        const float along = s.orientation == 1 ? ex : (s.orientation == 2 ? ey : ez);
        if (!(along >= 0.0f && along < s.a)) return false;
        const float x = s.px + ex, y = s.py + ey, z = s.pz + ez;      // the image itself, for the base triangle's signs
        float pu, pv, u1, v1, u2, v2, u3, v3;
        if (s.plane == 4) { pu = x; pv = y; u1 = s.px; v1 = s.py; u2 = s.b2x; v2 = s.b2y; u3 = s.b3x; v3 = s.b3y; }
        else if (s.plane == 5) { pu = x; pv = z; u1 = s.px; v1 = s.pz; u2 = s.b2x; v2 = s.b2z; u3 = s.b3x; v3 = s.b3z; }
        else { pu = y; pv = z; u1 = s.py; v1 = s.pz; u2 = s.b2y; v2 = s.b2z; u3 = s.b3y; v3 = s.b3z; }
        const float d1 = gr_prism_sign(pu, pv, u1, v1, u2, v2), d2 = gr_prism_sign(pu, pv, u2, v2, u3, v3), d3 = gr_prism_sign(pu, pv, u3, v3, u1, v1);
        const bool has_neg = (d1 < 0.0f) || (d2 < 0.0f) || (d3 < 0.0f), has_pos = (d1 > 0.0f) || (d2 > 0.0f) || (d3 > 0.0f);
        return !(has_neg && has_pos);
    }
    }
    return false;
}
GR_HD bool gr_shape_inside_tric(const GrShapeDev &s, float x, float y, float z, const GrBox &b) {
#pragma clang fp contract(off)
    float dx = x - s.px, dy = y - s.py, dz = z - s.pz;
    float k = rintf(dz / b.cz);
    dx = dx - k * b.cx; dy = dy - k * b.cy; dz = dz - k * b.cz;
    k = rintf(dy / b.by);
    dx = dx - k * b.bx; dy = dy - k * b.by;
    k = rintf(dx / b.ax);
    dx = dx - k * b.ax;
    // Which images: every point of the body lies within its REACH of the anchor (rectangle: its diagonal; cylinder: |(radius,
    // height)|; prism: height + the longer base edge from the anchor), and the brick-reduced difference is at most half the
    // brick's diagonal long, so only lattice vectors with |t| <= T = reach + D / 2 can bring the point inside: the loops visit the
    // (k, j, i) whose components can stay below T -- 3 x 5 x 5 or fewer for bodies smaller than an ordinary cell, as many as it
    // takes in a flat one (rounds 1-3 visited |i|, |j|, |k| <= 2 whatever the cell and the body).
    double reach;
    if (s.kind == GR_SH_RECTANGULAR) reach = sqrt((double)s.a * s.a + (double)s.b * s.b + (double)s.c * s.c);
    else if (s.kind == GR_SH_CYLINDER) reach = sqrt((double)s.a * s.a + (double)s.b * s.b);
    else {
        const double e2 = ((double)s.b2x - s.px) * ((double)s.b2x - s.px) + ((double)s.b2y - s.py) * ((double)s.b2y - s.py) + ((double)s.b2z - s.pz) * ((double)s.b2z - s.pz);
        const double e3 = ((double)s.b3x - s.px) * ((double)s.b3x - s.px) + ((double)s.b3y - s.py) * ((double)s.b3y - s.py) + ((double)s.b3z - s.pz) * ((double)s.b3z - s.pz);
        reach = (double)s.a + sqrt(e2 > e3 ? e2 : e3);
    }
    const double T = (reach + 0.5 * sqrt((double)b.ax * b.ax + (double)b.by * b.by + (double)b.cz * b.cz)) * (1.0 + 1e-6);
    const double ia = 1.0 / b.ax, ib = 1.0 / b.by, ic = 1.0 / b.cz;    // (the bounds are generous by 1e-6 T: no division per level needed)
    const int kmax = (int)floor(T * ic);
    for (int kc = -kmax; kc <= kmax; ++kc) {
        const double cyk = (double)kc * b.cy, cxk = (double)kc * b.cx;
        const int jlo = (int)ceil((-T - cyk) * ib), jhi = (int)floor((T - cyk) * ib);
        for (int kb = jlo; kb <= jhi; ++kb) {
            const double x0 = (double)kb * b.bx + cxk;
            const int ilo = (int)ceil((-T - x0) * ia), ihi = (int)floor((T - x0) * ia);
            for (int ka = ilo; ka <= ihi; ++ka) {
                const float tx = ((float)ka * b.ax + (float)kb * b.bx) + (float)kc * b.cx;
                const float ty = (float)kb * b.by + (float)kc * b.cy;
                const float tz = (float)kc * b.cz;
                if (gr_shape_free_inside(s, dx + tx, dy + ty, dz + tz)) return true;
            }
        }
    }
    return false;
}

// how many lattice images gr_shape_inside_tric visits per point at most (host: a body / cell pair whose walk is unreasonable is
// refused with GR_E_UNSUPPORTED_BOX instead of enumerated per atom on the device)
inline double gr_shape_tric_walk(const GrShapeDev &s, const GrBox &b) {
    if (b.ortho || s.kind == GR_SH_SPHERE) return 1.0;
    double reach;
    if (s.kind == GR_SH_RECTANGULAR) reach = sqrt((double)s.a * s.a + (double)s.b * s.b + (double)s.c * s.c);
    else if (s.kind == GR_SH_CYLINDER) reach = sqrt((double)s.a * s.a + (double)s.b * s.b);
    else {
        const double e2 = ((double)s.b2x - s.px) * ((double)s.b2x - s.px) + ((double)s.b2y - s.py) * ((double)s.b2y - s.py) + ((double)s.b2z - s.pz) * ((double)s.b2z - s.pz);
        const double e3 = ((double)s.b3x - s.px) * ((double)s.b3x - s.px) + ((double)s.b3y - s.py) * ((double)s.b3y - s.py) + ((double)s.b3z - s.pz) * ((double)s.b3z - s.pz);
        reach = (double)s.a + sqrt(e2 > e3 ? e2 : e3);
    }
    const double T = reach + 0.5 * sqrt((double)b.ax * b.ax + (double)b.by * b.by + (double)b.cz * b.cz);
    if (!(T == T) || !(b.ax > 0.0f && b.by > 0.0f && b.cz > 0.0f)) return 1.0e300;
    return (2.0 * T / b.cz + 1.0) * (2.0 * T / b.by + 1.0) * (2.0 * T / b.ax + 1.0);
}
#define GR_SHAPE_WALK_MAX 4096.0

template <int NC = GR_MAX_CAND>
GR_HD bool gr_shape_inside_pbc(const GrShapeDev &s, float x, float y, float z, const GrBox &box) {
    if (!box.ortho && s.kind != GR_SH_SPHERE) return gr_shape_inside_tric(s, x, y, z, box);   // (the sphere is the minimum-image distance either way)
    switch (s.kind) {
    case GR_SH_SPHERE:   // :114-116
        return gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, 7, box) < s.a;
    case GR_SH_RECTANGULAR: {   // :169-184
        float dx = gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, 1, box); if (dx < 0.0f) dx += box.ax;
        float dy = gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, 2, box); if (dy < 0.0f) dy += box.by;
        float dz = gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, 3, box); if (dz < 0.0f) dz += box.cz;
        return dx <= s.a && dy <= s.b && dz <= s.c;
    }
    case GR_SH_CYLINDER: {   // :256-275
        float da = gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, s.orientation, box);
        if (da < 0.0f) da += gr_box_len(box, s.orientation);
        return !(da > s.b || gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, s.plane, box) > s.a);
    }
    case GR_SH_PRISM: {   // :435-460 (the base itself is not periodic, the height is)
        float d = gr_distance<NC, true>(x, y, z, s.px, s.py, s.pz, s.orientation, box);
        if (d < 0.0f) d += gr_box_len(box, s.orientation);
        if (d >= s.a) return false;
        float pu, pv, u1, v1, u2, v2, u3, v3;
        if (s.plane == 4) { pu = x; pv = y; u1 = s.px; v1 = s.py; u2 = s.b2x; v2 = s.b2y; u3 = s.b3x; v3 = s.b3y; }
        else if (s.plane == 5) { pu = x; pv = z; u1 = s.px; v1 = s.pz; u2 = s.b2x; v2 = s.b2z; u3 = s.b3x; v3 = s.b3z; }
        else { pu = y; pv = z; u1 = s.py; v1 = s.pz; u2 = s.b2y; v2 = s.b2z; u3 = s.b3y; v3 = s.b3z; }
        const float d1 = gr_prism_sign(pu, pv, u1, v1, u2, v2), d2 = gr_prism_sign(pu, pv, u2, v2, u3, v3), d3 = gr_prism_sign(pu, pv, u3, v3, u1, v1);
        const bool has_neg = (d1 < 0.0f) || (d2 < 0.0f) || (d3 < 0.0f), has_pos = (d1 > 0.0f) || (d2 > 0.0f) || (d3 > 0.0f);
        return !(has_neg && has_pos);
    }
    }
    return false;
}

GR_HD bool gr_shape_inside_naive(const GrShapeDev &s, float x, float y, float z) {
    switch (s.kind) {
    case GR_SH_SPHERE:   // :473-475
        return gr_distance_naive(x, y, z, s.px, s.py, s.pz, 7) < s.a;
    case GR_SH_CYLINDER: {   // :482-487
        const float d = gr_distance_naive(x, y, z, s.px, s.py, s.pz, s.orientation);
        return d >= 0.0f && d < s.b && gr_distance_naive(x, y, z, s.px, s.py, s.pz, s.plane) < s.a;
    }
    case GR_SH_RECTANGULAR: {   // :494-500
        const float dx = x - s.px, dy = y - s.py, dz = z - s.pz;
        return dx >= 0.0f && dx <= s.a && dy >= 0.0f && dy <= s.b && dz >= 0.0f && dz <= s.c;
    }
    }
    return false;   // the reference has no naive triangular prism (rejected on the host)
}

#if defined(__HIPCC__)
// One bit per atom of the selection (ordinal order): has a position and lies inside every shape.  The host turns the
// mask into AtomContainer blocks -- groups are host objects; the mask is n/8 bytes (125 KB per 1e6 atoms).
// The box and the shapes come in by value: scalar loads, SGPR operands.
// (round 5: the box and the shapes through uniform pointers -- by value and indexed at run time, they were copied into
//  816 bytes of scratch memory per lane)
__global__ __launch_bounds__(256) void k_shape_mask(const float *__restrict__ xyz, GrSel sel, const GrBox *__restrict__ boxp, const GrShapeSet *__restrict__ shapesp,
                                                     unsigned long long *__restrict__ mask) {
    const GrBox &box = *boxp;
    const GrShapeSet &shapes = *shapesp;
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool in = false;
    if (j < sel.n) {
        const uint32_t a = sel.contiguous ? sel.start + j : sel.idx[j];
        float x, y, z;
        gr_pos_load(xyz, a, x, y, z);
        in = (x == x);
        for (int q = 0; q < shapes.n && in; ++q)
            in = shapes.naive ? gr_shape_inside_naive(shapes.s[q], x, y, z) : gr_shape_inside_pbc(shapes.s[q], x, y, z, box);
    }
    const unsigned long long bits = __ballot(in);
    if ((threadIdx.x & 63u) == 0) mask[j >> 6] = bits;
}
#endif
