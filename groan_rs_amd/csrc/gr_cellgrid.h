// gr_cellgrid.h -- cut-off pair search through a cell grid built on the device.
//
// Reference: CellGrid::new / neighbors_iter (src/structures/cellgrid.rs:301-409, 434-476: cells = floor(L / size).max(1) per
// axis, atoms wrapped into the box and binned by floor(pos / cell), the 27 neighbouring cells of a point with periodic wrap,
// no cell visited twice) as its consumer uses it (src/system/hbonds.rs:248-265: every candidate from the neighbouring cells,
// self pairs skipped, kept when distance <= max_distance).  The grid only prunes: the result is the set of pairs
// {(i, j): i in g1, j in g2, i != j, distance(x_j, x_i) <= cutoff}, reported by i then j -- the reference leaves the visiting
// order undefined.  Orthogonal boxes as in the reference; triclinic boxes through a grid in fractional coordinates (below).
//
// Device pipeline (all on the context's stream):
//   k_cg_assign   group-2 atoms -> cell id                                        (12 B/atom read)
//   rocprim::radix_sort_pairs  (cell id, atom), stable: atoms of a cell stay in group order
//   k_cg_starts   first sorted position of every cell (binary search per cell)
//   k_cg_pairs<false>  one lane per group-1 atom: count its pairs                 -> rocprim::exclusive_scan
//   k_cg_pairs<true>   the same walk, writing (i, j, d) at the atom's offset, then the lane sorts its own short segment by j
#pragma once
#include <rocprim/rocprim.hpp>
#include "gr_math.h"
#include "gr_kernels.h"

struct GrCellGrid {
    uint32_t nc[3];          // cells per axis (orthorhombic: along x, y, z; triclinic: along the box vectors a, b, c)
    float inv_cell[3];       // orthorhombic: 1 / cell length; triclinic: cells per unit of the fractional coordinate (= nc)
    uint32_t ncells;
    int tric;
};

// cells per axis for a cut-off: the cell is at least (1 + 1e-5) x cutoff thick so that rounding at a cell border can never
// put two atoms within the cut-off two cells apart (the reference bins with the bare cut-off; the pruned set is the same).
// Triclinic boxes (extension; the reference's CellGrid needs an orthogonal box, cellgrid.rs:411-430): the grid lives in
// FRACTIONAL coordinates -- nc_a slabs between lattice planes parallel to (b, c), and so on -- and a slab's thickness is the
// cell's perpendicular height over nc: h_a = V / |b x c|, h_b = V / |c x a|, h_c = V / |a x b| = cz.  Two atoms whose
// minimum-image distance is within the cut-off then sit in the same or in periodically adjacent slabs along every axis, so
// the 27-cell walk of the orthorhombic grid finds them.
static inline GrCellGrid gr_cellgrid_make(const GrBox &b, float cutoff) {
    GrCellGrid g;
    const float cs = cutoff * 1.00001f;
    g.ncells = 1;
    g.tric = b.ortho ? 0 : 1;
    double h[3] = { b.ax, b.by, b.cz };
    if (g.tric) {
        const double ax = b.ax, bx = b.bx, by = b.by, cx = b.cx, cy = b.cy, cz = b.cz, V = ax * by * cz;
        const double bc[3] = { by * cz, -bx * cz, bx * cy - by * cx };          // b x c
        h[0] = V / sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
        h[1] = V / (ax * sqrt(cz * cz + cy * cy));                              // |c x a| = ax sqrt(cy^2 + cz^2)
        h[2] = cz;
    }
    for (int a = 0; a < 3; ++a) {
        float n = floorf((float)h[a] / cs);
        if (!(n >= 1.0f)) n = 1.0f;
        if (n > 1024.0f) n = 1024.0f;    // bounded table: more cells than this only make the lists shorter than one atom
        g.nc[a] = (uint32_t)n;
        g.inv_cell[a] = g.tric ? (float)g.nc[a] : (float)g.nc[a] / (float)h[a];
        g.ncells *= g.nc[a];
    }
    return g;
}

__device__ __forceinline__ void gr_cg_cell_of(float x, float y, float z, const GrBox &box, const GrCellGrid &g, uint32_t (&c)[3]) {
    gr_wrap(x, y, z, box);                                       // pos.wrap(simbox), cellgrid.rs:463
    float p[3] = { x, y, z };
    if (g.tric) {   // fractional coordinates of the wrapped position, taken modulo 1 (the brick is not the parallelepiped)
        const float fc = z / box.cz, fb = (y - fc * box.cy) / box.by, fa = (x - fb * box.bx - fc * box.cx) / box.ax;
        p[0] = fa - floorf(fa); p[1] = fb - floorf(fb); p[2] = fc - floorf(fc);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int k = (int)floorf(p[a] * g.inv_cell[a]);
        k = k < 0 ? 0 : k;
        c[a] = (uint32_t)k % g.nc[a];                            // pos == L lands in cell 0 (rem_euclid, :472-474)
    }
}

__global__ __launch_bounds__(256) void k_cg_assign(const float *__restrict__ xyz, GrSel sel, const GrBox box, const GrCellGrid g,
                                                   uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ bad) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= sel.n) return;
    const uint32_t a = sel.contiguous ? sel.start + k : sel.idx[k];
    float x, y, z;
    gr_pos_load(xyz, a, x, y, z);
    uint32_t c[3] = { 0, 0, 0 };
    if (x != x) atomicMin(bad, k);                              // ordinal of the first atom without position (group order)
    else gr_cg_cell_of(x, y, z, box, g, c);
    keys[k] = (c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0];
    vals[k] = a;
}

// starts[c] = first position in the sorted key array whose key is >= c (starts[ncells] = n)
__global__ __launch_bounds__(256) void k_cg_starts(const uint32_t *__restrict__ sorted_keys, uint32_t n, uint32_t ncells, uint32_t *__restrict__ starts) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c > ncells) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted_keys[mid] < c) lo = mid + 1; else hi = mid; }
    starts[c] = lo;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_cg_pairs(const float *__restrict__ xyz, GrSel s1, const GrBox box, const GrCellGrid g, float cutoff,
                                                  const uint32_t *__restrict__ sorted_atoms, const uint32_t *__restrict__ starts,
                                                  unsigned long long *__restrict__ counts, const unsigned long long *__restrict__ offsets,
                                                  unsigned long long cap, uint32_t *__restrict__ out_i, uint32_t *__restrict__ out_j, float *__restrict__ out_d,
                                                  uint32_t *__restrict__ bad) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= s1.n) return;
    const uint32_t i = s1.contiguous ? s1.start + k : s1.idx[k];
    float x, y, z;
    gr_pos_load(xyz, i, x, y, z);
    if (x != x) { if (!WRITE) { atomicMin(bad + 1, k); counts[k] = 0; } return; }
    uint32_t c[3];
    gr_cg_cell_of(x, y, z, box, g, c);
    // neighbour offsets per axis without visiting a cell twice (CellNeighbors::convert): 3 cells -> -1 0 1, 2 -> 0 1, 1 -> 0
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = g.nc[a] >= 3 ? -1 : 0; hi[a] = g.nc[a] >= 2 ? 1 : 0; }
    unsigned long long n = 0;
    const unsigned long long base = WRITE ? offsets[k] : 0ull;
    for (int dz = lo[2]; dz <= hi[2]; ++dz)
        for (int dy = lo[1]; dy <= hi[1]; ++dy)
            for (int dx = lo[0]; dx <= hi[0]; ++dx) {
                const uint32_t cx = (uint32_t)((int)c[0] + dx + (int)g.nc[0]) % g.nc[0], cy = (uint32_t)((int)c[1] + dy + (int)g.nc[1]) % g.nc[1],
                               cz = (uint32_t)((int)c[2] + dz + (int)g.nc[2]) % g.nc[2];
                const uint32_t cell = (cz * g.nc[1] + cy) * g.nc[0] + cx;
                for (uint32_t q = starts[cell]; q < starts[cell + 1]; ++q) {
                    const uint32_t j = sorted_atoms[q];
                    if (j == i) continue;                                                     // hbonds.rs:250
                    float jx, jy, jz;
                    gr_pos_load(xyz, j, jx, jy, jz);
                    const float d = gr_distance<4, true>(jx, jy, jz, x, y, z, 7, box);              // acceptor.distance(donor), :261
                    if (d > cutoff) continue;                                                 // :262-264
                    if (WRITE && base + n < cap) { out_i[base + n] = i; out_j[base + n] = j; out_d[base + n] = d; }
                    ++n;
                }
            }
    if (!WRITE) { counts[k] = n; return; }
    // my pairs arrived cell by cell; order them by j (short segment: insertion sort in place)
    const unsigned long long m = (base + n <= cap) ? n : (base < cap ? cap - base : 0ull);
    for (unsigned long long a = 1; a < m; ++a) {
        const uint32_t vj = out_j[base + a]; const float vd = out_d[base + a];
        unsigned long long b = a;
        while (b > 0 && out_j[base + b - 1] > vj) { out_j[base + b] = out_j[base + b - 1]; out_d[base + b] = out_d[base + b - 1]; --b; }
        out_j[base + b] = vj; out_d[base + b] = vd;
    }
}
