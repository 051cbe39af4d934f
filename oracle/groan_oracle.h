/*
 * groan_oracle.h -- CPU parity oracle for the groan_rs per-frame geometry path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * (Ladme/groan_rs v0.11.3) CPU algorithm for the hot path: f32 arithmetic, the
 * same pass structure, the same sequential summation order.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (libgroan_hip.so) never links or calls it.
 *
 * Parity status: PINNED for orthorhombic boxes -- checked against the reference's
 * own known-answer tests (tests/test_oracle_golden.py lists file:line for each).
 * Triclinic ("go_tric_*" code paths, taken when v2x/v3x/v3y != 0) has NO reference
 * arithmetic (the reference returns SimBoxError::NotOrthogonal,
 * src/structures/simbox.rs:230-236): PARITY UNPINNED there; those paths are
 * validated against an fp64 brute-force image search in tests/ instead and
 * degenerate bit-for-bit to the orthorhombic code when the off-diagonals are zero.
 *
 * Conventions
 *   box9   : 9 floats in gro order v1x v2y v3z v1y v1z v2x v2z v3x v3y (simbox.rs:13-26)
 *   pos    : strided float[3] records: record i at (char*)pos + i*pos_stride
 *   mass   : strided float records (NULL => every mass is 1.0, i.e. "center of geometry")
 *   idx    : explicit, ordered atom indices of the selection (AtomContainer iteration order)
 *   "no position" / "no mass" (Rust Option::None) are encoded as NaN in x / in the mass.
 *   rotation matrices are written column-major (nalgebra storage order).
 */
#ifndef GROAN_ORACLE_H
#define GROAN_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: numerically identical to include/groan_hip.h */
enum {
    GO_OK = 0,
    GO_E_NO_BOX = 1,
    GO_E_NOT_ORTHOGONAL = 2,
    GO_E_ZERO_BOX = 3,
    GO_E_EMPTY_GROUP = 4,
    GO_E_INCONSISTENT_GROUP = 5,
    GO_E_NO_POSITION = 6,
    GO_E_NO_MASS = 7,
};

/* Dimension (src/structures/dimension.rs:13-23) */
enum { GO_DIM_NONE = 0, GO_DIM_X, GO_DIM_Y, GO_DIM_Z, GO_DIM_XY, GO_DIM_XZ, GO_DIM_YZ, GO_DIM_XYZ };

/* when non-zero, non-orthogonal boxes are rejected with GO_E_NOT_ORTHOGONAL exactly as the
 * reference does (simbox_check, simbox.rs:230-236).  Default 0 = triclinic extension enabled. */
void go_set_strict_orthogonal(int on);
/* 0 (default): long sums in f32, sequentially, exactly like the reference.  1: the same f32 terms summed
 * in double (the reference's own sequential-f32 rounding grows ~S*2^-24 and passes 1e-5 nm beyond a few
 * thousand atoms; large-S GPU parity is checked against mode 1, and mode 0 is bounded against mode 1). */
void go_set_accumulate_f64(int on);

/* ---- scalar/vector primitives (src/structures/vector3d.rs) ---- */
float go_floor_mod(float x, float y);                         /* :28-30  */
float go_wrap_coordinate(float coor, float box_len);          /* :398-417 */
float go_min_image(float dx, float box_len);                  /* :575-592 */
void  go_wrap(float p[3], const float box9[9]);               /* :380-384 (+ triclinic ext) */
void  go_vector_to(const float from[3], const float to[3], const float box9[9], float out[3]); /* :561-569 */
float go_distance(const float a[3], const float b[3], int dim, const float box9[9]);           /* :458-486 */
float go_distance_naive(const float a[3], const float b[3], int dim);                          /* :522-533 */
void  go_box_center(const float box9[9], float out[3]);       /* src/system/mod.rs:298-308 */
int   go_box_is_orthogonal(const float box9[9]);              /* simbox.rs:185-188 */
void  go_box_from_lengths_angles(const float len[3], const float ang_deg[3], float box9[9]); /* simbox.rs:96-123 */

/* ---- AtomContainer (src/structures/container.rs); blocks are inclusive [start,end] ---- */
/* each returns the number of blocks written to out_start/out_end (capacity: n inputs) */
size_t go_container_from_indices(const uint64_t *indices, size_t n, uint64_t n_atoms,
                                 uint64_t *out_start, uint64_t *out_end);     /* :51-104  */
size_t go_container_from_ranges(const uint64_t *start, const uint64_t *end, size_t n, uint64_t n_atoms,
                                uint64_t *out_start, uint64_t *out_end);      /* :122-215 */
size_t go_container_union(const uint64_t *s1, const uint64_t *e1, size_t n1,
                          const uint64_t *s2, const uint64_t *e2, size_t n2,
                          uint64_t *out_start, uint64_t *out_end);            /* :268-276 */
size_t go_container_intersection(const uint64_t *s1, const uint64_t *e1, size_t n1,
                                 const uint64_t *s2, const uint64_t *e2, size_t n2,
                                 uint64_t *out_start, uint64_t *out_end);     /* :278-291 */
uint64_t go_container_n_atoms(const uint64_t *s, const uint64_t *e, size_t n); /* :161-165 */
/* expands into out (capacity go_container_n_atoms); iteration order of next_index :381-411 */
size_t go_container_expand(const uint64_t *s, const uint64_t *e, size_t n, uint64_t *out);
int    go_container_isin(const uint64_t *s, const uint64_t *e, size_t n, uint64_t index); /* :241-258 */

/* ---- centers (src/structures/iterators.rs, src/auxiliary.rs:59-99) ---- */
/* err_index receives the atom index for GO_E_NO_POSITION / GO_E_NO_MASS */
int go_center_naive(const void *pos, size_t pos_stride, const void *mass, size_t mass_stride,
                    const uint64_t *idx, size_t n, float out[3], uint64_t *err_index);      /* :886-903,946-967 */
int go_estimate_center(const void *pos, size_t pos_stride, const void *mass, size_t mass_stride,
                       const uint64_t *idx, size_t n, const float *box9, float out[3],
                       uint64_t *err_index);                                                  /* :1152-1191,1314-1357 */
/* get_center (mass==NULL) / get_com: the centre estimate is ALWAYS the unweighted one (:1405-1407) */
int go_get_center(const void *pos, size_t pos_stride, const void *mass, size_t mass_stride,
                  const uint64_t *idx, size_t n, const float *box9, float out[3],
                  uint64_t *err_index);                                                       /* :1237-1266,1404-1438 */

/* ---- distances (src/system/analysis.rs) ---- */
int go_group_all_distances(const void *pos, size_t pos_stride,
                           const uint64_t *idx1, size_t n1, const uint64_t *idx2, size_t n2,
                           int dim, const float *box9, float *out_row_major,
                           uint64_t *err_index);                                              /* :401-427 */

/* ---- translate / wrap (iterators.rs:1520-1553, atom.rs:498-545) ---- */
int go_translate(void *pos, size_t pos_stride, const uint64_t *idx, size_t n,
                 const float v[3], const float *box9, uint64_t *err_index);
int go_wrap_atoms(void *pos, size_t pos_stride, const uint64_t *idx, size_t n,
                  const float *box9, uint64_t *err_index);
/* atoms_center (weighted=0) / atoms_center_mass (weighted=1): utility.rs:109-185.
 * ref_idx = the reference group, all_idx = all atoms of the system. */
int go_atoms_center(void *pos, size_t pos_stride, const void *mass, size_t mass_stride,
                    const uint64_t *ref_idx, size_t n_ref, const uint64_t *all_idx, size_t n_all,
                    int dim, int weighted, const float *box9, uint64_t *err_index);

/* ---- Kabsch / RMSD (src/system/rmsd.rs) ---- */
/* kabsch_rmsd :547-603.  p,q packed float[n][3]; R column-major; t = centroid_q - centroid_p */
void go_kabsch_rmsd(const float *p, const float *q, const float *w, size_t n,
                    const float centroid_p[3], const float centroid_q[3], float sum_w,
                    float R_colmajor[9], float t[3], float *rmsd);
/* extract_data_from_system :425-446 -> coords packed float[n][3] (shifted+wrapped), box centre */
int go_rmsd_extract(const void *pos, size_t pos_stride, const void *mass, size_t mass_stride,
                    const uint64_t *idx, size_t n, const float *box9,
                    float *coords_out, float box_center_out[3], uint64_t *err_index);
/* calc_rmsd_rot_trans :141-166.  The reference system and the current system each have their own
 * positions / masses / selection / box; weights come from the REFERENCE's masses (:154-155). */
int go_calc_rmsd(const void *ref_pos, size_t ref_pos_stride, const void *ref_mass, size_t ref_mass_stride,
                 const uint64_t *ref_idx, size_t n_ref, const float *ref_box9,
                 const void *cur_pos, size_t cur_pos_stride, const void *cur_mass, size_t cur_mass_stride,
                 const uint64_t *cur_idx, size_t n_cur, const float *cur_box9,
                 float R_colmajor[9], float *rmsd, uint64_t *err_index, uint64_t counts[2]);
/* fit_structure :508-528: every atom in all_idx is moved; group COM is recomputed with get_com */
int go_fit_structure(void *pos, size_t pos_stride, const void *mass, size_t mass_stride,
                     const uint64_t *grp_idx, size_t n_grp, const uint64_t *all_idx, size_t n_all,
                     const float *box9, const float ref_group_com[3], const float R_colmajor[9]);
/* calc_rmsd_and_fit :131-139 (current system mutated only on success) */
int go_calc_rmsd_and_fit(const void *ref_pos, size_t ref_pos_stride, const void *ref_mass, size_t ref_mass_stride,
                         const uint64_t *ref_idx, size_t n_ref, const float *ref_box9,
                         void *cur_pos, size_t cur_pos_stride, const void *cur_mass, size_t cur_mass_stride,
                         const uint64_t *cur_idx, size_t n_cur, const uint64_t *all_idx, size_t n_all,
                         const float *cur_box9, float *rmsd, uint64_t *err_index, uint64_t counts[2]);

/* 3x3 helper used by go_kabsch_rmsd: R = U diag(1,1,sign det(U V^T)) V^T of H (row-major in, col-major out) */
void go_kabsch_rotation(const float H_rowmajor[9], float R_colmajor[9]);

/* ---- CPU baseline ("reference CPU path" stand-in) ----
 * Runs calc_rmsd_and_fit on n_frames frames with frames assigned round-robin to n_threads threads
 * exactly as src/system/parallel.rs:424-448 (thread t takes frames t, t+T, ...).
 * layout 0: "faithful" -- every frame is first scattered into a 232-byte AoS atom record array
 *           (position Option + mass Option inside the record, src/structures/atom.rs:23-71) the way
 *           update_system does (molly_xtc.rs:294-307), and the reference's pass structure with its
 *           per-pass heap allocations runs on that.
 * layout 1: "soa" -- same arithmetic directly on the packed arrays.
 * frames: packed float[n_frames][n_atoms][3]; fitted frames are written back in place.
 * Reference-side data is cached once like RMSDConverterAnalyzer::new (rmsd.rs:186-203).
 * Returns the busiest worker's seconds spent inside the per-frame analysis (frame ingest and the
 * AoS scatter of update_system are not timed, as decode/H2D are not on the GPU side). */
double go_baseline_rmsd_fit(float *frames, size_t n_frames, size_t n_atoms,
                            const float *ref_xyz, const float *masses,
                            const float *box9, int n_threads, int layout, float *rmsd_out);

/* ---- geometry selection: src/structures/shape.rs:110-185,252-276,431-461 (PBC), :466-505 (naive);
 * Group::apply_geometry / apply_geometries src/structures/group.rs:119-175 ---- */
enum { GO_SHAPE_SPHERE = 1, GO_SHAPE_RECTANGULAR = 2, GO_SHAPE_CYLINDER = 3, GO_SHAPE_TRIANGULAR_PRISM = 4 };
typedef struct go_shape {
    int kind;
    float position[3];        /* sphere centre / box origin / centre of the cylinder base / base1 of the prism */
    float size[3];            /* sphere: radius ; rectangular: x y z ; cylinder: radius height ; prism: height */
    float base2[3], base3[3]; /* prism */
    int orientation, plane;   /* GO_DIM_* ; cylinder and prism */
} go_shape;
/* TriangularPrism::new (:343-378): derives orientation / plane; returns 0, or 1 / 2 where the reference panics
 * ("does not lie in xy, xz, nor yz plane" / "can not be constructed") */
int go_shape_prism_init(go_shape *s, const float b1[3], const float b2[3], const float b3[3], float height);
int go_shape_inside(const go_shape *s, const float point[3], const float box9[9]);
int go_shape_inside_naive(const go_shape *s, const float point[3]);   /* -1: the reference has no naive prism */
/* atoms of idx[] (in order) that have a position and lie inside ALL shapes; returns the count */
size_t go_group_from_geometries(const void *pos, size_t pos_stride, const uint64_t *idx, size_t n, const float *box9,
                                const go_shape *shapes, size_t n_shapes, int naive, uint64_t *out_idx);

/* ---- cut-off pair search: what a CellGrid walk + distance filter produces (src/structures/cellgrid.rs:301-409 as used by
 * src/system/hbonds.rs:248-265: candidates from the neighbouring cells, kept when distance <= cutoff, self pairs skipped).
 * Brute force over all pairs here -- the grid only prunes, it never changes the set.  Pairs come out by i (order of idx1)
 * then j (order of idx2).  Returns the number of pairs (writes at most max_pairs). */
size_t go_pairs_within(const void *pos, size_t pos_stride, const uint64_t *idx1, size_t n1, const uint64_t *idx2, size_t n2,
                       const float *box9, float cutoff, size_t max_pairs, uint64_t *out_i, uint64_t *out_j, float *out_d);

#ifdef __cplusplus
}
#endif
#endif
