/*
 * groan_hip.h -- C ABI of libgroan_hip.so: the MI355X (gfx950) per-frame geometry engine that
 * replaces groan_rs's CPU hot path (PBC distances, centres of geometry/mass, Kabsch RMSD / RMSD-fit,
 * translate / wrap / centre).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Each entry point names the reference interface it replaces (file:line relative to the groan_rs
 * v0.11.3 root).  INTEGRATION.md shows the Rust `extern "C"` block + safe wrappers that bind these,
 * in the style of the reference's existing xdrfile FFI (src/io/xdrfile.rs:27-120).
 *
 * Model
 *   gr_ctx   device mirror of one `System` (src/system/mod.rs:38-73): n_atoms, masses, named groups
 *            (AtomContainer block lists), and `n_slots` resident frames (positions float[n][3] + box).
 *            Not re-entrant; distinct contexts are independent (one per worker / per GPU), like the
 *            per-thread System clones of traj_iter_map_reduce (src/system/parallel.rs:236).
 *   slot     one frame resident in HBM.  A trajectory reader uploads decoded frames into slots
 *            (gr_frame_upload takes the rvec[n] + box exactly as xdrfile/molly deliver them,
 *            src/io/xdrfile.rs:28-36, src/io/xtc_io/molly_xtc.rs:294-307).
 *   plan     cached reference-side RMSD data = RMSDConverterAnalyzer (src/system/rmsd.rs:170-203).
 *
 * Conventions
 *   box9     gro order v1x v2y v3z v1y v1z v2x v2z v3x v3y (src/structures/simbox.rs:13-26); NULL = no box
 *   Option   a missing position / mass (Rust None) is NaN in x / in the mass (the library stores such an atom with NaN in y and z as well:
 *            gr_frame_download returns NaN, NaN, NaN for it whatever y and z the caller had sent)
 *   matrices rotation matrices are column-major (nalgebra storage)
 *   status   every call returns an int: 0 = OK (xdrfile convention, src/io/xtc_io/xdrfile_xtc.rs:63-83)
 *   Non-orthogonal boxes: the reference rejects them (SimBoxError::NotOrthogonal,
 *   src/structures/simbox.rs:230-236).  By default this library computes with the triclinic extension
 *   described in DESIGN.md; gr_ctx_set_strict_orthogonal(ctx,1) restores the reference's error.
 */
#ifndef GROAN_HIP_H
#define GROAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gr_ctx gr_ctx;
typedef struct gr_rmsd_plan gr_rmsd_plan;
typedef struct gr_xtc gr_xtc;

/* status codes; 1..7 map onto the reference's error enums (src/errors.rs) */
enum {
    GR_OK = 0,
    GR_E_NO_BOX = 1,             /* SimBoxError::DoesNotExist            errors.rs:556-566 */
    GR_E_NOT_ORTHOGONAL = 2,     /* SimBoxError::NotOrthogonal (strict mode only)           */
    GR_E_ZERO_BOX = 3,           /* reference panics: vector3d.rs:402-404,576-578           */
    GR_E_EMPTY_GROUP = 4,        /* GroupError::EmptyGroup / RMSDError::EmptyGroup          */
    GR_E_INCONSISTENT_GROUP = 5, /* RMSDError::InconsistentGroup(name, n_ref, n_cur)        */
    GR_E_NO_POSITION = 6,        /* PositionError::NoPosition(index)     errors.rs:570-574 */
    GR_E_NO_MASS = 7,            /* MassError::NoMass(index)             errors.rs:578-582 */
    GR_E_GROUP_NOT_FOUND = 8,    /* GroupError::NotFound / RMSDError::NonexistentGroup      */
    GR_E_OUT_OF_RANGE = 9,       /* AtomError::OutOfRange(index)         errors.rs:290-305 */
    GR_E_INVALID_ARG = 10,       /* bad slot / NULL pointer / size mismatch (caller bug)    */
    GR_E_GROUP_EXISTS = 11,      /* GroupError::AlreadyExistsWarning                        */
    GR_E_HIP = 12,               /* HIP runtime failure: see gr_last_error                  */
    GR_E_NO_DEVICE = 13,         /* no usable gfx950 device: the library has NO CPU fallback */
    GR_E_UNSUPPORTED_BOX = 14,   /* non-orthogonal box that needs more than 16 +- pairs of lattice vectors in the minimum-image table
                                    (a FLAT cell: one box vector much shorter than the skew of the others); ordinary triclinic,
                                    dodecahedral and octahedral cells need 4-8.  Refused rather than answered approximately. */
    GR_E_IO = 15,                /* ReadTrajError::FileNotFound / read failure              */
    GR_E_FORMAT = 16,            /* ReadTrajError::NotXtc / FrameNotFound (corrupt stream)  */
    GR_E_INVALID_NAME = 17       /* GroupError::InvalidName (auxiliary.rs:37-51)            */
};

/* Dimension (src/structures/dimension.rs:13-23) */
enum { GR_DIM_NONE = 0, GR_DIM_X, GR_DIM_Y, GR_DIM_Z, GR_DIM_XY, GR_DIM_XZ, GR_DIM_YZ, GR_DIM_XYZ };

/* centre kinds: group_get_center_naive / group_estimate_center / group_get_center and the _com
 * variants (src/system/analysis.rs:52-320) */
enum { GR_CENTER_NAIVE = 0, GR_CENTER_ESTIMATE = 1, GR_CENTER_PBC = 2 };

/* ---------------------------------------------------------------- library / context */
const char *gr_version(void);
const char *gr_status_string(int status);
int gr_device_count(int *count);

/* System::new / System::clone per worker.  device = HIP ordinal.  n_slots resident frames. */
gr_ctx *gr_ctx_create(int device, uint64_t n_atoms, uint32_t n_slots, int *status);
void gr_ctx_destroy(gr_ctx *ctx);
const char *gr_last_error(const gr_ctx *ctx);
/* detail of the last GR_E_NO_POSITION / GR_E_NO_MASS / GR_E_OUT_OF_RANGE (atom index) and of
 * GR_E_INCONSISTENT_GROUP (n_ref, n_cur): what the reference carries inside its error variants */
uint64_t gr_last_error_index(const gr_ctx *ctx);
void gr_last_error_counts(const gr_ctx *ctx, uint64_t counts[2]);
int gr_ctx_set_strict_orthogonal(gr_ctx *ctx, int on);
uint64_t gr_n_atoms(const gr_ctx *ctx);
uint32_t gr_n_slots(const gr_ctx *ctx);
int gr_sync(gr_ctx *ctx);

/* Atom::set_mass for all atoms (src/structures/atom.rs).  masses[n_atoms], NaN = no mass. */
int gr_set_masses(gr_ctx *ctx, const float *masses, uint64_t n);

/* ---------------------------------------------------------------- AtomContainer (host, bit-exact)
 * Pure functions mirroring src/structures/container.rs; blocks are inclusive [start,end].
 * Each returns the number of blocks written (capacity needed: the number of inputs). */
size_t gr_container_from_indices(const uint64_t *indices, size_t n, uint64_t n_atoms,
                                 uint64_t *out_start, uint64_t *out_end);          /* container.rs:51-104  */
size_t gr_container_from_ranges(const uint64_t *start, const uint64_t *end, size_t n, uint64_t n_atoms,
                                uint64_t *out_start, uint64_t *out_end);           /* container.rs:122-215 */
size_t gr_container_union(const uint64_t *s1, const uint64_t *e1, size_t n1,
                          const uint64_t *s2, const uint64_t *e2, size_t n2,
                          uint64_t *out_start, uint64_t *out_end);                 /* container.rs:268-276 */
size_t gr_container_intersection(const uint64_t *s1, const uint64_t *e1, size_t n1,
                                 const uint64_t *s2, const uint64_t *e2, size_t n2,
                                 uint64_t *out_start, uint64_t *out_end);          /* container.rs:278-291 */
uint64_t gr_container_n_atoms(const uint64_t *s, const uint64_t *e, size_t n);     /* container.rs:161-165 */
size_t gr_container_expand(const uint64_t *s, const uint64_t *e, size_t n, uint64_t *out); /* :381-411 */
int gr_container_isin(const uint64_t *s, const uint64_t *e, size_t n, uint64_t index);     /* :241-258 */
/* May this block list index a system of n_atoms atoms?  GR_OK, or GR_E_OUT_OF_RANGE with *bad_index = the offending index.
 * AtomContainer::from_indices never range-checks its smallest index (container.rs:66): from_indices([n], n) is the block
 * (n, n) and the reference panics on the first access through it.  Every entry point that takes a selection (groups,
 * gr_sel_*) applies this check and creates / computes nothing when it fails -- no kernel bounds-checks a selection. */
int gr_container_validate(const uint64_t *s, const uint64_t *e, size_t n, uint64_t n_atoms, uint64_t *bad_index);

/* ---------------------------------------------------------------- groups (src/system/groups.rs)
 * System::group_create_from_ranges / _from_indices.  "all" exists from creation (System::new).
 * Creating an existing name overwrites it and returns GR_E_GROUP_EXISTS (the reference's warning).
 * A block outside [0, n_atoms) (see gr_container_validate) is GR_E_OUT_OF_RANGE + gr_last_error_index; no group is created. */
int gr_group_create_from_ranges(gr_ctx *ctx, const char *name, const uint64_t *start,
                                const uint64_t *end_inclusive, size_t n_ranges);
int gr_group_create_from_indices(gr_ctx *ctx, const char *name, const uint64_t *indices, size_t n);
int gr_group_remove(gr_ctx *ctx, const char *name);
int gr_group_exists(const gr_ctx *ctx, const char *name);
/* the context's groups, for front ends that match group names (the selection language's `group r'...'`): */
uint64_t gr_group_count(const gr_ctx *ctx);
int gr_group_name(const gr_ctx *ctx, uint64_t i, char *name, size_t capacity);
int gr_group_n_atoms(const gr_ctx *ctx, const char *name, uint64_t *n);            /* group_get_n_atoms */
int gr_group_n_blocks(const gr_ctx *ctx, const char *name, size_t *n_blocks);
int gr_group_blocks(const gr_ctx *ctx, const char *name, uint64_t *out_start, uint64_t *out_end);

/* ---------------------------------------------------------------- frames
 * TrajRead::update_system (src/io/traj_read.rs:160-186; molly_xtc.rs:294-307; xdrfile_xtc.rs:88-104):
 * positions as the xtc readers deliver them (rvec[n_atoms], 12-byte records) + box.  The copy runs on the
 * context's own copy stream, asynchronously when xyz is pinned host memory (gr_host_alloc): kernels on other
 * slots keep running (double buffering); any later call that touches `slot` waits for it, and the copy itself
 * waits for kernels that still read the slot.  Keep xyz valid until gr_frame_upload_wait / gr_sync. */
int gr_frame_upload(gr_ctx *ctx, uint32_t slot, const float *xyz, const float *box9);
/* block until the upload into `slot` has left the host buffer (the staging buffer may then be reused) */
int gr_frame_upload_wait(gr_ctx *ctx, uint32_t slot);
int gr_frame_download(gr_ctx *ctx, uint32_t slot, float *xyz);        /* blocking */
int gr_frame_set_box(gr_ctx *ctx, uint32_t slot, const float *box9); /* System::set_box / reset_box (NULL) */
int gr_frame_get_box(const gr_ctx *ctx, uint32_t slot, float box9[9]); /* GR_E_NO_BOX if none */
int gr_frame_copy(gr_ctx *ctx, uint32_t dst_slot, uint32_t src_slot);
/* pinned host staging buffers for the decode -> H2D double buffer */
void *gr_host_alloc(size_t bytes);
void gr_host_free(void *p);

/* ---------------------------------------------------------------- centres
 * System::group_get_center_naive / group_estimate_center / group_get_center (weighted = 0) and
 * group_get_com_naive / group_estimate_com / group_get_com (weighted = 1): analysis.rs:52-320 over
 * iterators.rs:886-967,1152-1191,1237-1266,1314-1357,1404-1438.
 * Check order as in the reference: group exists -> non-empty -> box -> positions/masses. */
int gr_group_center(gr_ctx *ctx, uint32_t slot, const char *group, int kind, int weighted, float out[3]);
/* GR_CENTER_PBC (get_center / get_com) of a contiguous group of at least `min_atoms` atoms is computed in ONE pass over the
 * frame instead of the reference's two dependent ones (Bai-Breen estimate, then the unwrapped mean): images about the
 * group's first atom + a proof that they are the images the reference would unwrap to.  Frames whose centre lies within the
 * proof's bound of a cell face additionally run the estimate pass (it selects the periodic copy the result lies in), frames
 * where the proof fails (groups wider than half a box) both passes -- one masked launch over the batch either way.
 * min_atoms = 0 switches the one-pass path off; default 4096.  gr_center_fallbacks counts the
 * frames that needed those extra passes since the context was created. */
int gr_ctx_set_center_onepass_min(gr_ctx *ctx, uint32_t min_atoms);
uint64_t gr_center_fallbacks(const gr_ctx *ctx);

/* ---------------------------------------------------------------- distances
 * System::group_distance analysis.rs:348-360; atoms_distance :459-471; group_all_distances :401-427
 * (row-major n1 x n2, signed for 1-D dims).  out_host may be pinned or pageable. */
int gr_group_distance(gr_ctx *ctx, uint32_t slot, const char *group1, const char *group2, int dim, float *out);
int gr_atoms_distance(gr_ctx *ctx, uint32_t slot, uint64_t index1, uint64_t index2, int dim, float *out);
int gr_group_all_distances(gr_ctx *ctx, uint32_t slot, const char *group1, const char *group2, int dim,
                           float *out_host, size_t out_capacity_floats);
/* same, result left in HBM (freed by the context; valid until the next call of this function);
 * *out_dev receives the device pointer, for consumers that keep working on the GPU */
int gr_group_all_distances_device(gr_ctx *ctx, uint32_t slot, const char *group1, const char *group2, int dim,
                                  float **out_dev, uint64_t *n1, uint64_t *n2);
/* the matrices of `n_frames` consecutive slots in one launch and one synchronisation (a trajectory loop of
 * group_all_distances over resident frames): matrix f starts at *out_dev + f * n1 * n2.  The caller bounds the memory
 * through n_frames (4 * n1 * n2 bytes each).  status_out[f] (may be NULL) is the frame's status -- a failed frame's
 * matrix is undefined -- and the return value is the first frame's error, as for the other batch calls. */
int gr_group_all_distances_batch_device(gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *group1, const char *group2, int dim,
                                        float **out_dev, uint64_t *n1, uint64_t *n2, int *status_out);
/* group_all_distances + what its callers do with the matrix (analysis.rs:401-427; :1420-1451 take its maximum and minimum), WITHOUT the
 * matrix: the kernels compute the same tiles, bit for bit, and only the reduction reaches memory (a 1e4 x 1e4 matrix is 400 MB per frame).
 *   GR_PD_MIN / GR_PD_MAX     out = float[n_frames][per_row ? n1 : 1]: the smallest / largest entry of every row, or of the matrix
 *                             (1-D dimensions are signed, as in the matrix)
 *   GR_PD_COUNT_BELOW         out = uint64_t[n_frames][per_row ? n1 : 1]: entries < param
 *   GR_PD_HIST                out = uint64_t[n_frames][nbins]: entries d with 0 <= d, bin (uint32_t)(d * (float)nbins / param) < nbins, i.e.
 *                             nbins (<= 4096) bins over [0, param); per_row must be 0
 * Errors and statuses as gr_group_all_distances_batch_device; the results equal the same reduction of that call's matrix exactly. */
enum { GR_PD_MIN = 1, GR_PD_MAX = 2, GR_PD_COUNT_BELOW = 3, GR_PD_HIST = 4 };
int gr_group_all_distances_reduce(gr_ctx *ctx, uint32_t slot, const char *group1, const char *group2, int dim, int op, int per_row, float param,
                                  uint32_t nbins, void *out, size_t out_capacity_bytes);
int gr_group_all_distances_reduce_batch(gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *group1, const char *group2, int dim, int op,
                                        int per_row, float param, uint32_t nbins, void *out, size_t out_capacity_bytes, int *status_out);
/* copy `bytes` from a device pointer this library handed out (e.g. *out_dev above) into host memory, after the
 * context's stream has drained */
int gr_device_read(gr_ctx *ctx, const void *dev, void *host, size_t bytes);

/* ---------------------------------------------------------------- translate / wrap / centre
 * System::atoms_translate / group_translate (modifying.rs:45-75), atoms_wrap / group_wrap (:201-222),
 * atoms_center / atoms_center_mass (utility.rs:109-185).  group == NULL means all atoms. */
int gr_group_translate(gr_ctx *ctx, uint32_t slot, const char *group, const float v[3]);
int gr_group_wrap(gr_ctx *ctx, uint32_t slot, const char *group);
int gr_atoms_center(gr_ctx *ctx, uint32_t slot, const char *reference_group, int dim, int weighted);

/* ---------------------------------------------------------------- anonymous selections: the iterator-level surface
 * The reference's atom iterators carry an AtomContainer (inclusive index blocks, container.rs:23-31) and a box, no name
 * (System::group_iter / atoms_iter / selection_iter, src/system/iterating.rs:43-140; unions and intersections of containers,
 * iterators.rs:1563-1604).  These entry points take the blocks directly -- (start[], end_inclusive[], n_blocks), exactly the
 * AtomBlock layout -- and compute what the iterator traits compute, with THEIR error behaviour, which differs from the
 * System-level calls in one place: an EMPTY selection is not an error, its centre is (NaN, NaN, NaN)
 * (iterators.rs:1186-1188,1259-1261).  Blocks are validated like groups (gr_container_validate: GR_E_OUT_OF_RANGE).
 *   gr_sel_center          AtomIterable::get_center_naive / get_com_naive (:886-967), AtomIteratorWithBox::estimate_center /
 *                          get_center / estimate_com / get_com (:1152-1438); kind / weighted as gr_group_center
 *   gr_sel_translate/wrap  MutAtomIteratorWithBox::translate / wrap (:1520-1553)
 *   gr_sel_all_distances   the double loop of group_all_distances (analysis.rs:414-424) over two iterators, row-major n1 x n2
 *   gr_sel_filter_geometry AtomIteratorWithBox::filter_geometry / ImmutableAtomIterable::filter_geometry_naive (:994-1004,
 *                          1094-1105): the atoms of the selection that have a position and lie inside every shape, returned as
 *                          the blocks of a new container (*n_out_blocks of them, *n_out_atoms atoms; pass NULL buffers to count) */
int gr_sel_center(gr_ctx *ctx, uint32_t slot, const uint64_t *start, const uint64_t *end_inclusive, size_t n_blocks, int kind, int weighted, float out[3]);
int gr_sel_translate(gr_ctx *ctx, uint32_t slot, const uint64_t *start, const uint64_t *end_inclusive, size_t n_blocks, const float v[3]);
int gr_sel_wrap(gr_ctx *ctx, uint32_t slot, const uint64_t *start, const uint64_t *end_inclusive, size_t n_blocks);
int gr_sel_all_distances(gr_ctx *ctx, uint32_t slot, const uint64_t *start1, const uint64_t *end1, size_t n_blocks1,
                         const uint64_t *start2, const uint64_t *end2, size_t n_blocks2, int dim, float *out_host, size_t out_capacity_floats);
struct gr_shape;
int gr_sel_filter_geometry(gr_ctx *ctx, uint32_t slot, const uint64_t *start, const uint64_t *end_inclusive, size_t n_blocks,
                           const struct gr_shape *shapes, size_t n_shapes, int naive,
                           uint64_t *out_start, uint64_t *out_end, size_t capacity_blocks, size_t *n_out_blocks, uint64_t *n_out_atoms);

/* ---------------------------------------------------------------- RMSD / RMSD-fit
 * gr_calc_rmsd / gr_calc_rmsd_and_fit = System::calc_rmsd / calc_rmsd_and_fit (rmsd.rs:75-166):
 * reference and current frame are two (context, slot) pairs on the same device (they may be the same
 * context); the group must exist in both; weights are the REFERENCE's masses (rmsd.rs:154-155).
 * R (optional, may be NULL) = optimal rotation, column-major. */
int gr_calc_rmsd(gr_ctx *ctx, uint32_t slot, gr_ctx *reference, uint32_t ref_slot, const char *group,
                 float *rmsd, float *R_colmajor9);
int gr_calc_rmsd_and_fit(gr_ctx *ctx, uint32_t slot, gr_ctx *reference, uint32_t ref_slot,
                         const char *group, float *rmsd);

/* RMSDConverterAnalyzer::new (rmsd.rs:186-203): extract + cache the reference side once. */
gr_rmsd_plan *gr_rmsd_plan_create(gr_ctx *reference, uint32_t ref_slot, gr_ctx *target,
                                  const char *group, int *status);
void gr_rmsd_plan_destroy(gr_rmsd_plan *plan);
/* FrameAnalyze::analyze (rmsd.rs:228-236) on `n_frames` consecutive slots in one batch.
 * rmsd_out[n_frames]; status_out[n_frames] per-frame status (may be NULL); R_out optional [n][9].
 * Returns the first non-OK per-frame status (error detail = that frame), else GR_OK. */
int gr_rmsd_batch(gr_rmsd_plan *plan, uint32_t first_slot, uint32_t n_frames,
                  float *rmsd_out, int *status_out, float *R_out);
/* FrameConvertAnalyze::convert_analyze (rmsd.rs:238-251): RMSD + in-place fit of ALL atoms of each
 * frame; a frame whose analysis fails is left unmodified (rmsd.rs:91). */
int gr_rmsd_fit_batch(gr_rmsd_plan *plan, uint32_t first_slot, uint32_t n_frames,
                      float *rmsd_out, int *status_out);
/* The same in two halves, for pipelines that keep the GPU busy while the host decodes / uploads the next frames:
 * begin issues every launch for `n_frames` (<= 1024) consecutive slots and returns without waiting; end waits and
 * delivers the results.  One batch in flight per CONTEXT; between the two calls the context accepts gr_frame_upload /
 * gr_frame_upload_wait / gr_xtc_read_frames_device / gr_trr_read_frames_device into slots OUTSIDE the batch and gr_host_*
 * only (uploads run on the copy stream beside the kernels) -- every other call, and an upload into one of the batch's own
 * slots (gr_rmsd_batch_end may still need those frames: it redoes the ones whose image proof failed), returns
 * GR_E_INVALID_ARG and changes nothing. */
int gr_rmsd_batch_begin(gr_rmsd_plan *plan, uint32_t first_slot, uint32_t n_frames, int fit);
int gr_rmsd_batch_end(gr_rmsd_plan *plan, float *rmsd_out, int *status_out, float *R_out);
/* number of frames of the last batch that left the single-pass path for the multi-pass exact path */
uint32_t gr_rmsd_plan_last_fallbacks(const gr_rmsd_plan *plan);
/* force the multi-pass exact path (parity testing of both paths) */
int gr_rmsd_plan_force_exact(gr_rmsd_plan *plan, int on);
/* Launch geometry and path selection of the batched RMSD calls.  Nothing in the library reads the environment for these:
 * results of a call depend on its arguments and on what the caller set here, never on the caller's environment.
 *   GR_TUNE_SUB_BATCH  frames per sums -> fit launch group (1 .. 1024, default 256)
 *   GR_TUNE_CHUNKS     workgroups per frame of the reduction kernels (0 = automatic)
 *   GR_TUNE_FIT_WGS    workgroups per frame of the fit kernel (0 = automatic: one 256-atom tile per wave)
 *   GR_TUNE_FUSE       1 (default): the sums kernel's last workgroup per frame closes the frame; 0: separate finalize launch
 *   GR_TUNE_TWO_PASS   1 (default): RMSD-fit = sums pass + fit pass that evaluates the rmsd; 0: closed-form single-pass rmsd
 *   GR_TUNE_RESIDENT   RMSD-fit as ONE pass over HBM, the frame waiting on chip for its rotation (gr_resident.h: one launch per
 *                      segment whose workgroups wait for one another; needs n_atoms <= ~1.04e6 on MI355X).  1 (default): when the
 *                      frames in flight (GR_TUNE_RESIDENT_STREAMS) fill at least 10/16 of the chip, the selection is at least 45 %
 *                      of the system and every stream gets 16 frames of the call or more; 0: never; 2: whenever it
 *                      fits.  One such launch runs per device and process at a time (a context that finds the device taken uses
 *                      the two-pass path); a launch whose workgroups do not all get onto the chip (a device shared with another
 *                      process) leaves without touching a frame and the segment runs on the two-pass path.  Same results as the two-pass path up to the order of the partial sums.
 *                      A launch in which a wait runs out of patience (a workgroup made no progress for seconds) is aborted from inside:
 *                      frames it had completed keep their results, frames nobody had touched are redone on the two-pass path, and a
 *                      frame that was caught half fitted -- possible only in the instant of the abort -- is reported as GR_E_HIP
 *                      in status_out with its index in gr_last_error_index (gr_ctx_stat counts aborts and redone frames).
 *   GR_TUNE_RESIDENT_STREAMS frames that fill half of the chip or less run as several frame STREAMS side by side in one resident
 *                      launch (stream s of S owns frames s, s + S, ... of the segment and its own share of the CUs).  0 (default):
 *                      as many as fit, up to 32, when GR_TUNE_RESIDENT is 1 (each stream needs 16 frames of the segment), one when
 *                      it is 2; 1 .. 32: at most so many.  Results do not depend on the number of streams.
 *   GR_TUNE_RESIDENT_FILL    sixteenths of the chip (1 .. 16, default 1; 10 until round 5) the streams of a launch must fill together
 *                      for GR_TUNE_RESIDENT = 1 to choose the pass
 *   GR_TUNE_RESIDENT_WG_GROUPS  4-atom groups per streaming workgroup of the resident pass: 0 (default) = 1024, two per lane; 64 ..
 *                      1024 in steps of 64 cuts a frame into more, smaller workgroups.  An experiment's knob: a turn of the pass is
 *                      bound by instruction issue and latency, not by the work per CU -- 768 groups instead of 1024 gave 5 % at
 *                      600 000 atoms, where frame streams give 40 % (DESIGN.md "Frame streams").  Same results to rounding.
 *   GR_TUNE_PAIRDIST_SYMMETRIC  1 (default): the all-pairs matrix of a group with ITSELF computes the tiles on and above the
 *                      diagonal and writes each of them twice (the mirror image transposed on chip); 0: every element on its own.
 *                      Same bits either way.
 *   GR_TUNE_RESIDENT_GROUPS  retired (round 3 removed the one-group shape of the resident pass): only the value 2 is accepted
 *   GR_TUNE_RMSD_FAST  1 (default): the RMSD WITHOUT fit (gr_rmsd_batch, gr_calc_rmsd) of a contiguous mass-weighted selection of at
 *                      least GR_TUNE_RMSD_FAST_MIN atoms runs as the fit path's sums pass with the closed-form RMSD's sums kept as short
 *                      f32 chains widened to fp64; a frame whose rmsd is too close to the rounding of those sums (a rigid copy of the
 *                      reference) is redone by the exact-product pass, counted in GR_STAT_RMSD_EXACT_REDOS.  0: always the exact pass
 *   GR_TUNE_RMSD_FAST_MIN  smallest selection (atoms, default 16384) that takes it
 *   GR_TUNE_RESIDENT_METRO_NS  the resident pass's METRONOME (gr_resident.h): the period, in nanoseconds per turn, at which the launch's
 *                      row requests sweep each frame in address order.  1 (default) = off, the waves run free; 0 = chosen and kept up to date
 *                      by the library from what its own launches report (GR_STAT_RES_METRO_PERIOD_NS / _LAST_TURN_NS / _LATE_PERMILLE: the
 *                      default for most of round 5 -- it bought 3 % while the pass was paced by memory, nothing since, and could settle on a
 *                      slow period after a cold first launch); 100 .. 1 000 000 = this period.  Results do not depend on it.
 *   GR_TUNE_RESIDENT_FIT_LAST  order of a turn of the resident pass: 1 = the fit of frame i - 6, then the sums of frame i (rounds 2-4);
 *                      2 = the sums first (the frame's record is needed later and published earlier: one more turn for the finalizers);
 *                      0 (default) = sums first when the streaming workgroups fill 9/10 of the chip.  Same results either way.
 *   GR_TUNE_STREAM_WGS_PER_CU  workgroups per CU of the grid-launched read-modify-write streams (translate / wrap / centre, the two-pass fit):
 *                      1 .. 8, 0 (default) = the library's choice.  A copy is fastest with 20-32 KiB of loads in flight per CU.
 *   GR_TUNE_CENTER_RESIDENT  1 (default): gr_atoms_center_batch about a contiguous reference group of at least 15 % of the system (3 % in non-orthogonal cells) runs as ONE pass
 *                      over HBM where the resident pass can take it (gr_resident.h MODE 1: every frame read once, written once -- 24 instead of
 *                      36 bytes per atom); 0: always the two passes (centre estimate, then translate + wrap).  Same bits either way.
 *   GR_TUNE_TRANSLATE_ROWS  1 (default): translate / wrap / centring of a contiguous selection in an ORTHORHOMBIC cell runs one 16-byte load and
 *                      store per lane, a 256-atom tile per workgroup (every coordinate wraps on its own there); 0: the three-rows-per-lane walk
 *                      that non-orthogonal cells and scattered selections take.  Same bits either way. */
enum { GR_TUNE_SUB_BATCH = 1, GR_TUNE_CHUNKS = 2, GR_TUNE_FIT_WGS = 3, GR_TUNE_FUSE = 4, GR_TUNE_TWO_PASS = 5, GR_TUNE_RESIDENT = 6, GR_TUNE_RESIDENT_GROUPS = 7,
       GR_TUNE_RESIDENT_STREAMS = 8, GR_TUNE_RESIDENT_FILL = 9, GR_TUNE_PAIRDIST_SYMMETRIC = 10, GR_TUNE_RESIDENT_WG_GROUPS = 11,
       GR_TUNE_RMSD_FAST = 12, GR_TUNE_RMSD_FAST_MIN = 13, GR_TUNE_RESIDENT_METRO_NS = 18, GR_TUNE_RESIDENT_FIT_LAST = 19, GR_TUNE_STREAM_WGS_PER_CU = 20, GR_TUNE_CENTER_RESIDENT = 21, GR_TUNE_TRANSLATE_ROWS = 22,
       GR_TUNE_MASKED_SELECTIONS = 15 /* 1 (default): a scattered selection that covers at least an eighth of the atoms between its first and its last one (>= 4096
                                         atoms) also gets a bit mask, and RMSD / RMSD-fit / get_com read its span coalesced instead of gathering it atom by atom;
                                         0: groups created afterwards keep to their index lists.  Same results to rounding. */,
       GR_TUNE_XTC_DEVICE_ENCODE = 16 /* 1 (default): gr_xtc_write_slots compresses outputs of >= 200 000 atoms (frames x atoms) on the device -- the same bytes as the
                                         host encoder, only the compressed stream crosses PCIe and no host thread encodes (GR_STAT_XTC_DEVICE_FRAMES counts the
                                         frames); 0: host threads encode (host_threads of the call).  2.0 / 2.5 k frames/s of 5e5 atoms (water-like / one dense chain)
                                         against 1.4-1.6 / 2.1-2.8 k for 16 host encoders: the rate the file takes on the test box (tools/xtc_write_bench.py) */,
       GR_TUNE_RMSD_FAST_SIGMAS = 14 /* multiples (default 6) of the pass's own rounding estimate a frame's rmsd must stand clear of to be kept; 0 keeps every frame: calibration runs only (tools/rmsd_calibrate.py) */,
       GR_TUNE_SMALL_CALLS = 17 /* largest selection (atoms; default 4096, 0 = never) for which ONE-frame calls -- gr_group_center, gr_sel_center, gr_rmsd /
                                    gr_rmsd_batch of one frame -- run as one single-wave dispatch whose result the host reads out of coherent host
                                    memory instead of a chain of launches, two small copies and a stream synchronisation (gr_small.h: 36 -> ~10 us for the
                                    centre of mass, 55 -> ~14 us for the RMSD of a 363-atom selection); same closing arithmetic, sums in another order */,
       GR_TUNE_TEST_RESIDENT_NO_START = 100 /* tests: the next resident launch behaves as if its workgroups never got onto the chip */,
       GR_TUNE_TEST_RESIDENT_ABORT_AT = 101 /* tests: the next resident launch is aborted from inside when it reaches this frame of its segment (< 0: never) */ };
int gr_ctx_set_tuning(gr_ctx *ctx, int key, int64_t value);
/* Read-only facts about a context and its device, and counters of what the batched RMSD path did since the context was created
 * (unknown key: GR_E_INVALID_ARG).
 *   GR_STAT_N_CUS                  compute units of the device
 *   GR_STAT_RES_MAX_WGS            workgroups of the resident pass the device holds at once (0: the pass cannot run / was switched off)
 *   GR_STAT_RES_LAUNCHES           resident launches that ran to the end
 *   GR_STAT_RES_HANDSHAKE_MISSES   resident launches that closed themselves at the start handshake (the segment then took the two-pass path)
 *   GR_STAT_RES_ABORTS             resident launches in which a wait ran out of patience (see gr_rmsd_fit_batch)
 *   GR_STAT_RES_REDONE_FRAMES      frames of such launches that were still untouched and were redone on the two-pass path
 *   GR_STAT_RES_LAST_STREAMS       frame streams of the context's last resident launch (GR_TUNE_RESIDENT_STREAMS)
 *   GR_STAT_RES_METRO_PERIOD_NS    the metronome period the last resident launch ran with (0: it ran free)
 *   GR_STAT_RES_LAST_TURN_NS       what a turn of that launch took on the device clock (first slot to the last wave's exit, per turn)
 *   GR_STAT_RES_LATE_PERMILLE      thousandths of its metronome slots that waves reached more than a quarter period late
 *   GR_STAT_RES_SCLK_MHZ           shader clock that launch ran at (shader-clock ticks over device-clock ticks of its first workgroup's walk)
 *   GR_STAT_CENTER_RES_LAUNCHES    resident atoms_center launches that started (GR_TUNE_CENTER_RESIDENT)
 *   GR_STAT_CENTER_RES_REDONE      frames those launches handed back to the two passes (an atom without position or mass, sums that are not finite, an abort) */
enum { GR_STAT_N_CUS = 1, GR_STAT_RES_MAX_WGS = 2, GR_STAT_RES_LAUNCHES = 3, GR_STAT_RES_HANDSHAKE_MISSES = 4, GR_STAT_RES_ABORTS = 5, GR_STAT_RES_REDONE_FRAMES = 6, GR_STAT_RES_LAST_STREAMS = 7,
       GR_STAT_RMSD_FAST_FRAMES = 8 /* frames of RMSD-without-fit calls closed by the f32-chain pass (GR_TUNE_RMSD_FAST) */,
       GR_STAT_RMSD_EXACT_REDOS = 9 /* ... and frames that pass handed back to the exact-product pass */,
       GR_STAT_XTC_DEVICE_FRAMES = 10 /* frames gr_xtc_write_slots compressed on the device (GR_TUNE_XTC_DEVICE_ENCODE) */,
       GR_STAT_SMALL_CALLS = 11 /* one-frame calls answered by a single-wave dispatch (GR_TUNE_SMALL_CALLS) */,
       GR_STAT_SMALL_SYNC_FALLBACKS = 12 /* ... of which the host gave up polling for the result (20 ms) and synchronised the stream instead */,
       GR_STAT_RES_METRO_PERIOD_NS = 13, GR_STAT_RES_LAST_TURN_NS = 14, GR_STAT_RES_LATE_PERMILLE = 15, GR_STAT_RES_SCLK_MHZ = 16,
       GR_STAT_CENTER_RES_LAUNCHES = 17, GR_STAT_CENTER_RES_REDONE = 18 };
int gr_ctx_stat(const gr_ctx *ctx, int key, uint64_t *value);

/* ---------------------------------------------------------------- text front end: gro structures, ndx index groups (host side)
 * read_gro (src/io/gro_io/structure.rs:120-231, gro_io/mod.rs:21-72) and Groups::from_ndx (src/io/ndx_io.rs:104-230): what is
 * needed to run the path from files -- positions, box, atom / residue names and numbers, index groups.  A failed parse returns
 * GR_E_IO (file not found) or GR_E_FORMAT with *detail_code = the reference's error variant and `detail` = its payload
 * (the offending line / number), exactly as the reference's tests pin them. */
enum { GR_PARSE_OK = 0, GR_PARSE_FILE_NOT_FOUND, GR_PARSE_LINE_NOT_FOUND, GR_PARSE_LINE, GR_PARSE_ATOM_LINE, GR_PARSE_BOX_LINE,
       GR_PARSE_UNSUPPORTED_BOX, GR_PARSE_INVALID_FLOAT, GR_PARSE_GROUP_NAME, GR_PARSE_INVALID_ATOM_INDEX };
typedef struct gr_structure gr_structure;
int gr_gro_read(const char *path, gr_structure **out, int *detail_code, char *detail, size_t detail_cap);
void gr_structure_free(gr_structure *s);
uint64_t gr_structure_n_atoms(const gr_structure *s);
const char *gr_structure_title(const gr_structure *s);
int gr_structure_box(const gr_structure *s, float box9[9]);          /* GR_E_NO_BOX when the box line is all zeros */
int gr_structure_positions(const gr_structure *s, float *xyz);       /* [n][3] */
int gr_structure_velocities(const gr_structure *s, float *vel);      /* [n][3], NaN rows for atoms without velocity */
int gr_structure_atom(const gr_structure *s, uint64_t i, uint64_t *resid, uint64_t *atomid, char resname[8], char atomname[8]);
typedef struct gr_ndx gr_ndx;
int gr_ndx_read(const char *path, uint64_t n_atoms, gr_ndx **out, int *detail_code, char *detail, size_t detail_cap);
void gr_ndx_free(gr_ndx *x);
size_t gr_ndx_n_groups(const gr_ndx *x);                             /* groups in file order (a repeated name appears again) */
const char *gr_ndx_group_name(const gr_ndx *x, size_t g);
size_t gr_ndx_group_size(const gr_ndx *x, size_t g);                 /* indices as written: 0-based, file order, duplicates kept */
int gr_ndx_group_indices(const gr_ndx *x, size_t g, uint64_t *out);
/* System::read_ndx: create the groups on a context (Group::from_indices); a group with an invalid name is skipped and counted,
 * a repeated / already existing name is overwritten and counted (the reference's two warnings) */
int gr_ndx_install(const gr_ndx *x, gr_ctx *ctx, size_t *n_invalid_names, size_t *n_duplicate_names);

/* ---------------------------------------------------------------- cut-off pair search (cell grid on the device)
 * What a CellGrid walk with a distance filter produces (CellGrid::new + neighbors_iter, src/structures/cellgrid.rs:301-409,
 * as src/system/hbonds.rs:248-265 uses them): all pairs (i in group1, j in group2, i != j) with distance(x_j, x_i) <= cutoff,
 * ordered by i then j (the reference leaves the order undefined).  The grid (cells of at least `cutoff`, periodic 27-cell
 * neighbourhood) is built on the device; it replaces the S1 x S2 distance matrix when only near pairs are wanted.
 * Needs a box (GR_E_NO_BOX); non-orthogonal boxes -- which the reference's CellGrid rejects (cellgrid.rs:411-430), and so does
 * this call in strict-orthogonal mode (GR_E_NOT_ORTHOGONAL) -- are binned in fractional coordinates with the triclinic
 * minimum-image distance as the filter; cutoff > 0 (GR_E_INVALID_ARG, the reference's
 * CellGridError::InvalidCellSize), positions for every atom of both groups (GR_E_NO_POSITION + index: group2 is checked
 * first, as the grid is built before it is queried).  *n_pairs receives the number of pairs found; at most max_pairs are
 * written -- when *n_pairs > max_pairs call again with larger buffers (the buffers may be NULL to just count). */
int gr_group_pairs_within(gr_ctx *ctx, uint32_t slot, const char *group1, const char *group2, float cutoff, uint64_t max_pairs,
                          uint32_t *i_out, uint32_t *j_out, float *dist_out, uint64_t *n_pairs);

/* ---------------------------------------------------------------- per-frame analyses over a batch of slots
 * The calls above for `n_frames` consecutive slots in ONE set of launches and one read-back (a trajectory loop of
 * group_get_com / atoms_center / atoms_wrap over resident frames; per-frame calls cost a launch + a synchronisation each).
 * Every frame is judged on its own: status_out[f] (may be NULL) receives the frame's status, a failed frame yields NaN /
 * is left untouched, and the return value is the first frame's error (message and index as for the single calls). */
int gr_group_center_batch(gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *group, int kind, int weighted,
                          float *out /* [n_frames][3] */, int *status_out);
int gr_group_translate_batch(gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *group, const float v[3], int *status_out);
int gr_group_wrap_batch(gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *group, int *status_out);
int gr_atoms_center_batch(gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *ref_group, int dim, int weighted, int *status_out);

/* ---------------------------------------------------------------- trr reader
 * TrrReader / TrrFrameData (src/io/trr_io.rs:30-135 over the vendored xdrfile's read_trr): an own reader of the GROMACS trr
 * format.  gr_trr_open indexes the file (frames are random-access, reads are thread-safe).  A frame carries any of positions /
 * velocities / forces / box, in single or double precision; *sections = 1 positions | 2 velocities | 4 forces | 8 box.
 * gr_trr_read_frame converts to f32 into the caller's buffers (any may be NULL); a section the frame does not carry reads as
 * zeros, which the reference turns into "no position / velocity / force" (trr_io.rs:101-126).  gr_trr_read_frames_device
 * copies the raw big-endian position sections of a batch of frames to the GPU and converts them there, straight into the
 * slots (asynchronous like gr_frame_upload; all-zero positions become the NaN marker of a missing position). */
typedef struct gr_trr gr_trr;
gr_trr *gr_trr_open(const char *path, int *status);
void gr_trr_close(gr_trr *trr);
uint64_t gr_trr_n_atoms(const gr_trr *trr);
uint64_t gr_trr_n_frames(const gr_trr *trr);
int gr_trr_frame_info(const gr_trr *trr, uint64_t frame, uint64_t *step, float *time, float *lambda, float box9[9], int *sections, int *double_precision);
int gr_trr_read_frame(const gr_trr *trr, uint64_t frame, float *xyz, float *velocities, float *forces, float box9[9], uint64_t *step, float *time, float *lambda);
int gr_trr_read_frames_device(const gr_trr *trr, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *ctx, uint32_t first_slot,
                              uint64_t *steps, float *times);
/* TrrWriter (src/io/trr_io.rs:441-520 over xdrfile's write_trr): single precision, box + the sections whose array is not
 * NULL (the reference always writes positions, velocities and forces, zeros where an atom has none), byte for byte the
 * reference writer's frames.  A position without value (NaN in x) is written as zeros; box9 NULL writes a zero matrix. */
typedef struct gr_trr_writer gr_trr_writer;
gr_trr_writer *gr_trr_writer_open(const char *path, int *status);
int gr_trr_writer_close(gr_trr_writer *w);
int gr_trr_write_frame(gr_trr_writer *w, uint64_t n_atoms, const float *xyz, const float *velocities, const float *forces, const float box9[9],
                       int64_t step, float time, float lambda);

/* ---------------------------------------------------------------- xtc writer (host side; the step after calc_rmsd_and_fit)
 * XtcWriter::new / write_frame (src/io/xtc_io/mod.rs:256-331 over xdrfile's write_xtc): the library's own encoder, byte for
 * byte the stream the reference writes for the same coordinates (the reference's golden fitted trajectories are files).
 * Atoms without a position are written as the origin, a frame without box as a zero matrix (xdrfile.rs:188-200).
 * gr_xtc_write_slots streams device frames out: D2H of slot k+1 overlaps the encoding of slot k (`host_threads` encoders,
 * 0 = up to 16), frames are written in slot order; `group` = NULL writes every atom, else the group's atoms in its order
 * (xtc_group_writer_init).  A coordinate whose value x precision does not fit the format's 32-bit integers (or a NaN in
 * y / z) makes the call fail with GR_E_OUT_OF_RANGE and that frame is not written (gr_xtc_write_slots stops there: the frames
 * before it are in the file, as after a loop of write_frame calls) -- the reference's C writer prints "Internal
 * overflow compressing coordinates." and converts anyway (undefined behaviour, external/xdrfile/xdrfile.c:1025-1030). */
typedef struct gr_xtc_writer gr_xtc_writer;
gr_xtc_writer *gr_xtc_writer_open(const char *path, int *status);
int gr_xtc_writer_close(gr_xtc_writer *w);
int gr_xtc_write_frame(gr_xtc_writer *w, uint64_t n_atoms, const float *xyz, const float box9[9], int64_t step, float time, float precision);
int gr_xtc_write_slots(gr_xtc_writer *w, gr_ctx *ctx, uint32_t first_slot, uint32_t n_frames, const char *group,
                       const int64_t *steps, const float *times, float precision, int host_threads);

/* ---------------------------------------------------------------- geometry selection
 * Shape::inside of Sphere / Rectangular / Cylinder / TriangularPrism (src/structures/shape.rs:110-185,252-276,431-461),
 * the PBC-free NaiveShape variants (:466-505), and System::group_create_from_geometry / _geometries
 * (src/system/groups.rs:94-188 over Group::apply_geometries, src/structures/group.rs:119-175): the new group holds, in
 * the source group's order, the atoms that have a position and lie inside EVERY shape.  Needs a box (GR_E_NO_BOX); the reference
 * also needs it orthogonal (groups.rs:108-110: GR_E_NOT_ORTHOGONAL in strict-orthogonal mode) -- otherwise a non-orthogonal box
 * takes the image-enumeration extension (some lattice image of the atom lies inside the shape); the name must be valid (GR_E_INVALID_NAME: empty, or one of
 * '"&|!@()<>= , auxiliary.rs:37-51); an existing group is replaced and GR_E_GROUP_EXISTS returned (the reference's
 * AlreadyExistsWarning).  The predicate runs on the device (one ballot bit per atom), the blocks are built on the host. */
enum { GR_SHAPE_SPHERE = 1, GR_SHAPE_RECTANGULAR = 2, GR_SHAPE_CYLINDER = 3, GR_SHAPE_TRIANGULAR_PRISM = 4 };
#define GR_MAX_SHAPES 8
typedef struct gr_shape {
    int kind;
    float position[3];         /* sphere centre / box origin / centre of the cylinder's base / first vertex of the prism's base */
    float size[3];             /* sphere: radius ; rectangular: x y z ; cylinder: radius, height ; prism: height */
    float base2[3], base3[3];  /* prism: the other two vertices of the base */
    int orientation, plane;    /* GR_DIM_* ; cylinder and prism (filled by the constructors) */
} gr_shape;
int gr_shape_sphere(gr_shape *s, const float position[3], float radius);                                   /* Sphere::new :95-97 */
int gr_shape_rectangular(gr_shape *s, const float position[3], float x, float y, float z);                 /* Rectangular::new :141-143 */
/* Cylinder::new :211-229 -- orientation GR_DIM_X / _Y / _Z, anything else is GR_E_INVALID_ARG (the reference panics) */
int gr_shape_cylinder(gr_shape *s, const float position[3], float radius, float height, int orientation);
/* TriangularPrism::new :343-378 -- GR_E_INVALID_ARG where the reference panics (base not in a coordinate plane / degenerate) */
int gr_shape_triangular_prism(gr_shape *s, const float base1[3], const float base2[3], const float base3[3], float height);
/* Shape::inside / NaiveShape::inside_naive for one point (host; box9 may be NULL when naive) */
int gr_shape_inside(const gr_shape *s, const float point[3], const float box9[9], int naive, int *inside);
int gr_group_create_from_geometries(gr_ctx *ctx, uint32_t slot, const char *name, const char *source_group,
                                    const gr_shape *shapes, size_t n_shapes, int naive);

/* ---------------------------------------------------------------- xtc reader (host side)
 * The stage in front of the path: XtcReader (src/io/xtc_io/mod.rs, molly_xtc.rs:96-308, xdrfile_xtc.rs:42-104).
 * gr_xtc_open indexes the file (frame offsets, steps, times, boxes) so frames are random-access -- what the
 * reference's with_range / with_step / per-thread skipping (traj_read.rs:215-246, parallel.rs:424-448) need -- and
 * gr_xtc_read_frame is thread-safe: one frame per host thread, decoded straight into the caller's buffer
 * (pinned memory from gr_host_alloc feeds gr_frame_upload without another copy).
 * step is the u32-wrapped simulation step (xdrfile_xtc.rs:98-100); box9 in gro order (xdrfile.rs:170-187). */
gr_xtc *gr_xtc_open(const char *path, int *status);
void gr_xtc_close(gr_xtc *xtc);
uint64_t gr_xtc_n_atoms(const gr_xtc *xtc);
uint64_t gr_xtc_n_frames(const gr_xtc *xtc);
int gr_xtc_frame_info(const gr_xtc *xtc, uint64_t frame, uint64_t *step, float *time, float box9[9], float *precision);
int gr_xtc_read_frame(const gr_xtc *xtc, uint64_t frame, float *xyz, float box9[9], uint64_t *step, float *time, float *precision);
/* The same for a batch, unpacked ON THE DEVICE: frames first_frame, first_frame + frame_step, ... (n_frames of them) land
 * in slots first_slot .. first_slot + n_frames - 1 of `ctx` with their boxes, like n gr_frame_upload calls -- but what
 * crosses PCIe is the compressed bit stream (~3.5 B/atom instead of 12) and the host only skims the group framing
 * (`host_threads` workers, 0 = one per frame up to 16), the unpacking of all frames runs as one kernel on the copy stream.
 * Asynchronous like gr_frame_upload (kernels that use the slots are ordered behind it; gr_frame_upload_wait to block).
 * Results are bit-identical to gr_xtc_read_frame.  steps / times may be NULL. */
int gr_xtc_read_frames_device(const gr_xtc *xtc, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *ctx,
                              uint32_t first_slot, int host_threads, uint64_t *steps, float *times);

/* ---------------------------------------------------------------- frame-sharded map-reduce over several GPUs
 * System::traj_iter_map_reduce (src/system/parallel.rs:208-481): frames are independent units; worker w of T takes frames
 * w, w + T, ... (parallel.rs:424-448); every worker owns a clone of the System; a shared error flag, polled every 10 frames,
 * stops the others when one fails (parallel.rs:28,453-475) and the call as a whole fails (:288-321); the per-worker results
 * are merged at the end (ParallelTrajData::reduce, :31-49).  Bulk frame data never crosses GPUs: no data-path collective.
 *
 * In-process form (one host process, several GPUs or several workers per GPU): gr_pool_create makes one context per entry of
 * `devices` (System::clone per worker) -- set masses / groups / plans on them through gr_pool_ctx.  gr_pool_map runs `body`
 * for frames 0 .. n_frames - 1 on the workers' own threads (worker w: frames w, w + T, ...); the body loads its frame into its
 * context (gr_frame_upload, gr_xtc_read_frames_device ...: the xtc / trr handles are thread-safe), analyses it and writes
 * `width` floats to `result`, which IS the frame's row of `results` -- the order is restored by construction, the gather is
 * host memory.  A non-zero return is the frame's error: the first one wins, becomes the return value of gr_pool_map
 * (*error_frame = its frame), and the other workers stop at their next flag check; *frames_done counts completed frames. */
typedef struct gr_pool gr_pool;
typedef int (*gr_pool_body)(gr_ctx *ctx, int worker, uint64_t frame, void *user, float *result);
gr_pool *gr_pool_create(const int *devices, int n_workers, uint64_t n_atoms, uint32_t n_slots, int *status);
void gr_pool_destroy(gr_pool *pool);
int gr_pool_size(const gr_pool *pool);
gr_ctx *gr_pool_ctx(gr_pool *pool, int worker);
const char *gr_pool_last_error(const gr_pool *pool);
int gr_pool_map(gr_pool *pool, uint64_t n_frames, gr_pool_body body, void *user, size_t width, float *results,
                uint64_t *frames_done, uint64_t *error_frame);
/* The same with the reference's range / step arguments and its progress printer (traj_iter_map_reduce's start_time, end_time, step,
 * progress_printer, parallel.rs:208-222; frame-index form: a reader's index turns times into frame numbers, gr_xtc_frame_info):
 * the frames visited are first_frame + k * step for k = 0, 1, ... while < end_frame; worker w of T takes k = w, w + T, ... -- it skips
 * w * step frames and then advances by step * T, exactly parallel.rs:425-448 -- and `result` is row k of `results` (rows in visiting
 * order, ceil((end_frame - first_frame) / step) of them).  `body` receives the FRAME number.  `progress` (may be NULL) is the
 * ProgressPrinter: called on worker 0's thread after every frame that worker completes (GR_PROGRESS_RUNNING; the reference
 * attaches the printer to the master thread's iterator only, parallel.rs:417-422) and once after all workers have joined with
 * GR_PROGRESS_COMPLETED and the last frame any worker read, or GR_PROGRESS_FAILED and the frame that failed (:288-321).
 * step = 0 is GR_E_INVALID_ARG (the reference's ReadTrajError::InvalidStep). */
enum { GR_PROGRESS_RUNNING = 0, GR_PROGRESS_COMPLETED = 1, GR_PROGRESS_FAILED = 2 };
typedef void (*gr_pool_progress)(void *progress_user, int status, uint64_t frame, uint64_t frames_done);
int gr_pool_map_range(gr_pool *pool, uint64_t first_frame, uint64_t end_frame, uint64_t step, gr_pool_body body, void *user,
                      size_t width, float *results, gr_pool_progress progress, void *progress_user,
                      uint64_t *frames_done, uint64_t *error_frame);
/* Multi-process form (one process per GPU, e.g. under torchrun / mpirun): an RCCL communicator over xGMI for the two exchanges
 * the path has -- the final gather of the per-frame results and the shared error flag.  Rank 0 calls gr_comm_unique_id and
 * hands the 128 bytes to every rank by whatever the launcher offers (a file, MPI_Bcast, torch.distributed ...); every rank
 * then calls gr_comm_create (collective).  gr_comm_gather_per_frame: `local` holds this rank's frames rank, rank + G, ...
 * ([n_local][width], host memory); every rank receives all n_total rows in frame order (out[f] = shard[f mod G][f div G]): ONE
 * ncclAllGather of ceil(n_total / G) x width floats per rank.  gr_comm_any_error: 1-int ncclAllReduce(MAX) of the flag.
 * RCCL is resolved at run time (gr_comm_library says from where); without it these calls return GR_E_NO_DEVICE. */
typedef struct gr_comm gr_comm;
int gr_comm_unique_id(void *id128);
gr_comm *gr_comm_create(int device, int rank, int world, const void *id128, int *status);
void gr_comm_destroy(gr_comm *comm);
const char *gr_comm_last_error(const gr_comm *comm);
const char *gr_comm_library(void);
/* use this RCCL build instead of the default search (the process's own copy, then librccl.so.1); call before the first gr_comm_* call
 * of the process -- the library is resolved once, and a call that comes after that returns GR_E_INVALID_ARG (thread-safe against a
 * concurrent first gr_comm_* call).  A path that cannot be loaded makes every gr_comm_* call return GR_E_NO_DEVICE and
 * gr_comm_library() say why. */
int gr_comm_set_library(const char *path);
int gr_comm_gather_per_frame(gr_comm *comm, const float *local, uint64_t n_total, size_t width, float *out);
int gr_comm_any_error(gr_comm *comm, int local_flag, int *any);
/* the host half of the gather on its own (no device, no RCCL): `gathered` = G shards of ceil(n_total / G) rows -> frame order */
void gr_shard_deinterleave(const float *gathered, int world, uint64_t n_total, size_t width, float *out);

/* Partial-frame reading = GroupXtcReader / System::group_xtc_iter (src/io/xtc_io/molly_xtc.rs:475-620 over the molly crate):
 * only the atoms of a group change, "all other atoms are left unchanged", the frame's box / step / time are set as usual.  The
 * bit stream is sequential, so everything UP TO the group's last atom is walked -- and nothing behind it: for a group near the
 * start of the structure (a peptide in front of its membrane and water) that is a small fraction of the frame.
 *   gr_xtc_read_frame_prefix        host decode of the first n_prefix atoms (= last atom of the group + 1) into xyz[n_prefix][3];
 *                                   reads only the needed prefix of the frame's stream (*stream_bytes_read, may be NULL)
 *   gr_xtc_read_frames_device_group like gr_xtc_read_frames_device, but the host reads + skims only up to the group's last atom,
 *                                   only that prefix crosses PCIe, and the unpack kernel writes the group's atoms only
 * Both are bit-identical to the full decode on the atoms they deliver. */
int gr_xtc_read_frame_prefix(const gr_xtc *xtc, uint64_t frame, uint64_t n_prefix, float *xyz, float box9[9], uint64_t *step, float *time, float *precision,
                             uint64_t *stream_bytes_read);
int gr_xtc_read_frames_device_group(const gr_xtc *xtc, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *ctx,
                                    uint32_t first_slot, const char *group, int host_threads, uint64_t *steps, float *times);

/* ---------------------------------------------------------------- measurement / synthetic data
 * HIP-event timing on the context's stream (the stream the kernels are launched on). */
int gr_timer_start(gr_ctx *ctx);
int gr_timer_stop(gr_ctx *ctx, float *milliseconds);
/* Per-kernel HIP-event profile of the batched RMSD path, recorded on the context's stream around
 * each launch while enabled: kernel 0 = sums pass (k_sums_pk / k_rmsd_accum), 1 = k_rmsd_finalize (separate launch only), 2 = fit pass (k_fit_pk), 3 = the resident single pass (k_fit_resident).  Enabling resets
 * the counters.  ms_total / launches = average launch duration; frames = frames those launches covered. */
int gr_profile_enable(gr_ctx *ctx, int on);
int gr_profile_read(const gr_ctx *ctx, int kernel, double *ms_total, uint64_t *launches, uint64_t *frames);
/* Seeded synthetic workload of SURVEY.md section 8(d), generated directly in HBM:
 *  gr_synth_reference: n_atoms points uniform in a ball of `radius` about the box centre -> slot
 *  gr_synth_frames   : frame f = R_f (x0 - c) + c + t_f + noise, wrapped into the cell, for
 *                      n_frames consecutive slots; R_f, t_f, noise from a counter-based hash of
 *                      (seed, first_frame_index + f*frame_index_stride, atom): a rank of a G-way
 *                      round-robin shard passes (rank, G).  Every slot inherits the reference slot's box.
 *  gr_synth_uniform  : n_atoms points uniform in the unit cell (config 3). */
int gr_synth_reference(gr_ctx *ctx, uint32_t slot, const float *box9, float radius, uint64_t seed);
int gr_synth_frames(gr_ctx *ctx, uint32_t ref_slot, uint32_t first_slot, uint32_t n_frames,
                    uint64_t first_frame_index, uint64_t frame_index_stride, float noise_sigma, uint64_t seed);
int gr_synth_uniform(gr_ctx *ctx, uint32_t slot, const float *box9, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* GROAN_HIP_H */
