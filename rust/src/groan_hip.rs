//! Bindings of libgroan_hip.so (include/groan_hip.h) for groan_rs, in the style of the crate's existing
//! xdrfile FFI (`src/io/xdrfile.rs:27-120`): raw `extern "C"` declarations, an opaque `#[repr(C)]` handle,
//! an RAII owner with `Drop`, and safe methods returning the crate's own error enums.
//!
//! NOT COMPILED in the build container (no cargo/rustc there).  Feature-gate as `hip` in Cargo.toml and
//! link with `println!("cargo:rustc-link-lib=dylib=groan_hip")` from build.rs.
#![allow(non_camel_case_types)]

use std::ffi::{c_char, c_float, c_int, c_void, CString};
use std::marker::PhantomData;

use crate::errors::{AtomError, GroupError, MassError, PositionError, RMSDError, SimBoxError};
use crate::structures::{container::AtomContainer, dimension::Dimension, simbox::SimBox, vector3d::Vector3D};
use crate::system::System;
use crate::structures::traj_convert::{FrameAnalyze, FrameConvertAnalyze};

#[repr(C)] pub struct gr_ctx { _private: [u8; 0] }
#[repr(C)] pub struct gr_rmsd_plan { _private: [u8; 0] }

pub const GR_OK: c_int = 0;
pub const GR_E_NO_BOX: c_int = 1;
pub const GR_E_NOT_ORTHOGONAL: c_int = 2;
pub const GR_E_ZERO_BOX: c_int = 3;
pub const GR_E_EMPTY_GROUP: c_int = 4;
pub const GR_E_INCONSISTENT_GROUP: c_int = 5;
pub const GR_E_NO_POSITION: c_int = 6;
pub const GR_E_NO_MASS: c_int = 7;
pub const GR_E_GROUP_NOT_FOUND: c_int = 8;

#[repr(C)] pub struct gr_pool { _private: [u8; 0] }
#[repr(C)] pub struct gr_comm { _private: [u8; 0] }
/// `body(ctx, worker, frame, user, result)` of gr_pool_map: non-zero return = the frame's error (the first one wins)
/// `progress(user, status, frame, frames_done)` of gr_pool_map_range: GR_PROGRESS_RUNNING = 0 after every frame of worker 0,
/// then once GR_PROGRESS_COMPLETED = 1 / GR_PROGRESS_FAILED = 2 (ProgressPrinter + ProgressStatus, src/progress.rs)
pub type gr_pool_progress = Option<unsafe extern "C" fn(user: *mut c_void, status: c_int, frame: u64, frames_done: u64)>;
pub type gr_pool_body = Option<unsafe extern "C" fn(ctx: *mut gr_ctx, worker: c_int, frame: u64, user: *mut c_void, result: *mut c_float) -> c_int>;
extern "C" {
    pub fn gr_ctx_create(device: c_int, n_atoms: u64, n_slots: u32, status: *mut c_int) -> *mut gr_ctx;
    pub fn gr_ctx_destroy(ctx: *mut gr_ctx);
    pub fn gr_last_error_index(ctx: *const gr_ctx) -> u64;
    pub fn gr_last_error_counts(ctx: *const gr_ctx, counts: *mut u64);
    pub fn gr_ctx_set_strict_orthogonal(ctx: *mut gr_ctx, on: c_int) -> c_int;
    pub fn gr_set_masses(ctx: *mut gr_ctx, masses: *const c_float, n: u64) -> c_int;
    pub fn gr_group_create_from_ranges(ctx: *mut gr_ctx, name: *const c_char, start: *const u64, end: *const u64, n: usize) -> c_int;
    pub fn gr_frame_upload(ctx: *mut gr_ctx, slot: u32, xyz: *const c_float, box9: *const c_float) -> c_int;
    pub fn gr_frame_download(ctx: *mut gr_ctx, slot: u32, xyz: *mut c_float) -> c_int;
    pub fn gr_group_center(ctx: *mut gr_ctx, slot: u32, group: *const c_char, kind: c_int, weighted: c_int, out: *mut c_float) -> c_int;
    pub fn gr_group_distance(ctx: *mut gr_ctx, slot: u32, g1: *const c_char, g2: *const c_char, dim: c_int, out: *mut c_float) -> c_int;
    pub fn gr_group_all_distances(ctx: *mut gr_ctx, slot: u32, g1: *const c_char, g2: *const c_char, dim: c_int, out: *mut c_float, cap: usize) -> c_int;
    pub fn gr_group_all_distances_batch_device(ctx: *mut gr_ctx, first_slot: u32, n_frames: u32, g1: *const c_char, g2: *const c_char, dim: c_int,
                                               out_dev: *mut *mut c_float, n1: *mut u64, n2: *mut u64, status_out: *mut c_int) -> c_int;
    // group_all_distances + the reduction its callers apply (analysis.rs:1420-1451), without the matrix: GR_PD_MIN = 1, _MAX = 2, _COUNT_BELOW = 3, _HIST = 4
    pub fn gr_group_all_distances_reduce(ctx: *mut gr_ctx, slot: u32, g1: *const c_char, g2: *const c_char, dim: c_int, op: c_int, per_row: c_int, param: c_float,
                                         nbins: u32, out: *mut c_void, out_capacity_bytes: usize) -> c_int;
    pub fn gr_group_all_distances_reduce_batch(ctx: *mut gr_ctx, first_slot: u32, n_frames: u32, g1: *const c_char, g2: *const c_char, dim: c_int, op: c_int,
                                               per_row: c_int, param: c_float, nbins: u32, out: *mut c_void, out_capacity_bytes: usize, status_out: *mut c_int) -> c_int;
    pub fn gr_trr_open(path: *const c_char, status: *mut c_int) -> *mut gr_trr;
    pub fn gr_trr_close(trr: *mut gr_trr);
    pub fn gr_trr_n_atoms(trr: *const gr_trr) -> u64;
    pub fn gr_trr_n_frames(trr: *const gr_trr) -> u64;
    pub fn gr_trr_frame_info(trr: *const gr_trr, frame: u64, step: *mut u64, time: *mut c_float, lambda: *mut c_float, box9: *mut c_float, sections: *mut c_int, double_precision: *mut c_int) -> c_int;
    pub fn gr_trr_read_frame(trr: *const gr_trr, frame: u64, xyz: *mut c_float, vel: *mut c_float, force: *mut c_float, box9: *mut c_float, step: *mut u64, time: *mut c_float, lambda: *mut c_float) -> c_int;
    pub fn gr_trr_writer_open(path: *const c_char, status: *mut c_int) -> *mut gr_trr_writer;
    pub fn gr_trr_writer_close(w: *mut gr_trr_writer) -> c_int;
    pub fn gr_trr_write_frame(w: *mut gr_trr_writer, n_atoms: u64, xyz: *const c_float, vel: *const c_float, force: *const c_float, box9: *const c_float, step: i64, time: c_float, lambda: c_float) -> c_int;
    pub fn gr_trr_read_frames_device(trr: *const gr_trr, first_frame: u64, n_frames: u32, frame_step: u64, ctx: *mut gr_ctx, first_slot: u32, steps: *mut u64, times: *mut c_float) -> c_int;
    pub fn gr_group_count(ctx: *const gr_ctx) -> u64;
    pub fn gr_group_name(ctx: *const gr_ctx, i: u64, name: *mut c_char, capacity: usize) -> c_int;
    pub fn gr_ctx_set_center_onepass_min(ctx: *mut gr_ctx, min_atoms: u32) -> c_int;
    pub fn gr_center_fallbacks(ctx: *const gr_ctx) -> u64;
    pub fn gr_device_read(ctx: *mut gr_ctx, dev: *const c_void, host: *mut c_void, bytes: usize) -> c_int;
    pub fn gr_group_translate(ctx: *mut gr_ctx, slot: u32, group: *const c_char, v: *const c_float) -> c_int;
    pub fn gr_group_wrap(ctx: *mut gr_ctx, slot: u32, group: *const c_char) -> c_int;
    pub fn gr_atoms_center(ctx: *mut gr_ctx, slot: u32, group: *const c_char, dim: c_int, weighted: c_int) -> c_int;
    pub fn gr_rmsd_plan_create(reference: *mut gr_ctx, ref_slot: u32, target: *mut gr_ctx, group: *const c_char, status: *mut c_int) -> *mut gr_rmsd_plan;
    pub fn gr_rmsd_plan_destroy(plan: *mut gr_rmsd_plan);
    pub fn gr_rmsd_batch(plan: *mut gr_rmsd_plan, first_slot: u32, n: u32, rmsd: *mut c_float, status: *mut c_int, rot: *mut c_float) -> c_int;
    pub fn gr_rmsd_fit_batch(plan: *mut gr_rmsd_plan, first_slot: u32, n: u32, rmsd: *mut c_float, status: *mut c_int) -> c_int;
    // anonymous selections = the iterator-level surface (src/structures/iterators.rs:886-1554): AtomContainer blocks in, no group name
    pub fn gr_sel_center(ctx: *mut gr_ctx, slot: u32, start: *const u64, end: *const u64, n_blocks: usize, kind: c_int, weighted: c_int, out: *mut c_float) -> c_int;
    pub fn gr_sel_translate(ctx: *mut gr_ctx, slot: u32, start: *const u64, end: *const u64, n_blocks: usize, v: *const c_float) -> c_int;
    pub fn gr_sel_wrap(ctx: *mut gr_ctx, slot: u32, start: *const u64, end: *const u64, n_blocks: usize) -> c_int;
    pub fn gr_sel_all_distances(ctx: *mut gr_ctx, slot: u32, s1: *const u64, e1: *const u64, n1: usize, s2: *const u64, e2: *const u64, n2: usize,
                                dim: c_int, out: *mut c_float, cap: usize) -> c_int;
    pub fn gr_sel_filter_geometry(ctx: *mut gr_ctx, slot: u32, start: *const u64, end: *const u64, n_blocks: usize, shapes: *const gr_shape, n_shapes: usize,
                                  naive: c_int, out_start: *mut u64, out_end: *mut u64, cap_blocks: usize, n_out_blocks: *mut usize, n_out_atoms: *mut u64) -> c_int;
    pub fn gr_frame_upload_wait(ctx: *mut gr_ctx, slot: u32) -> c_int;
    // geometry selection (src/structures/shape.rs, src/system/groups.rs:94-188)
    pub fn gr_shape_sphere(s: *mut gr_shape, position: *const c_float, radius: c_float) -> c_int;
    pub fn gr_shape_rectangular(s: *mut gr_shape, position: *const c_float, x: c_float, y: c_float, z: c_float) -> c_int;
    pub fn gr_shape_cylinder(s: *mut gr_shape, position: *const c_float, radius: c_float, height: c_float, orientation: c_int) -> c_int;
    pub fn gr_shape_triangular_prism(s: *mut gr_shape, b1: *const c_float, b2: *const c_float, b3: *const c_float, height: c_float) -> c_int;
    pub fn gr_group_create_from_geometries(ctx: *mut gr_ctx, slot: u32, name: *const c_char, source: *const c_char,
                                           shapes: *const gr_shape, n: usize, naive: c_int) -> c_int;
    // xtc frames unpacked on the device straight into frame slots (src/io/xtc_io/*: XtcReader + update_system)
    pub fn gr_xtc_open(path: *const c_char, status: *mut c_int) -> *mut gr_xtc;
    pub fn gr_xtc_close(xtc: *mut gr_xtc);
    pub fn gr_xtc_read_frames_device(xtc: *const gr_xtc, first_frame: u64, n_frames: u32, frame_step: u64, ctx: *mut gr_ctx,
                                     first_slot: u32, host_threads: c_int, steps: *mut u64, times: *mut c_float) -> c_int;
    // asynchronous batches, one-shot calls, batched per-frame calls, tuning, host memory, diagnostics
    pub fn gr_rmsd_batch_begin(plan: *mut gr_rmsd_plan, first_slot: u32, n_frames: u32, fit: c_int) -> c_int;
    pub fn gr_rmsd_batch_end(plan: *mut gr_rmsd_plan, rmsd: *mut c_float, status: *mut c_int, rot: *mut c_float) -> c_int;
    pub fn gr_rmsd_plan_last_fallbacks(plan: *const gr_rmsd_plan) -> u32;
    pub fn gr_calc_rmsd(ctx: *mut gr_ctx, slot: u32, reference: *mut gr_ctx, ref_slot: u32, group: *const c_char, rmsd: *mut c_float, rot: *mut c_float) -> c_int;
    pub fn gr_calc_rmsd_and_fit(ctx: *mut gr_ctx, slot: u32, reference: *mut gr_ctx, ref_slot: u32, group: *const c_char, rmsd: *mut c_float) -> c_int;
    pub fn gr_group_center_batch(ctx: *mut gr_ctx, first_slot: u32, n_frames: u32, group: *const c_char, kind: c_int, weighted: c_int, out: *mut c_float, status: *mut c_int) -> c_int;
    pub fn gr_group_translate_batch(ctx: *mut gr_ctx, first_slot: u32, n_frames: u32, group: *const c_char, v: *const c_float, status: *mut c_int) -> c_int;
    pub fn gr_group_wrap_batch(ctx: *mut gr_ctx, first_slot: u32, n_frames: u32, group: *const c_char, status: *mut c_int) -> c_int;
    pub fn gr_atoms_center_batch(ctx: *mut gr_ctx, first_slot: u32, n_frames: u32, ref_group: *const c_char, dim: c_int, weighted: c_int, status: *mut c_int) -> c_int;
    pub fn gr_ctx_set_tuning(ctx: *mut gr_ctx, key: c_int, value: i64) -> c_int;   // GR_TUNE_* (include/groan_hip.h)
    pub fn gr_host_alloc(bytes: usize) -> *mut c_void;                              // pinned memory: asynchronous gr_frame_upload
    pub fn gr_host_free(p: *mut c_void);
    pub fn gr_sync(ctx: *mut gr_ctx) -> c_int;
    pub fn gr_last_error(ctx: *const gr_ctx) -> *const c_char;
    pub fn gr_status_string(status: c_int) -> *const c_char;
    // traj_iter_map_reduce across GPUs (src/system/parallel.rs:208-481): in one process ...
    pub fn gr_pool_create(devices: *const c_int, n_workers: c_int, n_atoms: u64, n_slots: u32, status: *mut c_int) -> *mut gr_pool;
    pub fn gr_pool_destroy(pool: *mut gr_pool);
    pub fn gr_pool_size(pool: *const gr_pool) -> c_int;
    pub fn gr_pool_ctx(pool: *mut gr_pool, worker: c_int) -> *mut gr_ctx;
    pub fn gr_pool_last_error(pool: *const gr_pool) -> *const c_char;
    pub fn gr_pool_map(pool: *mut gr_pool, n_frames: u64, body: gr_pool_body, user: *mut c_void, width: usize, results: *mut c_float,
                       frames_done: *mut u64, error_frame: *mut u64) -> c_int;
    /// traj_iter_map_reduce's start_time / end_time / step (frame-index form) and its progress printer (parallel.rs:208-222,417-448)
    pub fn gr_pool_map_range(pool: *mut gr_pool, first_frame: u64, end_frame: u64, step: u64, body: gr_pool_body, user: *mut c_void, width: usize,
                             results: *mut c_float, progress: gr_pool_progress, progress_user: *mut c_void, frames_done: *mut u64, error_frame: *mut u64) -> c_int;
    pub fn gr_comm_set_library(path: *const c_char) -> c_int;
    pub fn gr_ctx_stat(ctx: *const gr_ctx, key: c_int, value: *mut u64) -> c_int;   // GR_STAT_* (include/groan_hip.h)
    // ... and one process per GPU (RCCL over xGMI: one final all-gather + the shared error flag)
    pub fn gr_comm_unique_id(id128: *mut c_void) -> c_int;
    pub fn gr_comm_create(device: c_int, rank: c_int, world: c_int, id128: *const c_void, status: *mut c_int) -> *mut gr_comm;
    pub fn gr_comm_destroy(comm: *mut gr_comm);
    pub fn gr_comm_last_error(comm: *const gr_comm) -> *const c_char;
    pub fn gr_comm_library() -> *const c_char;
    pub fn gr_comm_gather_per_frame(comm: *mut gr_comm, local: *const c_float, n_total: u64, width: usize, out: *mut c_float) -> c_int;
    pub fn gr_comm_any_error(comm: *mut gr_comm, local_flag: c_int, any: *mut c_int) -> c_int;
    pub fn gr_shard_deinterleave(gathered: *const c_float, world: c_int, n_total: u64, width: usize, out: *mut c_float);
    // GroupXtcReader (partial-frame reads, molly_xtc.rs:475-560)
    pub fn gr_xtc_read_frame_prefix(xtc: *const gr_xtc, frame: u64, n_prefix: u64, xyz: *mut c_float, box9: *mut c_float, step: *mut u64, time: *mut c_float,
                                    precision: *mut c_float, stream_bytes_read: *mut u64) -> c_int;
    pub fn gr_xtc_read_frames_device_group(xtc: *const gr_xtc, first_frame: u64, n_frames: u32, frame_step: u64, ctx: *mut gr_ctx, first_slot: u32,
                                           group: *const c_char, host_threads: c_int, steps: *mut u64, times: *mut c_float) -> c_int;
}

#[repr(C)] pub struct gr_xtc { _private: [u8; 0] }
#[repr(C)] pub struct gr_trr { _private: [u8; 0] }
#[repr(C)] pub struct gr_trr_writer { _private: [u8; 0] }
/// `gr_shape` of include/groan_hip.h: filled by the `gr_shape_*` constructors from the fields of
/// `Sphere` / `Rectangular` / `Cylinder` / `TriangularPrism` (src/structures/shape.rs:17-68).
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct gr_shape { pub kind: c_int, pub position: [c_float; 3], pub size: [c_float; 3], pub base2: [c_float; 3], pub base3: [c_float; 3],
                      pub orientation: c_int, pub plane: c_int }

/// Device mirror of one `System` (one per worker thread / per GPU, like the clones of `traj_iter_map_reduce`).
pub struct HipSystem { ctx: *mut gr_ctx, n_atoms: usize, _not_sync: PhantomData<*mut ()> }
unsafe impl Send for HipSystem {}

impl Drop for HipSystem { fn drop(&mut self) { unsafe { gr_ctx_destroy(self.ctx) } } }

fn simbox_err(status: c_int) -> SimBoxError {
    if status == GR_E_NOT_ORTHOGONAL { SimBoxError::NotOrthogonal } else { SimBoxError::DoesNotExist }
}

impl HipSystem {
    /// Mirror masses and groups of `system` on `device`; positions follow with `upload`.
    pub fn new(system: &System, device: i32, n_slots: u32) -> Option<Self> {
        let mut st = 0;
        let n = system.get_n_atoms();
        let ctx = unsafe { gr_ctx_create(device, n as u64, n_slots, &mut st) };
        if ctx.is_null() { return None; }
        let masses: Vec<f32> = system.atoms_iter().map(|a| a.get_mass().unwrap_or(f32::NAN)).collect();
        unsafe { gr_set_masses(ctx, masses.as_ptr(), n as u64) };
        let me = HipSystem { ctx, n_atoms: n, _not_sync: PhantomData };
        for (name, group) in system.get_groups().iter() {
            me.group_from_container(name, group.get_atoms());
        }
        Some(me)
    }

    pub fn group_from_container(&self, name: &str, container: &AtomContainer) {
        let (s, e): (Vec<u64>, Vec<u64>) = container.blocks().map(|(a, b)| (a as u64, b as u64)).unzip();
        let cname = CString::new(name).unwrap();
        unsafe { gr_group_create_from_ranges(self.ctx, cname.as_ptr(), s.as_ptr(), e.as_ptr(), s.len()) };
    }

    /// `TrajRead::update_system` for the device copy: `coords` is the `[[f32; 3]]` the xtc readers produce.
    pub fn upload(&self, slot: u32, coords: &[[f32; 3]], simbox: Option<&SimBox>) {
        assert_eq!(coords.len(), self.n_atoms);
        let b9 = simbox.map(|b| [b.v1x, b.v2y, b.v3z, b.v1y, b.v1z, b.v2x, b.v2z, b.v3x, b.v3y]);
        let bp = b9.as_ref().map_or(std::ptr::null(), |b| b.as_ptr());
        unsafe { gr_frame_upload(self.ctx, slot, coords.as_ptr() as *const c_float, bp) };
    }

    fn group_error(&self, status: c_int, name: &str) -> GroupError {
        let idx = unsafe { gr_last_error_index(self.ctx) } as usize;
        match status {
            GR_E_GROUP_NOT_FOUND => GroupError::NotFound(name.to_owned()),
            GR_E_EMPTY_GROUP => GroupError::EmptyGroup(name.to_owned()),
            GR_E_NO_POSITION => GroupError::InvalidPosition(PositionError::NoPosition(idx)),
            GR_E_NO_MASS => GroupError::InvalidMass(MassError::NoMass(idx)),
            s => GroupError::InvalidSimBox(simbox_err(s)),
        }
    }

    /// `System::group_get_com` (src/system/analysis.rs:258-274)
    pub fn group_get_com(&self, slot: u32, name: &str) -> Result<Vector3D, GroupError> {
        let cname = CString::new(name).unwrap();
        let mut out = [0f32; 3];
        match unsafe { gr_group_center(self.ctx, slot, cname.as_ptr(), 2, 1, out.as_mut_ptr()) } {
            GR_OK => Ok(Vector3D::new(out[0], out[1], out[2])),
            s => Err(self.group_error(s, name)),
        }
    }

    /// `System::group_distance` (analysis.rs:348-360)
    pub fn group_distance(&self, slot: u32, g1: &str, g2: &str, dim: Dimension) -> Result<f32, GroupError> {
        let (a, b) = (CString::new(g1).unwrap(), CString::new(g2).unwrap());
        let mut out = 0f32;
        match unsafe { gr_group_distance(self.ctx, slot, a.as_ptr(), b.as_ptr(), dim as c_int, &mut out) } {
            GR_OK => Ok(out),
            s => Err(self.group_error(s, g1)),
        }
    }
}

/// The iterator-level surface on the device: what `AtomIterable` / `AtomIteratorWithBox` / `MutAtomIteratorWithBox`
/// (src/structures/iterators.rs:842-1554) compute for an `AtomContainer`, as ONE call each into the anonymous-selection entry
/// points -- no named group is created.  Obtain one from `HipSystem::iter_container` with the container of
/// `system.group_iter(..)`, `atoms_iter()`, a union / intersection (`container.rs:268-291`) ...
/// Error behaviour is the iterator traits': box first (`AtomError::InvalidSimBox`), then the first atom without position /
/// mass; an empty container yields `Vector3D(NaN, NaN, NaN)`, not an error (iterators.rs:1186-1188).
pub struct HipAtomIterator<'a> { sys: &'a HipSystem, slot: u32, start: Vec<u64>, end: Vec<u64> }

impl HipSystem {
    pub fn iter_container(&self, slot: u32, container: &AtomContainer) -> HipAtomIterator<'_> {
        let (start, end): (Vec<u64>, Vec<u64>) = container.blocks().map(|(a, b)| (a as u64, b as u64)).unzip();
        HipAtomIterator { sys: self, slot, start, end }
    }
    fn atom_error(&self, status: c_int) -> AtomError {
        let idx = unsafe { gr_last_error_index(self.ctx) } as usize;
        match status {
            GR_E_NO_POSITION => AtomError::InvalidPosition(PositionError::NoPosition(idx)),
            GR_E_NO_MASS => AtomError::InvalidMass(MassError::NoMass(idx)),
            9 /* GR_E_OUT_OF_RANGE */ => AtomError::OutOfRange(idx),
            s => AtomError::InvalidSimBox(simbox_err(s)),
        }
    }
}

impl<'a> HipAtomIterator<'a> {
    fn center(&self, kind: c_int, weighted: c_int) -> Result<Vector3D, AtomError> {
        let mut out = [0f32; 3];
        match unsafe { gr_sel_center(self.sys.ctx, self.slot, self.start.as_ptr(), self.end.as_ptr(), self.start.len(), kind, weighted, out.as_mut_ptr()) } {
            GR_OK => Ok(Vector3D::new(out[0], out[1], out[2])),
            s => Err(self.sys.atom_error(s)),
        }
    }
    pub fn get_center_naive(&self) -> Result<Vector3D, AtomError> { self.center(0, 0) }   // iterators.rs:886-903
    pub fn get_com_naive(&self) -> Result<Vector3D, AtomError> { self.center(0, 1) }      // :946-967
    pub fn estimate_center(&self) -> Result<Vector3D, AtomError> { self.center(1, 0) }    // :1152-1191
    pub fn estimate_com(&self) -> Result<Vector3D, AtomError> { self.center(1, 1) }       // :1314-1357
    pub fn get_center(&self) -> Result<Vector3D, AtomError> { self.center(2, 0) }         // :1237-1266
    pub fn get_com(&self) -> Result<Vector3D, AtomError> { self.center(2, 1) }            // :1404-1438
    /// `MutAtomIteratorWithBox::translate` (:1520-1524)
    pub fn translate(&self, v: &Vector3D) -> Result<(), AtomError> {
        let a = [v.x, v.y, v.z];
        match unsafe { gr_sel_translate(self.sys.ctx, self.slot, self.start.as_ptr(), self.end.as_ptr(), self.start.len(), a.as_ptr()) } { GR_OK => Ok(()), s => Err(self.sys.atom_error(s)) }
    }
    /// `MutAtomIteratorWithBox::wrap` (:1548-1553)
    pub fn wrap(&self) -> Result<(), AtomError> {
        match unsafe { gr_sel_wrap(self.sys.ctx, self.slot, self.start.as_ptr(), self.end.as_ptr(), self.start.len()) } { GR_OK => Ok(()), s => Err(self.sys.atom_error(s)) }
    }
    /// `AtomIteratorWithBox::filter_geometry` / `filter_geometry_naive` (:994-1004,1094-1105): the blocks of the filtered container
    pub fn filter_geometry(&self, shape: &gr_shape, naive: bool) -> Result<Vec<(usize, usize)>, AtomError> {
        let cap = self.start.iter().zip(&self.end).map(|(a, b)| (b - a + 1) as usize).sum::<usize>().max(1);
        let (mut os, mut oe) = (vec![0u64; cap], vec![0u64; cap]);
        let (mut nb, mut na) = (0usize, 0u64);
        match unsafe { gr_sel_filter_geometry(self.sys.ctx, self.slot, self.start.as_ptr(), self.end.as_ptr(), self.start.len(), shape, 1, naive as c_int,
                                              os.as_mut_ptr(), oe.as_mut_ptr(), cap, &mut nb, &mut na) } {
            GR_OK => Ok(os[..nb].iter().zip(&oe[..nb]).map(|(a, b)| (*a as usize, *b as usize)).collect()),
            s => Err(self.sys.atom_error(s)),
        }
    }
}

/// Batched form of `HipRmsd` for trajectory loops that can look ahead: `push` stages a frame into the next free slot (one
/// copy out of the `System`, one asynchronous upload), `flush` runs ONE `gr_rmsd_batch` over everything staged -- the
/// library's batched path instead of one launch + synchronisation per frame.
/// THIS is the adapter to use for throughput.  The per-frame `HipRmsd` below keeps the reference's `FrameAnalyze` contract (one
/// result per `analyze` call) and therefore runs batches of ONE frame: a launch + a synchronisation per frame and, for systems
/// that fill the chip, never the single-pass resident kernel (which needs >= 16 frames per call) --
/// about a tenth of the batched rate at 1e6 atoms.  Use a capacity of >= 64 (256-1024 for large systems).
pub struct HipRmsdBatch { inner: HipRmsd, staged: u32, capacity: u32, pinned: Vec<Vec<[f32; 3]>> }
impl HipRmsdBatch {
    pub fn new(reference: &System, target: &System, group: &str, device: i32, capacity: u32) -> Result<Self, RMSDError> {
        let mut inner = HipRmsd::new(reference, target, group, device)?;
        inner.target = HipSystem::new(target, device, capacity).ok_or_else(|| RMSDError::NonexistentGroup(group.to_owned()))?;
        let cname = CString::new(group).unwrap();
        let mut st = 0;
        unsafe { gr_rmsd_plan_destroy(inner.plan) };
        inner.plan = unsafe { gr_rmsd_plan_create(inner._reference.ctx, 0, inner.target.ctx, cname.as_ptr(), &mut st) };
        if inner.plan.is_null() { return Err(rmsd_error(&inner._reference, st, group)); }
        Ok(HipRmsdBatch { inner, staged: 0, capacity, pinned: (0..capacity).map(|_| Vec::new()).collect() })
    }
    pub fn is_full(&self) -> bool { self.staged == self.capacity }
    pub fn push(&mut self, system: &System) {
        let buf = &mut self.pinned[self.staged as usize];
        buf.clear();
        buf.extend(system.atoms_iter().map(|a| a.get_position().map_or([f32::NAN; 3], |p| [p.x, p.y, p.z])));
        self.inner.target.upload(self.staged, buf, system.get_box());
        self.staged += 1;
    }
    pub fn flush(&mut self) -> Result<Vec<f32>, RMSDError> {
        let n = self.staged;
        self.staged = 0;
        let (mut r, mut st) = (vec![0f32; n as usize], vec![0 as c_int; n as usize]);
        match unsafe { gr_rmsd_batch(self.inner.plan, 0, n, r.as_mut_ptr(), st.as_mut_ptr(), std::ptr::null_mut()) } {
            GR_OK => Ok(r),
            s => Err(rmsd_error(&self.inner.target, s, &self.inner.group)),
        }
    }
}

/// `RMSDConverterAnalyzer` (src/system/rmsd.rs:170-251) on the GPU: same trait impls, so
/// `system.xtc_iter(f)?.analyze(HipRmsd::new(..)?)` / `.convert_and_analyze(..)` work unchanged.
pub struct HipRmsd { plan: *mut gr_rmsd_plan, target: HipSystem, _reference: HipSystem, group: String, scratch: Vec<[f32; 3]> }

impl Drop for HipRmsd { fn drop(&mut self) { unsafe { gr_rmsd_plan_destroy(self.plan) } } }

impl HipRmsd {
    pub fn new(reference: &System, target: &System, group: &str, device: i32) -> Result<Self, RMSDError> {
        let r = HipSystem::new(reference, device, 1).ok_or_else(|| RMSDError::NonexistentGroup(group.to_owned()))?;
        let t = HipSystem::new(target, device, 1).ok_or_else(|| RMSDError::NonexistentGroup(group.to_owned()))?;
        let coords: Vec<[f32; 3]> = reference.atoms_iter().map(|a| a.get_position().map_or([f32::NAN; 3], |p| [p.x, p.y, p.z])).collect();
        r.upload(0, &coords, reference.get_box());
        let cname = CString::new(group).unwrap();
        let mut st = 0;
        let plan = unsafe { gr_rmsd_plan_create(r.ctx, 0, t.ctx, cname.as_ptr(), &mut st) };
        if plan.is_null() { return Err(rmsd_error(&r, st, group)); }
        Ok(HipRmsd { plan, target: t, _reference: r, group: group.to_owned(), scratch: Vec::new() })
    }

    fn stage(&mut self, system: &System) {
        self.scratch.clear();
        self.scratch.extend(system.atoms_iter().map(|a| a.get_position().map_or([f32::NAN; 3], |p| [p.x, p.y, p.z])));
        self.target.upload(0, &self.scratch, system.get_box());
    }
}

fn rmsd_error(sys: &HipSystem, status: c_int, group: &str) -> RMSDError {
    let idx = unsafe { gr_last_error_index(sys.ctx) } as usize;
    match status {
        GR_E_GROUP_NOT_FOUND => RMSDError::NonexistentGroup(group.to_owned()),
        GR_E_EMPTY_GROUP => RMSDError::EmptyGroup(group.to_owned()),
        GR_E_NO_POSITION => RMSDError::InvalidPosition(PositionError::NoPosition(idx)),
        GR_E_NO_MASS => RMSDError::InvalidMass(MassError::NoMass(idx)),
        GR_E_INCONSISTENT_GROUP => {
            let mut c = [0u64; 2];
            unsafe { gr_last_error_counts(sys.ctx, c.as_mut_ptr()) };
            RMSDError::InconsistentGroup(group.to_owned(), c[0] as usize, c[1] as usize)
        }
        s => RMSDError::InvalidSimBox(simbox_err(s)),
    }
}

impl FrameAnalyze for HipRmsd {
    type Error = RMSDError;
    type AnalysisResult = f32;
    fn analyze(&mut self, system: &System) -> Result<f32, RMSDError> {
        self.stage(system);
        let (mut r, mut st) = (0f32, 0);
        match unsafe { gr_rmsd_batch(self.plan, 0, 1, &mut r, &mut st, std::ptr::null_mut()) } {
            GR_OK => Ok(r),
            s => Err(rmsd_error(&self.target, s, &self.group)),
        }
    }
}

impl FrameConvertAnalyze for HipRmsd {
    type Error = RMSDError;
    type AnalysisResult = f32;
    fn convert_analyze(&mut self, system: &mut System) -> Result<f32, RMSDError> {
        self.stage(system);
        let (mut r, mut st) = (0f32, 0);
        match unsafe { gr_rmsd_fit_batch(self.plan, 0, 1, &mut r, &mut st) } {
            GR_OK => {
                unsafe { gr_frame_download(self.target.ctx, 0, self.scratch.as_mut_ptr() as *mut c_float) };
                for (atom, p) in system.atoms_iter_mut().zip(self.scratch.iter()) {
                    atom.set_position(Vector3D::new(p[0], p[1], p[2]));
                }
                Ok(r)
            }
            s => Err(rmsd_error(&self.target, s, &self.group)),
        }
    }
}
