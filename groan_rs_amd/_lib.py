"""ctypes binding of libgroan_hip.so (the C ABI declared in include/groan_hip.h).

The shared library is the product; this module only declares argument types.  It fails loudly if
the library has not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C groan_rs_amd/csrc`): there is no Python/CPU fallback for any operation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GR_LIB_PATH") or os.path.join(_HERE, "libgroan_hip.so")   # GR_LIB_PATH: A/B builds only

c_u64p = C.POINTER(C.c_uint64)
c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int)

# status codes (include/groan_hip.h)
(OK, E_NO_BOX, E_NOT_ORTHOGONAL, E_ZERO_BOX, E_EMPTY_GROUP, E_INCONSISTENT_GROUP, E_NO_POSITION, E_NO_MASS,
 E_GROUP_NOT_FOUND, E_OUT_OF_RANGE, E_INVALID_ARG, E_GROUP_EXISTS, E_HIP, E_NO_DEVICE, E_UNSUPPORTED_BOX, E_IO, E_FORMAT,
 E_INVALID_NAME) = range(18)

CENTER_NAIVE, CENTER_ESTIMATE, CENTER_PBC = 0, 1, 2

# every symbol include/groan_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gr_version": (C.c_char_p, []),
    "gr_status_string": (C.c_char_p, [C.c_int]),
    "gr_device_count": (C.c_int, [c_i32p]),
    "gr_ctx_create": (C.c_void_p, [C.c_int, C.c_uint64, C.c_uint32, c_i32p]),
    "gr_ctx_destroy": (None, [C.c_void_p]),
    "gr_last_error": (C.c_char_p, [C.c_void_p]),
    "gr_last_error_index": (C.c_uint64, [C.c_void_p]),
    "gr_last_error_counts": (None, [C.c_void_p, c_u64p]),
    "gr_ctx_set_strict_orthogonal": (C.c_int, [C.c_void_p, C.c_int]),
    "gr_n_atoms": (C.c_uint64, [C.c_void_p]),
    "gr_n_slots": (C.c_uint32, [C.c_void_p]),
    "gr_sync": (C.c_int, [C.c_void_p]),
    "gr_set_masses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "gr_container_from_indices": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]),
    "gr_container_from_ranges": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]),
    "gr_container_union": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "gr_container_intersection": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "gr_container_n_atoms": (C.c_uint64, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "gr_container_expand": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gr_container_isin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]),
    "gr_container_validate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, c_u64p]),
    "gr_group_create_from_ranges": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gr_group_create_from_indices": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "gr_group_remove": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gr_group_exists": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gr_group_count": (C.c_uint64, [C.c_void_p]),
    "gr_group_name": (C.c_int, [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]),
    "gr_group_n_atoms": (C.c_int, [C.c_void_p, C.c_char_p, c_u64p]),
    "gr_group_n_blocks": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]),
    "gr_group_blocks": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p]),
    "gr_frame_upload": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "gr_frame_upload_wait": (C.c_int, [C.c_void_p, C.c_uint32]),
    "gr_frame_download": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "gr_frame_set_box": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "gr_frame_get_box": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "gr_frame_copy": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "gr_host_alloc": (C.c_void_p, [C.c_size_t]),
    "gr_host_free": (None, [C.c_void_p]),
    "gr_group_center": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "gr_sel_center": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "gr_sel_translate": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gr_sel_wrap": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gr_sel_all_distances": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]),
    "gr_sel_filter_geometry": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), c_u64p]),
    "gr_group_distance": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, c_f32p]),
    "gr_atoms_distance": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, c_f32p]),
    "gr_group_all_distances": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]),
    "gr_group_all_distances_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), c_u64p, c_u64p]),
    "gr_group_all_distances_batch_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), c_u64p, c_u64p, C.c_void_p]),
    "gr_group_all_distances_reduce": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_size_t]),
    "gr_group_all_distances_reduce_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gr_device_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gr_trr_open": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "gr_trr_close": (None, [C.c_void_p]),
    "gr_trr_n_atoms": (C.c_uint64, [C.c_void_p]),
    "gr_trr_n_frames": (C.c_uint64, [C.c_void_p]),
    "gr_trr_frame_info": (C.c_int, [C.c_void_p, C.c_uint64, c_u64p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gr_trr_read_frame": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_u64p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gr_trr_writer_open": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "gr_trr_writer_close": (C.c_int, [C.c_void_p]),
    "gr_trr_write_frame": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float]),
    "gr_trr_read_frames_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "gr_group_translate": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p]),
    "gr_group_wrap": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
    "gr_atoms_center": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int]),
    "gr_calc_rmsd": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, c_f32p, C.c_void_p]),
    "gr_calc_rmsd_and_fit": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, c_f32p]),
    "gr_rmsd_plan_create": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p, c_i32p]),
    "gr_rmsd_plan_destroy": (None, [C.c_void_p]),
    "gr_rmsd_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gr_rmsd_fit_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "gr_rmsd_batch_begin": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "gr_rmsd_batch_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gr_rmsd_plan_last_fallbacks": (C.c_uint32, [C.c_void_p]),
    "gr_rmsd_plan_force_exact": (C.c_int, [C.c_void_p, C.c_int]),
    "gr_ctx_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "gr_ctx_stat": (C.c_int, [C.c_void_p, C.c_int, c_u64p]),
    "gr_ctx_set_center_onepass_min": (C.c_int, [C.c_void_p, C.c_uint32]),
    "gr_center_fallbacks": (C.c_uint64, [C.c_void_p]),
    "gr_gro_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "gr_structure_free": (None, [C.c_void_p]),
    "gr_structure_n_atoms": (C.c_uint64, [C.c_void_p]),
    "gr_structure_title": (C.c_char_p, [C.c_void_p]),
    "gr_structure_box": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gr_structure_positions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gr_structure_velocities": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gr_structure_atom": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_char_p]),
    "gr_ndx_read": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "gr_ndx_free": (None, [C.c_void_p]),
    "gr_ndx_n_groups": (C.c_size_t, [C.c_void_p]),
    "gr_ndx_group_name": (C.c_char_p, [C.c_void_p, C.c_size_t]),
    "gr_ndx_group_size": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "gr_ndx_group_indices": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "gr_ndx_install": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "gr_group_pairs_within": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "gr_group_center_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gr_group_translate_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p]),
    "gr_group_wrap_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p]),
    "gr_atoms_center_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "gr_xtc_writer_open": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int)]),
    "gr_xtc_writer_close": (C.c_int, [C.c_void_p]),
    "gr_xtc_write_frame": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float]),
    "gr_xtc_write_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int]),
    "gr_xtc_read_frames_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "gr_xtc_read_frame_prefix": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, c_u64p, c_f32p, c_f32p, c_u64p]),
    "gr_xtc_read_frames_device_group": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gr_shape_sphere": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float]),
    "gr_shape_rectangular": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "gr_shape_cylinder": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int]),
    "gr_shape_triangular_prism": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]),
    "gr_shape_inside": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gr_group_create_from_geometries": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int]),
    "gr_xtc_open": (C.c_void_p, [C.c_char_p, c_i32p]),
    "gr_xtc_close": (None, [C.c_void_p]),
    "gr_xtc_n_atoms": (C.c_uint64, [C.c_void_p]),
    "gr_xtc_n_frames": (C.c_uint64, [C.c_void_p]),
    "gr_xtc_frame_info": (C.c_int, [C.c_void_p, C.c_uint64, c_u64p, c_f32p, C.c_void_p, c_f32p]),
    "gr_xtc_read_frame": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, c_u64p, c_f32p, c_f32p]),
    "gr_pool_create": (C.c_void_p, [C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_uint32, c_i32p]),
    "gr_pool_destroy": (None, [C.c_void_p]),
    "gr_pool_size": (C.c_int, [C.c_void_p]),
    "gr_pool_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gr_pool_last_error": (C.c_char_p, [C.c_void_p]),
    "gr_pool_map": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, c_u64p, c_u64p]),
    "gr_pool_map_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, c_u64p, c_u64p]),
    "gr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gr_comm_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_void_p, c_i32p]),
    "gr_comm_destroy": (None, [C.c_void_p]),
    "gr_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "gr_comm_library": (C.c_char_p, []),
    "gr_comm_set_library": (C.c_int, [C.c_char_p]),
    "gr_comm_gather_per_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p]),
    "gr_comm_any_error": (C.c_int, [C.c_void_p, C.c_int, c_i32p]),
    "gr_shard_deinterleave": (None, [C.c_void_p, C.c_int, C.c_uint64, C.c_size_t, C.c_void_p]),
    "gr_timer_start": (C.c_int, [C.c_void_p]),
    "gr_timer_stop": (C.c_int, [C.c_void_p, c_f32p]),
    "gr_synth_reference": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_uint64]),
    "gr_synth_frames": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.c_uint64]),
    "gr_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gr_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), c_u64p, c_u64p]),
    "gr_synth_uniform": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
}

_LIB = None


class LibraryMissing(RuntimeError):
    pass


def load():
    """dlopen libgroan_hip.so and declare every entry point.  Raises if the HIP library is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            "groan_rs_amd/libgroan_hip.so has not been built (run __graft_entry__.build() or "
            "`make -C groan_rs_amd/csrc`); the HIP extension is required, there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("GR_LIB_PATH") and not hasattr(lib, name):
            continue              # A/B builds of older sources (tools/ab_bench.sh) may lack newer entry points
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
