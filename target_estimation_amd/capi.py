"""ctypes binding of libtarget_estimation_amd.so (the C ABI in include/target_estimation_amd/).

There is no fallback: if the HIP library has not been built, importing the handle raises.
"""
import ctypes as C
import os

from ._build import LIB

_lib = None

c_uint_p = C.POINTER(C.c_uint)
c_double_p = C.POINTER(C.c_double)
c_ubyte_p = C.POINTER(C.c_ubyte)

# every symbol include/target_estimation_amd/*.h declares: name -> (restype, argtypes)
class BatchSequence(C.Structure):
    """target_batch_sequence_c of target_batch_c.h"""
    _fields_ = [("meas_dev", C.c_void_p), ("tick_stride", C.c_long), ("ld", C.c_long),
                ("has_meas_dev", C.c_void_p), ("has_stride", C.c_long),
                ("delta_dev", C.c_void_p), ("pose_dev", C.c_void_p), ("ring_ticks", C.c_long)]


class StreamSpec(C.Structure):
    """target_stream_c of target_batch_c.h"""
    _fields_ = [("model", C.c_int), ("seed", C.c_ulonglong), ("first_target", C.c_long), ("dt", C.c_double),
                ("availability", C.c_double), ("rpy_noise", C.c_double)]


SIGNATURES = {
    # target_manager_c.h (the reference's ten symbols)
    "target_manager_new": (C.c_void_p, [C.c_char_p]),
    "target_manager_init": (None, [C.c_void_p, C.c_uint, C.c_double, c_double_p, C.c_double]),
    "target_manager_update_meas": (None, [C.c_void_p, C.c_uint, C.c_double, c_double_p]),
    "target_manager_update": (None, [C.c_void_p, C.c_uint, C.c_double]),
    "target_manager_get_est_pose": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_est_twist": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_est_acceleration": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_n_measurements": (C.c_int, [C.c_void_p, C.c_uint]),
    "target_manager_log": (None, [C.c_void_p]),
    "target_manager_delete": (None, [C.c_void_p]),
    # target_batch_c.h
    "target_manager_new_ex": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int]),
    "target_manager_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "target_manager_synchronize": (C.c_int, [C.c_void_p]),
    "target_manager_set_log_directory": (C.c_int, [C.c_void_p, C.c_char_p]),
    "target_manager_last_error": (C.c_char_p, []),
    "target_intersection_solver_new": (C.c_void_p, [C.c_void_p, C.c_uint]),
    "target_intersection_solver_delete": (None, [C.c_void_p]),
    "target_intersection_solver_get_time_with_sphere": (C.c_double, [C.c_void_p, C.c_uint, C.c_double, c_double_p, C.c_double]),
    "target_intersection_solver_get_pose_with_sphere": (C.c_bool, [C.c_void_p, C.c_uint, C.c_double, C.c_double, C.c_double, c_double_p,
                                                                   C.c_double, c_double_p]),
    "target_intersection_solver_last_errors": (None, [C.c_void_p, c_double_p, c_double_p]),
    "target_manager_set_log_targets": (C.c_int, [C.c_void_p, c_uint_p, C.c_long]),
    "target_manager_set_keep_measurement": (C.c_int, [C.c_void_p, C.c_int]),
    "target_manager_get_measured_pose": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_period_estimate": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_estimated_transform": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_n": (C.c_int, [C.c_void_p, C.c_uint]),
    "target_manager_get_m": (C.c_int, [C.c_void_p, C.c_uint]),
    "target_manager_get_model_matrices": (C.c_bool, [C.c_void_p, C.c_uint, c_double_p, c_double_p, c_double_p]),
    "target_manager_init_typed": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_double, C.c_double, c_double_p,
                                            c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "target_manager_init_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, C.c_double, C.c_double, c_double_p,
                                             c_double_p, c_double_p]),
    "target_manager_init_batch_typed": (C.c_long, [C.c_void_p, C.c_int, c_uint_p, C.c_long, C.c_double, C.c_double,
                                                   c_double_p, c_double_p, c_double_p, C.c_int, c_double_p,
                                                   c_double_p, c_double_p]),
    "target_manager_init_batch_classes": (C.c_long, [C.c_void_p, C.c_int, c_uint_p, C.c_long, C.c_double, C.c_double, C.c_long,
                                                     c_double_p, c_double_p, c_double_p, c_uint_p, c_double_p, c_double_p, c_double_p]),
    "target_batch_num_classes": (C.c_int, [C.c_void_p]),
    "target_manager_erase": (C.c_int, [C.c_void_p, C.c_uint]),
    "target_manager_erase_batch": (C.c_long, [C.c_void_p, C.POINTER(C.c_uint), C.c_long]),
    "target_manager_size": (C.c_long, [C.c_void_p]),
    "target_manager_get_available_targets": (C.c_long, [C.c_void_p, c_uint_p, C.c_long]),
    "target_manager_update_meas_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, C.c_double, c_double_p, c_ubyte_p]),
    "target_manager_update_all": (C.c_int, [C.c_void_p, C.c_double]),
    "target_manager_get_est_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, c_double_p, c_double_p, c_double_p, c_ubyte_p]),
    "target_manager_get_est_at_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, C.c_double, c_double_p, c_double_p,
                                                   c_double_p, c_ubyte_p]),
    "target_manager_get_state_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, c_double_p, c_double_p]),
    "target_manager_get_time": (C.c_int, [C.c_void_p, C.c_uint, c_double_p]),
    "target_manager_get_intersection_time_with_sphere": (C.c_double, [C.c_void_p, C.c_uint, C.c_double, c_double_p, C.c_double]),
    "target_manager_get_intersection_pose_with_sphere": (C.c_bool, [C.c_void_p, C.c_uint, C.c_double, c_double_p, C.c_double,
                                                                    c_double_p, c_double_p]),
    "target_manager_intersect_sphere_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, C.c_double, c_double_p, C.c_double,
                                                         c_double_p, c_double_p, c_ubyte_p]),
    "target_manager_intersect_sphere_converged_batch": (C.c_long, [C.c_void_p, c_uint_p, C.c_long, C.c_double, C.c_double, C.c_double,
                                                                   c_double_p, C.c_double, C.c_int, c_double_p, c_double_p,
                                                                   c_ubyte_p, c_ubyte_p, c_double_p]),
    "target_batch_intersect_sphere_converged_dev": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, c_double_p, C.c_double,
                                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "target_batch_gate_update_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "target_batch_intersect_sphere_dev": (C.c_int, [C.c_void_p, C.c_double, c_double_p, C.c_double, C.c_void_p, C.c_void_p]),
    "target_comm_unique_id": (C.c_int, [C.c_char_p]),
    "target_comm_new": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int]),
    "target_comm_delete": (None, [C.c_void_p]),
    "target_manager_gather_pose_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_long), C.c_void_p]),
    "target_manager_gather_pose_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "target_manager_gather_pose_wait_for": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_float)]),
    "target_ingest_new": (C.c_void_p, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "target_ingest_delete": (None, [C.c_void_p]),
    "target_ingest_set_expiration_time": (None, [C.c_void_p, C.c_double]),
    "target_ingest_set_token_name": (None, [C.c_void_p, C.c_char_p]),
    "target_ingest_push": (C.c_int, [C.c_void_p, C.c_uint, C.c_double, c_double_p]),
    "target_ingest_push_named": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, c_double_p]),
    "target_ingest_tick": (C.c_long, [C.c_void_p, C.c_double, C.c_double, c_uint_p, c_double_p, C.c_long]),
    "target_manager_num_batches": (C.c_int, [C.c_void_p]),
    "target_manager_get_batch": (C.c_void_p, [C.c_void_p, C.c_int]),
    "target_manager_get_batch_of_type": (C.c_void_p, [C.c_void_p, C.c_int]),
    "target_batch_size": (C.c_long, [C.c_void_p]),
    "target_batch_type": (C.c_int, [C.c_void_p]),
    "target_batch_dtype": (C.c_int, [C.c_void_p]),
    "target_batch_state_dim": (C.c_int, [C.c_void_p]),
    "target_batch_meas_dim": (C.c_int, [C.c_void_p]),
    "target_batch_lanes_per_target": (C.c_int, [C.c_void_p]),
    "target_batch_is_symmetric_packed": (C.c_int, [C.c_void_p]),
    "target_batch_layout": (C.c_int, [C.c_void_p]),
    "target_batch_algorithmic_bytes": (C.c_long, [C.c_void_p]),
    "target_batch_resident_bytes_per_target": (C.c_double, [C.c_void_p]),
    "target_batch_slot_ids": (C.c_long, [C.c_void_p, c_uint_p, C.c_long]),
    "target_batch_step": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_long, C.c_void_p]),
    "target_batch_step_host": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_long, C.c_void_p]),
    "target_batch_step_sequence": (C.c_int, [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_int]),
    "target_batch_step_sequence_ring": (C.c_int, [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int]),
    "target_manager_population_tick": (C.c_int, [C.c_void_p]),
    "target_manager_step_sequence_all": (C.c_int, [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_long, C.c_int, c_double_p, C.c_double, C.c_int]),
    "target_batch_step_fused": (C.c_int, [C.c_void_p, C.c_long, C.c_double, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long]),
    "target_batch_live_start": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_long,
                                          C.c_long, C.c_double]),
    "target_batch_live_set_pose_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long]),
    "target_batch_live_post": (C.c_int, [C.c_void_p, C.c_long]),
    "target_batch_live_post_each": (C.c_int, [C.c_void_p, C.c_long]),
    "target_batch_live_done": (C.c_long, [C.c_void_p]),
    "target_batch_live_wait": (C.c_int, [C.c_void_p, C.c_long, C.c_double]),
    "target_batch_live_stop": (C.c_long, [C.c_void_p]),
    "target_batch_live_capacity": (C.c_long, [C.c_void_p]),
    "target_batch_live_running": (C.c_int, [C.c_void_p]),
    "target_manager_live_start_all": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_double, C.c_int,
                                                c_double_p, C.c_double]),
    "target_manager_live_post_all": (C.c_int, [C.c_void_p, C.c_long, C.c_int]),
    "target_manager_live_done_all": (C.c_long, [C.c_void_p]),
    "target_manager_live_wait_all": (C.c_int, [C.c_void_p, C.c_long, C.c_double]),
    "target_manager_live_stop_all": (C.c_long, [C.c_void_p]),
    "target_batch_get_est_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double]),
    "target_stream_fill_dev": (C.c_int, [C.POINTER(StreamSpec), C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_long,
                                         C.c_void_p, C.c_long, C.c_void_p]),
    "target_stream_truth_dev": (C.c_int, [C.POINTER(StreamSpec), C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]),
    "target_batch_pack_meas_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long]),
}


def lib():
    """The loaded shared library; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise RuntimeError(
            "target_estimation_amd: %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB)
    # One HIP runtime per process: torch wheels bundle their own libamdhip64.so (SONAME
    # libamdhip64.so.7, the same as /opt/rocm's).  If torch is imported first, the dynamic loader
    # resolves this library's DT_NEEDED against the copy already mapped; loaded the other way round
    # the process ends up with two runtimes and the second one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    handle = C.CDLL(LIB)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def last_error():
    return lib().target_manager_last_error().decode()
