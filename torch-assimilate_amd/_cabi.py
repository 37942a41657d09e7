"""ctypes binding of the C ABI declared in include/mia_letkf.h.

There is no CPU fallback: if the gfx950 library is missing or a call fails, this raises.
"""
import ctypes as C
import os

from ._build import LIB_PATH

_lib = None

i32, i64, f32, f64, vp, sz = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_size_t


class KernelOp(C.Structure):
    """mia_kernel_op_t"""
    _fields_ = [("op", C.c_int32), ("reserved", C.c_int32), ("value", C.c_double)]


_PROTOS = {
    "mia_version": ([], i32),
    "mia_status_string": ([i32], C.c_char_p),
    "mia_set_option": ([C.c_char_p, i32], i32),
    "mia_get_option": ([C.c_char_p, C.POINTER(C.c_int)], i32),
    "mia_last_analysis_kernel": ([C.c_char_p, i32], i32),
    "mia_gaspari_cohn_f64": ([vp, i64, vp, vp], i32),
    "mia_gaspari_cohn_f32": ([vp, i64, vp, vp], i32),
    "mia_gaspari_cohn_inf_f64": ([vp, i64, vp, vp], i32),
    "mia_gaspari_cohn_inf_f32": ([vp, i64, vp, vp], i32),
    "mia_letkf_localize_taper_f64": ([i32, vp, i64, i64, vp, i64, i32, C.POINTER(C.c_int32), C.POINTER(f64), i32, f64,
                                      i32, vp, vp, vp, vp, vp, sz, vp], i32),
    "mia_letkf_localize_from_dist_taper_f64": ([i32, vp, vp, i64, i32, C.POINTER(f64), i32, f64, vp, vp, vp, vp, vp],
                                               i32),
    "mia_lketkf_kernel_analysis_packed_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32,
                                               C.POINTER(KernelOp), i32, vp, i64, i64, vp, vp, vp], i32),
    "mia_lketkf_kernel_analysis_packed_f64": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f64,
                                               C.POINTER(KernelOp), i32, vp, i64, i64, vp, vp, vp], i32),
    "mia_letkf_localize_workspace_bytes": ([i64, i32, C.POINTER(sz)], i32),
    "mia_letkf_localize_f64": ([vp, i64, i64, vp, i64, i32, C.POINTER(C.c_int32), C.POINTER(f64), i32, f64,
                                i32, vp, vp, vp, vp, vp, sz, vp], i32),
    "mia_letkf_localize_from_dist_f64": ([vp, vp, i64, i32, C.POINTER(f64), i32, f64, vp, vp, vp, vp, vp], i32),
    "mia_letkf_analysis_workspace_bytes": ([i32, i64, i32, C.POINTER(sz)], i32),
    "mia_letkf_analysis_f32": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, vp, vp, vp, i32, i32, f32,
                                vp, i64, i64, vp, vp, vp, sz, vp], i32),
    "mia_letkf_analysis_f64": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, vp, vp, vp, i32, i32, f64,
                                vp, i64, i64, vp, vp, vp, sz, vp], i32),
    "mia_letkf_pack_obs_f32": ([vp, vp, i32, i64, vp, vp], i32),
    "mia_letkf_pack_obs_f64": ([vp, vp, i32, i64, vp, vp], i32),
    "mia_letkf_analysis_packed_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32,
                                       vp, i64, i64, vp, vp, vp], i32),
    "mia_letkf_analysis_packed_f64": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f64, f64,
                                       vp, i64, i64, vp, vp, vp], i32),
    "mia_letkf_analysis_matfun_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32,
                                       vp, i64, i64, vp, vp, vp], i32),
    "mia_letkf_weights_matfun_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32,
                                      vp, i64, i64, vp, vp, vp, vp], i32),
    "mia_letkf_weights_retry_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32,
                                     vp, i64, i64, vp, vp, vp], i32),
    "mia_letkf_index_build_f64": ([vp, i64, i32, C.POINTER(C.c_int32), C.POINTER(f64), i32, vp, sz, vp], i32),
    "mia_letkf_analysis_matfun_fused_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, i32, C.POINTER(C.c_int32),
                                             C.POINTER(f64), i32, f64, vp, sz, i32, f32, f32, vp, i64, i64, vp, vp, vp,
                                             vp], i32),
    "mia_letkf_tile_lists_bytes": ([i64, i32, i32, C.POINTER(sz)], i32),
    "mia_letkf_localize_tiles_f64": ([i32, vp, i64, i64, vp, i64, i32, C.POINTER(C.c_int32), C.POINTER(f64), i32, f64,
                                      i32, i32, vp, sz, vp, vp, sz, vp], i32),
    "mia_letkf_split_record_bytes": ([i32, C.POINTER(sz)], i32),
    "mia_letkf_pack_split_f32": ([vp, vp, i32, i64, vp, vp], i32),
    "mia_letkf_analysis_tiles_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, i32, i32, f32, vp, i64, i64, vp, vp, vp], i32),
    "mia_letkf_tiles_cover": ([i32, i32, i32, i32, i64, i64, i64, i64, f32], i32),
    "mia_lketkf_rbf_analysis_tiles_f32": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, vp, i32, i32, f32, f32, vp, i64, i64,
                                           vp, vp, vp], i32),
    "mia_letkf_weights_tiles_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, i32, i32, f32, vp, i64, i64, vp, vp, vp, vp], i32),
    "mia_letkf_analysis_retry_f32": ([vp, i64, i32, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32,
                                      vp, i64, i64, vp, vp], i32),
    "mia_lketkf_rbf_analysis_f32": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, vp, vp, vp, i32, i32, f32, f32,
                                     vp, i64, i64, vp, vp, vp, sz, vp], i32),
    "mia_lketkf_rbf_analysis_f64": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, vp, vp, vp, i32, i32, f64, f64,
                                     vp, i64, i64, vp, vp, vp, sz, vp], i32),
    "mia_etkf_workspace_bytes": ([i32, i64, i32, C.POINTER(sz)], i32),
    "mia_etkf_weights_f32": ([vp, vp, i32, i64, f32, vp, vp, vp, sz, vp], i32),
    "mia_etkf_weights_f64": ([vp, vp, i32, i64, f64, vp, vp, vp, sz, vp], i32),
    "mia_ketkf_workspace_bytes": ([i32, i64, i32, C.POINTER(sz)], i32),
    "mia_ketkf_weights_f32": ([vp, vp, i32, i64, f32, C.POINTER(KernelOp), i32, vp, vp, vp, sz, vp], i32),
    "mia_ketkf_weights_f64": ([vp, vp, i32, i64, f64, C.POINTER(KernelOp), i32, vp, vp, vp, sz, vp], i32),
    "mia_apply_weights_f32": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, i64, vp], i32),
    "mia_apply_weights_f64": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, i64, vp], i32),
    "mia_apply_local_weights_f32": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, i64, vp], i32),
    "mia_apply_local_weights_f64": ([vp, i64, i32, i32, i64, i64, vp, vp, i64, i64, vp], i32),
    "mia_lienks_update_f32": ([vp, i64, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp], i32),
    "mia_lienks_update_f64": ([vp, i64, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f64, f64, vp, vp, vp], i32),
    "mia_lienks_update_matfun_f32": ([vp, i64, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp], i32),
    "mia_lienks_update_retry_f32": ([vp, i64, i32, i64, i64, vp, i64, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp], i32),
    "mia_obs_space_uncorr_f32": ([vp, i64, vp, vp, i32, i64, vp, i64, vp, vp, vp], i32),
    "mia_obs_space_uncorr_f64": ([vp, i64, vp, vp, i32, i64, vp, i64, vp, vp, vp], i32),
    "mia_obs_space_corr_workspace_bytes": ([i32, i64, i32, C.POINTER(sz)], i32),
    "mia_obs_space_corr_f32": ([vp, i64, vp, vp, i32, i64, vp, i64, vp, vp, vp, vp, sz, vp], i32),
    "mia_obs_space_corr_f64": ([vp, i64, vp, vp, i32, i64, vp, i64, vp, vp, vp, vp, sz, vp], i32),
    "mia_comm_load": ([C.c_char_p], i32),
    "mia_comm_unique_id": ([vp], i32),
    "mia_comm_create": ([vp, i32, i32, C.POINTER(vp)], i32),
    "mia_comm_create_partition": ([i32, i32, C.POINTER(vp)], i32),
    "mia_comm_create_custom": ([i32, i32, vp, vp, vp, C.POINTER(vp)], i32),
    "mia_comm_set_place_stream": ([vp, vp], i32),
    "mia_letkf_step_submit": ([vp, i64, i32, i32, vp, vp, i64, vp, vp, i32, C.POINTER(C.c_int32), C.POINTER(f64), i32, f64,
                               f32, f32, i32, i32, vp, i32, i32, vp, vp, vp, vp, sz, vp, vp, vp, i32,
                               vp, vp, vp, C.POINTER(vp), vp, vp, C.POINTER(vp)], i32),
    "mia_letkf_step_submit_args": ([vp, vp], i32),
    "mia_letkf_step_run_args": ([vp, vp], i32),
    "mia_letkf_step_collect": ([vp, vp, vp, vp, i32, vp, vp], i32),
    "mia_timing_event_acquire": ([C.POINTER(vp)], i32),
    "mia_timing_event_release": ([vp], i32),
    "mia_timing_event_elapsed_ms": ([vp, vp, C.POINTER(C.c_float)], i32),
    "mia_letkf_step_join": ([vp], i32),
    "mia_letkf_step_join_info": ([vp, C.POINTER(C.c_int)], i32),
    "mia_letkf_step_coalesce_stats": ([C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], i32),
    "mia_debug_step_trace": ([C.POINTER(C.c_longlong), i32], i32),
    "mia_letkf_step_drain": ([], i32),
    "mia_letkf_step_readback": ([vp, vp, vp, vp, C.POINTER(vp)], i32),
    "mia_event_synchronize": ([vp], i32),
    "mia_stream_wait_event": ([vp, vp], i32),
    "mia_stream_wait_stream": ([vp, vp, C.POINTER(vp)], i32),
    "mia_event_destroy": ([vp], i32),
    "mia_letkf_step_launch_stats": ([C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)], i32),
    "mia_comm_peer_alloc": ([vp, sz, i32, vp], i32),
    "mia_comm_peer_open": ([vp, vp], i32),
    "mia_comm_peer_attach": ([vp, i32, C.POINTER(vp), vp], i32),
    "mia_comm_peer_wait_bound": ([vp, i32], i32),
    "mia_comm_peer_buffer": ([vp, i32], vp),
    "mia_comm_peer_sync_area": ([vp], vp),
    "mia_comm_peer_exchange": ([vp, i32, i32, i64, i64, i64, vp, vp], i32),
    "mia_comm_peer_rewait": ([vp, i32, vp, vp], i32),
    "mia_comm_destroy": ([vp], i32),
    "mia_comm_last_error": ([], C.c_char_p),
    "mia_letkf_sharded_step_workspace_bytes": ([i64, i32, i32, i64, i32, i32, i32, i32, C.POINTER(sz)], i32),
    "mia_letkf_step_workspace_release": ([vp], i32),
    "mia_letkf_sharded_step_f32": ([vp, i64, i32, i32, vp, vp, i64, vp, vp, i32, C.POINTER(C.c_int32), C.POINTER(f64),
                                    i32, f64, f32, f32, i32, i32, vp, i32, i32, vp, vp, vp, vp, sz, vp, vp], i32),
    "mia_letkf_step_timing_events": ([vp, vp], i32),
    "mia_letkf_sharded_step_streams_f32": ([vp, i64, i32, i32, vp, vp, i64, vp, vp, i32, C.POINTER(C.c_int32),
                                            C.POINTER(f64), i32, f64, f32, f32, i32, i32, vp, i32, i32, vp, vp, vp, vp,
                                            sz, vp, vp, vp, i32], i32),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)
# callbacks of mia_comm_create_custom
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, sz, vp)
ALLREDUCE_MAX_I32_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int, vp)


class MiaError(RuntimeError):
    pass


def lib():
    """Load (once) the in-tree gfx950 library; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiaError(
                "HIP extension %s not found: run `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(there is no CPU fallback for the LETKF hot path)" % LIB_PATH)
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7) but links it as
        # "libamdhip64.so"; loading ours first would pull /opt/rocm's copy and leave the process
        # with two HIP runtimes.  Import torch first so both bind to the same runtime.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (args, res) in _PROTOS.items():
            fn = getattr(h, name)      # AttributeError if the library does not export it
            fn.argtypes, fn.restype = args, res
        _lib = h
    return _lib


def set_option(name: str, value: int) -> int:
    """mia_set_option: returns the previous value (so that a caller / test can restore it)."""
    old = C.c_int(0)
    check(lib().mia_get_option(name.encode(), C.byref(old)), "mia_get_option(%s)" % name)
    check(lib().mia_set_option(name.encode(), int(value)), "mia_set_option(%s)" % name)
    return old.value


class StepArgs(C.Structure):
    """mia_step_args_t (include/mia_letkf.h): mia_letkf_step_submit's arguments as one block a pipeline slot keeps."""
    _fields_ = [("X", C.c_void_p), ("G", C.c_int64), ("m", C.c_int32), ("k", C.c_int32), ("Yb", C.c_void_p), ("d", C.c_void_p),
                ("P", C.c_int64), ("grid_xyz", C.c_void_p), ("obs_xyz", C.c_void_p), ("n_coord", C.c_int32),
                ("coord_group", C.c_int32 * 3), ("gc_c", C.c_double * 3), ("n_r", C.c_int32), ("gc_eps", C.c_double),
                ("inf_factor", C.c_float), ("gamma", C.c_float), ("method", C.c_int32), ("p_max_assumed", C.c_int32),
                ("comm", C.c_void_p), ("n_chunks", C.c_int32), ("phase", C.c_int32), ("Xa", C.c_void_p), ("flags", C.c_void_p),
                ("counters", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("stream", C.c_void_p),
                ("comm_stream", C.c_void_p), ("prep_stream", C.c_void_p), ("step_flags", C.c_int32), ("host8", C.c_void_p),
                ("after_stream", C.c_void_p), ("on_stream", C.c_void_p), ("done_event", C.POINTER(C.c_void_p)),
                ("time_start_event", C.c_void_p), ("time_stop_event", C.c_void_p), ("caller_stream", C.c_void_p),
                ("in_event", C.POINTER(C.c_void_p))]


class TimingEvent:
    """A timing event from the library's pool (mia_timing_event_acquire): what ShardedLetkf.time_next_step hands to the step's
    launch.  ``elapsed_time`` as torch.cuda.Event's; released to the pool, never destroyed, when the object goes away."""
    __slots__ = ("cuda_event",)
    _free = []          # handles ready for use on this side of the boundary (acquired thirty-two at a time: a library call per event
                        # was 3 us of every timed step's submission)

    def __init__(self):
        free = TimingEvent._free
        if not free:
            l = lib()
            ev = C.c_void_p()
            for _ in range(32):
                check(l.mia_timing_event_acquire(C.byref(ev)), "mia_timing_event_acquire")
                free.append(ev.value)
        self.cuda_event = free.pop()

    def elapsed_time(self, other) -> float:
        ms = C.c_float(0.0)
        check(lib().mia_timing_event_elapsed_ms(self.cuda_event, other.cuda_event, C.byref(ms)), "mia_timing_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.cuda_event:
                TimingEvent._free.append(self.cuda_event)      # (reused, never destroyed: a launch thread that still holds it touches a live event)
        except Exception:       # (interpreter shutdown)
            pass


def step_coalesce_stats():
    """mia_letkf_step_coalesce_stats: (analysis launches made by the launch thread's collector, steps they carried) so far."""
    a, b = C.c_longlong(0), C.c_longlong(0)
    check(lib().mia_letkf_step_coalesce_stats(C.byref(a), C.byref(b)), "mia_letkf_step_coalesce_stats")
    return int(a.value), int(b.value)


def last_analysis_kernel() -> str:
    """mia_last_analysis_kernel: the analysis kernel launched last, as rocprofv3 names it ('' before the first launch)."""
    buf = C.create_string_buffer(160)
    check(lib().mia_last_analysis_kernel(buf, 160), "mia_last_analysis_kernel")
    return buf.value.decode()


def check(status, what):
    if status != 0:
        msg = lib().mia_status_string(status).decode()
        if status == -6:
            msg += ": " + lib().mia_comm_last_error().decode()
        raise MiaError("%s failed with status %d: %s" % (what, status, msg))
