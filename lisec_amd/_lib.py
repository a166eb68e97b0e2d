"""ctypes binding of the C ABI (include/lisec_hip.h).  No fallback: a missing library is an error."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_void_p  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblisec_hip.so")


class LisecError(RuntimeError):
    pass


class VoxelCfg(Structure):
    _fields_ = [("xSize", c_double), ("ySize", c_double), ("zSize", c_double),
                ("maxVoxelX", c_int), ("maxVoxelY", c_int), ("maxVoxelZ", c_int),
                ("sampleSize", c_int)]


class VfeParams(Structure):
    _fields_ = [("kernel", c_void_p * 3), ("gamma", c_void_p * 3), ("beta", c_void_p * 3),
                ("moving_mean", c_void_p * 3), ("moving_var", c_void_p * 3)]


class VfeGrads(Structure):
    _fields_ = [("kernel", c_void_p * 3), ("gamma", c_void_p * 3), ("beta", c_void_p * 3)]


class RpnCfg(Structure):
    _fields_ = [("outX", c_int), ("outY", c_int), ("vx", c_double), ("vy", c_double),
                ("anchors", (c_double * 4) * 2)]


class BnSinkDesc(ctypes.Structure):           # lisec_bn_sink
    _fields_ = [("acc", c_void_p), ("kind", c_int), ("unbiased_moving", c_int), ("n_rows", c_double),
                ("gamma", c_void_p), ("beta", c_void_p), ("moving_mean", c_void_p), ("moving_var", c_void_p),
                ("bnstate", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("coef", c_void_p)]


class ConvExtras(ctypes.Structure):           # lisec_conv_extras
    _fields_ = [("out_mask", ctypes.c_void_p), ("bwd_y", ctypes.c_void_p), ("bwd_bnstate", ctypes.c_void_p),
                ("bwd_relu", ctypes.c_int), ("sink", POINTER(BnSinkDesc)), ("queue", ctypes.c_void_p),
                ("tail_w", ctypes.c_void_p), ("tail_out", ctypes.c_void_p),
                ("in_y", ctypes.c_void_p), ("in_fold_bnstate", ctypes.c_void_p), ("in_fold_coef", ctypes.c_void_p),
                ("in_fold_relu", ctypes.c_int), ("dense_dw", ctypes.c_void_p)]


class ConvPlan(Structure):                     # lisec_conv_plan
    _fields_ = [(n, c_int) for n in ("kernel", "cols", "tiles", "tail_tile0", "k_slices", "plane_pair", "parity_classes",
                                     "workgroups", "launches", "double_buffered")]


class WgradPlan(Structure):                    # lisec_wgrad_plan
    _fields_ = [(n, c_int) for n in ("halo", "mirrored", "taps_per_group", "groups", "tile_rows", "staging_passes", "tiles",
                                     "slabs", "tiles_per_slab", "workgroups", "lane_reduce", "combine_in_kernel", "ring",
                                     "runs_per_column", "lines_per_run")]


class Tuning(Structure):                       # lisec_tuning
    _fields_ = [(n, c_int) for n in ("struct_bytes", "max_splitk", "splitk_min_steps", "min_splitk", "plane_pair", "dense64",
                                     "half_n", "vfe_shape", "field_seg", "field_tpw", "wgrad_blocks", "debug_sync",
                                     "force_splitk", "wgrad_combine_max", "wgrad_batch_blocks", "lone_db", "wgrad_per_cu",
                                     "wgrad_ring", "wgrad_ring_slots", "wide_tile")]


KERNEL_NAMES = {0: "igemm", 1: "halo2", 2: "halo3", 3: "dense64", 4: "queue", 5: "wide"}


class CopyDesc(Structure):                     # lisec_copy_desc
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_int), ("cols", c_int),
                ("src_stride", ctypes.c_longlong), ("dst_stride", ctypes.c_longlong)]


class PackDesc(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("tap_stride", ctypes.c_longlong),
                ("k_stride", ctypes.c_longlong), ("n_stride", ctypes.c_longlong), ("start", ctypes.c_longlong),
                ("ntaps", c_int), ("K", c_int), ("N", c_int), ("Kp", c_int), ("Np", c_int), ("pad_", c_int)]


class ConvGeom(Structure):
    _fields_ = [(n, c_int) for n in ("mode", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "KD", "KH", "KW",
                                     "sd", "sh", "sw", "pd", "ph", "pw", "Cin", "in_stride", "Cout",
                                     "out_stride", "ps", "ps_channels")]


class WgradItem(Structure):                    # lisec_wgrad_item
    _fields_ = [("g", POINTER(ConvGeom)), ("in_", c_void_p), ("in_bnstate", c_void_p), ("flags", c_int), ("dy", c_void_p),
                ("transpose_out", c_int), ("dW", c_void_p)]


HOST_CALL = ctypes.CFUNCTYPE(c_int, c_void_p)    # int (*)(void*): lisec_step_plan_host_call

ROW_STATS_REPLICAS = 16                       # LISEC_ROW_STATS_* of include/lisec_hip.h
ROW_STATS_MOMENT_WORDS = ROW_STATS_REPLICAS * 27 * 2
ROW_STATS_WORDS = ROW_STATS_MOMENT_WORDS + 4 * 64 * 2

_lib = None


def _declare(lib):
    P = c_void_p
    lib.lisec_last_error.restype = c_char_p
    lib.lisec_last_error.argtypes = []
    lib.lisec_abi_version.restype = c_int
    lib.lisec_device_info.restype = c_int
    lib.lisec_device_info.argtypes = [ctypes.c_char_p, c_int]
    lib.lisec_voxelize_workspace_bytes.restype = c_size_t
    lib.lisec_voxelize_workspace_bytes.argtypes = [POINTER(VoxelCfg), c_int]
    lib.lisec_voxelize.restype = c_int
    lib.lisec_voxelize.argtypes = [POINTER(VoxelCfg), P, c_int, c_int, c_int, P, c_size_t, c_int,
                                   P, P, P, P, P, P, P, P, P, P]
    lib.lisec_voxel_rows_to_padded.restype = c_int
    lib.lisec_voxel_rows_to_padded.argtypes = [P, P, P, P, c_int, c_int, P, P]
    lib.lisec_vfe_saved_floats.restype = c_size_t
    lib.lisec_vfe_saved_floats.argtypes = [c_int]
    lib.lisec_vfe_workspace_bytes.restype = c_size_t
    lib.lisec_vfe_workspace_bytes.argtypes = []
    LL = ctypes.c_longlong
    lib.lisec_rpn_to_region_workspace_bytes.restype = c_size_t
    lib.lisec_rpn_to_region_workspace_bytes.argtypes = [POINTER(RpnCfg), c_int]
    lib.lisec_rpn_to_region.restype = c_int
    lib.lisec_rpn_to_region.argtypes = [POINTER(RpnCfg), P, c_int, P, c_int, c_double, c_int, P, c_size_t, P, P, P, P]
    lib.lisec_rpn_labels_workspace_bytes.restype = c_size_t
    lib.lisec_rpn_labels_workspace_bytes.argtypes = [c_int]
    lib.lisec_rpn_labels.restype = c_int
    lib.lisec_rpn_labels.argtypes = [POINTER(RpnCfg), P, c_int, c_double, c_double, P, c_size_t, P, P, P, P]
    lib.lisec_lidar_transform.restype = c_int
    lib.lisec_lidar_transform.argtypes = [P, c_int, c_int, POINTER(c_double), POINTER(c_double), P, P]
    lib.lisec_vfe_grid_from_saved.restype = c_int
    lib.lisec_vfe_grid_from_saved.argtypes = [P, P, c_int, c_int, P, P, P]
    lib.lisec_vfe_backward_workspace_bytes.restype = c_size_t
    lib.lisec_vfe_backward_workspace_bytes.argtypes = [c_int, c_int]
    lib.lisec_vfe_backward.restype = c_int
    lib.lisec_vfe_backward.argtypes = [POINTER(VfeParams), P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P,
                                       POINTER(VfeGrads), c_int, P, c_size_t, P]
    lib.lisec_vfe_saved_field_offset.restype = c_size_t
    lib.lisec_vfe_saved_field_offset.argtypes = [c_int, c_int]
    lib.lisec_conv_tap_sums_workspace_bytes.restype = c_size_t
    lib.lisec_conv_tap_sums_workspace_bytes.argtypes = [POINTER(ConvGeom)]
    lib.lisec_conv_tap_sums.restype = c_int
    lib.lisec_conv_tap_sums.argtypes = [POINTER(ConvGeom), P, P, P, c_size_t, P]
    lib.lisec_conv_tap_sums_bn.restype = c_int
    lib.lisec_conv_tap_sums_bn.argtypes = [POINTER(ConvGeom), P, P, P, P, P, P, P, c_size_t, P]
    lib.lisec_conv_tap_sums_finish.restype = c_int
    lib.lisec_conv_tap_sums_finish.argtypes = [POINTER(ConvGeom), P, c_size_t, P, P]
    lib.lisec_conv_field_forward_workspace_bytes.restype = c_size_t
    lib.lisec_conv_field_forward_workspace_bytes.argtypes = [POINTER(ConvGeom), c_int]
    lib.lisec_conv_field_forward.restype = c_int
    lib.lisec_conv_field_forward.argtypes = [POINTER(ConvGeom), P, P, P, P, P, c_int, P, P, P, POINTER(BnSinkDesc), P,
                                             c_size_t, P]
    lib.lisec_const_field_grads.restype = c_int
    lib.lisec_const_field_grads.argtypes = [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P]
    lib.lisec_conv_packed_floats.restype = c_size_t
    lib.lisec_conv_packed_floats.argtypes = [c_int, c_int, c_int]
    lib.lisec_conv_pack_weights.restype = c_int
    lib.lisec_conv_pack_weights.argtypes = [P, c_int, c_int, c_int, LL, LL, LL, P, P]
    lib.lisec_conv_pack_weights_batched.restype = c_int
    lib.lisec_conv_pack_weights_batched.argtypes = [P, c_int, LL, P]
    lib.lisec_conv_num_mblocks.restype = c_int
    lib.lisec_conv_num_mblocks.argtypes = [POINTER(ConvGeom)]
    lib.lisec_conv_forward.restype = c_int
    lib.lisec_conv_forward.argtypes = [POINTER(ConvGeom), P, P, P, P, c_int, P, P, P, c_size_t, P, P, c_int, P]
    lib.lisec_copy2d_batched.restype = c_int
    lib.lisec_copy2d_batched.argtypes = [P, c_int, P]
    lib.lisec_conv_forward_ex.restype = c_int
    lib.lisec_conv_forward_ex.argtypes = [POINTER(ConvGeom), P, P, P, P, c_int, P, POINTER(ConvExtras), P, P, c_size_t, P, P,
                                          c_int, P]
    lib.lisec_conv_winograd_packed_floats.restype = c_size_t
    lib.lisec_conv_winograd_packed_floats.argtypes = [c_int, c_int, c_int]
    lib.lisec_conv_pack_weights_winograd.restype = c_int
    lib.lisec_conv_pack_weights_winograd.argtypes = [P, c_int, c_int, c_int, LL, LL, LL, c_int, P, P]
    lib.lisec_conv_winograd_supported.restype = c_int
    lib.lisec_conv_winograd_supported.argtypes = [POINTER(ConvGeom), c_int, c_int, POINTER(ConvExtras)]
    lib.lisec_conv_forward_winograd.restype = c_int
    lib.lisec_conv_forward_winograd.argtypes = [POINTER(ConvGeom), P, P, P, P, c_int, P, POINTER(ConvExtras), P]
    lib.lisec_conv_wgrad_winograd_supported.restype = c_int
    lib.lisec_conv_wgrad_winograd_supported.argtypes = [POINTER(ConvGeom)]
    lib.lisec_conv_wgrad_winograd_workspace_bytes.restype = c_size_t
    lib.lisec_conv_wgrad_winograd_workspace_bytes.argtypes = [POINTER(ConvGeom)]
    lib.lisec_conv_wgrad_winograd.restype = c_int
    lib.lisec_conv_wgrad_winograd.argtypes = [POINTER(ConvGeom), P, P, P, c_size_t, P, P]
    lib.lisec_dense_dw_slabs.restype = c_int
    lib.lisec_dense_dw_slabs.argtypes = []
    lib.lisec_dense_dw_reduce.restype = c_int
    lib.lisec_dense_dw_reduce.argtypes = [P, P, P]
    lib.lisec_conv_num_mblocks_bwd.restype = c_int
    lib.lisec_conv_num_mblocks_bwd.argtypes = [POINTER(ConvGeom)]
    lib.lisec_bn_backward_apply.restype = c_int
    lib.lisec_bn_backward_apply.argtypes = [P, c_int, P, P, LL, c_int, c_int, P, c_int, P, P, P, P, c_size_t, P]
    lib.lisec_bn_sink_words.restype = c_size_t
    lib.lisec_bn_sink_words.argtypes = [c_int]
    lib.lisec_bn_backward_apply_coef.restype = c_int
    lib.lisec_bn_backward_apply_coef.argtypes = [P, c_int, P, P, LL, c_int, c_int, P, P, P]
    lib.lisec_conv_forward_masked.restype = c_int
    lib.lisec_conv_forward_masked.argtypes = [POINTER(ConvGeom), P, P, P, P, c_int, P, P, P, P, c_size_t, P, P, c_int, P]
    lib.lisec_conv_forward_workspace_bytes.restype = c_size_t
    lib.lisec_conv_forward_workspace_bytes.argtypes = [POINTER(ConvGeom)]
    lib.lisec_conv_forward_rows_workspace_bytes.restype = c_size_t
    lib.lisec_conv_forward_rows_workspace_bytes.argtypes = [POINTER(ConvGeom), c_int]
    lib.lisec_conv_wgrad_workspace_bytes.restype = c_size_t
    lib.lisec_conv_wgrad_workspace_bytes.argtypes = [POINTER(ConvGeom), c_int]
    lib.lisec_conv_wgrad.restype = c_int
    lib.lisec_conv_wgrad.argtypes = [POINTER(ConvGeom), P, P, c_int, P, P, P, c_size_t, c_int, P, P, P, c_int, P]
    lib.lisec_eltwise_workspace_bytes.restype = c_size_t
    lib.lisec_eltwise_workspace_bytes.argtypes = []
    lib.lisec_bn_backward.restype = c_int
    lib.lisec_bn_backward.argtypes = [P, c_int, P, P, LL, c_int, c_int, P, P, P, P, P, c_size_t, P]
    lib.lisec_relu_mask.restype = c_int
    lib.lisec_relu_mask.argtypes = [P, P, LL, P]
    lib.lisec_colsum.restype = c_int
    lib.lisec_colsum.argtypes = [P, c_int, LL, c_int, P, P, c_size_t, P]
    lib.lisec_rpn_loss.restype = c_int
    lib.lisec_rpn_loss.argtypes = [P, P, P, LL, c_int, c_float, P, P, P, c_size_t, P]
    lib.lisec_sgd_nesterov_step.restype = c_int
    lib.lisec_sgd_nesterov_step.argtypes = [P, P, P, LL, c_float, c_float, P]
    lib.lisec_sgd_nesterov_step_dev.restype = c_int
    lib.lisec_sgd_nesterov_step_dev.argtypes = [P, P, P, LL, c_double, c_double, c_float, P, P]
    lib.lisec_sgd_nesterov_step_dev_part.restype = c_int
    lib.lisec_sgd_nesterov_step_dev_part.argtypes = [P, P, P, LL, c_double, c_double, c_float, P, P]
    lib.lisec_fold_depth.restype = c_int
    lib.lisec_fold_depth.argtypes = [P, P, c_int, LL, c_int, c_int, P, P]
    lib.lisec_scale.restype = c_int
    lib.lisec_scale.argtypes = [P, LL, c_float, P]
    lib.lisec_comm_unique_id.restype = c_int
    lib.lisec_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.lisec_comm_init.restype = c_int
    lib.lisec_comm_init.argtypes = [c_int, c_int, ctypes.c_char_p, POINTER(c_void_p)]
    lib.lisec_comm_destroy.restype = c_int
    lib.lisec_comm_destroy.argtypes = [P]
    lib.lisec_allreduce_grads.restype = c_int
    lib.lisec_allreduce_grads.argtypes = [P, P, LL, c_int, P]
    lib.lisec_head_compose.restype = c_int
    lib.lisec_head_compose.argtypes = [P, P, P, c_int, c_int, c_int, LL, LL, P, P, P, P]
    lib.lisec_head_compose_backward.restype = c_int
    lib.lisec_head_compose_backward.argtypes = [P, LL, LL, P, P, P, P, c_int, c_int, c_int, P, P, P, P]
    lib.lisec_head_shuffle.restype = c_int
    lib.lisec_head_shuffle.argtypes = [P, c_int, c_int, c_int, POINTER(c_void_p), POINTER(c_int), c_int, P]
    lib.lisec_workspace_init.restype = c_int
    lib.lisec_workspace_init.argtypes = [P, c_size_t, P]
    lib.lisec_rpn_decode.restype = c_int
    lib.lisec_rpn_decode.argtypes = [POINTER(RpnCfg), P, c_int, P, c_int, P, P, P, P]
    lib.lisec_box_geometry.restype = c_int
    lib.lisec_box_geometry.argtypes = [P, c_int, P, P, P, P]
    lib.lisec_step_plan_create.restype = c_int
    lib.lisec_step_plan_create.argtypes = [POINTER(c_void_p)]
    for name in ("lisec_step_plan_begin", "lisec_step_plan_end", "lisec_step_plan_run", "lisec_step_plan_size",
                 "lisec_step_plan_destroy"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [P]
    lib.lisec_step_plan_recording.restype = c_int
    lib.lisec_step_plan_recording.argtypes = []
    lib.lisec_step_plan_host_call.restype = c_int
    lib.lisec_step_plan_host_call.argtypes = [HOST_CALL, P]
    lib.lisec_event_record.restype = c_int
    lib.lisec_event_record.argtypes = [P, P]
    lib.lisec_stream_wait_event.restype = c_int
    lib.lisec_stream_wait_event.argtypes = [P, P]
    lib.lisec_comm_probe.restype = c_int
    lib.lisec_comm_probe.argtypes = []
    lib.lisec_comm_count.restype = c_int
    lib.lisec_comm_count.argtypes = [P, POINTER(c_int)]
    lib.lisec_conv_plan_query.restype = c_int
    lib.lisec_conv_plan_query.argtypes = [POINTER(ConvGeom), c_int, c_int, POINTER(ConvExtras), c_int, c_size_t, c_int, c_int,
                                          POINTER(ConvPlan)]
    lib.lisec_conv_wgrad_batched_workspace_bytes.restype = c_size_t
    lib.lisec_conv_wgrad_batched_workspace_bytes.argtypes = [POINTER(WgradItem), c_int]
    lib.lisec_conv_wgrad_batched.restype = c_int
    lib.lisec_conv_wgrad_batched.argtypes = [POINTER(WgradItem), c_int, P, c_size_t, P]
    lib.lisec_conv_wgrad_plan_query.restype = c_int
    lib.lisec_conv_wgrad_plan_query.argtypes = [POINTER(ConvGeom), c_int, c_int, c_int, c_int, POINTER(WgradPlan)]
    lib.lisec_tuning_get.restype = c_int
    lib.lisec_tuning_get.argtypes = [POINTER(Tuning)]
    lib.lisec_tuning_set.restype = c_int
    lib.lisec_tuning_set.argtypes = [POINTER(Tuning)]
    for name in ("lisec_debug_igemm_stamps", "lisec_debug_wgrad_stamps", "lisec_debug_vfe_stamps", "lisec_debug_field_stamps",
                 "lisec_debug_wino_stamps"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = [P]
    lib.lisec_bn_finalize.restype = c_int
    lib.lisec_bn_finalize.argtypes = [P, c_int, c_int, c_double, P, P, P, P, c_int, P, P]
    lib.lisec_bn_fold.restype = c_int
    lib.lisec_bn_fold.argtypes = [P, P, P, P, c_int, P, P]
    lib.lisec_vfe_forward.restype = c_int
    lib.lisec_vfe_forward.argtypes = [POINTER(VfeParams), P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P,
                                      c_size_t, P, P]
    lib.lisec_vfe_saved_floats_rows.restype = c_size_t
    lib.lisec_vfe_saved_floats_rows.argtypes = [c_int, c_int]


def load():
    """Returns the loaded library; raises LisecError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LisecError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C lisec_amd/csrc`.  There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
        spec = os.environ.get("LISEC_TUNING")       # measurement aid: "max_splitk=8,half_n=0" (see set_tuning)
        if spec:
            set_tuning(**{k.strip(): int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv.strip())
                          if k.strip() in dict(Tuning._fields_)})
    return _lib


# Grow-on-demand device buffers (VFE saved state / backward scratch, field workspace, weight-gradient scratch, row-list
# gradient rows, split-K scratch, voxeliser scratch) are reallocated by ordinary eager calls; a recorded step plan
# (lisec_step_plan_*) holds their RAW addresses.  Every reallocation bumps this counter; a plan remembers the value it was
# recorded under and refuses to replay (network.StalePlanError) / is re-recorded (Model._captured_step) once it differs.
_ALLOC_GEN = [0]


def alloc_generation():
    return _ALLOC_GEN[0]


def bump_alloc_generation():
    _ALLOC_GEN[0] += 1


def knob(name, default):
    """Schedule knobs of the Python host layer (measurement aids; the defaults are the measured optimum): read once from
    the same LISEC_TUNING="key=value,..." string as the library's lisec_tuning record."""
    spec = os.environ.get("LISEC_TUNING", "")
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            if k.strip() == name:
                return type(default)(v) if not isinstance(default, bool) else v.strip() not in ("0", "false", "")
    return default


def get_tuning():
    """The library's launch-plan knobs (lisec_tuning) as a dict."""
    t = Tuning()
    check(load().lisec_tuning_get(ctypes.byref(t)))
    return {n: getattr(t, n) for n, _ in Tuning._fields_ if n != "struct_bytes"}


def set_tuning(**kw):
    """Changes launch-plan knobs of the library (defaults = the measured optimum; tools/ measure the alternatives).
    Only between calls: the record is process-wide.  Returns the previous values of the changed knobs."""
    t = Tuning()
    check(load().lisec_tuning_get(ctypes.byref(t)))
    prev = {}
    for k, v in kw.items():
        if k not in dict(Tuning._fields_) or k == "struct_bytes":
            raise KeyError(f"unknown tuning knob {k!r}")
        prev[k] = getattr(t, k)
        setattr(t, k, int(v))
    check(load().lisec_tuning_set(ctypes.byref(t)))
    return prev


def check(rc):
    if rc != 0:
        raise LisecError(f"lisec C ABI error {rc}: {load().lisec_last_error().decode()}")


def ptr(t):
    """Device address of a torch tensor as a plain int (ctypes converts it for `void*` parameters), or None."""
    if t is None:
        return None
    return t.data_ptr()


# A caller that issues many launches on a stream it already knows (LisecNet.forward / backward) pins the handle here
# instead of asking torch for the current stream on every launch; None = ask torch.
_pinned_stream = None


def pin_stream(handle):
    """Pins the stream handle returned by current_stream(); returns the previous pin (restore it when done)."""
    global _pinned_stream
    prev = _pinned_stream
    _pinned_stream = handle
    return prev


def current_stream():
    if _pinned_stream is not None:
        return _pinned_stream
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    """The device this process computes on.  Under a multi-process launch (WORLD_SIZE > 1) that is cuda:LOCAL_RANK,
    bound here before anything is allocated (one process per GPU; lisec_amd.parallel.select_device)."""
    import os
    import torch
    if not torch.cuda.is_available():
        raise LisecError("lisec_amd needs an MI355X (torch.cuda.is_available() is False); "
                         "there is no CPU fallback for the hot path")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("LISEC_DEVICE"):
        from .parallel import select_device
        return select_device()
    return torch.device("cuda", torch.cuda.current_device())


# ---- device-scope events ----------------------------------------------------------------------------------------------
# torch.cuda.Event records with a system-scope release (the host may inspect it): ~6 us of the recording queue per
# record.  The forks between the two backward streams only order work on ONE device, so they use events created with
# hipEventDisableTiming | hipEventReleaseToDevice through the HIP runtime the process already has loaded.
_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreateWithFlags.argtypes = [POINTER(c_void_p), ctypes.c_uint]
        _hip.hipEventDestroy.argtypes = [c_void_p]
        _hip.hipEventElapsedTime.argtypes = [POINTER(ctypes.c_float), c_void_p, c_void_p]
    return _hip


def stream_priority_range():
    """(least, greatest) stream priority of the current device (hipDeviceGetStreamPriorityRange): numerically lower = higher."""
    lo, hi = c_int(0), c_int(0)
    rc = _hiprt().hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    if rc != 0:
        raise LisecError(f"hipDeviceGetStreamPriorityRange failed ({rc})")
    return lo.value, hi.value


def create_stream(priority):
    """A non-blocking HIP stream at `priority` (hipStreamCreateWithPriority); returns the raw handle (int)."""
    h = c_void_p()
    rc = _hiprt().hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), c_int(priority))   # 1 = hipStreamNonBlocking
    if rc != 0:
        raise LisecError(f"hipStreamCreateWithPriority failed ({rc})")
    return h.value


_HOST_CALLS = []      # ctypes trampolines of recorded host calls: a plan may call them for as long as the process lives


def host_call(fn):
    """Runs fn() now and, if this thread is recording a step plan, at this place of the sequence in every replay
    (lisec_step_plan_host_call).  Exceptions inside a replayed call are reported as a failed step."""
    if not load().lisec_step_plan_recording():
        fn()                                        # nothing records: no trampoline to keep alive
        return
    err = []

    def tramp(_arg):
        try:
            fn()
            return 0
        except Exception as e:                      # noqa: BLE001 -- must not unwind through the C frame
            err.append(e)
            return 1
    cb = HOST_CALL(tramp)
    _HOST_CALLS.append(cb)
    rc = load().lisec_step_plan_host_call(cb, None)
    if rc != 0:
        raise (err[-1] if err else LisecError(load().lisec_last_error().decode()))


class DeviceEvent:
    """hipEvent_t with device-scope release: record(stream_handle) / wait(stream_handle); handles are the integers
    torch's Stream.cuda_stream gives."""
    FLAGS = 0x2 | 0x40000000          # hipEventDisableTiming | hipEventReleaseToDevice

    def __init__(self, timing=False):
        h = c_void_p()
        rc = _hiprt().hipEventCreateWithFlags(ctypes.byref(h), 0 if timing else self.FLAGS)
        if rc != 0:
            raise LisecError(f"hipEventCreateWithFlags failed ({rc})")
        self.handle = h

    def record(self, stream_handle):
        # through the library (lisec_event_record), so that a recording step plan sees the edge
        check(load().lisec_event_record(self.handle, c_void_p(stream_handle)))

    def wait(self, stream_handle):
        check(load().lisec_stream_wait_event(c_void_p(stream_handle), self.handle))

    def elapsed_ms(self, later):
        """Milliseconds from this event to `later` (both created with timing=True and complete)."""
        ms = ctypes.c_float(0)
        rc = _hiprt().hipEventElapsedTime(ctypes.byref(ms), self.handle, later.handle)
        if rc != 0:
            raise LisecError(f"hipEventElapsedTime failed ({rc})")
        return ms.value

    def __del__(self):
        try:
            if self.handle:
                _hiprt().hipEventDestroy(self.handle)
        except Exception:
            pass
