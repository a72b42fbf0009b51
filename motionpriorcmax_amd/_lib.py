"""ctypes binding of the C ABI declared in include/mpcmax.h.

There is NO fallback: if libmpcmax.so is missing or a call fails, a RuntimeError is raised."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmpcmax.so')
if os.environ.get('MPC_AB_LIB'):          # diagnostics: another build of the SAME library beside the product one (bounds-checked: tools/bounds_run.sh; A/B timing)
    LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])

# flags, mirror of include/mpcmax.h
F_SCALE_BY_DT = 1 << 0
F_MASK_BORDER = 1 << 1
F_POLARITY_SPLIT = 1 << 2
F_NORM_L2 = 1 << 3
F_OBJ_VARIANCE = 1 << 4
F_DIST_L1 = 1 << 5
F_SCHEME_IWD = 1 << 6
F_WANT_NEXT = 1 << 7
F_NO_WARP = 1 << 8
F_UNIT_WEIGHT = 1 << 9
F_ATOMIC_PATH = 1 << 10
F_NO_BWD_RECORDS = 1 << 11

E_NULL, E_SHAPE, E_UNSUPPORTED = -1, -2, -4        # include/mpcmax.h
SCAL_LOSS, SCAL_FOCUS, SCAL_SMOOTH, SCAL_VAL, SCAL_GCOEF, SCAL_COUNT = 0, 1, 2, 3, 4, 8

EXPORTS = ['mpc_version', 'mpc_last_error_string', 'mpc_workspace_bytes', 'mpc_knn_lut_fwd',
           'mpc_knn_lut_bwd', 'mpc_event_splat_fwd', 'mpc_contrast_fwd', 'mpc_lut_smooth',
           'mpc_finalize', 'mpc_event_splat_bwd', 'mpc_scale', 'mpc_voxel_workspace_bytes', 'mpc_voxel_grid', 'mpc_ingest_workspace_bytes', 'mpc_ingest_count',
           'mpc_ingest_scatter', 'mpc_dense_flow', 'mpc_flow_error_workspace_bytes', 'mpc_flow_error',
           'mpc_knn_fail_list_offset', 'mpc_knn_list_offsets', 'mpc_knn_tail_counters_offset', 'mpc_knn_state_floats', 'mpc_focus_fwd', 'mpc_focus_bwd',
           'mpc_event_lut_strips', 'mpc_event_order_workspace_bytes', 'mpc_event_bucket_order', 'mpc_event_splat_bwd_ordered',
           'mpc_profile_start', 'mpc_profile_stop', 'mpc_event_splat_fwd_fixed', 'mpc_iwe_from_fixed',
           'mpc_ingest_ordered_workspace_bytes', 'mpc_ingest_scatter_ordered', 'mpc_pool2_fwd', 'mpc_pool2_bwd_add', 'mpc_event_pos_grad', 'mpc_pe_warp', 'mpc_pe_grad', 'mpc_pe_grad_ordered', 'mpc_pe_grad_ordered_supported', 'mpc_bounds_check', 'mpc_curve_traj_fwd', 'mpc_curve_traj_bwd',
           'mpc_pe_tile_rows', 'mpc_pe_tile_rows_bwd', 'mpc_pe_basis_field', 'mpc_pe_rows_grad_finish']


class Shape(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in
                ('B', 'M', 'Mp', 'nb', 'T', 'H', 'W', 'sp', 'hq', 'wq', 'n', 'K')] + \
               [('flags', ctypes.c_uint32)]


class FocusBuffers(ctypes.Structure):
    """include/mpcmax.h: struct mpc_focus_buffers."""
    _fields_ = [(k, ctypes.c_void_p) for k in
                ('traj', 'events', 't_ref', 'flow_lut', 'flow_next', 'knn_state', 'smooth_grad', 'iwe_raw', 'iwe_blur',
                 'grad_iwe', 'scal')] + [('smooth_weight', ctypes.c_float), ('event_offsets', ctypes.c_void_p), ('scal_out', ctypes.c_void_p)]


class VoxShape(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ('B', 'N', 'C', 'H', 'W', 'norm')] + [('quantile', ctypes.c_float), ('keep', ctypes.c_float)]


class IngestShape(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ('B', 'N', 'H', 'W', 'nb')]


class FlowShape(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ('B', 'C', 'n', 'patch', 'H', 'W')]


class ErrShape(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ('B', 'H', 'W')]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -m motionpriorcmax_amd.build` '
            '(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for this path.')
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    sp = ctypes.POINTER(Shape)
    L.mpc_version.restype = ctypes.c_int
    L.mpc_version.argtypes = []
    L.mpc_last_error_string.restype = ctypes.c_char_p
    L.mpc_last_error_string.argtypes = []
    L.mpc_workspace_bytes.restype = i64
    L.mpc_workspace_bytes.argtypes = [sp]
    L.mpc_knn_fail_list_offset.argtypes = [sp]
    L.mpc_knn_tail_counters_offset.argtypes = [sp]
    L.mpc_knn_list_offsets.argtypes = [sp, ctypes.POINTER(ctypes.c_int64)]
    L.mpc_knn_state_floats.argtypes = [sp]
    L.mpc_knn_lut_fwd.argtypes = [sp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_knn_lut_bwd.argtypes = [sp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_event_splat_fwd.argtypes = [sp, vp, vp, vp, vp, vp, vp]
    L.mpc_contrast_fwd.argtypes = [sp, vp, vp, vp, vp, vp]
    L.mpc_lut_smooth.argtypes = [sp, vp, i32, i32, f32, vp, vp, vp]
    L.mpc_finalize.argtypes = [sp, i32, i32, f32, vp, vp, vp]
    L.mpc_event_splat_bwd.argtypes = [sp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_event_pos_grad.argtypes = [sp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_pe_warp.argtypes = [sp, vp, vp, vp, ctypes.c_int32, vp, vp, vp]
    L.mpc_pe_grad.argtypes = [sp, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp]
    L.mpc_pe_grad_ordered.argtypes = [sp, vp, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, ctypes.c_int32, vp]
    L.mpc_pe_grad_ordered_supported.argtypes = [sp, ctypes.c_int32]
    L.mpc_pe_grad_ordered_supported.restype = ctypes.c_int32
    L.mpc_scale.argtypes = [vp, vp, vp, i64, vp]
    fb = ctypes.POINTER(FocusBuffers)
    L.mpc_focus_fwd.argtypes = [sp, fb, vp, vp]
    L.mpc_focus_bwd.argtypes = [sp, fb, vp, vp, vp, vp, vp, vp]
    vsp = ctypes.POINTER(VoxShape)
    L.mpc_voxel_workspace_bytes.argtypes = [vsp]
    L.mpc_voxel_grid.argtypes = [vsp, vp, vp, vp, vp, vp]
    for name in EXPORTS[3:]:
        getattr(L, name).restype = ctypes.c_int
    L.mpc_voxel_workspace_bytes.restype = i64
    L.mpc_knn_fail_list_offset.restype = i64
    L.mpc_knn_tail_counters_offset.restype = i64
    L.mpc_knn_state_floats.restype = i64
    L.mpc_event_lut_strips.argtypes = [sp]
    L.mpc_event_order_workspace_bytes.argtypes = [sp]
    L.mpc_event_order_workspace_bytes.restype = i64
    L.mpc_event_bucket_order.argtypes = [sp, vp, vp, vp, vp, vp]
    L.mpc_event_splat_bwd_ordered.argtypes = [sp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    isp = ctypes.POINTER(IngestShape)
    L.mpc_ingest_workspace_bytes.argtypes = [isp]
    L.mpc_ingest_workspace_bytes.restype = i64
    L.mpc_ingest_count.argtypes = [isp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_ingest_scatter.argtypes = [isp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.mpc_ingest_ordered_workspace_bytes.argtypes = [isp, sp]
    L.mpc_ingest_ordered_workspace_bytes.restype = i64
    L.mpc_ingest_scatter_ordered.argtypes = [isp, sp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mpc_dense_flow.argtypes = [ctypes.POINTER(FlowShape), vp, vp, vp, vp, vp]
    L.mpc_flow_error_workspace_bytes.argtypes = [ctypes.POINTER(ErrShape)]
    L.mpc_flow_error_workspace_bytes.restype = i64
    L.mpc_flow_error.argtypes = [ctypes.POINTER(ErrShape), vp, vp, vp, vp, vp, vp, vp]
    L.mpc_event_splat_fwd_fixed.argtypes = [sp, vp, vp, vp, vp, vp, vp]
    L.mpc_iwe_from_fixed.argtypes = [vp, vp, i64, vp]
    L.mpc_pool2_fwd.argtypes = [vp, vp, i32, i32, i32, vp]
    L.mpc_pool2_bwd_add.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.mpc_profile_start.argtypes = []
    L.mpc_bounds_check.argtypes = []
    L.mpc_curve_traj_fwd.argtypes = [vp, vp, vp, f32, vp, i32, i32, i32, i32, vp]
    L.mpc_curve_traj_bwd.argtypes = [vp, vp, f32, vp, i32, i32, i32, i32, vp]
    L.mpc_pe_tile_rows.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.mpc_pe_tile_rows_bwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.mpc_pe_basis_field.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.mpc_pe_rows_grad_finish.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.mpc_profile_stop.argtypes = [ctypes.c_char_p, i32, ctypes.POINTER(f32), i32]
    if L.mpc_version() != 107:
        raise RuntimeError(f'libmpcmax.so version {L.mpc_version()} does not match the binding (107)')
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().mpc_last_error_string().decode(errors='replace')
        raise RuntimeError(f'{what} failed (rc={rc}): {msg}')
