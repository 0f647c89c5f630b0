"""ctypes binding of liblaff_hip.so (include/laff_hip.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LAFF_HIP_LIB') or os.path.join(_HERE, 'lib', 'liblaff_hip.so')   # env override: debug builds

# enums of include/laff_hip.h
ACT = {None: 0, False: 0, '': 0, 'none': 0, 'tanh': 1, 'relu': 2, 'sigmoid': 3}
ATT_WITH_AVE, ATT_MUL, ATT_L2NORM_EACH_HEAD, ATT_NO_SPLIT_HEAD, ATT_JUST_AVERAGE = 1, 2, 4, 8, 16
PREC = {'fp32': 0, 'fp16': 1, 'bf16': 2, 'fp16x3': 3, 'bf16x3': 4}
ABI_VERSION = 22


class Plane(C.Structure):
    _fields_ = [('src', C.c_void_p), ('ld', C.c_int), ('tile', C.c_int), ('scale', C.c_void_p), ('shift', C.c_void_p), ('act', C.c_int),
                ('indptr', C.c_void_p), ('indices', C.c_void_p), ('values', C.c_void_p), ('wt', C.c_void_p), ('ldwt', C.c_int),
                ('dk', C.c_int), ('bias', C.c_void_p), ('row_scale', C.c_void_p)]


class RankSide(C.Structure):
    _fields_ = [('side', C.c_int), ('gt_col', C.c_void_p), ('col0', C.c_int), ('Ev', C.c_void_p), ('Nv', C.c_int), ('s_gt64', C.c_void_p),
                ('band', C.c_void_p), ('band_v', C.c_void_p), ('count', C.c_void_p), ('pairs', C.c_void_p), ('partials', C.c_void_p),
                ('tickets', C.c_void_p)]


class FcFusedProblem(C.Structure):
    _fields_ = [('X', C.c_void_p), ('ldx', C.c_int), ('x_rscale', C.c_void_p), ('N', C.c_int), ('Dk', C.c_int),
                ('Ws', C.c_void_p), ('w_rscale', C.c_void_p), ('bias', C.c_void_p), ('bn_scale', C.c_void_p),
                ('bn_shift', C.c_void_p), ('D', C.c_int), ('act', C.c_int), ('Y', C.c_void_p), ('ldy', C.c_int)]


class FcStripProblem(C.Structure):
    _fields_ = [('X', C.c_void_p), ('ldx', C.c_int), ('N', C.c_int), ('img', C.c_void_p), ('D', C.c_int), ('act', C.c_int),
                ('Y', C.c_void_p), ('ldy', C.c_int)]


class FcProblem(C.Structure):
    _fields_ = [('X', C.c_void_p), ('N', C.c_int), ('Dk', C.c_int), ('ldx', C.c_int), ('W', C.c_void_p), ('ldw', C.c_int),
                ('bias', C.c_void_p), ('bn_scale', C.c_void_p), ('bn_shift', C.c_void_p), ('D', C.c_int), ('act', C.c_int),
                ('Y', C.c_void_p), ('ldy', C.c_int)]


class FcSplitProblem(C.Structure):
    _fields_ = [('Xs', C.c_void_p), ('x_rscale', C.c_void_p), ('N', C.c_int), ('Dk', C.c_int), ('Ws', C.c_void_p),
                ('w_rscale', C.c_void_p), ('bias', C.c_void_p), ('bn_scale', C.c_void_p), ('bn_shift', C.c_void_p),
                ('D', C.c_int), ('act', C.c_int), ('Y', C.c_void_p), ('ldy', C.c_int)]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    'laff_abi_version': (C.c_int, []),
    'laff_last_error': (C.c_char_p, []),
    'laff_ctx_create': (C.c_int, [_I, _P, C.POINTER(_P)]),
    'laff_ctx_set_stream': (C.c_int, [_P, _P]),
    'laff_ctx_destroy': (C.c_int, [_P]),
    'laff_device_info': (C.c_int, [_P, C.POINTER(_I)]),
    'laff_stamp': (C.c_int, [_P, _P]),
    'laff_wall_clock_khz': (C.c_int, [_P, C.POINTER(_I)]),
    'laff_fc_act_bn': (C.c_int, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P, _I]),
    'laff_fc_act_bn_grouped': (C.c_int, [_P, C.POINTER(FcProblem), _I]),
    'laff_fc_gather_act_bn': (C.c_int, [_P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P, _I]),
    'laff_margin_loss_workspace_bytes': (C.c_int, [_I, _I, _I, C.POINTER(C.c_size_t)]),
    'laff_margin_loss': (C.c_int, [_P, _P, _P, _I, _I, _I, C.c_float, C.c_uint, _P, _P, _P, _P, C.c_size_t]),
    'laff_row_scales_grouped': (C.c_int, [_P, _I, C.POINTER(C.c_void_p), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I),
                                          C.POINTER(C.c_void_p)]),
    'laff_fc_act_bn_fused_grouped': (C.c_int, [_P, C.POINTER(FcFusedProblem), _I]),
    'laff_fc_strip_pack_bytes': (C.c_int, [_I, _I, C.POINTER(C.c_size_t)]),
    'laff_fc_strip_pack': (C.c_int, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    'laff_fc_act_bn_strip_grouped': (C.c_int, [_P, C.POINTER(FcStripProblem), _I]),
    'laff_frame_fuse_grouped': (C.c_int, [_P, _I, C.POINTER(C.c_void_p), _P, _I, _I, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.c_uint, C.POINTER(C.c_void_p)]),
    'laff_frame_fuse_grouped_mask': (C.c_int, [_P, _I, C.POINTER(C.c_void_p), _P, _I, _I, _I, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.c_uint, C.POINTER(C.c_void_p)]),
    'laff_split_rows_bytes': (C.c_int, [_I, _I, C.POINTER(C.c_size_t)]),
    'laff_split_rows': (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    'laff_split_rows_grouped': (C.c_int, [_P, _I, C.POINTER(C.c_void_p), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'laff_fc_act_bn_split_grouped': (C.c_int, [_P, C.POINTER(FcSplitProblem), _I]),
    'laff_plane_row_norms': (C.c_int, [_P, C.POINTER(Plane), _I, _I, _I, _I, C.c_uint, _P]),
    'laff_fuse': (C.c_int, [_P, C.POINTER(Plane), _I, _I, _I, _I, _P, _P, _P, C.c_uint, _P, _P]),
    'laff_fuse_packed': (C.c_int, [_P, C.POINTER(Plane), _I, _I, _I, _I, _P, _P, _P, C.c_uint, _P, _P, _P, _I, _F]),
    'laff_fuse_packed_rank': (C.c_int, [_P, C.POINTER(Plane), _I, _I, _I, _I, _P, _P, _P, C.c_uint, _P, _P, _P, _I, _F, C.POINTER(RankSide)]),
    'laff_frame_fuse': (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P, C.c_uint, _P]),
    'laff_packed_bytes': (C.c_int, [_I, _I, _I, C.POINTER(C.c_size_t)]),
    'laff_pack_rows': (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    'laff_sim_gemm': (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _P, _I, _P, _I, _P, _P]),
    'laff_row_dot_gt': (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _P, _I, _P, _P]),
    'laff_rank_prepare': (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _P]),
    'laff_match_ids': (C.c_int, [C.c_char_p, C.c_size_t, _I, C.c_char_p, C.c_size_t, _I, _P]),
    'laff_rank_prepare_part': (C.c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _P]),
    'laff_rank_prepare_emit': (C.c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _P]),
    'laff_rank_export_pairs': (C.c_int, [_P, _P, _P, _P, _I, _I, _P, C.c_uint, _P, _I, _I, _P, C.c_uint, _P]),
    'laff_sim_gemm_banded': (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, C.c_uint]),
    'laff_rank_resolve': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, C.c_uint]),
    'laff_rank_resolve_metrics': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, C.c_uint, _I, _P, _P, _I]),
    'laff_gather_gt': (C.c_int, [_P, _P, _I, _I, _I, _P, _I, _P]),
    'laff_rank_count': (C.c_int, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _I]),
    'laff_topk_rows': (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'laff_v2t_count': (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _I, _P]),
    'laff_comm_unique_id': (C.c_int, [_P]),
    'laff_comm_init': (C.c_int, [_P, _I, _I, _P, C.POINTER(_P)]),
    'laff_comm_set_stream': (C.c_int, [_P, _P]),
    'laff_comm_destroy': (C.c_int, [_P]),
    'laff_allgather_rows': (C.c_int, [_P, _P, _P, C.c_size_t]),
    'laff_allreduce_i32_sum': (C.c_int, [_P, _P, C.c_size_t]),
    'laff_allreduce_f64_max': (C.c_int, [_P, _P, C.c_size_t]),
    'laff_v2t_count_exact': (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, C.c_uint]),
    'laff_rank_metrics': (C.c_int, [_P, _P, _I, _I, _P, C.POINTER(C.c_double)]),
    'laff_rank_metrics_async': (C.c_int, [_P, _P, _I, _I, _P, _P]),
}

_lib = None


def load():
    """Load the library once; raises if it has not been built (python -m laff_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('liblaff_hip.so is missing (%s): build it with `python -m laff_amd.build`; '
                               'laff_amd has no CPU fallback' % LIB_PATH)
        # torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  It must be in the process BEFORE
        # liblaff_hip.so is loaded so that both resolve to ONE HIP runtime (shared streams / allocations); loaded
        # the other way round the second runtime sees no device.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.laff_abi_version() != ABI_VERSION:
            raise RuntimeError('liblaff_hip.so ABI %d != binding ABI %d: rebuild' % (lib.laff_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError('liblaff_hip: %s (rc=%d)' % (load().laff_last_error().decode(), rc))
