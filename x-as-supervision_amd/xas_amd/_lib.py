"""ctypes binding of libxas_hip.so (include/xas_hip.h)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XAS_HIP_LIB') or os.path.join(_HERE, 'libxas_hip.so')   # override: ablation / A-B builds (tools/build_abl.py, tools/gpu/ab_lib.sh)
_lib = None
ABI_VERSION = 3          # include/xas_hip.h / csrc/abi.hip: xas_abi_version()

_T = {'p': ctypes.c_void_p, 'i': ctypes.c_int, 'l': ctypes.c_long, 'f': ctypes.c_float, 'u': ctypes.c_uint,
      'd': ctypes.c_double, 'z': ctypes.c_size_t}


class ConvShape(ctypes.Structure):
    """xas_conv_shape"""
    _fields_ = [(n, ctypes.c_int) for n in ('N', 'Hi', 'Wi', 'Cin', 'Cout', 'R', 'S', 'stride', 'pad', 'Ho', 'Wo', 'mode')] + [
        ('grad_amax', ctypes.c_void_p),      # device pointer to max |v| of the call's tensor operand (x of forward-type, dy of
                                             # gradient launches), or None
        ('x_amax', ctypes.c_void_p)]         # weight gradient: the same for its x argument


# xas_hip.h XAS_PREC_*: arithmetic of the MFMA convolutions.  ConvShape.mode = 0 (process default) or 1 + one of these.
PREC_F32, PREC_BF16, PREC_BF16X6, PREC_F16X3 = 0, 1, 2, 3
PREC_NAMES = {'f32': PREC_F32, 'bf16': PREC_BF16, 'bf16x6': PREC_BF16X6, 'f16x3': PREC_F16X3}


# name -> (argument codes, return code).  's' = pointer to ConvShape.  Last 'p' is the stream
# for every launch function.
SIGNATURES = {
    'xas_abi_version': ('', 'i'),
    'xas_set_tuning': ('i', 'i'),
    'xas_set_precision': ('i', 'i'),
    'xas_get_precision': ('', 'i'),
    'xas_conv_weight_planes': ('si', 'i'),
    'xas_conv_kernel_class': ('si', 'i'),
    'xas_f16_weight_overflow': ('i', 'i'),
    'xas_f16_weight_overflow_peek': ('pp', 'i'),
    'xas_split_weight_bytes': ('lli', 'z'),
    'xas_split_weight': ('ppllip', 'i'),
    'xas_prepare_weights': ('pilp', 'i'),
    'xas_head_workspace_floats': ('iii', 'z'),
    'xas_head_softargmax_fwd': ('piiiiipppippp', 'i'),
    'xas_head_softargmax_bwd': ('ppppiiiiippp', 'i'),
    'xas_head_softargmax_bwd_amax': ('ppppiiiiipppp', 'i'),
    'xas_patch_to_world_fwd': ('ppppppiiiffipp', 'i'),
    'xas_patch_to_world_bwd': ('pppppppiiiffipp', 'i'),
    'xas_lines_nblk': ('i', 'i'),
    'xas_draw_lines_max_fwd': ('plliippiufipp', 'i'),
    'xas_draw_lines_max_bwd': ('plliippiufipppp', 'i'),
    'xas_conv_fwd': ('ppppsp', 'i'),
    'xas_conv_fwd_head_chunks': ('sii', 'i'),
    'xas_conv_fwd_head': ('ppppsiipp', 'i'),
    'xas_head_softargmax_from_partials': ('piiiiiipppipp', 'i'),
    'xas_conv_fwd_bnstats_workspace_floats': ('si', 'z'),
    'xas_conv_fwd_bnstats': ('pppsippplppppfp', 'i'),
    'xas_conv_dgrad_bn_bwd_workspace_floats': ('si', 'z'),
    'xas_conv_dgrad_bn_bwd': ('ppspppppfidppppppp', 'i'),
    'xas_conv_dgrad': ('pppsp', 'i'),
    'xas_conv_dgrad_acc': ('pppsp', 'i'),
    'xas_conv_dgrad_acc_masked': ('pppsppp', 'i'),
    'xas_conv_wgrad_workspace_floats': ('s', 'z'),
    'xas_conv_wgrad': ('ppppsp', 'i'),
    'xas_conv_wgrad_oihw': ('ppppsp', 'i'),
    'xas_conv_wgrad_acc': ('ppppsp', 'i'),
    'xas_pack_weight': ('ppiiiiip', 'i'),
    'xas_unpack_weight': ('ppiiiiip', 'i'),
    'xas_bn_workspace_floats': ('lii', 'z'),
    'xas_bn_stats': ('pliipplppppflp', 'i'),
    'xas_bn_stats_from_partials': ('pliilppplppppfp', 'i'),
    'xas_bn_bwd_sums_from_partials': ('pliippppp', 'i'),
    'xas_bn_sync_merge': ('piiilppppfp', 'i'),
    'xas_col_sum': ('plippp', 'i'),
    'xas_col_sum_acc': ('plippp', 'i'),
    'xas_bn_apply': ('ppppppfiliippp', 'i'),
    'xas_bn_apply_amax': ('ppppppfiliipppp', 'i'),
    'xas_abs_max': ('plpp', 'i'),
    'xas_bn_update_running': ('ppppfliip', 'i'),
    'xas_bn_bwd_reduce': ('pppppppfiliipppppp', 'i'),
    'xas_bn_bwd_apply': ('ppppppppfiliidpppp', 'i'),
    'xas_bn_bwd_apply_amax': ('ppppppppfiliidppppp', 'i'),
    'xas_maxpool3x3s2_fwd': ('piiiippp', 'i'),
    'xas_maxpool3x3s2_bwd': ('ppiiiipp', 'i'),
    'xas_upsample2x_fwd': ('piiiipp', 'i'),
    'xas_upsample2x_bwd': ('piiiipp', 'i'),
    'xas_sigmoid_fwd': ('plpp', 'i'),
    'xas_sigmoid_bwd': ('pplpp', 'i'),
    'xas_nchw_to_nhwc': ('piiiipp', 'i'),
    'xas_nhwc_to_nchw': ('piiiipp', 'i'),
    'xas_warp_affine_u8': ('ppppiiipp', 'i'),
    'xas_mask_blur_threshold': ('piipp', 'i'),
    'xas_patch_finish': ('pppppiiippp', 'i'),
    'xas_geodesic_workspace_bytes': ('ii', 'z'),
    'xas_geodesic_weight': ('pppiipppp', 'i'),
    'xas_geodesic_weight_multi': ('ppiipiipppp', 'i'),
    'xas_loss_nblk': ('l', 'i'),
    'xas_mask_loss_fwd': ('pppliPpp'.replace('P', 'p'), 'i'),
    'xas_mask_loss_bwd': ('pppliPppp'.replace('P', 'p'), 'i'),
    'xas_pose_loss_fwd': ('ppiiiifffpp', 'i'),
    'xas_pose_loss_bwd': ('ppiiiifffppppp', 'i'),
    'xas_lsgan_fwd': ('piifppp', 'i'),
    'xas_lsgan_bwd': ('piifpppp', 'i'),
    'xas_graph_aggregate': ('ppiiipp', 'i'),
    'xas_gln_workspace_floats': ('lii', 'z'),
    'xas_gln_fwd': ('ppppliifpppp', 'i'),
    'xas_gln_bwd': ('pppppliifppppp', 'i'),
    'xas_smpl_lbs_fwd': ('ppppppppiiipppp', 'i'),
    'xas_smpl_lbs_bwd_workspace_floats': ('ii', 'z'),
    'xas_smpl_lbs_bwd': ('ppppppppiiippppppp', 'i'),
    'xas_adam_step': ('pppplffffip', 'i'),
    'xas_eval_select': ('pppiiiifippppp', 'i'),
    'xas_projection_matrix': ('pppipp', 'i'),
    'xas_triangulate_dlt': ('ppiiipp', 'i'),
    'xas_pose_metrics': ('pppiififppppp', 'i'),
}


def load():
    """Load the library (idempotent).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libxas_hip.so is missing (%s): run `python __graft_entry__.py` to build the HIP '
                           'library; the MI355X path has no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.xas_abi_version.restype = ctypes.c_int
    lib.xas_abi_version.argtypes = []
    if lib.xas_abi_version() != ABI_VERSION:
        # (layout-breaking changes between versions: xas_conv_shape fields, the 1024-float slots of recorded maxima)
        raise RuntimeError('%s speaks ABI version %d, this binding expects %d (include/xas_hip.h): rebuild it with '
                           '`python __graft_entry__.py`' % (LIB_PATH, lib.xas_abi_version(), ABI_VERSION))
    lib.xas_last_error.restype = ctypes.c_char_p
    lib.xas_last_error.argtypes = []
    _lib = lib
    mode = os.environ.get('XAS_PRECISION', '')          # '' = library default (3, f16x3: launches without operand maxima run as bf16x6); '2' = bf16x6; '0' = exact fp32 MFMA; '1' = bf16
    if mode != '':
        lib.xas_set_precision.argtypes = [ctypes.c_int]
        lib.xas_set_precision.restype = ctypes.c_int
        if lib.xas_set_precision(int(mode)) != 0:
            raise RuntimeError('XAS_PRECISION=%s: %s' % (mode, lib.xas_last_error().decode()))
    global PREC_DEFAULT
    lib.xas_get_precision.restype = ctypes.c_int
    PREC_DEFAULT = int(lib.xas_get_precision())         # the process default: the library's, or XAS_PRECISION's
    return lib


PREC_DEFAULT = None


_bound = {}


def fn(name):
    """Bind one entry point (argtypes from SIGNATURES) on first use."""
    f = _bound.get(name)
    if f is None:
        args, ret = SIGNATURES[name]
        f = getattr(load(), name)      # AttributeError if the symbol is not exported
        f.argtypes = [ctypes.POINTER(ConvShape) if c == 's' else _T[c] for c in args]
        f.restype = _T[ret]
        _bound[name] = f
    return f


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('xas_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback' % t.device)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


profiler = None      # set by xas_amd.prof.KernelTimer: brackets selected entry points with HIP events


def call(name, *args):
    """Launch wrapper: appends the current HIP stream and turns a non-zero status into RuntimeError."""
    if profiler is not None and name in profiler.names:
        return profiler.timed_call(name, args)
    rc = fn(name)(*args, stream())
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, load().xas_last_error().decode()))


def query(name, *args):
    return fn(name)(*args)
