"""ctypes binding of libdvd_hip.so (the C ABI in include/dvd_hip.h).

No torch types cross the boundary: tensors are passed as raw device pointers
(`tensor.data_ptr()`), shapes as ints, the stream as the hipStream_t handle of
torch's current stream.  A missing library is a hard error (no CPU fallback).
"""
import ctypes
import os
import threading

from . import build as _build

c_float_p = ctypes.c_void_p   # device pointers travel as void*
c_size_t = ctypes.c_size_t
c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p

DVD_OK, DVD_EINVAL, DVD_EHIP, DVD_ENOSPC = 0, -1, -2, -3
ABI_VERSION = 7


class Cameras(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')]


class WarpCfg(ctypes.Structure):
    _fields_ = [('B', c_int), ('H', c_int), ('W', c_int), ('midas_mask', c_int), ('crit_l2', c_int),
                ('disp_mode', c_int), ('loss_on_sf', c_int), ('flow_mul', c_float), ('disp_mul', c_float)]


SURFACE_KEYS = ('global_p1', 'warped_global_p2', 'sf_by_depth', 'staticflow_1_2', 'dflow_1_2', 'depth_image_1_2',
                'depth_warp_1_2', 'p1_camera_2', 'warped_p2_camera_2')


class Surfaces(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in SURFACE_KEYS]


class MlpDesc(ctypes.Structure):
    _fields_ = [('n_freq_xyz', c_int), ('n_freq_t', c_int), ('time_dependent', c_int), ('freqs_xyz', c_void_p),
                ('freqs_t', c_void_p), ('stash_f16', c_int), ('fwd_monitor', c_void_p)]


c_longlong = ctypes.c_longlong


class BnParams(ctypes.Structure):           # struct dvd_bn_params
    _fields_ = [('gamma', c_void_p), ('beta', c_void_p), ('mean', c_void_p), ('var', c_void_p), ('eps', ctypes.c_float)]

class XPackItem(ctypes.Structure):          # struct dvd_xpack_item
    _fields_ = [('w', c_void_p), ('packed', c_void_p), ('gamma', c_void_p), ('var', c_void_p), ('eps', ctypes.c_float),
                ('Cout', c_int), ('Cin', c_int), ('KS', c_int), ('groups', c_int), ('transposed', c_int)]


PtrArr6 = c_void_p * 6
PtrArr5 = c_void_p * 5

# name -> (restype, argtypes); must list every symbol declared in include/dvd_hip.h
SIGNATURES = {
    'dvd_abi_version': (c_int, []),
    'dvd_last_error': (ctypes.c_char_p, []),
    'dvd_device_cu_count': (c_int, []),
    'dvd_flop_counters': (c_int, [c_void_p, c_int, c_int]),
    'dvd_byte_counters': (c_int, [c_void_p, c_int, c_int]),
    'dvd_unproject_fwd': (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    'dvd_unproject_bwd': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    'dvd_warp_loss_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'dvd_warp_loss_select': (c_int, [c_int, c_int, c_int]),
    'dvd_warp_loss_strip_select': (c_int, [c_int, c_int]),
    'dvd_warp_loss_fused': (c_int, [ctypes.POINTER(WarpCfg)] + [c_void_p] * 5 + [ctypes.POINTER(Cameras),
                                                                               c_void_p, c_size_t] +
                            [c_void_p] * 4 + [c_void_p]),
    'dvd_warp_loss_fwd': (c_int, [ctypes.POINTER(WarpCfg)] + [c_void_p] * 5 + [ctypes.POINTER(Cameras), c_void_p,
                                                                             c_size_t, c_void_p, c_void_p]),
    'dvd_loss_finalize': (c_int, [ctypes.POINTER(WarpCfg), c_void_p, c_void_p, c_void_p]),
    'dvd_sf_mlp_in_channels': (c_int, [ctypes.POINTER(MlpDesc)]),
    'dvd_sf_mlp_packed_bytes': (c_size_t, [ctypes.POINTER(MlpDesc)]),
    'dvd_sf_mlp_stash_bytes': (c_size_t, [ctypes.POINTER(MlpDesc), c_longlong]),
    'dvd_sf_mlp_gstash_bytes': (c_size_t, [c_longlong]),
    'dvd_sf_mlp_pack': (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(PtrArr6), ctypes.POINTER(PtrArr6), c_void_p,
                                c_void_p]),
    'dvd_sf_mlp_fwd': (c_int, [ctypes.POINTER(MlpDesc), c_void_p, c_void_p, c_void_p, c_float, c_float, c_longlong,
                               c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dvd_sf_mlp_bwd_dx': (c_int, [ctypes.POINTER(MlpDesc), c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p,
                                  c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    'dvd_mul_mask': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p]),
    'dvd_scale_add': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_longlong, c_void_p]),
    'dvd_subsample2_fwd': (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_void_p]),
    'dvd_subsample2_bwd': (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_void_p]),
    'dvd_avgpool_fwd': (c_int, [c_void_p, c_void_p, c_int, c_longlong] + [c_int] * 5 + [c_void_p]),
    'dvd_avgpool_bwd': (c_int, [c_void_p, c_void_p, c_int, c_longlong] + [c_int] * 5 + [c_void_p]),
    'dvd_depth_tail_fwd': (c_int, [c_void_p, c_void_p, c_longlong, c_void_p]),
    'dvd_depth_tail_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    'dvd_acc_reg_workspace_bytes': (c_size_t, []),
    'dvd_acc_reg': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p]),
    'dvd_adam_step': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float,
                              c_float, c_float, c_float, c_int, c_void_p]),
    'dvd_warp_surfaces': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Cameras),
                                  ctypes.POINTER(Surfaces), c_int, c_int, c_int, c_void_p]),
    'dvd_warp_surfaces_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Cameras),
                                      ctypes.POINTER(Surfaces), c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'dvd_flow_warp_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_flow_warp_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_bnrelu_fwd': (c_int, [c_void_p] * 6 + [c_float, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_bnrelu_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'dvd_bnrelu_bwd': (c_int, [c_void_p] * 6 + [c_float] + [c_void_p] * 5 + [c_size_t, c_int, c_int, c_int, c_int,
                                                                           c_void_p, c_void_p]),
    'dvd_upsample_bilinear_fwd': (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_upsample_bilinear_bwd': (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gconv3x3_c8_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gconv3x3_c8_bwd_data': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gconv3x3_c8_wgrad_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'dvd_gconv3x3_c8_bwd_weight': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_int,
                                           c_int, c_void_p]),
    'dvd_gconv3x3_c32_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'dvd_gconv3x3_c32_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    'dvd_gconv3x3_c32_bwd_data': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    'dvd_gconv3x3_c32_bwd_weight': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int,
                                            c_int, c_int, c_void_p]),
    'dvd_xconv_packed_bytes': (c_size_t, [c_int] * 5),
    'dvd_xconv_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_xconv_pack_table_bytes': (c_size_t, [c_int]),
    'dvd_xconv_pack_many': (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    'dvd_amax': (c_int, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p]),
    'dvd_chansum_workspace_bytes': (c_size_t, [c_int, c_int]),
    'dvd_chansum': (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dvd_xconv_fwd': (c_int, [c_void_p] * 6 + [ctypes.POINTER(BnParams), c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    'dvd_xconv_select': (c_int, [c_int]),
    'dvd_xconv_pack_scaled': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                      c_void_p]),
    'dvd_convbn_finalize': (c_int, [c_void_p] * 6 + [c_float, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dvd_xwgrad3_workspace_bytes': (c_size_t, [c_int] * 6),
    'dvd_xwgrad1s_workspace_bytes': (c_size_t, [c_int] * 5),
    'dvd_xwgrad_select': (c_int, [c_int]),
    'dvd_sf_mlp_select': (c_int, [c_int]),
    'dvd_xwgradk_workspace_bytes': (c_size_t, [c_int] * 6),
    'dvd_xwgradk': (c_int, [c_void_p] * 6 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    'dvd_xwgrad1s': (c_int, [c_void_p] * 6 + [c_size_t] + [c_int] * 6 + [c_void_p]),
    'dvd_xwgrad1s_rowsum': (c_int, [c_void_p] * 7 + [c_size_t] + [c_int] * 6 + [c_void_p]),
    'dvd_xwgrad3': (c_int, [c_void_p] * 6 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    'dvd_xwgrad_workspace_bytes': (c_size_t, [c_int] * 6),
    'dvd_xwgrad': (c_int, [c_void_p] * 4 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    'dvd_flow_consistency_mask': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'dvd_sf_mlp_bwd_dw': (c_int, [ctypes.POINTER(MlpDesc), c_void_p, c_void_p, c_longlong, ctypes.POINTER(PtrArr5),
                                  ctypes.POINTER(PtrArr5), c_void_p]),
    # fp16 activation storage (configs[4])
    'dvd_xconv_fwd_h': (c_int, [c_void_p] * 5 + [ctypes.POINTER(BnParams), c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    'dvd_xwgrad3_h': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    'dvd_xwgrad1s_h': (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 6 + [c_void_p]),
    'dvd_bnrelu_fwd_t': (c_int, [c_void_p] * 6 + [c_float, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_bnrelu_bwd_t': (c_int, [c_void_p] * 6 + [c_float] + [c_void_p] * 5 + [c_size_t, c_int, c_void_p, c_int, c_int, c_int,
                                                                             c_int, c_void_p, c_void_p]),
    'dvd_bnrelu_fwd_m': (c_int, [c_void_p] * 6 + [c_float, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dvd_bnrelu_bwd_m': (c_int, [c_void_p] * 6 + [c_float] + [c_void_p] * 5 + [c_size_t, c_int, c_void_p, c_int, c_int, c_int,
                                                                             c_int, c_void_p, c_void_p, c_void_p]),
    'dvd_upsample_bilinear_fwd_t': (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_upsample_bilinear_bwd_t': (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gconv3x3_c8_fwd_t': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gconv3x3_c8_bwd_data_t': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gconv3x3_c8_bwd_weight_t': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p, c_int,
                                             c_int, c_int, c_int, c_void_p]),
    'dvd_head1x1_fwd': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_head1x1_bwd_workspace_bytes': (c_size_t, [c_int]),
    'dvd_head1x1_bwd': (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_size_t, c_int, c_int, c_int, c_int, c_void_p]),
    'dvd_gscale_init': (c_int, [c_void_p, c_float, c_void_p]),
    'dvd_gscale_step_begin': (c_int, [c_void_p, c_void_p]),
    'dvd_gscale_begin': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'dvd_gscale_end': (c_int, [c_void_p, c_void_p]),
    'dvd_cast_scale_f32': (c_int, [c_void_p, c_int, c_void_p, c_longlong, c_void_p, c_void_p]),
    'dvd_maxpool3s2_fwd': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    'dvd_maxpool3s2_bwd': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    'dvd_adam_step_guarded': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float,
                                      c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
}

_lock = threading.Lock()
_lib = None


def library_path():
    return os.environ.get('DVD_HIP_LIB') or _build.lib_path()        # an empty DVD_HIP_LIB counts as unset


def load():
    """Load (once) and type the library.  Raises RuntimeError if it is absent:
    build it with `python -m dvd_hip.build` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libdvd_hip.so must bind to the HIP runtime torch has loaded (the one that owns the
    # tensors and streams handed to it), not pull a second libamdhip64 of its own into the process
    import torch  # noqa: F401
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path) and not os.environ.get('DVD_HIP_LIB'):
            try:                                  # fresh checkout: compile in-tree (hipcc cross-compiles gfx950)
                _build.build_library()
            except Exception as e:                # noqa: BLE001 -- reported below as the missing-library error
                raise RuntimeError('libdvd_hip.so not found at %s and building it failed (%s); '
                                   'dvd_hip has no CPU fallback' % (path, e))
        if not os.path.exists(path):
            raise RuntimeError('libdvd_hip.so not found at %s -- run `python -m dvd_hip.build`; '
                               'dvd_hip has no CPU fallback' % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.dvd_abi_version() != ABI_VERSION:
            raise RuntimeError('libdvd_hip.so ABI %d != binding %d' % (lib.dvd_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(status, what):
    if status != DVD_OK:
        msg = load().dvd_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what, status, (msg or b'').decode('utf-8', 'replace')))
