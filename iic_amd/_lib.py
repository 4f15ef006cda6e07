"""ctypes binding of libiic_hip.so (the C ABI declared in include/iic_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
cannot be resolved, importing any op raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C iic_amd/csrc``).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_int32, c_long, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two flavours of the same sources (iic_amd/csrc/Makefile): libiic_hip.so, the product -- exports exactly the C ABI of
# include/iic_hip.h, every measurement switch a compile-time constant -- and libiic_hip_dbg.so (`make dbg`), the same code
# with the iic_debug_* switches compiled in.  IIC_HIP_LIB=dbg selects the instrumented one for a whole process (tests
# marked `hooks`, tools/*.py); nothing in the product path depends on it.
HAS_HOOKS = os.environ.get("IIC_HIP_LIB", "") == "dbg"
LIB_PATH = os.path.join(_HERE, "libiic_hip_dbg.so" if HAS_HOOKS else "libiic_hip.so")

IIC_MAX_TAPS = 32
IIC_STAT_STRIPES = 32


class ConvGeom(Structure):
  """Mirror of ``iic_conv_geom`` (include/iic_hip.h)."""
  _fields_ = [
    ("N", c_int32), ("MY", c_int32), ("MX", c_int32),
    ("in_Hp", c_int32), ("in_Wp", c_int32), ("Cin", c_int32),
    ("sy", c_int32), ("sx", c_int32), ("oy", c_int32), ("ox", c_int32),
    ("out_Hp", c_int32), ("out_Wp", c_int32), ("Cout", c_int32),
    ("ty", c_int32), ("tx", c_int32), ("py", c_int32), ("px", c_int32),
    ("ntaps", c_int32),
    ("tap_off", c_int32 * IIC_MAX_TAPS),
    ("tap_w", c_int32 * IIC_MAX_TAPS),
    ("NP", c_int32),
    ("NP256", c_int32),
    ("NP64", c_int32),
    ("MP", c_int32),
  ]


_P = c_void_p
_SIGNATURES = {
  "iic_version": (c_int, []),
  "iic_iid_nsplit": (c_int, [c_int]),
  "iic_iid_workspace_bytes": (c_long, [c_int, c_int]),
  "iic_iid_joint_raw": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_long, c_long, c_int, _P]),
  "iic_iid_loss_from_joint": (c_int, [_P, c_int, c_int, c_int, c_double, c_double, _P, _P, _P, _P, _P, _P]),
  "iic_iid_grad": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_long, c_long, _P]),
  "iic_seg_joint_nsplit": (c_int, [c_int, c_int, c_int, c_int]),
  "iic_seg_joint_raw": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_seg_loss_from_joint": (c_int, [_P, c_int, c_int, c_int, c_double, c_double, _P, _P, _P, _P, _P, c_int, _P]),
  "iic_seg_grad_workspace_bytes": (c_long, [c_int, c_int]),
  "iic_seg_grad": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
  "iic_affine_warp_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_affine_warp_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_conv_lds_bytes": (c_long, [POINTER(ConvGeom), c_int]),
  "iic_conv_igemm": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P, c_int, _P]),
  "iic_conv_wgrad_nsplit": (c_int, [POINTER(ConvGeom)]),
  "iic_conv_wgrad": (c_int, [POINTER(ConvGeom), _P, _P, _P, c_int, c_int, _P]),
  "iic_conv_wgrad_reduce": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
  "iic_weight_prep": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
  "iic_conv_igemm_frag_supported": (c_int, [POINTER(ConvGeom)]),
  "iic_conv_igemm_frag": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P, c_int, _P]),
  "iic_conv_igemm_red_supported": (c_int, [POINTER(ConvGeom)]),
  "iic_conv_igemm_frag_red": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
  "iic_weight_prep_frag": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_weight_prep_multi_blocks": (c_long, [c_int, c_int, c_int]),
  "iic_weight_prep_multi": (c_int, [_P, c_int, c_long, _P]),
  "iic_bn_finalize": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_long, c_long, c_float, c_float, c_int, _P]),
  "iic_stat_bytes": (c_long, [c_int]),
  "iic_bn_running_update": (c_int, [c_int, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(c_int),
                                    c_float, _P]),
  "iic_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_bn_bwd_finalize": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_long, _P]),
  "iic_bn_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_stem_stats": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_stem_apply_pool": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_stem_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_stem_wgrad_partial_floats": (c_long, []),
  "iic_stem_bwd_wgrad": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_contingency": (c_int, [_P, _P, c_long, c_int, c_int, _P, _P]),
  "iic_count_equal": (c_int, [_P, _P, c_long, _P, _P]),
  "iic_augment": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P, c_int, _P, _P]),
  "iic_stem_bwd_fused": (c_int, [_P, _P, _P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_int, _P]),
  "iic_stem_wgrad_combine": (c_int, [_P, c_int, _P, _P, c_int, _P]),
  "iic_firstconv_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_firstconv_wgrad_partial_floats": (c_long, []),
  "iic_firstconv_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_maxpool2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_maxpool2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_bn_relu_maxpool2_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_bn_relu_maxpool2_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_sobel": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_avgpool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_avgpool_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
  "iic_gemm_f32": (c_int, [_P, c_long, c_long, _P, c_long, c_long, _P, _P, c_long, c_int, c_int, c_int, c_int, _P]),
  "iic_gemm_f32_ws_floats": (c_long, [c_long, c_long, c_long, c_long, c_int, c_int, c_int]),
  "iic_gemm_f32_ws": (c_int, [_P, c_long, c_long, _P, c_long, c_long, _P, _P, c_long, c_int, c_int, c_int, c_int, _P, c_long, _P]),
  "iic_gemm_f32_splitk": (c_int, [_P, c_long, c_long, _P, c_long, c_long, _P, c_long, c_int, c_int, c_int, c_int, _P]),
  "iic_seg_window_gather": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_seg_window_scatter": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_seg_head_supported": (c_int, [c_int, c_int]),
  "iic_seg_head_wgrad_chunks": (c_int, [c_long]),
  "iic_seg_head_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_seg_head_bwd_dx": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_seg_head_wgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_bilinear_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_bilinear_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_softmax_fwd": (c_int, [_P, _P, c_int, c_int, _P]),
  "iic_softmax_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P]),
  "iic_colsum_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
  "iic_adam_step": (c_int, [c_int, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(c_long),
                            c_float, c_float, c_float, c_float, c_int, _P]),
  "iic_adam_step_dev": (c_int, [c_int, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(c_long),
                                c_float, c_float, c_float, c_float, _P, _P]),
  "iic_adam_step_devlr": (c_int, [c_int, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(c_long),
                                  _P, c_float, c_float, c_float, c_float, _P, _P]),
  "iic_f32_conv": (c_int, [POINTER(ConvGeom), _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
  "iic_f32_wgrad": (c_int, [POINTER(ConvGeom), _P, _P, _P, c_int, c_int, _P]),
  "iic_f32_bn_apply": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_bn_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_avgpool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_avgpool_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
  "iic_f32_maxpool_s2p1_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_maxpool_s2p1_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_maxpool2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_maxpool2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_window_gather": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_window_scatter": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_f32_nchw_to_pt": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
  "iic_probe_tr16": (c_int, [_P, _P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


class IICLibraryError(RuntimeError):
  pass


def lib():
  """Load (once) and return the ctypes handle; raise loudly when it is absent."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise IICLibraryError(
        "%s not found -- build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU/PyTorch "
        "fallback for the IIC hot path." % LIB_PATH)
    h = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
      try:
        fn = getattr(h, name)
      except AttributeError as e:
        raise IICLibraryError("libiic_hip.so lacks symbol %s" % name) from e
      fn.restype = res
      fn.argtypes = args
    # A/B switches for measurements (instrumented library only), e.g. IIC_HIP_LIB=dbg IIC_DEBUG="iic_debug_bd_ms=2":
    # calls the named iic_debug_* setters (int argument) once at load.  Unset = defaults.
    for item in filter(None, os.environ.get("IIC_DEBUG", "").split(",")):
      name, _, val = item.partition("=")
      if not name.startswith("iic_debug_"):
        raise IICLibraryError("IIC_DEBUG: %r is not an iic_debug_* switch" % name)
      if not HAS_HOOKS:
        raise IICLibraryError("IIC_DEBUG needs the instrumented library: IIC_HIP_LIB=dbg (make -C iic_amd/csrc dbg)")
      getattr(h, name)(int(val or 1))
    _lib = h
  return _lib


_ERR = {-1: "IIC_ERR_ARG", -2: "IIC_ERR_LAUNCH", -3: "IIC_ERR_UNSUPPORTED"}


def check(rc, what=""):
  if rc != 0:
    raise IICLibraryError("%s failed: %s (%d)" % (what or "libiic_hip call", _ERR.get(rc, "?"), rc))


def ptr(t):
  """Device pointer of a torch tensor (None -> NULL)."""
  return None if t is None else t.data_ptr()


_RAW_STREAM = None


def stream_ptr():
  """hipStream_t of torch's current stream on the current device.  Uses torch's raw-stream
  accessor (0.2 us) instead of building a torch.cuda.Stream object per launch (several us: at
  ~1100 launches per step the host was close to becoming the bottleneck)."""
  global _RAW_STREAM
  import torch
  if _RAW_STREAM is None:
    _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)
  if _RAW_STREAM:
    return _RAW_STREAM(torch.cuda.current_device())
  return torch.cuda.current_stream().cuda_stream
