"""ctypes binding of libhawkeye_hip.so (C ABI declared in include/hawkeye_hip.h).

The library is loaded AFTER torch so that its NEEDED libamdhip64.so.7 resolves to
the HIP runtime torch already mapped (one runtime => torch's stream handles are
valid inside the kernels' launches).  There is no CPU fallback: if the library
is missing or a tensor is not on a HIP device, the op raises.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the .so is dlopen'ed)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libhawkeye_hip.so')

c_f = ctypes.c_void_p       # device float*
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_ll = ctypes.c_longlong
c_fl = ctypes.c_float

# name -> (restype, argtypes); mirrors include/hawkeye_hip.h one to one
SIGNATURES = {
    'hk_version': (ctypes.c_char_p, []),
    'hk_tuning_set': (c_i, [ctypes.c_char_p, c_i]),
    'hk_tuning_get': (c_i, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    'hk_bcnn_pool_ws_bytes': (c_sz, [c_i, c_i, c_i]),
    'hk_bcnn_pool_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_pool_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_pool_bwd_tdot': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_ssqrt_ws_bytes': (c_sz, [c_i, c_i, c_i]),
    'hk_bcnn_ssqrt_pool_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_ssqrt_pool_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_ssqrt_pool_fwd_parts': (c_i, [c_f, c_f, c_f, ctypes.c_void_p, c_i, c_i, c_i, c_f]),     # (4th: int* on the HOST - pass ctypes.byref)
    'hk_linear_fwd_ssq': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_ssqrt_pool_bwd_tdot': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_ssqrt_pool_fwd_unscaled': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_ssqrt_pool_bwd_unscaled': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_colsum_norm': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_bcnn_gram_norm': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_bcnn_bwd_gemm': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_bcnn_bwd_rank1': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_cov_pool_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_cov_pool_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_ns_sqrtm_ws_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'hk_ns_sqrtm_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_ns_sqrtm_fwd_sym': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_ns_sqrtm_triu_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_ns_sqrtm_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_ns_sqrtm_bwd_general': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_triu_vec_fwd': (c_i, [c_f, c_f, c_i, c_i, c_f]),
    'hk_triu_vec_bwd': (c_i, [c_f, c_f, c_i, c_i, c_f]),
    'hk_cbp_plan_bytes': (c_sz, [c_i, c_i]),
    'hk_cbp_plan_build': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f]),
    'hk_cbp_plan_destroy': (c_i, [c_f]),
    'hk_cbp_ws_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'hk_cbp_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_cbp_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_cbp_bin_matrix': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_cbp_unbin_matrix': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_cbp_loc_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_cbp_loc_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_trunk_ws_bytes': (c_sz, [c_i]),
    'hk_bias_relu_fwd': (c_i, [c_f, c_f, c_f, c_ll, c_i, c_f]),
    'hk_bias_relu_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_ll, c_i, c_f, c_sz, c_f]),
    'hk_bias_relu_pool_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_bias_relu_pool_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_conv1_ws_bytes': (c_sz, [c_i]),
    'hk_conv1_bias_relu_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'hk_conv1_bias_relu_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_add_relu_fwd': (c_i, [c_f, c_f, c_ll, c_f]),
    'hk_relu_mask_bwd': (c_i, [c_f, c_f, c_f, c_ll, c_f]),
    'hk_cbp_rect_plan_bytes': (c_sz, [c_i, c_i, c_i]),
    'hk_cbp_rect_plan_build': (c_i, [c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_f]),
    'hk_cbp_rect_bin_matrix': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_cbp_rect_unbin_matrix': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_cbp_rect_loc_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'hk_cbp_rect_loc_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'hk_att_pool_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_att_pool_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_att_pool3_fwd': (c_i, [c_f] * 8 + [c_i] * 5 + [c_f]),
    'hk_att_pool3_bwd': (c_i, [c_f] * 14 + [c_i] * 5 + [c_f]),
    'hk_att_roi_select3': (c_i, [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_fl, c_f, c_f]),     # host arrays passed as void*
    'hk_att_roi_select': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_f]),
    'hk_roi_crop_resize_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'hk_roi_crop_resize_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'hk_roi_boxes': (c_i, [c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_fl, c_f, c_f, c_i, c_f]),
    'hk_osme_gap': (c_i, [c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_osme_scale_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_osme_scale_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_linear_ws_bytes': (c_sz, [c_i, c_i, c_i]),
    'hk_linear_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_linear_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_linear_fwd_scaled': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_linear_bwd_scaled': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_npairs_ws_bytes': (c_sz, [c_i, c_i]),
    'hk_npairs_loss': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_cin_sci_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_cin_sci_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_f]),
    'hk_cin_cci_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'hk_cin_cci_ws_bytes': (c_sz, [c_i, c_i]),
    'hk_cin_cci_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_sz, c_f]),
    'hk_image_finalize': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'hk_bgemm_f32': (c_i, [c_f, c_i, c_ll, c_i, c_f, c_i, c_ll, c_i, c_f, c_i, c_ll, c_i, c_i, c_i, c_i,
                           c_fl, c_fl, c_fl, c_f]),
}

_lib = None


HK_OK, HK_ERR_BAD_ARG, HK_ERR_WORKSPACE, HK_ERR_UNSUPPORTED = 0, -1, -2, -3      # hk_common.h


class HawkeyeHipError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and attach prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HawkeyeHipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C hawkeye_amd/csrc` (there is no CPU fallback for the HIP ops)')
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError => header/library drifted apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise HawkeyeHipError(f'{what} failed with code {rc} '
                              f'({"HK_ERR" if rc < 0 else "hipError_t"})')


def ptr(t):
    """Device pointer of a contiguous fp32/int32 HIP tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HawkeyeHipError('hawkeye_amd ops run on MI355X only: got a CPU tensor (no CPU fallback; '
                              'the CPU reference lives in oracle/ and is test infrastructure)')
    # dense memory: row-major, or NHWC behind a [N,C,H,W] view (the trunk epilogues take channels_last maps as they lie)
    if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        raise HawkeyeHipError('internal error: non-contiguous tensor handed to the C ABI')
    if t.device.index != torch.cuda.current_device():
        # the library launches on the CURRENT device's stream (one process per GPU: Trainer / Tester / bench.py call
        # torch.cuda.set_device once); a tensor of another device would be touched from the wrong stream
        raise HawkeyeHipError(f'tensor lives on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: '
                              f'call torch.cuda.set_device({t.device.index}) (one process per GPU) or wrap the call in '
                              f'`with torch.cuda.device(...)`')
    return ctypes.c_void_p(t.data_ptr())


class tuning:
    """`with tuning(ns_tn=64, bcnn_generic=1): ...` - flips A/B knobs of the library (hk_tuning_set) for tests and
    benchmarks and restores the previous values on exit.  Not used by the product path."""

    def __init__(self, **knobs):
        self.knobs = knobs
        self.saved = {}

    def __enter__(self):
        lib = load()
        for name, value in self.knobs.items():
            old = ctypes.c_int(0)
            check(lib.hk_tuning_get(name.encode(), ctypes.byref(old)), f'hk_tuning_get({name})')
            self.saved[name] = old.value
            check(lib.hk_tuning_set(name.encode(), int(value)), f'hk_tuning_set({name})')
        return self

    def __exit__(self, *exc):
        lib = load()
        for name, value in self.saved.items():
            lib.hk_tuning_set(name.encode(), value)
        return False


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
