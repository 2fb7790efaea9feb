"""ctypes binding of librick_hip.so (C ABI in include/rick_hip.h).

The product path has NO fallback: if the shared library is missing the import fails, and
every op raises on non-CUDA tensors.  PyTorch is used only for device memory, streams and
autograd bookkeeping.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'librick_hip.so')
if os.environ.get('RICK_HIP_LIB'):          # kernel experiments (ablation builds): never silent
    import warnings
    LIB_PATH = os.environ['RICK_HIP_LIB']
    warnings.warn(f'rick_amd: RICK_HIP_LIB overrides the product library with {LIB_PATH}', RuntimeWarning)
MAX_TAPS = 16

c_fp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f = ctypes.c_float


class ConvGeom(ctypes.Structure):
    """rick_conv_geom (include/rick_hip.h)."""
    _fields_ = [('N', c_int), ('IH', c_int), ('IW', c_int), ('Ci', c_int),
                ('OH', c_int), ('OW', c_int), ('Co', c_int),
                ('GH', c_int), ('GW', c_int),
                ('is_', c_int), ('os', c_int), ('oy0', c_int), ('ox0', c_int),
                ('ntaps', c_int), ('nslices', c_int),
                ('dy', c_int * MAX_TAPS), ('dx', c_int * MAX_TAPS), ('wt', c_int * MAX_TAPS),
                ('split', c_int), ('alpha', c_f)]


class ConvEpilogue(ctypes.Structure):
    """rick_conv_epilogue (include/rick_hip.h)."""
    _fields_ = [('bias', c_fp), ('noise', c_fp), ('noise_w', c_fp), ('noise_nb', c_int), ('act', c_int),
                ('slope', c_f), ('gain', c_f), ('amax', c_fp), ('split_out', c_fp), ('split_hdr', c_fp), ('split_bound', c_fp),
                ('split_coef', c_f), ('split_scale', c_fp)]


class ColsumItem(ctypes.Structure):
    """rick_colsum_item (include/rick_hip.h)"""
    _fields_ = [('partials', ctypes.c_void_p), ('out', ctypes.c_void_p), ('out2', ctypes.c_void_p), ('nb', ctypes.c_int),
                ('stride', ctypes.c_int), ('ncols', ctypes.c_int), ('col0', ctypes.c_int), ('split', ctypes.c_int),
                ('accumulate', ctypes.c_int)]


class SplitOut(ctypes.Structure):
    """rick_split_out (include/rick_hip.h)."""
    _fields_ = [('split_out', c_fp), ('split_hdr', c_fp), ('bound0', c_fp), ('bound1', c_fp), ('bound_coef', c_f),
                ('amax', c_fp), ('accumulate', c_int), ('no_f32', c_int), ('chan_scale', c_fp), ('adj_ref', c_fp),
                ('adj_slope', c_f), ('adj_gain', c_f), ('adj_partials', c_fp)]


# name -> (restype, argtypes); every symbol declared in include/rick_hip.h
SIGNATURES = {
    'rick_abi_version': (c_int, []),
    'rick_upfirdn2d_f32': (c_int, [c_fp, c_fp, c_fp, c_i64] + [c_int] * 13 + [c_fp]),
    'rick_upfirdn2d_act_f32': (c_int, [c_fp, c_fp, c_fp, c_i64] + [c_int] * 13 + [ctypes.POINTER(ConvEpilogue), c_fp]),
    'rick_bias_act_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f,
                                  c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_fp]),
    'rick_upfirdn2d_any': (c_int, [c_fp, c_fp, c_fp, c_int, c_i64] + [c_int] * 12 + [c_fp]),
    'rick_bias_act_any': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_f, c_f, c_fp]),
    'rick_bias_act_bwd_blocks': (c_int, [c_i64, c_int]),
    'rick_bias_act_bwd_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_i64, c_i64, c_i64,
                                      c_f, c_f, c_fp, c_int, c_fp]),
    'rick_bias_act_bwd_dot_ok': (c_int, [c_i64, c_int, c_i64]),
    'rick_bias_act_bwd_dot_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_i64, c_i64, c_i64,
                                          c_f, c_f, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'rick_amax_f32': (c_int, [c_fp, c_i64, c_fp, c_fp]),
    'rick_stream_capture_id': (c_int, [c_fp, ctypes.POINTER(ctypes.c_ulonglong)]),
    'rick_split_pack_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_i64, c_int, c_fp]),
    'rick_split_unpack_f32': (c_int, [c_fp, c_fp, c_fp, c_i64, c_int, c_fp]),
    'rick_conv_wgrad_split_supported': (c_int, [ctypes.POINTER(ConvGeom)]),
    'rick_conv_wgrad_split_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, ctypes.POINTER(ConvGeom), c_int,
                                          c_fp, c_fp]),
    'rick_saturation_count': (c_int, [ctypes.POINTER(ctypes.c_uint), c_int]),
    'rick_conv_tuning': (c_int, [c_int, c_int]),
    'rick_upfirdn2d_ex_f32': (c_int, [c_fp, c_fp, c_fp, c_i64] + [c_int] * 13 + [ctypes.POINTER(ConvEpilogue),
                                                                                 ctypes.POINTER(SplitOut), c_fp]),
    'rick_bias_act_bwd_split_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_i64,
                                            c_i64, c_i64, c_f, c_f, c_fp, c_int, c_fp]),
    'rick_d_input_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_int, c_f, c_f, ctypes.POINTER(SplitOut), c_fp]),
    'rick_add_scale_split_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_f, c_fp]),
    'rick_conv_igemm_split_supported': (c_int, [ctypes.POINTER(ConvGeom)]),
    'rick_conv_igemm_split_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(ConvGeom), ctypes.POINTER(ConvEpilogue),
                                          c_fp, c_fp]),
    'rick_convt2_split_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp] + [c_int] * 7 + [c_f, c_fp, c_fp, c_fp]),
    'rick_upfirdn2d_adjoint_rows': (c_i64, [c_i64, c_int, c_int]),
    'rick_colsum_f32': (c_int, [c_fp, c_fp, c_i64, c_int, c_int, c_int, c_fp]),
    'rick_bound_tail_f32': (c_int, [c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_int, c_f, c_fp, c_int, c_fp]),
    'rick_bias_act_bwd_split2_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64,
                                             c_int, c_i64, c_i64, c_i64, c_f, c_f, c_fp, c_int, c_fp]),
    'rick_conv_packed_bytes': (c_i64, [c_int, c_int, c_int]),
    'rick_conv_pack_weight': (c_int, [c_fp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_f, c_int, c_fp, c_fp]),
    'rick_conv_pack_blocks': (c_int, [c_int, c_int]),
    'rick_conv_pack_weights_multi': (c_int, [c_fp, c_int, c_int, c_int, c_fp]),
    'rick_conv_igemm_workspace_bytes': (c_i64, [ctypes.POINTER(ConvGeom)]),
    'rick_conv_igemm_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(ConvGeom), c_fp, c_fp]),
    'rick_conv_igemm_act_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(ConvGeom),
                                        ctypes.POINTER(ConvEpilogue), c_fp, c_fp]),
    'rick_conv_igemm_multi_workspace_bytes': (c_i64, [ctypes.POINTER(ConvGeom), c_int]),
    'rick_conv_igemm_multi_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(ConvGeom), c_int, c_fp, c_fp]),
    'rick_convt2_workspace_bytes': (c_i64, [c_int] * 7),
    'rick_convt2_plan': (c_int, [c_int] * 7 + [ctypes.POINTER(c_int)]),
    'rick_convt2_posmap': (c_int, [c_int] * 7 + [ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(c_int)]),
    'rick_convt2_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp] + [c_int] * 8 + [c_f, c_fp, c_fp]),
    'rick_conv_wgrad_workspace_bytes': (c_i64, [ctypes.POINTER(ConvGeom)]),
    'rick_conv_wgrad_f32': (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp,
                                    ctypes.POINTER(ConvGeom), c_int, c_fp, c_fp]),
    'rick_thin_fwd_f32': (c_int, [c_fp, c_fp, c_i64, c_fp, c_fp, c_int, c_i64, c_int, c_int, c_fp]),
    'rick_thin_bwdx_f32': (c_int, [c_fp, c_fp, c_i64, c_fp, c_int, c_i64, c_int, c_int, c_fp]),
    'rick_torgb_fwd_f32': (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_int, c_fp]),
    'rick_torgb_bwdx_f32': (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_int, c_i64, c_int, c_int, c_fp]),
    'rick_torgb_bwdx_acc_f32': (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_int, c_i64, c_int, c_int, c_fp]),
    'rick_thin_wgrad_blocks': (c_int, [c_i64]),
    'rick_thin_wgrad_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_int, c_fp, c_fp]),
    'rick_chan_scale_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_fp]),
    'rick_hw_dot_blocks': (c_int, [c_i64]),
    'rick_hw_dot_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_fp, c_fp, c_fp]),
    'rick_hw_dot_scale_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_fp, c_fp]),
    'rick_hw_dot_act_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_i64, c_int, c_fp, c_fp, c_fp, c_int, c_f, c_f, c_fp, c_fp,
                                    c_fp]),
    'rick_add_scale_f32': (c_int, [c_fp, c_fp, c_fp, c_i64, c_f, c_fp]),
    'rick_mbstd_fwd_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    'rick_mbstd_bwd_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    'rick_sq_accumulate_f32': (c_int, [c_fp, c_fp, c_i64, c_fp]),
    'rick_filter_reduce_f32': (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_fp]),
    'rick_masked_adam_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_fp]),
    'rick_wsq_f32': (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_f, c_fp]),
    'rick_demod_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_f, c_fp]),
    'rick_demod_bwd_s_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    'rick_demod_bwd_w_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_f, c_int, c_fp]),
    'rick_colsum_multi_f32': (c_int, [ctypes.POINTER(ColsumItem), c_int, c_fp]),
    'rick_demod_blocks_wsq': (c_int, [c_int, c_int]),
    'rick_demod_blocks': (c_int, [c_int]),
    'rick_demod_blocks_bwd_s': (c_int, [c_int]),
    'rick_wsq_multi_f32': (c_int, [c_fp, c_int, c_int, c_fp]),
    'rick_demod_multi_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_f, c_fp]),
    'rick_demod_bwd_w_multi_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    'rick_demod_bwd_s_multi_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    'rick_modbank_blocks': (c_int, [c_int]),
    'rick_modbank_fwd_f32': (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_int, c_int, c_f, c_fp, c_fp]),
    'rick_modbank_bwd_f32': (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_int, c_int, c_f, c_fp, c_int, c_fp]),
    'rick_equal_linear_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_f, c_f, c_int, c_f, c_f, c_int, c_fp]),
    'rick_linear_fwd_workspace_floats': (c_i64, [c_int, c_int, c_int]),
    'rick_linear_fwd_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_f, c_f, c_fp, c_fp]),
    'rick_linear_dgrad_workspace_floats': (c_i64, [c_int, c_int, c_int]),
    'rick_linear_dgrad_f32': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_f, c_fp, c_fp]),
    'rick_linear_wgrad_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_f, c_f, c_int, c_fp]),
    'rick_adam_prepare_f32': (c_int, [c_fp, c_int, c_int, c_f, c_f, c_fp, c_fp]),
    'rick_masked_adam_dev_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_f, c_f, c_f, c_f, c_fp, c_fp]),
    'rick_ema_f32': (c_int, [c_fp, c_fp, c_i64, c_f, c_fp]),
    'rick_image_batch_f32': (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    'rick_png_unfilter': (c_int, [c_fp, c_int, c_int, c_int]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} not found: build the HIP extension first '
        f'(python -c "import __graft_entry__ as g; g.build()" or make -C rick_amd/csrc). '
        'rick_amd has no CPU or eager fallback.')

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == missing export
    _fn.restype = _res
    _fn.argtypes = _args
# kernel-form experiments (same-box A/B of bench.py, tools/ab_env2.sh): RICK_TUNE="key=value,..." -> rick_conv_tuning; never silent
if os.environ.get('RICK_TUNE'):
    import warnings
    warnings.warn(f"rick_amd: RICK_TUNE={os.environ['RICK_TUNE']} changes kernel-form selection", RuntimeWarning)
    for _kv in os.environ['RICK_TUNE'].split(','):
        _k, _v = _kv.split('=')
        assert lib.rick_conv_tuning(int(_k), int(_v)) >= 0, f'RICK_TUNE: unknown key {_k}'


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (raw accessor: ~0.3 us vs ~3.5 us for
    torch.cuda.current_stream().cuda_stream; every kernel launch pays it)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed with status {rc} '
                           f'({"invalid argument" if rc == 22 else "hipError " + str(rc - 1000)})')


def require_cuda_f32(*tensors):
    """Mirrors CHECK_CUDA of the reference binding (op/upfirdn2d.cpp:8,15-16): the HIP path is
    the only path; CPU tensors are an error, never a silent fallback."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('rick_amd ops require CUDA/HIP tensors (no CPU fallback); got device '
                               + str(t.device))
        if t.dtype != torch.float32:
            raise RuntimeError(f'rick_amd ops compute in float32; got {t.dtype}')


DTYPE_CODE = {torch.float64: 1, torch.float16: 2}      # rick_*_any entries (float32 has its own, fast entries)


def require_cuda_float(*tensors):
    """The two drop-in ops (upfirdn2d, fused_leaky_relu) accept the reference extension's dtypes — half, float, double
    (op/upfirdn2d_kernel.cu:311, op/fused_bias_act_kernel.cu:79) — all of one dtype per call; returns that dtype."""
    dt = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('rick_amd ops require CUDA/HIP tensors (no CPU fallback); got device ' + str(t.device))
        if t.dtype not in (torch.float32, torch.float64, torch.float16):
            raise RuntimeError(f'rick_amd ops support float16 / float32 / float64; got {t.dtype}')
        if dt is not None and t.dtype != dt:
            raise RuntimeError(f'rick_amd ops need one dtype per call; got {dt} and {t.dtype}')
        dt = t.dtype
    return dt


def ptr(t):
    return None if t is None else t.data_ptr()
