"""ctypes binding of libcrank_hip.so (the C ABI of include/crank_hip.h).

There is exactly one compute backend.  If the library is missing or a call fails the
error is raised here, loudly; nothing falls back to torch or to the CPU oracle.
"""
import ctypes
import os
from .config import cfg
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = cfg.lib_path or os.path.join(_HERE, "libcrank_hip.so")  # (CRANK_AMD_LIB: instrumented builds)
_lib = None

ERRORS = {1: "invalid argument", 2: "HIP runtime error", 3: "unsupported shape/configuration"}


class NetDesc(ctypes.Structure):
    _fields_ = [
        ("kind", c_int), ("in_ch", c_int), ("out_ch", c_int), ("kernel_size", c_int), ("layers", c_int),
        ("stacks", c_int), ("res_ch", c_int), ("gate_ch", c_int), ("skip_ch", c_int), ("aux_ch", c_int),
        ("conv_ch", c_int), ("causal", c_int), ("use_bias", c_int), ("slope", c_float), ("dropout", c_float),
    ]


COLLATE_MAX_STREAMS = 8


class CollateStream(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("ld", c_int), ("col0", c_int), ("ncols", c_int), ("dst", c_void_p)]


class CollateDesc(ctypes.Structure):
    _fields_ = [
        ("n_streams", c_int), ("streams", CollateStream * COLLATE_MAX_STREAMS), ("utt_start", c_void_p),
        ("utt_spk", c_void_p), ("n_utt", c_int), ("n_spk", c_int), ("lcf0_raw", c_void_p),
        ("spk_lcf0_mean", c_void_p), ("spk_lcf0_std", c_void_p), ("raw", c_void_p), ("raw_start", c_void_p),
        ("fftl", c_int), ("hop", c_int),
    ]


P, I, LL, ULL, F, D = c_void_p, c_int, c_longlong, c_ulonglong, c_float, c_double

SIGNATURES = {
    "crk_net_create": (P, [ctypes.POINTER(NetDesc)]),
    "crk_net_destroy": (None, [P]),
    "crk_net_param_count": (LL, [P]),
    "crk_net_conv_count": (I, [P]),
    "crk_net_conv_info": (I, [P, I, ctypes.POINTER(c_longlong)]),
    "crk_net_saved_bytes": (LL, [P, I, I]),
    "crk_net_reserve": (I, [P, I, I]),
    "crk_net_scratch_bytes": (LL, [P, I, I]),
    "crk_debug_alloc_count": (LL, []),
    "crk_debug_net_paths": (I, [P, I, I]),
    "crk_net_set_wgrad_stream": (I, [P, P]),
    "crk_seed_next": (I, [P, P, P]),
    "crk_nets_wnorm_bwd": (I, [I, P, P]),
    "crk_nets_prepare": (I, [I, P, P, ULL, P, P]),
    "crk_net_forward": (I, [P, P, ULL, P, I, P, I, P, I, P, I, I, I, ULL, P]),
    "crk_net_backward": (I, [P, P, ULL, P, P, I, P, I, P, I, P, I, F, P, I, P, I, I, I, ULL, P]),
    "crk_net_backward_scaled": (I, [P, P, ULL, P, P, I, P, I, P, I, P, I, F, P, I, P, I, I, I, ULL, P, P, P]),
    "crk_vq_forward": (I, [P, I, P, I, I, I, P, P, I, P, I, P]),
    "crk_vq_forward_fused": (I, [P, I, P, I, P, I, P, I, I, I, P, P, I, P, I, P, P, P, P, P]),
    "crk_vq_image_bytes": (LL, [I, I]),
    "crk_vq_image_build_multi": (I, [I, P, P, I, P, P]),
    "crk_vq_ema_scratch_bytes": (LL, [I, I, I]),
    "crk_vq_ema_stats": (I, [P, I, P, I, I, I, P, P, P, P]),
    "crk_vq_ema_apply": (I, [P, P, P, P, P, I, I, D, D, P]),
    "crk_vq_ema_partial": (I, [P, I, P, I, I, I, P, P]),
    "crk_vq_ema_reduce_multi": (I, [I, P, P, P, P, P, P, P]),
    "crk_vq_ema_apply_multi": (I, [I, P, P, P, P, P, P, P, D, D, P]),
    "crk_vq_ema_partial_multi": (I, [I, P, P, P, P, P, P, P, P]),
    "crk_vq_ema_reduce_size_multi": (I, [I, P, P, P, P, P, P, P, D, D, P]),
    "crk_vq_ema_blend_multi": (I, [I, P, P, P, P, P, P, D, P]),
    "crk_vq_ema_blend_image_multi": (I, [I, P, P, P, P, P, P, D, P, P]),
    "crk_stft_loss_multi_fwd": (I, [P, I, P, I, I, I, I, I, P, P, P, P, F, P, P, P]),
    "crk_stft_loss_multi_fwd_grad": (I, [P, I, P, I, I, I, I, I, P, P, P, P, F, P, P, I, P, P]),
    "crk_stft_loss_multi_bwd": (I, [P, I, P, I, I, I, I, I, P, P, P, P, F, P, P, I, P]),
    "crk_loss_scratch_floats": (I, []),
    "crk_masked_loss_fwd": (I, [P, I, P, I, F, P, LL, I, I, P, P, P]),
    "crk_masked_loss_bwd": (I, [P, I, P, I, F, P, LL, I, I, P, P, P, I, P, I, P]),
    "crk_masked_loss_both_fwd": (I, [P, I, P, I, P, LL, I, P, P, P]),
    "crk_weighted_sum": (I, [I, P, P, F, P, P]),
    "crk_weighted_sum_bwd": (I, [I, P, P, P, P]),
    "crk_recon_supported": (I, [I, I, P, P, P]),
    "crk_stft_twiddle_floats": (LL, [I, I]),
    "crk_stft_twiddles": (I, [I, I, P, P, P]),
    "crk_recon_grad_floats": (LL, [I, I, I, I, P, P]),
    "crk_recon_loss_fwd": (I, [P, I, P, I, P, I, I, I, I, P, P, P, P, F, P, P, P, P]),
    "crk_recon_loss_bwd": (I, [P, I, P, I, P, I, I, I, I, P, P, P, P, P, P, P, P, P, I, P]),
    "crk_masked_loss_bwd_acc": (I, [P, I, P, I, F, P, LL, I, I, P, P, P, I, P, I, P, I, P, P]),
    "crk_vq_commit_bwd": (I, [P, I, P, I, P, LL, I, P, P, P, I, P, I, P, I, P, I, P, I, P]),
    "crk_ce_fwd": (I, [P, I, P, LL, I, I, P, P, P, P]),
    "crk_ce_bwd": (I, [P, LL, I, P, P, P, P]),
    "crk_stft_loss_fwd": (I, [P, I, P, I, I, I, I, I, I, I, P, F, F, I, P, P, P]),
    "crk_stft_loss_bwd": (I, [P, I, P, I, I, I, I, I, I, I, P, F, F, P, P, I, P]),
    "crk_adam_step": (I, [P, P, P, P, LL, P, P, F, F, F, I, P]),
    "crk_radam_step": (I, [P, P, P, P, LL, P, P, D, D, D, I, P]),
    "crk_lamb_step": (I, [P, P, P, P, P, P, I, P, I, P, P, P, P, D, D, D, I, P]),
    "crk_lamb_tile": (I, []),
    "crk_concat_embed": (I, [P, I, I, P, I, I, P, I, P, LL, P, I, P]),
    "crk_embed_bwd_scratch_floats": (LL, [LL, I, I]),
    "crk_embed_bwd": (I, [P, I, I, I, P, LL, I, P, P, P]),
    "crk_concat_embed_run": (I, [P, I, I, P, I, I, P, I, P, LL, LL, P, I, P]),
    "crk_embed_bwd_run": (I, [P, I, I, I, P, LL, LL, I, P, P, P]),
    "crk_logmel_fwd": (I, [P, I, I, I, I, I, I, I, P, P, I, F, P, P, P, I, I, P]),
    "crk_scaler_apply": (I, [P, I, P, I, LL, I, P, P, I, P]),
    "crk_collate_batch": (I, [ctypes.POINTER(CollateDesc), P, I, I, P, P, P, P, P, P, P, P, P]),
    "crk_decode_f0": (I, [P, P, I, I, P, P, D, D, I, P, P, P, P, P, P]),
    "crk_mcd_scratch_bytes": (LL, [I, I, I, I, I]),
    "crk_mcd_fastdtw": (I, [P, P, P, P, I, I, I, I, I, P, P, P, LL, P, P, P]),
    "crk_prof_enable": (I, [I]),
    "crk_prof_report": (I, [I, ctypes.POINTER(c_longlong), ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
    "crk_prof_report_bytes": (I, [I, ctypes.POINTER(c_double)]),
    "crk_debug_vq_set_f16": (I, [I]),
    "crk_debug_flush_before": (I, [c_longlong]),
    "crk_debug_vq_flags": (I, [ctypes.POINTER(c_ulonglong), I]),
    "crk_version": (c_char_p, []),
}


def lib():
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C crank_amd/csrc`).  crank_amd has no fallback compute path."
            )
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libcrank_hip: {what} failed: {ERRORS.get(rc, rc)}")


_raw_stream = None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  Through the raw C accessors:
    ``torch.cuda.current_stream().cuda_stream`` builds a Stream object per call (~11 us; 39 calls per training step were
    14 % of the host time of a step)."""
    global _raw_stream
    if _raw_stream is None:
        import torch

        get_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        get_dev = getattr(torch._C, "_cuda_getDevice", None)
        if get_stream is not None and get_dev is not None:
            _raw_stream = lambda: get_stream(get_dev())  # noqa: E731
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    return _raw_stream()


def ptr(t):
    return 0 if t is None else t.data_ptr()
