"""ctypes binding of librovit_hip.so (C ABI declared in include/rovit_hip.h).

No libtorch linkage: tensors cross the boundary as raw device pointers plus the caller's current HIP stream.
There is NO fallback: if the shared library is missing, or a tensor is not on a CUDA/HIP device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get('ROVIT_HIP_LIB') or os.path.join(_PKG_ROOT, 'lib', 'librovit_hip.so')   # env override: developer A/B builds

_lib: Optional[C.CDLL] = None
ABI_VERSION = 410          # rovit_version() this binding matches (csrc/api.hip)

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# mlp_path of rovit_vit_forward / rovit_vit_backward (include/rovit_hip.h); rovit_gemm_nt's per-call tile flags
MLP_AUTO, MLP_TWO_LAUNCH, MLP_ONE_LAUNCH = 0, 1, 2
GEMM_TILED_192, GEMM_TILED_96 = 0x100, 0x200
MLP_FUSED_MIN_ROWS = 34000          # ROVIT_MLP_AUTO's threshold (csrc/vit.hip): one-launch MLP half from this many token rows

# name -> (restype, argtypes); mirrors include/rovit_hip.h one to one
SIGNATURES = {
    'rovit_version': (_i, []),
    'rovit_last_error_string': (C.c_char_p, []),
    'rovit_kan_basis': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'rovit_kan_layer_fwd': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'rovit_kan_prepare': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'rovit_kan_stack_fwd': (_i, [_vp] * 6 + [_i, _vp, _vp, _vp, _i, _vp]),
    'rovit_kan_mfma_prepared_floats': (_sz, [_i, _i, _i]),
    'rovit_kan_prepare_mfma': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'rovit_kan_stack_fwd_mfma': (_i, [_vp] * 5 + [_i, _vp, _vp, _vp, _i, _vp]),
    'rovit_kan_stack_bwd': (_i, [_vp] * 11 + [_i, _vp, _vp, _vp, _i, _vp]),
    'rovit_kan_layer_bwd': (_i, [_vp] * 10 + [_i] * 6 + [_vp]),
    'rovit_linear_fwd': (_i, [_vp] * 5 + [_i] * 4 + [_vp]),
    'rovit_linear_bwd': (_i, [_vp] * 9 + [_i] * 4 + [_vp]),
    'rovit_heads_fwd': (_i, [_vp] * 8 + [_i] * 5 + [_vp]),
    'rovit_heads_bwd': (_i, [_vp] * 12 + [_i] * 5 + [_vp]),
    'rovit_head_phase_fwd': (_i, [_vp, _vp]),
    'rovit_head_phase_bwd': (_i, [_vp, _vp]),
    'rovit_head_phase_bwd_params': (_i, [_vp, _vp]),
    'rovit_vit_num_params': (_i, [_i]),
    'rovit_vit_prep_bytes': (_sz, [_i]),
    'rovit_vit_workspace_bytes': (_sz, [_i, _i, _i]),
    'rovit_vit_f32_workspace_bytes': (_sz, [_i]),
    'rovit_vit_forward_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'rovit_vit_workspace_field': (_i, [_i, _i, _i, _i, _vp, _vp]),
    'rovit_vit_prepare': (_i, [_vp, _vp, _i, _vp]),
    'rovit_vit_forward': (_i, [_vp] * 5 + [_i] * 4 + [_vp]),
    'rovit_vit_forward_prepare': (_i, [_vp] * 5 + [_i] * 5 + [_vp]),
    'rovit_vit_forward_taps': (_i, [_vp] * 7 + [_i, _i, _vp]),
    'rovit_attention_probs': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    'rovit_vit_backward': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'rovit_vit_backward_notify': (_i, [_vp] * 6 + [_i] * 5 + [_vp] + [_vp]),
    'rovit_gemm_nt': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    'rovit_gemm_resid_ln': (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp]),
    'rovit_mlp_stream_bytes': (_sz, []),
    'rovit_mlp_prepare_stream': (_i, [_vp, _vp, _vp, _vp]),
    'rovit_mlp_fused_fwd': (_i, [_vp] * 9 + [_f, _i, _i, _vp]),
    'rovit_mlp_prepare_stream_tail': (_i, [_vp] * 6),
    'rovit_block_tail_fwd': (_i, [_vp] * 14 + [_f, _i, _i, _vp]),
    'rovit_mlp_fused_bwd': (_i, [_vp] * 8 + [_i, _vp]),
    'rovit_gemm_ln_bwd': (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rovit_wgrad_splits': (_i, [_i, _i, _i]),
    'rovit_wgrad_workspace_bytes': (_sz, [_i, _i, _i]),
    'rovit_wgrad': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'rovit_wgrad_multi': (_i, [_vp] * 7 + [_i, _i, _i, _vp]),
    'rovit_wgrad_multi_ex': (_i, [_vp] * 9 + [_i, _i, _i, _vp]),
    'rovit_wgrad_reduce': (_i, [_vp, _i, _i, _i] + [_vp] * 8 + [_vp]),
    'rovit_attention_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    'rovit_attention_bwd': (_i, [_vp] * 5 + [_i] * 4 + [_f, _vp]),
    'rovit_attention_cls_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    'rovit_attention_cls_bwd': (_i, [_vp] * 5 + [_i] * 4 + [_f, _vp]),
    'rovit_layernorm_fwd': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    'rovit_layernorm_bwd': (_i, [_vp] * 5 + [_i, _i, _vp]),
    'rovit_im2col': (_i, [_vp, _vp, _i, _vp]),
    'rovit_patch_embed_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'rovit_patch_embed_wgrad': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    'rovit_cls_rows': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'rovit_cls_norm_fwd': (_i, [_vp] * 6 + [_i, _i, _f, _vp]),
    'rovit_cls_norm_bwd': (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    'rovit_pos_grad': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'rovit_prep_weight': (_i, [_vp] * 7 + [_i, _i, _vp]),
    'rovit_joint_loss': (_i, [_vp] * 7 + [_i] + [_vp] * 7 + [_i, _i, _f, _f, _f, _f, _vp]),
    'rovit_scale_buffers': (_i, [_vp, _vp, _i, _vp, _vp]),
    'rovit_sq_norm_accum': (_i, [_vp, _sz, _vp, _vp, _vp]),
    'rovit_mix_images': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _vp]),
    'rovit_clip_coef': (_i, [_vp, _f, _vp, _vp, _vp]),
    'rovit_adamw_flat': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _f, _i, _vp]),
    'rovit_sq_norm_clip': (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'rovit_adamw_flat_multi': (_i, [_vp] * 7 + [_i, _vp, _f, _f, _f, _f, _vp]),
}


class HeadPhase(C.Structure):
    """``rovit_head_phase`` of include/rovit_hip.h, field for field (HOST arrays of device pointers inside)."""
    _fields_ = [('batch', _i), ('embed', _i), ('hid', _i), ('num_classes', _i), ('stage', _i), ('kan_layers', _i),
                ('kan_dims', _i * 5), ('kan_knots', _i * 4), ('kan_acts', _i * 4), ('drop_p', _f),
                ('seed', C.c_ulonglong), ('offset', C.c_ulonglong),
                ('features', _vp), ('head_params', _vp * 14), ('masks', _vp * 3), ('kan_w', _vp * 4), ('kan_knots_p', _vp * 4),
                ('kan_lw', _vp * 4), ('kan_lb', _vp * 4),
                ('hidden', _vp), ('cls', _vp), ('ord', _vp), ('mu', _vp), ('lv', _vp), ('kan_out', _vp * 4),
                ('g_cls', _vp), ('g_ord', _vp), ('g_mu', _vp), ('g_lv', _vp), ('g_kan', _vp),
                ('d_features', _vp), ('dpre', _vp), ('kan_gz', _vp * 4), ('head_grads', _vp * 14), ('kan_dw', _vp * 4),
                ('kan_dlw', _vp * 4), ('kan_dlb', _vp * 4), ('want_param_grads', _i)]


# entry points only the developer library exports (round-2 / round-3 experiments that lost; tools/ A/B them)
DEV_SIGNATURES = {
    'rovit_gemm_mlp_bwd': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    'rovit_mlp_prepare_stream_tail_bwd': (_i, [_vp] * 5),
    'rovit_block_tail_bwd': (_i, [_vp] * 9 + [_i, _vp]),
}


class RovitHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RovitHipError(
                f'{LIB_PATH} not found: build it with `python __graft_entry__.py` (or `make -C csrc`). '
                'The RoViT-KAN HIP path has no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if hasattr(lib, 'rovit_dev_set_knob'):           # developer library (make -C csrc dev; tools/ only)
            lib.rovit_dev_set_knob.restype, lib.rovit_dev_set_knob.argtypes = _i, [_i, _i, _i]
            for name, (res, args) in DEV_SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            # A/B of whole steps with the developer library: ROVIT_DEV_KNOBS="id=value,id=value" (common.h RovitKnob ids).
            # The product library exports no such entry point, so the variable does nothing there.
            for kv in filter(None, os.environ.get('ROVIT_DEV_KNOBS', '').split(',')):
                k, v = kv.split('=')
                lib.rovit_dev_set_knob(int(k), int(v), 0)
        if lib.rovit_version() != ABI_VERSION:
            raise RovitHipError(f'{LIB_PATH} has ABI version {lib.rovit_version()}, this binding was written for {ABI_VERSION}: '
                                'rebuild the library (`make -C csrc`); argument lists changed between versions')
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().rovit_last_error_string()
        raise RovitHipError(f'{what} failed (code {rc}): {msg.decode() if msg else ""}')


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RovitHipError('RoViT-KAN HIP kernels need tensors on a CUDA/HIP device (got a CPU tensor); '
                            'there is no CPU fallback in the product path')
    if not t.is_contiguous():
        raise RovitHipError('non-contiguous tensor passed to a HIP kernel')
    if t.device.index != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's stream (stream_ptr): one process (or at least one current device)
        # per GPU, as the data-parallel launcher sets it up; a tensor of another device would be read by the wrong GPU
        raise RovitHipError(f'tensor on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: '
                            'call torch.cuda.set_device() / use `with torch.cuda.device(...)` around the model call')
    return t.data_ptr()


def ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    """HOST array of device pointers (kept alive by the caller for the duration of the call)."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
