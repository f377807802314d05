"""ctypes binding of libmi355flow.so (C ABI: include/mi355_flow.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C flow-factory_amd/csrc`.
There is NO fallback: if the shared object is missing or a call fails, this module raises
(reference error convention: fail fast, never degrade silently -- constraints.md:144-145).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmi355flow.so")

F32, BF16, F16 = 0, 1, 2
ODE, FLOW_SDE, DANCE_SDE, CPS = 0, 1, 2, 3
DYNAMICS = {"ODE": ODE, "Flow-SDE": FLOW_SDE, "Dance-SDE": DANCE_SDE, "CPS": CPS}


class ModelCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("patch_size", C.c_int32),
        ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("head_dim", C.c_int32),
        ("joint_attention_dim", C.c_int32), ("pooled_projection_dim", C.c_int32),
        ("pos_embed_max_size", C.c_int32), ("time_proj_dim", C.c_int32), ("ff_mult", C.c_int32),
        ("dual_layer_mask", C.c_uint64), ("eps", C.c_float),
    ]


class VaeCfg(C.Structure):
    _fields_ = [
        ("latent_channels", C.c_int32), ("out_channels", C.c_int32), ("num_blocks", C.c_int32),
        ("layers_per_block", C.c_int32), ("norm_num_groups", C.c_int32), ("block_out_channels", C.c_int32 * 8),
        ("eps", C.c_float), ("scaling_factor", C.c_float), ("shift_factor", C.c_float),
    ]


class FluxCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("num_layers", C.c_int32), ("num_single_layers", C.c_int32), ("num_heads", C.c_int32),
        ("head_dim", C.c_int32), ("joint_attention_dim", C.c_int32), ("pooled_projection_dim", C.c_int32),
        ("guidance_embeds", C.c_int32), ("time_proj_dim", C.c_int32), ("axes_dims_rope", C.c_int32 * 3), ("eps", C.c_float),
    ]


class WanCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
        ("head_dim", C.c_int32), ("ffn_dim", C.c_int32), ("text_dim", C.c_int32), ("freq_dim", C.c_int32),
        ("patch_t", C.c_int32), ("patch_h", C.c_int32), ("patch_w", C.c_int32), ("eps", C.c_float),
    ]


class WvaeCfg(C.Structure):
    _fields_ = [
        ("z_dim", C.c_int32), ("base_dim", C.c_int32), ("num_res_blocks", C.c_int32), ("out_channels", C.c_int32),
        ("dim_mult", C.c_int32 * 4), ("temporal_upsample", C.c_int32 * 3), ("latents_mean", C.c_float * 16), ("latents_std", C.c_float * 16),
    ]


class QwenCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("head_dim", C.c_int32),
        ("joint_attention_dim", C.c_int32), ("time_proj_dim", C.c_int32), ("axes_dims_rope", C.c_int32 * 3),
        ("scale_rope", C.c_int32), ("eps", C.c_float),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_int64

# name -> (restype, argtypes); must list every symbol declared in include/mi355_flow.h
SIGNATURES = {
    "mi355_version": (_I, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_engine_create": (_I, [C.POINTER(ModelCfg), C.POINTER(_P)]),
    "mi355_engine_destroy": (_I, [_P]),
    "mi355_engine_bind_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _P]),
    "mi355_engine_weights_ready": (_I, [_P]),
    "mi355_engine_attention_info": (_I, [_P, _P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_F)]),
    "mi355_engine_num_params": (_I, [_P]),
    "mi355_engine_param_name": (C.c_char_p, [_P, _I]),
    "mi355_plan_create": (_I, [_P, _I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "mi355_plan_destroy": (_I, [_P]),
    "mi355_plan_workspace_bytes": (_L, [_P]),
    "mi355_transformer_forward": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P]),
    "mi355_sde_step": (_I, [_P, _I, _L, _P, _P, _I, _F, _P, _I, _P, _P, _I, _P, _P, _P, _I, _F, _I, _I,
                            _P, _P, _P, _P, _P, _P, _P]),
    "mi355_sde_step_bwd": (_I, [_P, _I, _L, _P, _P, _F, _P, _I, _P, _I, _P, _P, _P, _I, _F, _I, _I, _P, _P, _P, _P]),
    "mi355_denoise_step": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _F, _P, _P, _I, _P, _P, _P, _I, _F, _I, _I,
                                _P, _P, _P, _P, _P, _P, _P]),
    "mi355_rollout": (_I, [_P, _P, _I, C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _I, _F, _P, _I, _I, _P,
                           _P, _P, _P, _P, C.POINTER(C.c_int32), _P, _P, _P, _I]),
    "mi355_engine_set_grad": (_I, [_P, C.c_char_p, _P]),
    "mi355_engine_set_grad_typed": (_I, [_P, C.c_char_p, _P, _I]),
    "mi355_engine_clear_grads": (_I, [_P]),
    "mi355_engine_set_train_scope": (_I, [_P, _I]),
    "mi355_engine_grad_supported": (_I, [_P, C.c_char_p]),
    "mi355_plan_training_bytes": (_L, [_P]),
    "mi355_denoise_step_train": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _F, _P, _I, _P, _P, _P, _I, _F, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mi355_denoise_step_backward": (_I, [_P, _P, _P, _I, _F, _P, _I, _P, _P, _P, _I, _F, _I, _I, _P, _P, _P]),
    "mi355_op_attention_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "mi355_op_linear": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "mi355_unipc_convert": (_I, [_P, _P, _P, _I, _F, _P, _I, _F, _P, _L]),
    "mi355_op_lincomb": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_I), C.POINTER(_F), _P, _I, _L]),
    "mi355_sched_trace": (_I, [_I]),
    "mi355_sched_trace_read": (C.c_longlong, [C.c_char_p, C.c_longlong]),
    "mi355_op_linear_trace": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mi355_op_linear_gate_res": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "mi355_op_wgrad": (_I, [_P, _P, C.c_int64, _P, C.c_int64, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mi355_clock_probe": (_I, [_P, _P, _I, _I]),
    "mi355_op_attention": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "mi355_op_ln_modulate": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F]),
    "mi355_vae_create": (_I, [C.POINTER(VaeCfg), C.POINTER(_P)]),
    "mi355_vae_destroy": (_I, [_P]),
    "mi355_vae_bind_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _P]),
    "mi355_vae_weights_ready": (_I, [_P]),
    "mi355_vae_num_params": (_I, [_P]),
    "mi355_vae_param_name": (C.c_char_p, [_P, _I]),
    "mi355_vae_plan_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "mi355_vae_plan_destroy": (_I, [_P]),
    "mi355_vae_plan_workspace_bytes": (_L, [_P]),
    "mi355_vae_decode": (_I, [_P, _P, _P, _I, _I, _P, _I, _I]),
    "mi355_flux_create": (_I, [C.POINTER(FluxCfg), C.POINTER(_P)]),
    "mi355_flux_destroy": (_I, [_P]),
    "mi355_flux_bind_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _P]),
    "mi355_flux_weights_ready": (_I, [_P]),
    "mi355_flux_num_params": (_I, [_P]),
    "mi355_flux_param_name": (C.c_char_p, [_P, _I]),
    "mi355_flux_plan_create": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "mi355_flux_plan_destroy": (_I, [_P]),
    "mi355_flux_plan_workspace_bytes": (_L, [_P]),
    "mi355_flux_forward": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "mi355_flux_rollout": (_I, [_P, _P, _I, C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _I, _F, _P, _I, _I, _P, _P, _P,
                                C.POINTER(C.c_int32), _P, _P, _P, _I]),
    "mi355_flux_set_grad": (_I, [_P, C.c_char_p, _P]),
    "mi355_flux_set_grad_typed": (_I, [_P, C.c_char_p, _P, _I]),
    "mi355_flux_clear_grads": (_I, [_P]),
    "mi355_flux_grad_supported": (_I, [_P, C.c_char_p]),
    "mi355_flux_plan_training_bytes": (_L, [_P]),
    "mi355_flux_forward_train": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "mi355_flux_backward": (_I, [_P, _P, _P]),
    "mi355_op_attention128_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "mi355_op_rope_norm_fwd_bwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F]),
    "mi355_op_attention128": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I]),
    "mi355_op_rope_norm": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F]),
    "mi355_op_norm_rope_full": (_I, [_P, _P, _L, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "mi355_wan_create": (_I, [C.POINTER(WanCfg), C.POINTER(_P)]),
    "mi355_wan_destroy": (_I, [_P]),
    "mi355_wan_bind_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _P]),
    "mi355_wan_weights_ready": (_I, [_P]),
    "mi355_wan_num_params": (_I, [_P]),
    "mi355_wan_param_name": (C.c_char_p, [_P, _I]),
    "mi355_wan_plan_create": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "mi355_wan_plan_destroy": (_I, [_P]),
    "mi355_wan_plan_workspace_bytes": (_L, [_P]),
    "mi355_wan_forward": (_I, [_P, _P, _P, _I, _P, _P, _P, _P]),
    "mi355_wan_rollout": (_I, [_P, _P, _I, C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _I, _F, _P, _I, _I, _P, _P, _P,
                               C.POINTER(C.c_int32), _P, _P, _P, _I]),
    "mi355_wan_set_grad": (_I, [_P, C.c_char_p, _P]),
    "mi355_wan_set_grad_typed": (_I, [_P, C.c_char_p, _P, _I]),
    "mi355_wan_clear_grads": (_I, [_P]),
    "mi355_wan_grad_supported": (_I, [_P, C.c_char_p]),
    "mi355_wan_plan_training_bytes": (_L, [_P]),
    "mi355_wan_forward_train": (_I, [_P, _P, _P, _I, _P, _P, _P, _P]),
    "mi355_wan_backward": (_I, [_P, _P, _P]),
    "mi355_op_norm_rope_full_fwd_bwd": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F]),
    "mi355_qwen_create": (_I, [C.POINTER(QwenCfg), C.POINTER(_P)]),
    "mi355_qwen_destroy": (_I, [_P]),
    "mi355_qwen_bind_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _P]),
    "mi355_qwen_weights_ready": (_I, [_P]),
    "mi355_qwen_num_params": (_I, [_P]),
    "mi355_qwen_param_name": (C.c_char_p, [_P, _I]),
    "mi355_qwen_plan_create": (_I, [_P, _I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "mi355_qwen_plan_destroy": (_I, [_P]),
    "mi355_qwen_plan_workspace_bytes": (_L, [_P]),
    "mi355_qwen_forward": (_I, [_P, _P, _P, _I, _P, _P, C.POINTER(C.c_int32), _F, _P, _P]),
    "mi355_qwen_rollout": (_I, [_P, _P, _I, C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _I, _F, _P, _I, _I, _P, _P,
                                C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, _P, _P, _I]),
    "mi355_qwen_set_grad": (_I, [_P, C.c_char_p, _P]),
    "mi355_qwen_set_grad_typed": (_I, [_P, C.c_char_p, _P, _I]),
    "mi355_qwen_clear_grads": (_I, [_P]),
    "mi355_qwen_grad_supported": (_I, [_P, C.c_char_p]),
    "mi355_qwen_plan_training_bytes": (_L, [_P]),
    "mi355_qwen_forward_train": (_I, [_P, _P, _P, _I, _P, _P, C.POINTER(C.c_int32), _F, _P, _P]),
    "mi355_qwen_backward": (_I, [_P, _P, _P]),
    "mi355_op_cfg_rescale_bwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _L, _I]),
    "mi355_op_cfg_rescale": (_I, [_P, _P, _P, _F, _P, _L, _I]),
    "mi355_op_rms_rows": (_I, [_P, _P, _P, _P, _I, _I, _F]),
    "mi355_wvae_create": (_I, [C.POINTER(WvaeCfg), C.POINTER(_P)]),
    "mi355_wvae_destroy": (_I, [_P]),
    "mi355_wvae_bind_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _P]),
    "mi355_wvae_weights_ready": (_I, [_P]),
    "mi355_wvae_num_params": (_I, [_P]),
    "mi355_wvae_param_name": (C.c_char_p, [_P, _I]),
    "mi355_wvae_plan_create": (_I, [_P, _I, _I, _I, _I, C.POINTER(_P)]),
    "mi355_wvae_plan_destroy": (_I, [_P]),
    "mi355_wvae_plan_workspace_bytes": (_L, [_P]),
    "mi355_wvae_decode": (_I, [_P, _P, _P, _I, _I, _P, _I, _I, _I]),
    "mi355_op_conv3d_causal": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "mi355_op_wan_rms": (_I, [_P, _P, _P, _P, _L, _I, _I, _I]),
    "mi355_op_conv3x3": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "mi355_op_conv_repack": (_I, [_P, _P, _I, _P, _I, _I, _I, _I]),
    "mi355_op_group_norm": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _F, _I]),
    "mi355_profile_enable": (_I, [_I]),
    "mi355_tune_set": (_I, [_I, _I]),
    "mi355_profile_collect": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (no GPU needed to load; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"mi355_flow: {LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C flow-factory_amd/csrc`).  There is no CPU / PyTorch fallback for the rollout hot path."
        )
    # One HIP runtime per process, and always the same one: torch bundles its own libamdhip64 / libhsa-runtime64, libmi355flow.so links the
    # system ROCm's.  Whichever loads first serves both (same sonames), so load torch FIRST -- every caller that reaches a compute call has
    # imported it anyway (tensors are how memory gets here).  (Round 5: one `__graft_entry__.py smoke` process -- build() then smoke(), the
    # library loaded before torch -- failed its first hipMalloc right behind a bench run; it did not reproduce with either order one call
    # later, profiles/r05f_smoke_order.txt.  The import keeps the order fixed regardless; engine_create now reports the HIP error string.)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # A/B of kernel / launch variants for a whole process (tests, bench, scripts): MI355_TUNE="key=value,..." -> mi355_tune_set
    for kv in filter(None, os.environ.get("MI355_TUNE", "").split(",")):
        k, v = kv.split("=")
        if lib.mi355_tune_set(int(k), int(v)) != 0:
            raise RuntimeError(f"mi355_flow: MI355_TUNE entry '{kv}' was rejected: {lib.mi355_last_error().decode()}")
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mi355_last_error()
        raise RuntimeError(f"mi355_flow {what} failed: {msg.decode() if msg else 'unknown error'}")
