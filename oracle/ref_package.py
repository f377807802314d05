"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the reference's WHOLE `flow_factory` package from /root/reference/src in this container (where `diffusers`, `peft`,
`deepspeed`, ... are not installed) so that tests can check the plugin binding (`mi355_flow.flow_factory_plugin`) against the
reference's real classes: `BaseAdapter`, `SD3_5Adapter`, `Flux1Adapter`, `Wan2_T2V_Adapter`, `BaseSample`, the SDE schedulers
and `GRPOTrainer`.

How: a meta-path finder serves every module below the ABSENT third-party roots as an auto-stub whose attributes are
placeholder classes, except for the handful of diffusers symbols whose behaviour the hot path depends on, which come from
`oracle.diffusers_stub` (BaseOutput, randn_tensor, retrieve_timesteps, FlowMatchEulerDiscreteScheduler).  Nothing is copied:
the reference's files are executed where they lie.  /root/reference does not exist on the GPU box: callers must
`pytest.skip` when `available()` is false.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("FLOW_FACTORY_REF", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")

ABSENT_ROOTS = ("diffusers", "peft", "deepspeed", "wandb", "swanlab", "imageio", "cv2", "flash_attn", "open_clip", "hpsv2", "ImageReward",
                "tensorboard", "bitsandbytes", "xformers", "torchvision", "kornia", "librosa", "soundfile", "av", "decord", "moviepy",
                "openai", "vllm", "lpips", "timm", "clip", "t2v_metrics", "image_reward", "torchaudio")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_SRC, "flow_factory", "models", "abc.py"))


class _Placeholder:
    """Stands in for any class of an absent dependency: constructible, subclassable, attribute access yields placeholders."""

    def __init__(self, *a, **k):
        pass

    def __init_subclass__(cls, **k):
        super().__init_subclass__()

    def __call__(self, *a, **k):
        return _Placeholder()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Placeholder()

    @classmethod
    def from_pretrained(cls, *a, **k):
        raise RuntimeError("placeholder of an absent dependency: no checkpoints exist in this environment")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        obj = type(name, (_Placeholder,), {"__module__": self.__name__})
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in ABSENT_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


def _populate(module) -> None:
    from . import diffusers_stub as D

    name = module.__name__
    known = {
        "diffusers.utils.outputs": {"BaseOutput": D.BaseOutput},
        "diffusers.utils.torch_utils": {"randn_tensor": D.randn_tensor},
        "diffusers.utils": {"BaseOutput": D.BaseOutput},
        "diffusers.pipelines.flux.pipeline_flux": {"retrieve_timesteps": D.retrieve_timesteps},
        "diffusers.pipelines.stable_diffusion_3.pipeline_stable_diffusion_3": {"retrieve_timesteps": D.retrieve_timesteps},
        "diffusers.schedulers.scheduling_flow_match_euler_discrete": {"FlowMatchEulerDiscreteScheduler": D.FlowMatchEulerDiscreteScheduler},
        "diffusers.schedulers": {"FlowMatchEulerDiscreteScheduler": D.FlowMatchEulerDiscreteScheduler,
                                 "UniPCMultistepScheduler": D.UniPCMultistepScheduler},
        "diffusers": {"FlowMatchEulerDiscreteScheduler": D.FlowMatchEulerDiscreteScheduler, "UniPCMultistepScheduler": D.UniPCMultistepScheduler},
        "diffusers.schedulers.scheduling_unipc_multistep": {"UniPCMultistepScheduler": D.UniPCMultistepScheduler},
        "diffusers.utils.import_utils": {"is_torch_available": lambda: True,
                                         "is_torch_version": _is_torch_version},
    }
    for k, v in known.get(name, {}).items():
        setattr(module, k, v)
    if name == "peft":
        # `isinstance(x, PeftModel)` must be a real class check (models/abc.py:557-597)
        module.PeftModel = type("PeftModel", (), {})


def _is_torch_version(op: str, version: str) -> bool:
    import operator

    import torch
    from packaging.version import parse

    ops = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}
    return ops[op](parse(parse(torch.__version__).base_version), parse(version))


_installed = False


def install() -> None:
    """Idempotent: put the stub finder in front of sys.meta_path and the reference's src/ on sys.path."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    for k in [k for k in sys.modules if k.split(".")[0] in ("flow_factory",) + ABSENT_ROOTS]:
        del sys.modules[k]            # e.g. the per-file loader of oracle/ref_loader.py ran earlier in this process
    sys.meta_path.insert(0, _Finder())
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    _installed = True


def load(*modules: str):
    """install() + import the named reference modules (default: the SD3.5 adapter); returns them."""
    install()
    names = modules or ("flow_factory.models.stable_diffusion.sd3_5",)
    out = [importlib.import_module(n) for n in names]
    return out[0] if len(out) == 1 else out
