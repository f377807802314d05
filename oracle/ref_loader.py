"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the reference's OWN Python for the scheduler / collectors / advantage math
straight from /root/reference (present only in the build container, never on the
GPU box) so that `make_golden.py` can generate fixtures from it.

The reference package `flow_factory` cannot be imported whole (its `__init__`
chains pull in `diffusers`, `peft`, `deepspeed`, ...), so
the few files on the hot path are loaded one by one under synthetic package
objects, with `oracle.diffusers_stub` standing in for `diffusers`.
Nothing is copied: the files are executed where they lie.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("FLOW_FACTORY_REF", "/root/reference")
REF_PKG = os.path.join(REF_ROOT, "src", "flow_factory")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_PKG, "scheduler", "flow_match_euler_discrete.py"))


def _pkg(name: str, path: str) -> types.ModuleType:
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _load(modname: str, relpath: str) -> types.ModuleType:
    if modname in sys.modules:
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_PKG, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load():
    """Return a namespace with the reference's own hot-path Python objects."""
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    from . import diffusers_stub

    diffusers_stub.install()
    _pkg("flow_factory", REF_PKG)
    _pkg("flow_factory.scheduler", os.path.join(REF_PKG, "scheduler"))
    _pkg("flow_factory.utils", os.path.join(REF_PKG, "utils"))
    base = _load("flow_factory.utils.base", "utils/base.py")
    _load("flow_factory.utils.logger_utils", "utils/logger_utils.py")
    sabc = _load("flow_factory.scheduler.abc", "scheduler/abc.py")
    fm = _load("flow_factory.scheduler.flow_match_euler_discrete", "scheduler/flow_match_euler_discrete.py")
    tc = _load("flow_factory.utils.trajectory_collector", "utils/trajectory_collector.py")
    ns = types.SimpleNamespace(
        FlowMatchEulerDiscreteSDEScheduler=fm.FlowMatchEulerDiscreteSDEScheduler,
        set_scheduler_timesteps=fm.set_scheduler_timesteps,
        calculate_shift=fm.calculate_shift,
        SDESchedulerOutput=sabc.SDESchedulerOutput,
        to_broadcast_tensor=base.to_broadcast_tensor,
        TrajectoryCollector=tc.TrajectoryCollector,
        CallbackCollector=tc.CallbackCollector,
        compute_trajectory_indices=tc.compute_trajectory_indices,
    )
    return ns


# SD3.5-medium scheduler_config.json as the reference would read it through
# `load_scheduler` (src/flow_factory/scheduler/loader.py:51-57): pipeline scheduler
# config merged with SchedulerArguments.  Values from memory of the public checkpoint.
SD35_SCHEDULER_CONFIG = dict(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=False)


def make_reference_scheduler(ns=None, **sde_kwargs):
    ns = ns or load()
    kw = dict(SD35_SCHEDULER_CONFIG)
    kw.update(sde_kwargs)
    return ns.FlowMatchEulerDiscreteSDEScheduler(**kw)
