"""Accelerator plug-in surface of the reference (neural_compressor/torch/utils/auto_accelerator.py:40-168, 427-456):
`register_accelerator(name, priority)`, the `Auto_Accelerator` interface and `auto_detect_accelerator()` with the reference's
selection order -- `INC_TARGET_DEVICE` (case-insensitive) > explicit `device_name` > highest-priority available class.

On ROCm the MI355X is PyTorch's device type "cuda" (HIP behind it), so the one accelerator registered here carries the
reference's name for that device string, "cuda": configs, `device=` arguments and saved checkpoints written for the
reference keep working.  There is deliberately NO "cpu" class: every algorithm of this package runs its arithmetic in
libinc_mi355x.so, and `INC_TARGET_DEVICE=cpu` (the reference's way to force its CPU path) is answered with an error that says
so instead of a silent fallback.  Third parties can still register further HIP devices the reference way.
"""

import os
from abc import ABC, abstractmethod
from functools import lru_cache
from typing import Any, Callable, List

import torch

from ...common.utils import logger

PRIORITY_HIP = 100


class AcceleratorRegistry:
    """name -> class, with a priority (reference :40-98)."""

    registered_accelerators = {}

    @classmethod
    def register_accelerator_impl(cls, name: str, priority: float = 0):
        def decorator(accelerator_cls):
            if name in cls.registered_accelerators:
                logger.warning("The accelerator %s is already registered, it is replaced.", name)
            cls.registered_accelerators[name] = (accelerator_cls, priority)
            return accelerator_cls

        return decorator

    @classmethod
    def get_sorted_accelerators(cls) -> List[Any]:
        return [c for c, _ in sorted(cls.registered_accelerators.values(), key=lambda item: item[1], reverse=True)]

    @classmethod
    def get_accelerator_cls_by_name(cls, name):
        pair = cls.registered_accelerators.get(name, None)
        return pair[0] if pair else None


accelerator_registry = AcceleratorRegistry()


def register_accelerator(name: str, priority: float = 0) -> Callable[..., Any]:
    """@register_accelerator(name="...", priority=N) class X(Auto_Accelerator): ...   (reference :100-113)."""
    return accelerator_registry.register_accelerator_impl(name=name, priority=priority)


class Auto_Accelerator(ABC):
    """The reference's accelerator interface (:115-168)."""

    @classmethod
    @abstractmethod
    def is_available(cls) -> bool: ...

    @abstractmethod
    def name(self) -> str: ...

    @abstractmethod
    def device_name(self, device_indx) -> str: ...

    @abstractmethod
    def set_device(self, device_index): ...

    @abstractmethod
    def current_device(self): ...

    @abstractmethod
    def current_device_name(self): ...

    @abstractmethod
    def device(self, device_index=None): ...

    @abstractmethod
    def empty_cache(self): ...

    @abstractmethod
    def synchronize(self): ...

    @abstractmethod
    def get_inc_accelerator_type(self): ...


@register_accelerator(name="cuda", priority=PRIORITY_HIP)
class HIPAccelerator(Auto_Accelerator):
    """An MI355X seen through PyTorch-ROCm's `cuda` device type (the reference's CUDA_Accelerator slot, :221-264)."""

    @classmethod
    def is_available(cls) -> bool:
        return torch.cuda.is_available()

    def name(self) -> str:
        return "cuda"

    def device_name(self, device_indx=None) -> str:
        return "cuda" if device_indx is None else f"cuda:{device_indx}"

    def synchronize(self):
        return torch.cuda.synchronize()

    def set_device(self, device_index):
        return torch.cuda.set_device(device_index)

    def current_device(self):
        return torch.cuda.current_device()

    def current_device_name(self):
        return f"cuda:{torch.cuda.current_device()}"

    def device(self, device_index=None):
        return torch.cuda.device(device_index)

    def empty_cache(self):
        return torch.cuda.empty_cache()

    def get_inc_accelerator_type(self):
        return "gfx950"  # the reference returns an INCAcceleratorType member; this framework has exactly one target


@lru_cache()
def _select(inc_target_device, device_name):
    if inc_target_device:
        cls = accelerator_registry.get_accelerator_cls_by_name(inc_target_device)
        if cls is None:
            raise RuntimeError(
                f"INC_TARGET_DEVICE={inc_target_device!r}: neural_compressor_amd only drives HIP devices (registered: "
                f"{sorted(accelerator_registry.registered_accelerators)}); there is no CPU path to force.")
        logger.warning("Force use %s accelerator.", inc_target_device)
        return cls()
    if device_name not in ("auto", None):
        base = str(device_name).split(":")[0]
        cls = accelerator_registry.get_accelerator_cls_by_name(base)
        if cls is not None:
            return cls()
        if base == "cpu":
            raise RuntimeError("device 'cpu' requested: neural_compressor_amd has no CPU implementation (tensors must live in HBM).")
        logger.warning("The device name %s is not supported, use auto detect instead.", device_name)
    for cls in accelerator_registry.get_sorted_accelerators():
        if cls.is_available():
            return cls()
    raise RuntimeError("No HIP device is visible (torch.cuda.is_available() is False). neural_compressor_amd has no CPU fallback.")


def auto_detect_accelerator(device_name="auto") -> Auto_Accelerator:
    """Reference auto_detect_accelerator (:427-456): INC_TARGET_DEVICE > device_name > priority."""
    env = os.environ.get("INC_TARGET_DEVICE", None)
    acc = _select(env.lower() if env else None, device_name)
    if not acc.is_available():
        raise RuntimeError("No HIP device is visible (torch.cuda.is_available() is False). neural_compressor_amd has no CPU fallback.")
    return acc
