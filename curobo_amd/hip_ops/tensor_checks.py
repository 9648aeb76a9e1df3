"""Strict tensor checks at the op boundary (reference ``cuda_ops/tensor_checks.py:20-83``):
device match, dtype and contiguity are verified BEFORE any launch and raise ``ValueError``
(the reference's ``log_and_raise``).  No ``.contiguous()`` is ever called implicitly: that
would allocate inside a captured graph."""

from __future__ import annotations

import torch


def _check(device, dtype, **tensors) -> None:
    for name, t in tensors.items():
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"{name} is not a tensor")
        if t.device != device:
            raise ValueError(f"{name} is on {t.device}, expected {device}")
        if t.dtype != dtype:
            raise ValueError(f"{name} has dtype {t.dtype}, expected {dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{name} is not contiguous")


def check_float32_tensors(device, **t):
    _check(device, torch.float32, **t)


def check_int16_tensors(device, **t):
    _check(device, torch.int16, **t)


def check_int8_tensors(device, **t):
    _check(device, torch.int8, **t)


def check_int32_tensors(device, **t):
    _check(device, torch.int32, **t)


def check_uint8_tensors(device, **t):
    _check(device, torch.uint8, **t)


def check_bool_tensors(device, **t):
    _check(device, torch.bool, **t)
