"""Helpers for the -m gpu parity tests: device buffers come from torch
(plumbing only); every decode goes through the C-ABI (rawspeed_amd/librsx.so)."""
import numpy as np
import torch

from rawspeed_amd import abi, capi

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def image_job_view(dim_x, dim_y, cpp, pitch, is_cfa=True):
    v = abi.Image()
    v.data = None
    v.pitch_bytes = pitch
    v.dim_x, v.dim_y, v.cpp = dim_x, dim_y, cpp
    v.is_cfa = 1 if is_cfa else 0
    return v
