"""
resize.py -- coefficient tables of Pillow's antialiased bicubic resize, for the device-side `resize-naive` path.

The reference resizes non-native frames with torchvision `TVF.resize(img, (224,224), interpolation=bicubic)` on PIL
images (processing_prismatic.py:136; 256x256 robot frames: experiments/robot/bridge/run_bridgev2_eval.py:161,170), which
IS `PIL.Image.resize` -- a separable two-pass convolution (horizontal, then vertical, uint8 intermediate) with 22-bit
fixed-point coefficients.  This module restates Pillow's `precompute_coeffs` / `normalize_coeffs_8bpc`
(src/libImaging/Resample.c) so the HIP kernels reproduce Pillow bit for bit; `resize_u8_reference` is the numpy
statement of the same two passes (pinned against the installed Pillow in tests/test_host_logic.py).
"""

from __future__ import annotations

import math
from functools import lru_cache
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC_SUPPORT = 2.0


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=32)
def bicubic_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out,ksize], ksize) exactly as Pillow."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    ki = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    # C casts truncate toward zero; astype(int64) on floats does the same
    return bounds, ki.astype(np.int32), ksize


def _pass(src: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One separable pass over `axis` (0 = rows / vertical, 1 = columns / horizontal) of [H,W,C] uint8."""
    out_n = bounds.shape[0]
    shape = list(src.shape)
    shape[axis] = out_n
    out = np.empty(shape, dtype=np.uint8)
    s64 = src.astype(np.int64)
    for o in range(out_n):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        k = kk[o, :n].astype(np.int64)
        acc = np.full(s64.take(0, axis=axis).shape, 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc = acc + s64.take(lo + t, axis=axis) * k[t]
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            out[o] = v
        else:
            out[:, o] = v
    return out


def resize_u8_reference(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [out_h,out_w,3], Pillow's order: horizontal pass first, then vertical."""
    h, w = img.shape[:2]
    cur = img
    if w != out_w:
        b, k, _ = bicubic_coeffs(w, out_w)
        cur = _pass(cur, b, k, axis=1)
    if h != out_h:
        b, k, _ = bicubic_coeffs(h, out_h)
        cur = _pass(cur, b, k, axis=0)
    return cur
