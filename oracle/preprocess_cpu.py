"""CPU restatement (numpy, integer arithmetic) of the image / label preprocessing in front of the hot path (SURVEY.md
§8 f2).  TEST INFRASTRUCTURE ONLY — imported by tests/ and nothing else.

What the reference does (evaluate.py:58-61, data/collator.py:82-91): images go through
``CLIPFeatureExtractor(size=S, crop_size=S)`` = resize so the SHORT edge is S with PIL bicubic, centre crop S x S,
x * (1/255) in float64 -> float32, (x - mean) / std in float32, CHW; labels are expanded to 3 channels and go through the
same extractor with ``resample=NEAREST, image_mean=0, image_std=1`` and come back as ``(pixel_values[:, 0] * 255).long()``.

Third-party arithmetic restated here: Pillow (12.2.0 in this image) ``src/libImaging/Resample.c`` — ``precompute_coeffs``,
``normalize_coeffs_8bpc`` (22-bit fixed point), ``ImagingResampleHorizontal_8bpc`` / ``Vertical`` (rounding constant
1 << 21, clip8) and the two-pass order (horizontal, then vertical, through an 8-bit intermediate);
``src/libImaging/Geometry.c`` ``ImagingScaleAffine`` (nearest: truncated source coordinate advanced by repeated addition).
Pinned by tests/test_preprocess_cpu.py against Pillow and transformers' CLIPImageProcessor themselves.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2) over the whole axis.
    Returns (bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
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
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(a: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass along `axis` of an HWC uint8 array."""
    in_size = a.shape[axis]
    if in_size == out_size:
        return a
    bounds, kk = resample_coeffs(in_size, out_size)
    a64 = np.moveaxis(a, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + a64.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(a64.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += a64[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(a: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL Image.resize((out_w, out_h), BICUBIC) of an HWC uint8 array: horizontal pass, then vertical."""
    return _resample_axis(_resample_axis(a, out_w, 1), out_h, 0)


def nearest_index(in_size: int, out_size: int) -> np.ndarray:
    """Geometry.c ImagingScaleAffine (the path Image.resize(NEAREST) takes): the source coordinate starts at scale * 0.5
    and is ADVANCED BY REPEATED ADDITION of scale (double), truncated per step — not (x + 0.5) * scale."""
    scale = in_size / out_size
    xo = scale * 0.5
    idx = np.empty(out_size, dtype=np.int32)
    for x in range(out_size):
        idx[x] = int(xo)
        xo += scale
    return np.minimum(idx, in_size - 1)


def resize_nearest_u8(a: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    return a[nearest_index(a.shape[0], out_h)][:, nearest_index(a.shape[1], out_w)]


def shortest_edge_size(h: int, w: int, size: int):
    """transformers get_resize_output_image_size(default_to_square=False): short edge -> size, long edge truncated."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)     # (new_h, new_w)


def normalize_lut(mean, std, rescale: float = 1 / 255) -> np.ndarray:
    """float32((float32(float64(v) * rescale) - mean) / std) for v in 0..255, per channel: [C, 256] float32."""
    v = (np.arange(256, dtype=np.float64) * rescale).astype(np.float32)
    m, s = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    return ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def label_lut() -> np.ndarray:
    """(float32 pixel value of the label extractor) * 255 truncated to int64 — data/collator.py:91 — for v in 0..255."""
    v = (np.arange(256, dtype=np.float64) * (1 / 255)).astype(np.float32)
    v = ((v - np.float32(0)) / np.float32(1)).astype(np.float32)
    return (v * np.float32(255)).astype(np.int64)


def clip_image(a: np.ndarray, size: int, crop: int, mean, std) -> np.ndarray:
    """HWC uint8 RGB -> [3, crop, crop] float32, the CLIPFeatureExtractor pipeline."""
    nh, nw = shortest_edge_size(a.shape[0], a.shape[1], size)
    r = resize_bicubic_u8(a, nh, nw)
    top, left = (nh - crop) // 2, (nw - crop) // 2
    c = r[top:top + crop, left:left + crop]
    lut = normalize_lut(mean, std)
    return np.stack([lut[ch][c[:, :, ch]] for ch in range(3)], axis=0)


def clip_label(lab: np.ndarray, size: int, crop: int) -> np.ndarray:
    """HW uint8 class map -> [crop, crop] int64 through the label extractor (nearest, mean 0, std 1, * 255, .long())."""
    nh, nw = shortest_edge_size(lab.shape[0], lab.shape[1], size)
    r = resize_nearest_u8(lab, nh, nw)
    top, left = (nh - crop) // 2, (nw - crop) // 2
    return label_lut()[r[top:top + crop, left:left + crop]]
