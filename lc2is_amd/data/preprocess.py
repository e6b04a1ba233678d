"""Image / label preprocessing on the MI355X, bit-exact with what the reference runs on the host.

The reference (evaluate.py:58-61, data/collator.py:82-91) pushes every PIL image through transformers'
``CLIPFeatureExtractor(size=S, crop_size=S)``: resize so the short edge is S (Pillow bicubic), centre crop, ``x * (1/255)``
in float64 -> float32, ``(x - mean) / std`` in float32, channels first; labels are expanded to three channels and pushed
through the same extractor with ``resample=NEAREST``, mean 0, std 1, then ``(pixel_values[:, 0] * 255).long()``.
Here the decoded uint8 image is the only thing that crosses PCIe; resampling (Pillow's 22-bit fixed-point separable
passes, ``lc2is_resample_u8``), nearest gathers, the crop and the float conversion (256-entry tables holding the
reference's exact float results) run as HIP kernels and write straight into the batch tensor.

    img = ClipImagePreprocessor(size=512, crop_size=512)           # CLIP mean / std defaults, like the extractor
    lab = ClipLabelPreprocessor(size=128, crop_size=128)
    batch = {"pixel_values": img(list_of_uint8_HWC), "label": lab(list_of_uint8_HW)}

Only the coefficient tables (a few KB per distinct image size, cached) are computed on the host, with the same double
arithmetic as Pillow's ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` / ``ImagingScaleAffine``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import ops

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # transformers.image_utils
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_PRECISION = 1 << (32 - 8 - 2)


def _cubic(x: float) -> float:            # Pillow Resample.c bicubic_filter, a = -0.5
    x = abs(x)
    if x < 1.0:
        return (1.5 * x - 2.5) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * -0.5
    return 0.0


def _bicubic_tables(n_in: int, n_out: int):
    """Per output sample: (first source index, tap count) and the taps as 22-bit fixed point (Resample.c)."""
    step = n_in / n_out
    fscale = max(step, 1.0)
    reach = 2.0 * fscale
    taps = int(math.ceil(reach)) * 2 + 1
    span = np.zeros((n_out, 2), dtype=np.int32)
    fixed = np.zeros((n_out, taps), dtype=np.int32)
    inv = 1.0 / fscale
    for o in range(n_out):
        mid = (o + 0.5) * step
        lo = max(int(mid - reach + 0.5), 0)
        hi = min(int(mid + reach + 0.5), n_in)
        w = [_cubic((t + lo - mid + 0.5) * inv) for t in range(hi - lo)]
        total = 0.0
        for v in w:                      # sequential double sum, like the C loop
            total += v
        for t, v in enumerate(w):
            if total != 0.0:
                v = v / total
            fixed[o, t] = int(v * _PRECISION - 0.5) if v < 0 else int(v * _PRECISION + 0.5)
        span[o] = (lo, hi - lo)
    return span, fixed


def _nearest_table(n_in: int, n_out: int) -> np.ndarray:
    step = n_in / n_out
    pos = step * 0.5
    out = np.empty(n_out, dtype=np.int32)
    for o in range(n_out):               # Geometry.c advances the coordinate by repeated addition
        out[o] = min(int(pos), n_in - 1)
        pos += step
    return out


def _target_size(h: int, w: int, size: int):
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


class _Base:
    def __init__(self, size: int, crop_size: int | None, device):
        self.size, self.crop = int(size), int(crop_size if crop_size is not None else size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("lc2is_amd.data: preprocessing runs on a HIP device; there is no CPU path "
                               "(the CPU oracle is oracle/preprocess_cpu.py, test-only)")
        self._tables = {}

    def _upload(self, a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        if a.dtype != torch.uint8:
            raise TypeError("lc2is_amd.data: images / labels must be uint8 (decoded pixels or class ids)")
        return a.to(self.device, non_blocking=True).contiguous()

    def _dev(self, key, build):
        t = self._tables.get(key)
        if t is None:
            t = tuple(torch.from_numpy(x).to(self.device) for x in build())
            self._tables[key] = t
        return t

    def _crop_origin(self, nh, nw):
        if nh < self.crop or nw < self.crop:
            raise ValueError("lc2is_amd.data: crop_size larger than the resized image is not supported")
        return (nh - self.crop) // 2, (nw - self.crop) // 2


class ClipImagePreprocessor(_Base):
    """``CLIPFeatureExtractor(size=, crop_size=)`` for RGB uint8 HWC images -> float32 [B,3,crop,crop] on the device."""

    def __init__(self, size: int = 224, crop_size: int | None = None, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD,
                 rescale_factor: float = 1 / 255, device="cuda"):
        super().__init__(size, crop_size, device)
        v = (np.arange(256, dtype=np.float64) * rescale_factor).astype(np.float32)
        m, s = np.array(image_mean, dtype=np.float32), np.array(image_std, dtype=np.float32)
        self.lut = torch.from_numpy(((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)).to(self.device).contiguous()

    def __call__(self, images) -> torch.Tensor:
        out = torch.empty(len(images), 3, self.crop, self.crop, dtype=torch.float32, device=self.device)
        for i, im in enumerate(images):
            x = self._upload(im)
            if x.dim() != 3 or x.shape[2] != 3:
                raise ValueError("lc2is_amd.data: images must be HWC with 3 channels")
            h, w = x.shape[0], x.shape[1]
            nh, nw = _target_size(h, w, self.size)
            if nw != w:
                x = ops.resample_u8(x, nw, 1, *self._dev(("cubic", w, nw), lambda: _bicubic_tables(w, nw)))
            if nh != h:
                x = ops.resample_u8(x, nh, 0, *self._dev(("cubic", h, nh), lambda: _bicubic_tables(h, nh)))
            top, left = self._crop_origin(nh, nw)
            ops.crop_lut(x, top, left, self.crop, lut_f32=self.lut, out_f32=out[i])
        return out


class ClipLabelPreprocessor(_Base):
    """The reference's label transform (evaluate.py:59, data/collator.py:89-91): nearest resize of the short edge, centre
    crop, float round trip, int64.  uint8 HW class maps -> int64 [B,crop,crop]."""

    def __init__(self, size: int = 224, crop_size: int | None = None, device="cuda"):
        super().__init__(size, crop_size, device)
        v = (np.arange(256, dtype=np.float64) * (1 / 255)).astype(np.float32)
        v = ((v - np.float32(0)) / np.float32(1)).astype(np.float32)
        self.lut = torch.from_numpy((v * np.float32(255)).astype(np.int64)).to(self.device)

    def __call__(self, labels) -> torch.Tensor:
        out = torch.empty(len(labels), self.crop, self.crop, dtype=torch.int64, device=self.device)
        for i, lab in enumerate(labels):
            x = self._upload(lab)
            if x.dim() == 2:
                x = x.unsqueeze(-1).contiguous()
            h, w = x.shape[0], x.shape[1]
            nh, nw = _target_size(h, w, self.size)
            if (nh, nw) != (h, w):
                yi, = self._dev(("near", h, nh), lambda: (_nearest_table(h, nh),))
                xi, = self._dev(("near", w, nw), lambda: (_nearest_table(w, nw),))
                x = ops.gather2d_u8(x, yi, xi)
            top, left = self._crop_origin(nh, nw)
            ops.crop_lut(x, top, left, self.crop, lut_i64=self.lut, out_i64=out[i])
        return out


class ADE20KCollator:
    """data/collator.py:168-180: features = [(img [1,3,H,W], label [1,h,w], metas)] -> (dict(pixel_values, label), metas)."""

    def __call__(self, features):
        img_list, label_list, metas_list = [list(f) for f in zip(*features)]
        return dict(pixel_values=torch.cat(img_list, dim=0), label=torch.cat(label_list, dim=0)), metas_list
