"""GPU parity of the device-side preprocessing — bit-exact against the CPU oracle (itself pinned to
Pillow / transformers in tests/test_preprocess_cpu.py) and, where Pillow is importable on the box, against Pillow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,size,crop", [(97, 131, 64, 64), (150, 101, 64, 64), (40, 56, 64, 48), (375, 500, 224, 224), (64, 64, 64, 64)])
def test_image_preprocessor_bit_exact(dev, h, w, size, crop):
    from lc2is_amd.data import ClipImagePreprocessor
    from lc2is_amd.data.preprocess import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
    from oracle import preprocess_cpu as P
    rng = np.random.default_rng(h * 7 + w)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (w, h, 3), dtype=np.uint8)]
    pre = ClipImagePreprocessor(size=size, crop_size=crop, device=dev)
    out = pre(imgs)
    assert out.shape == (2, 3, crop, crop) and out.dtype == torch.float32
    for i, a in enumerate(imgs):
        ref = P.clip_image(a, size, crop, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD)
        assert np.array_equal(out[i].cpu().numpy(), ref)
    try:
        from PIL import Image
    except ImportError:
        return
    nh, nw = P.shortest_edge_size(h, w, size)
    pil = np.asarray(Image.fromarray(imgs[0]).resize((nw, nh), resample=Image.BICUBIC))
    from lc2is_amd import ops
    from lc2is_amd.data.preprocess import _bicubic_tables
    x = torch.from_numpy(imgs[0]).to(dev)
    if nw != w:
        x = ops.resample_u8(x, nw, 1, *[torch.from_numpy(t).to(dev) for t in _bicubic_tables(w, nw)])
    if nh != h:
        x = ops.resample_u8(x, nh, 0, *[torch.from_numpy(t).to(dev) for t in _bicubic_tables(h, nh)])
    assert np.array_equal(x.cpu().numpy(), pil)


@pytest.mark.parametrize("h,w,size", [(97, 131, 32), (150, 101, 32), (375, 500, 128), (20, 30, 64)])
def test_label_preprocessor_bit_exact(dev, h, w, size):
    from lc2is_amd.data import ClipLabelPreprocessor
    from oracle import preprocess_cpu as P
    rng = np.random.default_rng(h + w)
    labs = [rng.integers(0, 151, (h, w), dtype=np.uint8), rng.integers(0, 151, (w, h), dtype=np.uint8)]
    out = ClipLabelPreprocessor(size=size, crop_size=size, device=dev)(labs)
    assert out.shape == (2, size, size) and out.dtype == torch.int64
    for i, a in enumerate(labs):
        assert np.array_equal(out[i].cpu().numpy(), P.clip_label(a, size, size))


def test_collator_and_errors(dev):
    from lc2is_amd.data import ADE20KCollator, ClipImagePreprocessor
    feats = [(torch.randn(1, 3, 8, 8, device=dev), torch.zeros(1, 2, 2, dtype=torch.long, device=dev), dict(size=(2, 2))) for _ in range(3)]
    inputs, metas = ADE20KCollator()(feats)
    assert inputs["pixel_values"].shape == (3, 3, 8, 8) and inputs["label"].shape == (3, 2, 2) and len(metas) == 3
    with pytest.raises(RuntimeError):
        ClipImagePreprocessor(device="cpu")
    with pytest.raises(TypeError):
        ClipImagePreprocessor(size=8, device=dev)([np.zeros((8, 8, 3), dtype=np.float32)])
