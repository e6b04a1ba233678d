"""CPU: the preprocessing oracle (oracle/preprocess_cpu.py) against Pillow and transformers' CLIPImageProcessor, the
libraries the reference calls (evaluate.py:58-61).  Bit-exact: uint8 images, float32 pixel values, int64 labels."""
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from oracle import preprocess_cpu as P  # noqa: E402


@pytest.mark.parametrize("h,w,oh,ow", [(97, 131, 64, 86), (131, 97, 86, 64), (40, 56, 64, 89), (683, 512, 170, 128), (33, 33, 64, 64)])
def test_bicubic_resize_matches_pillow(h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.BICUBIC))
    assert np.array_equal(P.resize_bicubic_u8(a, oh, ow), ref)


@pytest.mark.parametrize("h,w,oh,ow", [(97, 131, 32, 43), (512, 683, 128, 170), (30, 20, 64, 43), (150, 101, 47, 32), (683, 512, 170, 128), (375, 500, 128, 170)])
def test_nearest_resize_matches_pillow(h, w, oh, ow):
    rng = np.random.default_rng(h + w)
    a = rng.integers(0, 151, (h, w), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.NEAREST))
    assert np.array_equal(P.resize_nearest_u8(a, oh, ow), ref)


def test_clip_pipeline_matches_transformers():
    tr = pytest.importorskip("transformers")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        img_p = tr.CLIPImageProcessor(size={"shortest_edge": 64}, crop_size={"height": 64, "width": 64})
        lab_p = tr.CLIPImageProcessor(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}, image_mean=[0, 0, 0],
                                      image_std=[1, 1, 1], resample=Image.NEAREST, do_convert_rgb=False)
    rng = np.random.default_rng(7)
    for (h, w) in ((97, 131), (150, 101), (64, 64)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = img_p([Image.fromarray(a)], return_tensors="np")["pixel_values"][0]
        got = P.clip_image(a, 64, 64, img_p.image_mean, img_p.image_std)
        assert got.dtype == np.float32 and np.array_equal(got, ref)
        lab = rng.integers(0, 151, (h, w), dtype=np.uint8)
        exp3 = np.repeat(lab[None], 3, axis=0)                                   # label.expand(3, -1, -1), data/collator.py:89
        lref = lab_p([exp3], return_tensors="np")["pixel_values"][0, 0]
        lref = (lref * np.float32(255)).astype(np.int64)                         # data/collator.py:91
        assert np.array_equal(P.clip_label(lab, 32, 32), lref)
    # the float32 round trip v/255*255 truncated to int64 happens to be the identity on 0..255 (checked, not assumed)
    assert np.array_equal(P.label_lut(), np.arange(256))
