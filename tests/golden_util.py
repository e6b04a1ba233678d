"""Deterministic weights shared by tools/make_golden.py (which loads them into the REFERENCE modules) and the
tests (which load them into the oracle / the HIP modules): fixtures then only need inputs and outputs."""
import torch


def make_weights(shapes: dict, seed: int) -> dict:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if "norm" in name and name.endswith("weight"):
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = torch.randn(shape, generator=g) * (max(fan_in, 1) ** -0.5)
    return out


def ftn_inputs(seed: int, B: int = 1):
    """Stage tensors of ftn.Decoder's hard-coded geometry (model/ftn.py:70,104) and an output gradient."""
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, h * h, c, generator=g) for h, c in zip((128, 64, 32, 16), (128, 256, 512, 1024))]
    dout = torch.randn(B, 16384, 512, generator=g) * 0.1
    return xs, dout


def prompt_inputs(seed: int, B: int = 2, K: int = 150, P: int = 256, d_model: int = 512, d_kv: int = 1024):
    """Inputs of the reference's PromptDecoder at its real use (model/model.py:183,200): tgt = K text embeddings,
    memory = the last Swin stage's P visual tokens of width d_kv; plus an output gradient."""
    g = torch.Generator().manual_seed(seed)
    tgt = torch.randn(B, K, d_model, generator=g)
    mem = torch.randn(B, P, d_kv, generator=g)
    dout = torch.randn(B, K, d_model, generator=g) * 0.1
    return tgt, mem, dout


def clip_full_inputs(seed: int, B: int = 2, size: int = 64, tokens: int = 17, C: int = 128):
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(B, 3, size, size, generator=g)
    dout = torch.randn(B, tokens, C, generator=g) * 0.1     # includes the CLS row (token 0)
    return pix, dout


def config1_batch(i: int, in_size: int = 128, out_size: int = 32, L: int = 16):
    """SURVEY.md §8d config 1 (the 16-image plumbing shard; ADE20K itself is not available offline): image i has
    pixel_values ~ N(0,1) from seed 1000+i, labels uniform over 151 classes and constant on 4x4 blocks, and the prompt
    BOS + 10 random tokens + EOS + 4 pads (mask 0).  Returns ``(inputs incl. "label", metas)`` like ADE20KCollator."""
    g = torch.Generator().manual_seed(1000 + i)
    pix = torch.randn(1, 3, in_size, in_size, generator=g)
    lab = torch.randint(0, 151, (1, out_size // 4, out_size // 4), generator=g)
    lab = lab.repeat_interleave(4, 1).repeat_interleave(4, 2)
    ids = torch.full((1, L), 49407, dtype=torch.int64)
    ids[0, 0] = 49406
    ids[0, 1:11] = torch.randint(1, 49405, (10,), generator=g)
    mask = torch.zeros(1, L, dtype=torch.int64)
    mask[0, :12] = 1
    return dict(pixel_values=pix, input_ids=ids, attention_mask=mask, label=lab), None


DROP_SITES = ("sa_p", "d1", "ca_p", "d2", "ff", "d3")     # call order of F.dropout inside torch's TransformerDecoderLayer


def dropout_case(seed: int, norm_first: bool, B=2, K=24, P=40, d_model=128, d_kv=192, nhead=2, F=256, layers=2, p=0.25):
    """Inputs + per-layer dropout multipliers keep/(1-p) for the six sites of a decoder layer (tiny dims)."""
    g = torch.Generator().manual_seed(seed)
    tgt = torch.randn(B, K, d_model, generator=g)
    mem = torch.randn(B, P, d_kv, generator=g)
    dout = torch.randn(B, K, d_model, generator=g) * 0.1
    shapes = dict(sa_p=(B, nhead, K, K), d1=(B, K, d_model), ca_p=(B, nhead, K, P), d2=(B, K, d_model), ff=(B, K, F),
                  d3=(B, K, d_model))
    drops = [{s: (torch.rand(shapes[s], generator=g) >= p).float() / (1.0 - p) for s in DROP_SITES} for _ in range(layers)]
    return tgt, mem, dout, drops


def swin_droppath_inputs(seed: int = 63, B: int = 4):
    """Inputs of the drop-path fixture (tools/make_golden.py make_swin_droppath): pixels, per-block keep decisions [8, B]
    (call order of the SwinDropPath modules) and the output gradients of the four hidden states of the tiny Swin."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, 176, 176, generator=g)
    keeps = (torch.rand(8, B, generator=g) >= 0.4).float()
    douts = [torch.randn(B, n, c, generator=g) * 0.2 for n, c in ((1936, 32), (484, 64), (121, 128), (36, 256))]
    return x, keeps, douts


def _prompt_ids(g, K: int, L: int = 8, vocab_hi: int = 500, bos: int = 510, eos: int = 511):
    """K prompts: BOS, 3-5 random tokens, EOS, EOS padding (mask 0 on the padding) — the reference's tokenizer layout."""
    ids = torch.full((K, L), eos, dtype=torch.int64)
    mask = torch.zeros(K, L, dtype=torch.int64)
    for k in range(K):
        n = 3 + (k % 3)
        ids[k, 0] = bos
        ids[k, 1:1 + n] = torch.randint(1, vocab_hi, (n,), generator=g)
        mask[k, :n + 2] = 1
    return ids, mask


def prompt_ftn_inputs(seed: int, K: int = 6):
    """One 512 x 512 image (the 128 x 128 token grid PromptFTN hard-codes), K prompts, labels over the K classes."""
    g = torch.Generator().manual_seed(seed)
    ids, mask = _prompt_ids(g, K)
    inputs = dict(pixel_values=torch.randn(1, 3, 512, 512, generator=g), input_ids=ids, attention_mask=mask)
    labels = torch.randint(0, K, (1, 512, 512), generator=g)
    return inputs, labels


def dense_clip_inputs(seed: int, B: int = 2, K: int = 5):
    g = torch.Generator().manual_seed(seed)
    ids, mask = _prompt_ids(g, K)
    inputs = dict(pixel_values=torch.randn(B, 3, 64, 64, generator=g), input_ids=ids, attention_mask=mask)
    ds = torch.randn(B, K, 4, 4, generator=g)
    do = torch.randn(B, 17, 256, generator=g) * 0.1
    return inputs, ds, do
