"""Calibration only (not a product path, not the oracle): the headline training step written with STOCK PyTorch / transformers
modules on the same GPU — HF CLIPVisionModel (ViT-B/16 @512) + CLIPTextModel, a torch pre-LN decoder layer with 512-wide text
memory, bicubic x4, Linear 768->512 on the upsampled tokens, prototype logits, CrossEntropyLoss, SGD — in the literal order of
the reference's forward, bf16 autocast, random init, synthetic batch of bench.py's shape.
It answers "what does the same step cost without this repo's kernels on an MI355X"; it is not checked for parity.
  python tools/torch_gpu_baseline.py [--batch 32] [--steps 10]"""
import argparse
import json
import time

import torch
import torch.nn.functional as F
from torch import nn
from transformers import CLIPTextConfig, CLIPTextModel, CLIPVisionConfig, CLIPVisionModel


class Decoder(nn.Module):   # x += SA(LN1 x); x += CA(LN2 x, mem); x += W2 relu(W1 LN3 x)   (pre-LN, d 768, memory 512, 8 heads, ff 2048)
    def __init__(self):
        super().__init__()
        self.sa = nn.MultiheadAttention(768, 8, batch_first=True, bias=False)
        self.ca = nn.MultiheadAttention(768, 8, kdim=512, vdim=512, batch_first=True)
        self.l1, self.l2 = nn.Linear(768, 2048, bias=False), nn.Linear(2048, 768, bias=False)
        self.n1, self.n2, self.n3 = nn.LayerNorm(768, bias=False), nn.LayerNorm(768, bias=False), nn.LayerNorm(768, bias=False)

    def forward(self, x, mem, pad):
        h = self.n1(x)
        x = x + self.sa(h, h, h, need_weights=False)[0]
        x = x + self.ca(self.n2(x), mem, mem, key_padding_mask=pad, need_weights=False)[0]
        return x + self.l2(F.relu(self.l1(self.n3(x))))


class Model(nn.Module):
    def __init__(self, in_size):
        super().__init__()
        self.vision = CLIPVisionModel(CLIPVisionConfig(image_size=in_size, patch_size=16))
        self.text = CLIPTextModel(CLIPTextConfig())
        self.dec = Decoder()
        self.visual, self.textual = nn.Linear(768, 512), nn.Linear(512, 512)
        self.prototypes = nn.Parameter(torch.randn(151, 512))
        self.grid = in_size // 16

    def forward(self, pixel_values, input_ids, attention_mask):
        v = self.vision(pixel_values=pixel_values).last_hidden_state[:, 1:, :]
        t = self.text(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        v = self.dec(v, t, attention_mask == 0)
        B, P, C = v.shape
        v = F.interpolate(v.transpose(1, 2).reshape(B, C, self.grid, self.grid), scale_factor=4, mode="bicubic")
        v = self.visual(v.flatten(2).transpose(1, 2))
        logits = v @ self.textual(self.prototypes).t()
        return logits.transpose(1, 2).reshape(B, 151, 4 * self.grid, 4 * self.grid)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Model(512).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-5)
    g = torch.Generator(device="cpu").manual_seed(2)
    px = torch.randn(a.batch, 3, 512, 512, generator=g).to(dev)
    labels = torch.randint(0, 151, (a.batch, 128, 128), generator=g).to(dev)
    ids = torch.randint(1, 49405, (a.batch, 16), generator=g)
    ids[:, 0], ids[:, 11:] = 49406, 49407
    mask = torch.ones(a.batch, 16, dtype=torch.long)
    mask[:, 12:] = 0
    ids, mask = ids.to(dev), mask.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(px, ids, mask)
        loss = F.cross_entropy(logits.float(), labels)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "stock PyTorch / transformers modules, bf16 autocast, literal order of the reference forward", "batch": a.batch,
                      "images_per_s": a.batch * a.steps / dt, "ms_per_step": dt / a.steps * 1e3, "loss": float(loss),
                      "torch": torch.__version__, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}), flush=True)


if __name__ == "__main__":
    main()
