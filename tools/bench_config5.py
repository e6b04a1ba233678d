#!/usr/bin/env python3
"""BASELINE.json configs[4] on one MI355X: the multi-scale decoder path on synthetic Swin-small stage tensors
(SURVEY.md §8d config 5): HierarchicalCrossA(in_dims=[96,192,384,768], depth=[1,1,1], dim=512, nhead=8, dropout=0)
+ score-map tail (normalize, einsum, bilinear x4) + CE at 512x512, forward + backward + SGD.
Prints one JSON line; `--profile-tail` additionally times the fused tail kernel alone (HBM roofline)."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import lc2is_amd.nn as N
from lc2is_amd import ops
from lc2is_amd.step import ParamArena


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--classes", type=int, default=150)
    ap.add_argument("--graph", action="store_true", help="capture the step into a hipGraph and time replays")
    ap.add_argument("--swin", action="store_true", help="end to end: Swin-small backbone (drop_path 0) produces the stage "
                    "tensors from 512x512 pixels instead of the synthetic ones")
    ap.add_argument("--prompt-ftn", action="store_true", help="the reference's own config-5 composition, lc2is_amd.nn.PromptFTN "
                    "(model/model.py:174-214): Swin-base -> frozen pooled CLIP text -> 8 prompt layers -> FTNDecoder -> score map + CE, "
                    "dropout / drop-path 0, K prompts of 8 tokens")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, K = args.batch, args.classes
    torch.manual_seed(5)
    dec = N.HierarchicalCrossA([96, 192, 384, 768], [1, 1, 1], 512, nhead=8, dropout=0).to(dev).train()
    tail = N.ScoreMapTail(4)
    swin = N.SwinTransformer(N.SWIN_S, drop_path_rate=0.0).to(dev).train() if args.swin else None
    pftn = None
    if args.prompt_ftn:
        a = N.SWIN_B
        pftn = N.PromptFTN(swin_arch=N.SwinArch(a.embed_dim, a.depths, a.num_heads, a.window, drop_path_rate=0.0), dropout=0.0).to(dev).train()
    model = torch.nn.ModuleList([pftn] if pftn is not None else [dec] + ([swin] if swin is not None else []))
    arena = ParamArena(model)
    g = torch.Generator().manual_seed(5)
    visual = [torch.randn(B, p, c, generator=g).to(dev) for p, c in zip((16384, 4096, 1024, 256), (96, 192, 384, 768))]
    text = torch.randn(B, K, 512, generator=g).to(dev)
    labels = torch.randint(0, K, (B, 512, 512), generator=g).to(dev)
    pixels = torch.randn(B, 3, 512, 512, generator=g).to(dev) if (args.swin or args.prompt_ftn) else None
    ids = torch.full((K, 8), 49407, dtype=torch.int64)
    ids[:, 0] = 49406
    ids[:, 1:5] = torch.randint(1, 49405, (K, 4), generator=g)
    mask = torch.zeros(K, 8, dtype=torch.int64)
    mask[:, :6] = 1
    pinputs = dict(pixel_values=pixels, input_ids=ids.to(dev), attention_mask=mask.to(dev)) if args.prompt_ftn else None

    def step():
        arena.zero_grad(set_to_none=True)
        if pftn is not None:
            loss = pftn.forward_loss(pinputs, labels)
            loss.backward()
            arena.finalize_grads()
            ops.sgd_step(arena.flat, arena.grad, None, 1e-5)
            for m in model.modules():
                if isinstance(m, N.HipModule):
                    m.invalidate_shadows()
            return loss
        t_in = text.detach().requires_grad_(True)   # a leaf made on the step's stream (its AccumulateGrad runs there: capturable)
        emb = dec(list(swin(pixels)) if swin is not None else visual, t_in)
        loss = tail.loss(emb, t_in, labels)
        loss.backward()
        arena.finalize_grads()
        ops.sgd_step(arena.flat, arena.grad, None, 1e-5)
        for m in model.modules():
            if isinstance(m, N.HipModule):
                m.invalidate_shadows()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    run = step
    if args.graph:   # the Swin path is ~1500 small launches per step and host-bound in eager mode: replay it as one hipGraph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            static_loss = step()

        def run():
            graph.replay()
            return static_loss
        for _ in range(2):
            run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # the fused tail kernel alone: reads [B,128,128,192] fp32 scores + int64 labels, writes the same-size gradient
    scores = torch.randn(B * 16384, 192, device=dev)
    for _ in range(2):
        ops.head_upsample_ce(scores, labels, B, 128, 128, K, 4, ops.INTERP_BILINEAR, want_grad=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.head_upsample_ce(scores, labels, B, 128, 128, K, 4, ops.INTERP_BILINEAR, want_grad=True)
    e1.record()
    torch.cuda.synchronize()
    t_tail = e0.elapsed_time(e1) / 5 * 1e-3
    alg_bytes = B * (16384 * 192 * 4 * 2 + 512 * 512 * 8)   # scores read + gradient write + labels
    ref_bytes = B * (K * 512 * 512 * 4) * 2                  # what the reference materialises: fp32 map write + read for CE
    print(json.dumps({
        "workload": "config5: HierarchicalCrossA + score-map tail + CE @512x512 on synthetic Swin-small stages",
        "batch": B, "hip_graph": bool(args.graph), "swin_backbone": bool(args.swin), "prompt_ftn": bool(args.prompt_ftn), "images_per_s": B / dt, "ms_per_step": dt * 1e3, "loss": float(loss.item()),
        "params_M": round(sum(p.numel() for p in model.parameters()) / 1e6, 2),
        "tail_kernel": {"us": t_tail * 1e6, "algorithmic_GBps": alg_bytes / t_tail / 1e9, "peak_GBps": 8000,
                        "frac": alg_bytes / t_tail / 8e12, "bytes_algorithmic": alg_bytes,
                        "bytes_reference_materialised_map": ref_bytes}}), flush=True)


if __name__ == "__main__":
    main()
