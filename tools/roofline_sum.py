#!/usr/bin/env python3
"""Per-kernel roofline figures of the headline step from rocprofv3 kernel traces.

usage: roofline_sum.py <full p_kernel_trace.csv> <steps in it> <encoder-only p_kernel_trace.csv> <steps in it> <commit> <out.json>
       [--pmc <*_hbm.json of tools/pmc_sum.py>]

* encoder_frac: the ViT encoder's algorithmic FLOPs (642.15 GF per image and train step) over the summed duration of
  the encoder's kernels INSIDE the full step, against 2.5 PFLOP/s.  A dispatch of the full step belongs to the encoder when its
  (kernel name, grid, block) signature occurs in the encoder-only trace (tools/encoder_only.py: same shapes, nothing else
  running); per step the full trace must hold at least as many dispatches of each signature as the encoder-only trace.
* bandwidth kernels: achieved GB/s = algorithmic bytes per launch / average duration, against 8 TB/s, for the LayerNorm
  backward of the vision tower, the fused upsample + CE head and the optimizer.
"""
import collections
import csv
import json
import sys

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
ENC_GF_PER_IMG, STEP_GF_PER_IMG = 642.15, 744.23
B, TOK, C = 32, 1025, 768


def load(path):
    sig = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
        sig[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return sig


def main():
    full, nfull, enc, nenc, commit, out_path = sys.argv[1], float(sys.argv[2]), sys.argv[3], float(sys.argv[4]), sys.argv[5], sys.argv[6]
    pmc = sys.argv[sys.argv.index("--pmc") + 1] if "--pmc" in sys.argv else None
    sq = sys.argv[sys.argv.index("--sq") + 1] if "--sq" in sys.argv else None
    F, E = load(full), load(enc)
    enc_s, rows, short = 0.0, [], []
    for k, durs in E.items():
        if "copyBuffer" in k[0] or "fillBuffer" in k[0] or "at::native" in k[0]:
            continue                                    # allocator / torch glue of the stand-alone script, not encoder kernels
        per_step_enc = len(durs) / nenc
        got = F.get(k, [])
        per_step_full = len(got) / nfull
        if per_step_full + 1e-9 < per_step_enc:
            short.append((k[0][:80], k[1], per_step_enc, per_step_full))
            continue
        # the signature may also serve other modules in the full step: take the encoder's share of the launches
        share = min(1.0, per_step_enc / per_step_full) if per_step_full else 0.0
        t = sum(got) / nfull * share
        enc_s += t
        rows.append(dict(kernel=k[0][:100], grid=k[1], launches_per_step=per_step_enc, ms_per_step=t * 1e3,
                         avg_us=sum(got) / len(got) * 1e6))
    rows.sort(key=lambda r: -r["ms_per_step"])
    total_s = sum(sum(v) for v in F.values()) / nfull
    out = dict(commit=commit, source="rocprofv3 --kernel-trace of bench.py --steps 3 --warmup 2 and of tools/encoder_only.py (tools/prof_roofline.sh)",
               kernel_ms_per_step=total_s * 1e3, encoder_kernel_ms_per_step=enc_s * 1e3,
               encoder_tflops=ENC_GF_PER_IMG * 1e9 * B / enc_s / 1e12 if enc_s else None,
               encoder_frac=ENC_GF_PER_IMG * 1e9 * B / enc_s / 1e12 / PEAK_TF if enc_s else None,
               encoder_signatures_missing_in_full_step=short, encoder_kernels=rows[:24])

    def bw(name_part, grid, bytes_per_launch, label):
        durs = [d for k, v in F.items() if name_part in k[0] and (grid is None or k[1] == grid) for d in v]
        if not durs:
            return None
        avg = sum(durs) / len(durs)
        return dict(kernel=label, launches_per_step=len(durs) / nfull, avg_us=avg * 1e6, algorithmic_bytes_per_launch=bytes_per_launch,
                    achieved_gbs=bytes_per_launch / avg / 1e9, frac_of_8tbs=bytes_per_launch / avg / 1e9 / PEAK_GBS)

    M = B * TOK
    bws = []
    # LayerNorm backward over the fp32 stream: reads dy (fp32) + x (fp32), writes dx fp32 + its bf16 twin = 18 B/element
    # (DESIGN.md §4); the vision tower's launches are the ones whose duration class is M x 768
    ln = [(k, v) for k, v in F.items() if "ln_bwd_kernel" in k[0]]
    if ln:
        k, v = max(ln, key=lambda kv: sum(kv[1]))
        bws.append(bw("ln_bwd_kernel", k[1], 18.0 * M * C, "ln_bwd (vision tower, M x 768 fp32 stream)"))
    # (the fused head moves 16x fewer bytes than the literal order by design: its time is the fp32 matrix pipe + the softmax
    #  arithmetic over 151 x 16384 scores per image, not HBM — the GB/s figure is reported because SURVEY.md §8d asks for it)
    bws.append(bw("head_ce_grp", None, 1.9e6 * B, "head_ce_grp, S = 4 (fused bicubic x4 + CE fwd/bwd, 1.9 MB/img; compute-bound by design; `head_ce_s4` before round 3)"))
    bws.append(bw("sgd_kernel", None, 12.0 * 157.09e6, "sgd (157.09 M fp32 parameters: read p, g, write p)"))
    out["bandwidth_kernels"] = [b for b in bws if b]
    if pmc:
        rec = json.load(open(pmc))
        out["dominant_kernel_hbm_bytes_per_launch"] = rec["hbm_bytes_per_launch"]
        out["dominant_kernel_hbm_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_pmc.sh) @ {rec.get('commit')}"
    if sq:   # SQ / GRBM counter evidence (tools/prof_sq.sh): matrix-pipe busy share and effective clock per kernel family
        rec = json.load(open(sq))
        out["sq_counters"] = {k: {f: v[f] for f in ("dispatches", "avg_us", "mfma_busy_frac", "clock_ghz", "active", "wait_inst", "wait_any")}
                              for k, v in rec.items()}
        dom = {k: v for k, v in rec.items() if k.startswith(("gemm_nt_pp_kernel", "gemm_nt_persist2_kernel", "gemm_nt_w384_kernel", "gemm_nt_dma_kernel<256, 256, 2, 4,",
                                                              "gemm_nt_dma_kernel<256, 256, 2, 4, 1", "gemm_nt_dma_kernel<256, 256, 2, 4, 2",
                                                              "gemm_nt_dma_kernel<256, 256, 2, 4, 3", "gemm_nt_dma_kernel<256, 256, 2, 4, 4"))}
        w = sum(v["dispatches"] * v["avg_us"] for v in dom.values())
        if w:
            out["dominant_kernel_mfma_busy_frac"] = sum(v["mfma_busy_frac"] * v["dispatches"] * v["avg_us"] for v in dom.values()) / w
            out["dominant_kernel_clock_ghz"] = sum(v["clock_ghz"] * v["dispatches"] * v["avg_us"] for v in dom.values()) / w
            out["dominant_kernel_counter_source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes over bench.py --steps 2 "
                                                     f"(tools/prof_sq.sh) @ {commit}; duration-weighted over {sorted(dom)}; the clock reads high on "
                                                     "dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back)")
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: out.get(k) for k in ("commit", "kernel_ms_per_step", "encoder_kernel_ms_per_step", "encoder_frac",
                                              "dominant_kernel_mfma_busy_frac", "dominant_kernel_clock_ghz")}))
    for b in out["bandwidth_kernels"]:
        print(f"  {b['kernel']}: {b['avg_us']:.1f} us, {b['achieved_gbs']:.0f} GB/s ({b['frac_of_8tbs']:.2f} of 8 TB/s)")
    if short:
        print("  WARNING: encoder signatures with fewer launches in the full step:", short[:5])


if __name__ == "__main__":
    main()
