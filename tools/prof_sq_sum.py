"""Summary of tools/prof_sq.sh: per kernel family, summed over its dispatches in the profiled run —
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (share of the run's SIMD-cycles with the matrix pipe busy)
  clock_GHz      = GRBM_GUI_ACTIVE / 8 / duration   (effective shader clock; reads high on dispatches shorter than ~0.3 ms)
  wave time split: SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY (issue stalls) / SQ_WAIT_ANY (s_waitcnt, barriers) over SQ_WAVE_CYCLES.
usage: python tools/prof_sq_sum.py <out-prefix> <pass dir> ...   (writes <out-prefix>.json, prints the table)"""
import collections
import csv
import glob
import json
import re
import sys

FAMILIES = ["gemm_nt_pp_kernel", "gemm_nt_w384_kernel", "gemm_nt_rows_kernel", "gemm_nt_dma_kernel", "gemm_nt_kernel", "gemm_tn_grouped_tbl_kernel",
            "gemm_tn_grouped_kernel", "gemm_tn_dma_kernel", "gemm_tn_kernel", "attn_fwd_kernel", "attn_bwd_dq2_kernel",
            "attn_bwd_dkdv_kernel", "ln_fwd_kernel", "ln_bwd_kernel", "head_ce_grp_kernel", "head_finish_kernel", "sgd_kernel",
            "bilinear_up_fwd_kernel", "bilinear_up_bwd_kernel", "add_n_kernel", "sr_gather_kernel", "sr_scatter_add_kernel", "l2norm_fwd_kernel", "l2norm_bwd_kernel"]


def short(k):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
    if not m or m.group(1) not in FAMILIES:
        return None
    return m.group(1) + (m.group(2) or "")


out_prefix, dirs = sys.argv[1], sys.argv[2:]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
ndisp = collections.defaultdict(int)
for d in dirs:
    grbm_pass = False
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            if n:
                cnt[n][r["Counter_Name"]] += float(r["Counter_Value"])
                grbm_pass |= r["Counter_Name"] == "GRBM_GUI_ACTIVE"
    if grbm_pass:   # durations from the pass that carries GRBM_GUI_ACTIVE (same dispatches as the clock's numerator)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = short(r["Kernel_Name"])
                if n:
                    dur[n] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    ndisp[n] += 1
res = {}
print(f"{'kernel':58s} {'disp':>5s} {'avg us':>8s} {'mfma_busy':>9s} {'clock GHz':>9s} {'active':>7s} {'wait_inst':>9s} {'wait_any':>8s} {'VALU/MFMA':>9s} {'LDS/MFMA':>8s} {'bank confl':>10s}")
for n in sorted(cnt, key=lambda k: -dur[k]):
    c = cnt[n]
    if not dur[n] or "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    mf = c.get("SQ_INSTS_MFMA", 0.0)
    row = dict(dispatches=ndisp[n], avg_us=dur[n] / ndisp[n] / 1e3, mfma_busy_frac=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024),
               clock_ghz=cyc / dur[n], active=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, wait_inst=c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
               wait_any=c.get("SQ_WAIT_ANY", 0.0) / wc, valu_per_mfma=(c.get("SQ_INSTS_VALU", 0.0) / mf) if mf else None,
               lds_per_mfma=(c.get("SQ_INSTS_LDS", 0.0) / mf) if mf else None,
               lds_bank_conflict_frac=(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None)
    res[n] = row
    f = lambda v, w, p: (f"{v:{w}.{p}f}" if v is not None else " " * (w - 1) + "-")   # noqa: E731
    print(f"{n[:58]:58s} {row['dispatches']:5d} {row['avg_us']:8.1f} {row['mfma_busy_frac']:9.3f} {row['clock_ghz']:9.2f} {row['active']:7.3f} "
          f"{row['wait_inst']:9.3f} {row['wait_any']:8.3f} {f(row['valu_per_mfma'], 9, 2)} {f(row['lds_per_mfma'], 8, 2)} {f(row['lds_bank_conflict_frac'], 10, 3)}")
json.dump(res, open(out_prefix + ".json", "w"), indent=1)
