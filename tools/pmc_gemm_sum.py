"""Per-kernel means of the counters collected by tools/pmc_gemm.sh, with kernel duration from the kernel trace:
effective clock = GRBM_GUI_ACTIVE / 8 / duration, mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt" not in k and "gemm_tn" not in k:
                continue
            m = re.search(r"gemm_\w+<[^>]*>|gemm_\w+", k)
            acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt" not in k and "gemm_tn" not in k:
                continue
            m = re.search(r"gemm_\w+<[^>]*>|gemm_\w+", k)
            dur[m.group(0)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for name in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    ns = sorted(dur[name])[len(dur[name]) // 2] if dur[name] else 0.0
    print(f"{name}   duration (median, under the counter passes) {ns / 1e3:.1f} us")
    wc = c.get("SQ_WAVE_CYCLES")
    for k in sorted(c):
        extra = f"  ({c[k] / wc:.3f} of SQ_WAVE_CYCLES)" if wc and k != "SQ_WAVE_CYCLES" else ""
        print(f"   {k:28s} {c[k]:16.0f}{extra}")
    if "GRBM_GUI_ACTIVE" in c and ns:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        print(f"   effective_clock_GHz          {cyc / ns:16.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"   mfma_busy_frac               {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):16.3f}")
