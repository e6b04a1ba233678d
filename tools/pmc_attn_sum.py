"""Per-kernel means of the counters collected by tools/pmc_attn.sh (rocprofv3 counter_collection CSVs) for the attention kernels,
with the derived quantities the verdicts ask for: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES ... per SIMD)
is reported as SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) when GRBM_GUI_ACTIVE is present."""
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn_" not in k:
                continue
            m = re.search(r"attn_\w+<[^>]*>", k)
            name = (m.group(0) if m else k[:60]) + " grid=" + r.get("Grid_Size", "?")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name)
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    wc = c.get("SQ_WAVE_CYCLES")
    for k in sorted(c):
        extra = f"  ({c[k] / wc:.3f} of SQ_WAVE_CYCLES)" if wc and k != "SQ_WAVE_CYCLES" else ""
        print(f"   {k:28s} {c[k]:16.0f}{extra}")
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # shader cycles of the dispatch (the counter sums the 8 XCDs)
        print(f"   mfma_busy_frac               {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):16.3f}  (MFMA-busy cycles / (dispatch cycles x 1024 SIMDs))")
