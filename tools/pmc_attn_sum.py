"""Per-kernel means of the counters collected by tools/pmc_attn.sh (rocprofv3 counter_collection CSVs), attention kernels at the
largest grid only (the S = 1025 vision shape)."""
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
