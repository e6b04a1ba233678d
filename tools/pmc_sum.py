"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.
usage: pmc_sum.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix> [commit]
Writes <prefix>_hbm_per_kernel.csv and <prefix>_hbm.json (the dominant kernel: every instantiation of
gemm_nt_dma_kernel<256, 256, 2, 4, *> and of its persistent forms gemm_nt_persist2_kernel<*> / gemm_nt_pp_kernel<*> aggregated).  Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE is KB
and tallies 128-B requests at 64 B on gfx950 -> bytes = 2 x 1000 x FETCH_SIZE; WRITE_SIZE is KB, exact."""
import collections, csv, json, re, sys

DOM = ("gemm_nt_dma_kernel<256, 256, 2, 4,", "gemm_nt_persist2_kernel<", "gemm_nt_pp_kernel<", "gemm_nt_w384_kernel")   # the 256x256 LDS-DMA NT GEMM: plain, persistent and ping-pong forms


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "attn_" in k:    # vision / text / decoder launches of one attention kernel differ by grid: keep them apart
            k = k + " grid=" + r.get("Grid_Size", "?")
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in fetch:
        if k not in write or nf[k] == 0:
            continue
        rd, wr = 2.0 * 1000.0 * fetch[k] / nf[k], 1000.0 * write[k] / nw[k]
        rows.append((k, nf[k], fetch[k] / nf[k], rd, wr, rd + wr))
    rows.sort(key=lambda r: -r[1] * r[5])
    with open(sys.argv[3] + "_hbm_per_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "fetch_size_kb_per_launch", "read_bytes_per_launch", "write_bytes_per_launch",
                    "hbm_bytes_per_launch"])
        for r in rows[:60]:
            w.writerow([r[0][:120] + (r[0][r[0].rfind(" grid="):] if " grid=" in r[0][120:] else ""), *r[1:]])
    dom = [r for r in rows if any(d in r[0] for d in DOM)]
    n = sum(r[1] for r in dom)
    rd = sum(r[1] * r[3] for r in dom) / n
    wr = sum(r[1] * r[4] for r in dom) / n
    out = {"commit": sys.argv[4] if len(sys.argv) > 4 else "unrecorded",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline (two separate passes, tools/prof_pmc.sh)",
           "corrections": "FETCH_SIZE is KB and tallies 128-B requests at 64 B on gfx950: bytes = 2 x 1000 x FETCH_SIZE; "
                          "WRITE_SIZE KB exact (MI355X_MICROARCH.md, HBM section)",
           "kernel": "gemm_nt_dma_kernel<256, 256, 2, 4, *> + gemm_nt_pp_kernel<*> / gemm_nt_persist2_kernel<*> + gemm_nt_w384_kernel (the large-tile LDS-DMA NT GEMM: 256x256 plain / persistent, 256x384; all epilogue instantiations, launch-weighted)", "launches": n,
           "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump(out, open(sys.argv[3] + "_hbm.json", "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
