#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries kept under profiles/.

usage: python profiles/summarize.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [--kernel k_lk_chain] [--idle-launches N]
 - copies *_kernel_stats.csv  -> profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats)
 - aggregates the two PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
   MI355X_MICROARCH.md 'rocprofv3 PMC slots' requires) for the dominant kernel into
   profiles/<tag>_lk_chain_pmc.json: hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the
   launches that did work (the frame-0 launch of every run exits immediately and is dropped).
   FETCH_SIZE is NOT doubled: the guide's x2 correction is calibrated for 16-B/lane coalesced streams only;
   this kernel gathers bytes, which the guide lists as uncalibrated — treat the number as a lower bound.
"""
import csv, glob, json, os, shutil, sys


def find(d, pat):
    g = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not g:
        raise SystemExit("no %s under %s" % (pat, d))
    return g[0]


def pmc_mean(d, counter, kernel, drop_small=True):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(find(d, "*counter_collection.csv")))
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    if drop_small and vals:
        thr = 0.05 * max(vals)
        vals = [v for v in vals if v > thr]
    return sum(vals) / max(len(vals), 1), len(vals)


def main():
    a = [x for i, x in enumerate(sys.argv[1:], 1) if not x.startswith("--") and sys.argv[i - 1] not in ("--seqs", "--kernel")]
    kernel = "k_lk_chain"
    if "--kernel" in sys.argv:
        kernel = sys.argv[sys.argv.index("--kernel") + 1]
    tag, stats = a[0], a[1]
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(find(stats, "*kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats.csv"))
    if len(a) >= 4:
        f, nf = pmc_mean(a[2], "FETCH_SIZE", kernel)
        w, nw = pmc_mean(a[3], "WRITE_SIZE", kernel)
        seqs = 32
        if "--seqs" in sys.argv:
            seqs = int(sys.argv[sys.argv.index("--seqs") + 1])
        out = {"kernel": kernel, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": [nf, nw],
               "sequences_per_launch": seqs, "hbm_bytes_per_launch": (f + w) * 1024,
               "note": "separate --pmc passes; units KB*1024; FETCH_SIZE uncorrected (byte-gather access, uncalibrated per MI355X_MICROARCH.md §HBM)"}
        json.dump(out, open(os.path.join(here, tag + "_lk_chain_pmc.json"), "w"), indent=1)
        print(out)


if __name__ == "__main__":
    main()
