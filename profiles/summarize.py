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


def sq_summary(tag, sq_dir, bench_json, extra_dirs=(), gui_dir=None, kernel="k_lk_chain"):
    """profiles/<tag>_lk_chain_sq.json from a `--pmc SQ_*` pass (<= 8 counters per pass: more "exceeds the capabilities of the
    hardware"), optional further passes (other counters) and an optional GRBM_GUI_ACTIVE pass for the clock the chip held;
    features per launch come from the bench line of the same configuration (mean_features_into_lk x sequences per launch)."""
    here = os.path.dirname(os.path.abspath(__file__))
    counters = {}
    for d in (sq_dir,) + tuple(extra_dirs):
        names = {r["Counter_Name"] for r in csv.DictReader(open(find(d, "*counter_collection.csv"))) if kernel in r["Kernel_Name"]}
        for n in sorted(names):
            counters[n], launches = pmc_mean(d, n, kernel)
    j = json.loads(open(bench_json).read().strip().splitlines()[-1])
    seqs = j["config"]["sequences_per_gpu"] // j["config"]["contexts_per_gpu"]
    feats = j["config"]["mean_features_into_lk"] * seqs
    out = {"kernel": j["roofline"]["kernel"], "sequences_per_launch": seqs, "features_per_launch": feats, "launches_averaged": launches,
           "counters_per_launch": counters, "valu_instructions_per_feature": counters["SQ_INSTS_VALU"] / feats,
           "newton_steps_per_feature": j["roofline"]["valu_flop"]["newton_steps_per_feature"],
           "level_visits_per_feature": j["roofline"]["valu_flop"]["level_visits_per_feature"],
           "kernel_avg_ms_unprofiled": j["roofline"]["kernel_avg_ms"],
           "valu_active_per_wave_slot_time": counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_WAVE_CYCLES"] if "SQ_ACTIVE_INST_VALU" in counters else None,
           "note": "rocprofv3 --pmc; SQ_WAVE_CYCLES / WAIT / ACTIVE are in quad-cycles summed over waves; valu_active_per_wave_slot_time = "
                   "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (multiply by the resident waves per SIMD — 6 at w = 21 since round 3, 4 before — "
                   "for a SIMD's view); valu_issue_frac (with the GRBM pass) = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x the launch's active "
                   "cycles): the share of the launch in which a SIMD issues a VALU instruction of this kernel"}
    if gui_dir:
        g, _ = pmc_mean(gui_dir, "GRBM_GUI_ACTIVE", kernel)
        kt = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(find(gui_dir, "*kernel_trace.csv"))) if kernel in r["Kernel_Name"]]
        kt = [t for t in kt if t > 0.05 * max(kt)]
        out["GRBM_GUI_ACTIVE_per_launch"] = g
        out["effective_clock_GHz"] = g / 8.0 / (sum(kt) / len(kt))      # the counter sums the 8 XCDs; duration in ns
        out["valu_issue_frac"] = counters["SQ_INSTS_VALU"] * 4.0 / (1024.0 * g / 8.0)   # passes are separate runs: a few per cent of launch-to-launch noise
    json.dump(out, open(os.path.join(here, tag + "_lk_chain_sq.json"), "w"), indent=1)
    print(out)


def main():
    if "--sq" in sys.argv:                  # summarize.py <tag> --sq <sq_dir> --bench <bench.json> [--extra dir ...] [--gui dir]
        a = sys.argv
        extra = [a[i + 1] for i, x in enumerate(a) if x == "--extra"]
        sq_summary(a[1], a[a.index("--sq") + 1], a[a.index("--bench") + 1], tuple(extra), a[a.index("--gui") + 1] if "--gui" in a else None)
        return
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
