"""Copy the summaries of tools/profile_round3.sh (gpurun_out/r3/prof) into profiles/ (tracked):
bench lines, kernel-stats CSVs, the PMC summaries, the text tables.    python tools/collect_profiles_r3.py"""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(ROOT, "gpurun_out/r3/prof"), os.path.join(ROOT, "profiles")
for w in ("c3", "c2", "c3-sqfa", "c4", "c5"):
    for kind in ("bench", "bench_profiled"):
        src = os.path.join(P, f"{w}_{kind}.json")
        if os.path.exists(src):
            lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
            open(os.path.join(OUT, f"r3_{w}_{kind}.json"), "w").write(lines[-1] + "\n")
    stats = glob.glob(os.path.join(P, f"{w}_stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(OUT, f"r3_{w}_kernel_stats.csv"))
for name in ("all_sizes", "shard_timings", "fit_benchmark", "gauss_pairs", "overlap_probe", "projection_kernel",
             "projection_dims", "clock_probe", "scale_probe", "gauss_sizes"):
    src = os.path.join(P, name + ".txt")
    if os.path.exists(src):
        txt = "\n".join(l for l in open(src).read().splitlines() if "amdgpu.ids" not in l)
        open(os.path.join(OUT, f"r3_{name}.txt"), "w").write(txt + "\n")
w = glob.glob(os.path.join(P, "pmc_w", "**", "*counter_collection.csv"), recursive=True)
f = glob.glob(os.path.join(P, "pmc_f", "**", "*counter_collection.csv"), recursive=True)
if w and f:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools/pmc_summary.py"), w[0], f[0], os.path.join(OUT, "r3_pmc_c3.json")], check=True)
if os.path.isdir(os.path.join(P, "pmc_pairs")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools/pmc_sq_summary.py"), os.path.join(P, "pmc_pairs"), os.path.join(OUT, "r3_pmc_pairs.json")], check=True)
# per-launch durations of the headline kernel in the profiled default command: mean (what `--stats` prints), median, and
# the mean without the launches that took more than twice the median (a pre-empted launch shifts the mean by percent)
import csv, json, statistics
tr = glob.glob(os.path.join(P, "c3_stats", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr[0]))
         if "pair_tile_kernel" in r["Kernel_Name"] and "float, 16, 4, 4" in r["Kernel_Name"]]
    med = statistics.median(d)
    kept = [x for x in d if x <= 2 * med]
    json.dump({"kernel": "pair_tile_kernel<PairCfg<float,16,4,4,8,4>>", "source": "rocprofv3 --kernel-trace of the default `python bench.py`",
               "launches": len(d), "mean_us": statistics.mean(d), "median_us": med, "min_us": min(d), "max_us": max(d),
               "launches_over_2x_median": len(d) - len(kept), "mean_us_without_those": statistics.mean(kept)},
              open(os.path.join(OUT, "r3_c3_kernel_trace_summary.json"), "w"), indent=1)
print(sorted(x for x in os.listdir(OUT) if x.startswith("r3_")))
