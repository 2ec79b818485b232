"""Copy the summaries of tools/profile_round4.sh (gpurun_out/r4/prof) into profiles/ (tracked): bench lines, the
kernel-stats CSV of the default command, the PMC traffic summary, the all-sizes table and the per-kernel split of one c4
pair shard.    python tools/collect_profiles_r4.py"""
import csv, glob, json, os, shutil, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(ROOT, "gpurun_out/r4/prof"), os.path.join(ROOT, "profiles")
for name in ("c3_bench", "c3_bench_profiled", "c3_f64_bench", "c2_bench", "c5_bench"):
    src = os.path.join(P, name + ".json")
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(OUT, f"r4_{name}.json"), "w").write(lines[-1] + "\n")
stats = glob.glob(os.path.join(P, "c3_stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(OUT, "r4_c3_kernel_stats.csv"))
src = os.path.join(P, "all_sizes.txt")
if os.path.exists(src):
    open(os.path.join(OUT, "r4_all_sizes.txt"), "w").write("\n".join(l for l in open(src).read().splitlines() if "amdgpu.ids" not in l) + "\n")
w = glob.glob(os.path.join(P, "pmc_write", "**", "*counter_collection.csv"), recursive=True)
f = glob.glob(os.path.join(P, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)
if w and f:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools/pmc_summary.py"), w[0], f[0], os.path.join(OUT, "r4_pmc_c3.json")], check=True)
tr = glob.glob(os.path.join(P, "c3_stats", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr[0]))
         if "pair_tile_kernel" in r["Kernel_Name"] and "float, 16, 4, 4" in r["Kernel_Name"]]
    if d:
        med = statistics.median(d)
        kept = [x for x in d if x <= 2 * med]
        json.dump({"kernel": "pair_tile_kernel<PairCfg<float,16,4,4,8,4>>", "source": "rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline`",
                   "launches": len(d), "mean_us": statistics.mean(d), "median_us": med, "min_us": min(d), "max_us": max(d),
                   "launches_over_2x_median": len(d) - len(kept), "mean_us_without_those": statistics.mean(kept)},
                  open(os.path.join(OUT, "r4_c3_kernel_trace_summary.json"), "w"), indent=1)
# per-kernel split of one c4 pair shard (tools/run_shard_c4_once.py under rocprofv3 --stats)
rows = []
for n in (1, 8):
    st = glob.glob(os.path.join(P, f"shard_c4_{n}", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        for r in csv.DictReader(open(st[0])):
            name = r["Name"].split("(")[0]
            for key in ("cholesky_kernel", "class_factor", "pair_tile_kernel", "finalize_kernel"):
                if key in name:
                    rows.append(f"  shard 0/{n}: {key:18s} {float(r['AverageNs']) / 1e3:9.1f} us average over {r['Calls']} launches")
if rows:
    path = os.path.join(OUT, "r4_shard_timings_c4.txt")
    txt = open(path).read() if os.path.exists(path) else ""
    if "per-kernel split" not in txt:
        open(path, "a").write("\nper-kernel split of the c4 pair stage (rocprofv3 --kernel-trace --stats of tools/run_shard_c4_once.py; K0, K0b and K2 are replicated on every rank):\n" + "\n".join(rows) + "\n")
print(sorted(x for x in os.listdir(OUT) if x.startswith("r4_")))
