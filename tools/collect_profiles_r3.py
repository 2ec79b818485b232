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
for name in ("all_sizes", "shard_timings", "fit_benchmark", "gauss_pairs", "overlap_probe", "projection_kernel"):
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
print(sorted(x for x in os.listdir(OUT) if x.startswith("r3_")))
