# Kernel-time budget of a whole fit() (run on the GPU box): rocprofv3 kernel statistics of tools/fit_benchmark.py <config>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
cfg=${1:-c5}
rm -rf gpurun_out/fitstats; rocprofv3 --kernel-trace --stats -d gpurun_out/fitstats -o s --output-format csv -- python3 tools/fit_benchmark.py $cfg > gpurun_out/fitstats.log 2>&1
grep "fit()" gpurun_out/fitstats.log
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/fitstats/**/s_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot/1e6:.1f} ms")
    for row in rows[:28]:
        print(f'{row["Name"][:90]:90s} calls {row["Calls"]:>6s} avg_us {float(row["AverageNs"])/1e3:8.2f} total_ms {float(row["TotalDurationNs"])/1e6:8.1f}')
PY
