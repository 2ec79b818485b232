# per-kernel averages of tools/time_gauss_pairs.py under rocprofv3 (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/gk; rocprofv3 --kernel-trace --stats -d gpurun_out/gk -o s --output-format csv -- python3 tools/time_gauss_pairs.py > gpurun_out/gk.log 2>&1
grep "C=" gpurun_out/gk.log
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/gk/**/s_kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:12]:
        print(f'{row["Name"][:100]:100s} calls {row["Calls"]:>5s} avg_us {float(row["AverageNs"])/1e3:9.2f} total_ms {float(row["TotalDurationNs"])/1e6:8.1f}')
PY
