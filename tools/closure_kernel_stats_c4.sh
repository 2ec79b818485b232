cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for spec in "1000 2048 32 smsqfa 10"; do
  rm -rf gpurun_out/cl; rocprofv3 --kernel-trace --stats -d gpurun_out/cl -o s --output-format csv -- python3 tools/run_closure_once.py $spec > gpurun_out/cl.log 2>&1
  echo "== $spec: $(tail -1 gpurun_out/cl.log)"
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/cl/**/s_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for row in rows[:16]:
        print(f'{row["Name"][:80]:80s} calls {row["Calls"]:>5s} avg_us {float(row["AverageNs"])/1e3:9.2f}')
PY
done
