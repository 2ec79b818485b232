cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests/test_lbfgs.py -x -q 2>&1 | tail -1
python3 tools/time_lbfgs.py
rm -rf gpurun_out/lbs; rocprofv3 --kernel-trace --stats -d gpurun_out/lbs -o s --output-format csv -- python3 tools/time_lbfgs.py > gpurun_out/lbs.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/lbs/**/s_kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:12]:
        if "sqfa" in row["Name"]: print(f'{row["Name"][:60]:60s} calls {row["Calls"]:>6s} avg_us {float(row["AverageNs"])/1e3:8.2f}')
PY
