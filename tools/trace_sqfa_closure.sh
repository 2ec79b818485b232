cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/cl; rocprofv3 --kernel-trace -d gpurun_out/cl -o s --output-format csv -- python3 tools/run_closure_once.py 1000 784 16 sqfa 60 > gpurun_out/cl.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/cl/**/s_kernel_trace.csv", recursive=True):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "pair_tile" in r["Kernel_Name"]]
    print(len(d), [round(x) for x in d])
PY
python3 tools/run_closure_once.py 1000 784 16 sqfa 60
python3 tools/time_pairs.py 2>&1 | head -3
