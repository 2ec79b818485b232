#!/bin/bash
# Developer aid: per-kernel average durations (rocprofv3 --kernel-trace --stats) of tools/run_pairs_once.py for
# several pre-built library variants:  tools/ab_kernel_stats.sh "SPECS" lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
specs=$1; shift
for lib in "$@"; do
  export SQFA_HIP_LIBRARY=$(realpath $lib)   # read by sqfa_amd/_lib.py: the installed library stays untouched
  name=$(basename $lib .so)
  rm -rf gpurun_out/ab_$name
  SQFA_REPS=30 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$name -o s --output-format csv -- python3 tools/run_pairs_once.py $specs > gpurun_out/ab_$name.log 2>&1
  echo "== $name"
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/ab_$name/**/s_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n=row["Name"]
        if "sqfa::" in n:
            print(f'{n[:70]:70s} calls {row["Calls"]:>5s} avg_us {float(row["AverageNs"])/1e3:9.2f}')
PY
done
