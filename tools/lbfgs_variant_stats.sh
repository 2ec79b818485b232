# A/B of L-BFGS kernel variants under rocprofv3 (developer aid): tools/lbfgs_variant_stats.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for lib in "$@"; do
  export SQFA_HIP_LIBRARY=$(realpath $lib)   # read by sqfa_amd/_lib.py: the installed library stays untouched
  rm -rf gpurun_out/lbs; rocprofv3 --kernel-trace --stats -d gpurun_out/lbs -o s --output-format csv -- python3 tools/time_lbfgs.py > gpurun_out/lbs.log 2>&1
  echo "== $lib"
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/lbs/**/s_kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:12]:
        if "lb_solve" in row["Name"]: print(f'{row["Name"][:40]:40s} calls {row["Calls"]:>6s} avg_us {float(row["AverageNs"])/1e3:8.2f}')
PY
done
