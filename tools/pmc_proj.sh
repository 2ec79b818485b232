# PMC passes over the projection kernel for the installed library and variants: bash tools/pmc_proj.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/r3/pmc_proj; mkdir -p $OUT
for lib in "$@"; do
  name=$(basename $lib .so)
  export SQFA_HIP_LIBRARY=$(realpath $lib)
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum -d $OUT/${name}_a -o a --output-format csv -- python3 tools/run_proj_once.py 768 784 > $OUT/${name}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RD_UNCACHED_32B_sum -d $OUT/${name}_b -o b --output-format csv -- python3 tools/run_proj_once.py 768 784 > $OUT/${name}_b.log 2>&1
  tail -2 $OUT/${name}_a.log $OUT/${name}_b.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "project_kernel" in row["Kernel_Name"]:
            acc[(row["Grid_Size"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    print(f.split("pmc_proj/")[1])
    for (g, c), v in sorted(acc.items()):
        print(f"   grid {g:>10s} {c:32s} {sum(v[2:])/max(len(v[2:]),1):16.0f}  (n={len(v)})")
PY
