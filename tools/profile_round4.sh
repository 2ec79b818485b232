#!/bin/bash
# Round-4 profile set in ONE gpurun call (run from the repo root on the GPU box):
#   bash tools/profile_round4.sh        -> gpurun_out/r4/prof/*, then tools/collect_profiles_r4.py copies summaries to profiles/r4_*
# rocprofv3 is always given the python interpreter itself after "--" (no env / shell hop), counters in their own passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4/prof
mkdir -p $O
# 1. the default bench line, plain and under rocprofv3 --kernel-trace --stats
python3 bench.py > $O/c3_bench.json 2> $O/c3_bench.err
rocprofv3 --kernel-trace --stats -d $O/c3_stats -o c3 --output-format csv -- python3 bench.py --no-cpu-baseline > $O/c3_bench_profiled.json 2> $O/c3_prof.err
# 2. float64 line and the other workloads (pair stage + closure), plain
python3 bench.py --dtype f64 --no-cpu-baseline --steps 60 --warmup 10 > $O/c3_f64_bench.json 2> $O/c3_f64.err
for w in c2 c5; do python3 bench.py --workload $w --no-cpu-baseline --no-c4-pairs --no-c4-closure > $O/${w}_bench.json 2> $O/${w}.err; done
# 3. HBM traffic of the closure's kernels: two --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 tools/run_closure_once.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 tools/run_closure_once.py > $O/pmc_write.log 2>&1
# 4. all sizes, float32 / float64
python3 tools/all_sizes.py > $O/all_sizes.txt 2>&1
# 5. per-kernel split of one c4 pair shard, N = 1 and 8
for n in 1 8; do rocprofv3 --kernel-trace --stats -d $O/shard_c4_$n -o s --output-format csv -- python3 tools/run_shard_c4_once.py $n > $O/shard_c4_$n.log 2>&1; done
echo done > $O/DONE
