#!/bin/bash
# Round profile set (run on the GPU box via gpurun): kernel-trace stats of the default bench command,
# HBM traffic counters of a c3 closure (separate FETCH_SIZE / WRITE_SIZE passes) and the SQ counter
# passes of the pair kernel.  Outputs under gpurun_out/r2/prof; summaries are copied to profiles/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
P=gpurun_out/r2/prof; mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P/bench -o b --output-format csv -- python3 bench.py > $P/bench_profiled.json 2> $P/bench_profiled.err
python3 bench.py > $P/bench.json 2> $P/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/pmc_w -o w --output-format csv -- python3 tools/run_closure_once.py > $P/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/pmc_f -o f --output-format csv -- python3 tools/run_closure_once.py > $P/pmc_f.log 2>&1
PMC_NAME=prof/pmc_pairs bash tools/pmc_pairs.sh > $P/pmc_pairs.log 2>&1
find $P -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
tail -c 400 $P/bench.json
