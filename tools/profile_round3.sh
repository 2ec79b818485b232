#!/bin/bash
# Round-3 profile set (run on the GPU box via gpurun; python3 directly after `--`, no exec hops): everything DESIGN.md
# section 6 quotes, under gpurun_out/r3/prof; tools/collect_profiles_r3.py copies the summaries into profiles/.
#   1 default bench line + rocprofv3 --kernel-trace --stats of the same command
#   2 the other BASELINE workloads through bench.py (line + kernel stats each)
#   3 HBM traffic counters of a c3 closure (separate FETCH_SIZE / WRITE_SIZE passes)
#   4 SQ counter passes of the pair kernel (m = 16, 17, 32, 33)
#   5 all matrix sizes (float32 / float64), shard timings, fit() wall-clock, Gaussian pair kernel, overlap probe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
P=gpurun_out/r3/prof; mkdir -p $P
echo "[1] default bench"; date
python3 bench.py > $P/c3_bench.json 2> $P/c3_bench.err
rocprofv3 --kernel-trace --stats -d $P/c3_stats -o b --output-format csv -- python3 bench.py > $P/c3_bench_profiled.json 2> $P/c3_bench_profiled.err
echo "[2] other workloads"; date
for w in c2 c3-sqfa c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline --no-c4-pairs --no-c4-closure > $P/${w}_bench.json 2> $P/${w}_bench.err
  rocprofv3 --kernel-trace --stats -d $P/${w}_stats -o b --output-format csv -- python3 bench.py --workload $w --no-cpu-baseline --no-c4-pairs --no-c4-closure > $P/${w}_bench_profiled.json 2> $P/${w}_bench_profiled.err
done
echo "[3] HBM traffic"; date
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/pmc_w -o w --output-format csv -- python3 tools/run_closure_once.py > $P/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/pmc_f -o f --output-format csv -- python3 tools/run_closure_once.py > $P/pmc_f.log 2>&1
echo "[4] SQ counters"; date
SPECS="1000:16:smsqfa 1000:16:sqfa 1000:32:smsqfa 1000:32:sqfa"
Q=$P/pmc_pairs; mkdir -p $Q
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $Q/a -o a --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $Q/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d $Q/b -o b --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $Q/b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $Q/c -o c --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $Q/c.log 2>&1
rocprofv3 --kernel-trace --stats -d $Q/s -o s --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $Q/s.log 2>&1
echo "[5] tables"; date
python3 tools/all_sizes.py > $P/all_sizes.txt 2>&1
python3 tools/time_shard.py > $P/shard_timings.txt 2>&1
python3 tools/fit_benchmark.py c1 c2 c2s c5 c3 > $P/fit_benchmark.txt 2>&1
python3 tools/time_gauss_pairs.py > $P/gauss_pairs.txt 2>&1
python3 tools/time_gauss_sizes.py > $P/gauss_sizes.txt 2>&1
python3 tools/overlap_probe.py c3 > $P/overlap_probe.txt 2>&1
python3 tools/overlap_probe.py c4 >> $P/overlap_probe.txt 2>&1
python3 tools/time_projection_kernel.py > $P/projection_kernel.txt 2>&1
python3 tools/time_projection_dims.py 736 768 784 800 816 832 896 960 1024 784 > $P/projection_dims.txt 2>&1
python3 tools/clock_probe.py 16 17 32 33 8 > $P/clock_probe.txt 2>&1
python3 tools/scale_probe.py > $P/scale_probe.txt 2>&1
date; echo done
