#!/bin/bash
# Round 4: re-tuning of the m=16 float32 row on the slot-exchange sweeps (variants built by tools/build_variant.sh)
set -e
V=variants/build
O=gpurun_out/r4/exchange_tune.txt
mkdir -p gpurun_out/r4
: >> $O
spec=${1:-1000:16:smsqfa}; shift || true
echo "== $spec: $*" | tee -a $O
libs=()
for n in "$@"; do if [ "$n" = "-" ]; then libs+=("-"); else libs+=("$V/$n.so"); fi; done
python tools/ab_pairs.py $spec "${libs[@]}" 2>&1 | tee -a $O
