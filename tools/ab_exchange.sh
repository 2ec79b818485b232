#!/bin/bash
# Round 4 A/B: slot-exchange ordering of the sweeps (pair_kernel.hpp, exchange_slots) against the two-owner tournament.
# variants/build/r4_base.so = the library before the change (git worktree add /tmp/base 84e2754 && make -C /tmp/base/sqfa_amd/csrc &&
# cp /tmp/base/sqfa_amd/lib/libsqfa_hip.so variants/build/r4_base.so; variants/ is git-ignored); "-" = the installed library.
set -e
V=variants/build
O=gpurun_out/r4/exchange.txt
mkdir -p gpurun_out/r4
: > $O
run() { echo "== $1" | tee -a $O; shift; python tools/ab_pairs.py "$@" 2>&1 | tee -a $O; }
for spec in 1000:12:smsqfa 1000:16:smsqfa 1000:16:sqfa 1000:24:smsqfa 1000:32:smsqfa 1000:32:sqfa 300:48:smsqfa \
            1000:12:smsqfa:f64 1000:16:smsqfa:f64 1000:16:sqfa:f64; do
  run "$spec: tournament (base) / slot exchange / base / slot exchange" $spec $V/r4_base.so - $V/r4_base.so -
done
