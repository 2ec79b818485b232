"""The fused loss+grad evaluation (K0 + K1 + K2) at C=1000 for every padded size, float32 and float64: ms per
evaluation, nominal-flop fraction of the FP32 peak, sweeps (tools/time_pairs.py data).  python tools/all_sizes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import time_pairs as t
for m in (4, 8, 12, 16, 17, 20, 24, 32, 33, 40, 48, 64):
    t.run(1000, m, False, torch.float32, reps=6 if m <= 24 else 3)
for m in (4, 8, 12, 16, 17, 20, 24, 32, 33, 48, 64):
    t.run(1000 if m <= 33 else 300, m, False, torch.float64, reps=4 if m <= 20 else 2)
