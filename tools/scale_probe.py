"""Developer aid: does the evaluation time scale with the number of pairs?  ms per evaluation and ns per pair at m=16 (and
m=32) for C = 500 ... 2000 (fixed costs, ramp and tail of the launch show up as a falling ns/pair)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import time_pairs as t
for m in (16, 32):
    for C in (250, 500, 708, 1000, 1414, 2000):
        t.run(C, m, False, torch.float32, reps=6)
