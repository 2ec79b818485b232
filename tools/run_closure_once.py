"""Developer aid for PMC / kernel-trace profiling: a few full closures (projection + pair kernel).
    python tools/run_closure_once.py [C D K model [steps]]     (default: the c3 closure, 3 steps)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sqfa_amd import _lib
lib = _lib.load()
a = sys.argv[1:]
C, D, K = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1000, 784, 16)
model = a[3] if len(a) >= 4 else "smsqfa"
steps = int(a[4]) if len(a) >= 5 else 3
r = bench.closure_benchmark(C, D, K, model, torch.device("cuda:0"), steps, lib)
print(r["ms_per_closure"])
