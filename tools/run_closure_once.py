"""Developer aid for PMC profiling: a few full c3 closures (projection + pair kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sqfa_amd import _lib
lib = _lib.load()
r = bench.closure_benchmark(1000, 784, 16, "smsqfa", torch.device("cuda:0"), 3, lib)
print(r["ms_per_closure"])
