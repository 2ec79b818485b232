"""Developer aid: fit() wall-clock with host-side vs device-side optimizer state for the small configurations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_benchmark as fb
import sqfa_amd._optim as o
fb.warm_up()
for rep in range(2):
    for limit in (8192, 0):
        o.HOST_SIDE_LBFGS_MAX_NUMEL_COMPACT = limit
        print("host limit", limit, flush=True)
        for name in ("c1", "c2", "c2s"):
            fb.run(name)
