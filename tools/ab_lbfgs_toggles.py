"""Developer aid: the c1 fit with device-side optimizer state under each L-BFGS feature toggle (same final loss expected)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fit_benchmark as fb
import sqfa_amd._optim as o
import sqfa_amd._lbfgs as l
fb.warm_up()
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
def run(tag):
    print(tag, flush=True); fb.run(name)
run("host (default)")
o.HOST_SIDE_LBFGS_MAX_NUMEL_COMPACT = 0
run("device, all features")
l.CompactLBFGS.speculate_descent_test = False
run("device, no speculation")
o.DEFERRED_CLOSURE = False
run("device, no speculation, no deferred closure")
l._History.native = False
run("device, torch compact form (no native kernels)")
l._History.native = True
o.COMPACT_LBFGS = False
run("device, torch.optim.LBFGS")
