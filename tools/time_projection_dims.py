"""Developer aid: sqfa_project_scatters alone for a list of D (K=16, C=1000): which part of the D=784 deficit is the
ragged last stripe and which the 64-byte misalignment of every other row?  python tools/time_projection_dims.py 768 784 800 832"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from sqfa_amd import _lib
lib = _lib.load()
def run(C, D, K, reps=30):
    Psi = torch.randn(C, D, D, device="cuda")
    F = torch.randn(K, D, device="cuda"); T = torch.empty(C, D, K, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(5): lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"C={C} D={D} K={K}: {ms:.3f} ms = {4.0*C*D*D/ms/1e6:.0f} GB/s   row {4*D} B = {4*D/128:.2f} lines, {D/64:.2f} stripes", flush=True)
for d in sys.argv[1:]:
    run(1000, int(d), 16)
