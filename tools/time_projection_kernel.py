import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from sqfa_amd import _lib
lib = _lib.load()
def run(C, D, K, reps=30):
    Psi = torch.randn(C, D, D, device="cuda"); Psi = Psi + Psi.transpose(1, 2)
    F = torch.randn(K, D, device="cuda"); T = torch.empty(C, D, K, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(5): lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"C={C} D={D} K={K}: {ms:.3f} ms = {4.0*C*D*D/ms/1e6:.0f} GB/s", flush=True)
run(1000, 784, 16); run(1000, 2048, 32); run(100, 3072, 16); run(1000, 784, 4); run(1000, 1024, 16); run(1000, 512, 8)
