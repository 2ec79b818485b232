"""Minimal driver for rocprofv3 passes over sqfa_project_scatters: python3 tools/run_proj_once.py D [D ...] (C=1000, K=16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from sqfa_amd import _lib
lib = _lib.load()
for d in sys.argv[1:]:
    D = int(d); C, K = 1000, 16
    Psi = torch.randn(C, D, D, device="cuda"); F = torch.randn(K, D, device="cuda"); T = torch.empty(C, D, K, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(int(os.environ.get("SQFA_REPS", "6"))):
        lib.sqfa_project_scatters(F.data_ptr(), K, D, Psi.data_ptr(), C, 0, T.data_ptr(), st)
    torch.cuda.synchronize()
    del Psi, F, T
