import os, sys, shutil, subprocess
code = r'''
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np, time_pairs as t
from conftest import load_golden, rel_err
from sqfa_amd import _native
t.run(1000,16,False,torch.float32,reps=8)
G1 = load_golden("g1_airm_self.npz")
worst=[0,0]
for key in ("C10_m16","C37_m16","C16_m8","C20_m12"):
    S = torch.tensor(G1[key+"_S"], dtype=torch.float32, device="cuda").requires_grad_(True)
    C=S.shape[0]; P=C*(C-1)//2
    loss, fl = _native.PairwiseLoss.apply(S,1.0,1e-6,True,-1.0/P,(0,1),None); loss.backward()
    worst[0]=max(worst[0], abs(loss.item()-float(G1[key+"_loss_f64"]))/abs(float(G1[key+"_loss_f64"])))
    worst[1]=max(worst[1], rel_err(S.grad.cpu(), G1[key+"_grad_f64"]))
# baseline-like data accuracy vs f64 kernel
from jacobi_emulation import baseline_like
S64 = torch.tensor(baseline_like(120, 784, 16), device="cuda")
def fused(S):
    S=S.clone().requires_grad_(True); C=S.shape[0]; P=C*(C-1)//2
    l,_=_native.PairwiseLoss.apply(S,1.0,1e-6,True,-1.0/P,(0,1),None); l.backward(); return l.item(), S.grad
l64,g64=fused(S64); l32,g32=fused(S64.float())
print("golden worst loss rel %.2e grad rel %.2e | baseline-like f32 vs f64: loss %.2e grad %.2e" % (worst[0], worst[1], abs(l32-l64)/abs(l64), rel_err(g32.cpu(), g64.cpu())))
'''
for lib in sys.argv[1:]:
    if os.path.abspath(lib) != os.path.abspath("sqfa_amd/lib/libsqfa_hip.so"):
        shutil.copy(lib, "sqfa_amd/lib/libsqfa_hip.so")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(lib, "\n".join(out.stdout.strip().splitlines()[-2:]) if out.stdout.strip() else out.stderr[-800:], flush=True)
