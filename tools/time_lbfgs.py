"""Developer aid: time sqfa_lbfgs_push + sqfa_lbfgs_direction (device-resident L-BFGS state) per iteration.
python tools/time_lbfgs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqfa_amd._lbfgs import _History

for n, h in ((49152, 100), (12544, 100), (65536, 100), (49152, 20)):
    for dtype in (torch.float32, torch.float64):
        like = torch.zeros(n, dtype=dtype, device="cuda")
        hist = _History(h, like)
        g = torch.Generator(device="cuda").manual_seed(1)
        for _ in range(h + 5):
            s = torch.randn(n, dtype=dtype, device="cuda", generator=g); y = s + 0.1 * torch.randn(n, dtype=dtype, device="cuda", generator=g)
            hist.push(y, s)
        grad = torch.randn(n, dtype=dtype, device="cuda", generator=g)
        H = torch.tensor(0.7, dtype=dtype, device="cuda")
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tp = td = 0.0
        for it in range(33):
            e[0].record(); hist.push(y, s); e[1].record(); d = hist.direction(grad, H); e[2].record(); torch.cuda.synchronize()
            if it >= 3:
                tp += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2])
        print(f"n={n} h={h} {str(dtype)[6:]}: push {tp/30*1e3:.1f} us  direction {td/30*1e3:.1f} us", flush=True)
