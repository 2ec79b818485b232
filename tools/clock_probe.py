"""Developer aid: what shader clock does the chip hold while the pair kernel runs back to back?  Launches the fused
evaluation of a given size for a few seconds and samples rocm-smi (sclk, power) from a thread meanwhile.
    python tools/clock_probe.py 16 32 [seconds]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sqfa_amd import _native
from jacobi_emulation import baseline_like

def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=10).stdout
            sclk = re.findall(r"sclk clock level[^\(]*\((\d+)Mhz\)", txt)
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", txt)
            out.append((time.time(), int(sclk[0]) if sclk else None, float(pw[0]) if pw else None, txt if not sclk else ""))
        except Exception as e:  # noqa
            out.append((time.time(), None, None, str(e)))
        time.sleep(0.25)

def run(m, seconds):
    C = 1000
    S = torch.tensor(baseline_like(200, 784 if m <= 17 else 2048, m), dtype=torch.float32)
    S = S.repeat(5, 1, 1) * (1 + 0.3 * torch.rand(C, 1, 1))
    N = torch.randn(C, m, m) * 0.02
    S = (S + N @ N.transpose(1, 2)).cuda()
    P = C * (C - 1) // 2
    f = lambda: _native.hip_pair_backend(S, None, scale=1.0, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                         shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)
    f(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            f()
        torch.cuda.synchronize(); n += 20
    dt = (time.time() - t0) / n
    stop.set(); th.join()
    clk = [s[1] for s in out if s[1]]
    pw = [s[2] for s in out if s[2]]
    print(f"m={m}: {dt*1e3:.3f} ms per evaluation over {seconds} s; sclk samples (MHz) min/median/max "
          f"{min(clk) if clk else None}/{sorted(clk)[len(clk)//2] if clk else None}/{max(clk) if clk else None} ({len(clk)} samples); "
          f"power (W) median {sorted(pw)[len(pw)//2] if pw else None}", flush=True)
    if not clk and out:
        print("   rocm-smi said:", out[0][3][:400].replace("\n", " | "))

secs = float(sys.argv[-1]) if len(sys.argv) > 2 and "." in sys.argv[-1] else 4.0
for a in sys.argv[1:]:
    if "." not in a:
        run(int(a), secs)
