"""A/B of the class factor pass in the plain inner product (K0b of round 3) vs in the metric of the mean class (round 4,
class_factor_mean_kernel), same library, alternating per-call policies (sqfa_airm_options::mean_metric_policy):
pair kernel ms (HIP events), whole evaluation ms, sweeps per wave round, for
  * bench.py's BASELINE generator (classes scattered around a multiple of I): m = 12, 16, 17, 20, 24, 32, 33
  * classes that share a dominant covariance, Sigma_c = Sbar^1/2 (I + E_c)^2 Sbar^1/2, cond(Sbar) = 1e2 ... 1e4
  * unrelated ill-conditioned classes (random eigenvectors, log-uniform eigenvalues)
    python tools/ab_mean_metric.py [f32|f64]"""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sqfa_amd import _lib, _native

dtype = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
dev = torch.device("cuda")
lib = _lib.load()


def shared_structure(C, m, cond, spread, seed):
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(m, m, generator=g, dtype=torch.float64))
    ev = torch.exp(torch.linspace(0.0, float(torch.log(torch.tensor(cond))), m, dtype=torch.float64))
    root = (q * ev.sqrt()) @ q.T
    E = torch.randn(C, m, m, generator=g, dtype=torch.float64) * spread / m ** 0.5
    M = torch.eye(m, dtype=torch.float64) + 0.5 * (E + E.transpose(1, 2))
    return root @ (M @ M.transpose(1, 2)) @ root


def unrelated(C, m, cond, seed):
    g = torch.Generator().manual_seed(seed)
    Q, _ = torch.linalg.qr(torch.randn(C, m, m, generator=g, dtype=torch.float64))
    ev = torch.exp((torch.rand(C, m, generator=g, dtype=torch.float64) - 0.5) * torch.log(torch.tensor(cond)))
    S = (Q * ev[:, None, :]) @ Q.transpose(1, 2)
    return 0.5 * (S + S.transpose(1, 2))


def measure(S, scale, reps):
    C = S.shape[0]
    P = C * (C - 1) // 2
    f = lambda: _native.hip_pair_backend(S, None, scale=scale, eps=1e-6, sqrt_mode=True, weights=None, uniform_weight=-1.0 / P,
                                         shard=(0, 1), want_loss=True, want_grad=True, want_dist=False, want_eig=False)
    res = {}
    for mode in (0, 1):
        with _native.policies(mean_metric=mode):
            cnt = torch.zeros(2, dtype=torch.int64, device=dev)
            with _native.policies(sweep_counter=cnt):
                out = f(); torch.cuda.synchronize()
            c = cnt.tolist()
            res[mode] = {"sweeps": c[0] / max(c[1], 1), "loss": out["loss"].item(), "grad": out["gradA"].double().cpu(), "k": [], "e": []}
    for rnd in range(4):                                   # alternate the two policies
        for mode in (0, 1):
            with _native.policies(mean_metric=mode):
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                lib.sqfa_airm_profile(1)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    f()
                b.record(); torch.cuda.synchronize()
                lib.sqfa_airm_profile(0)
                ms, n = ctypes.c_double(0), ctypes.c_int(0)
                lib.sqfa_airm_profile_read(ctypes.byref(ms), ctypes.byref(n))
                res[mode]["k"].append(ms.value / max(n.value, 1)); res[mode]["e"].append(a.elapsed_time(b) / reps)
    return res


def line(name, S, scale=1.0):
    S = S.to(dtype).to(dev).contiguous()
    m = S.shape[1]
    r = measure(S, scale, 20 if m <= 20 else 6)
    p, q = r[0], r[1]
    k0, k1, e0, e1 = (statistics.median(v) for v in (p["k"], q["k"], p["e"], q["e"]))
    dg = (torch.linalg.norm(q["grad"] - p["grad"]) / torch.linalg.norm(p["grad"])).item()
    print(f"{name:44s} m={m:2d}  plain: kernel {k0:7.4f} eval {e0:7.4f} ms sweeps {p['sweeps']:.2f} | mean metric: kernel {k1:7.4f} eval {e1:7.4f} ms "
          f"sweeps {q['sweeps']:.2f} | eval {100 * (e1 / e0 - 1):+5.1f} %  loss diff {abs(q['loss'] - p['loss']) / abs(p['loss']):.1e} grad diff {dg:.1e}", flush=True)


print(f"dtype {dtype}; C=1000 unless stated")
for K, model in ((12, "smsqfa"), (16, "smsqfa"), (16, "sqfa"), (20, "smsqfa"), (24, "smsqfa"), (32, "smsqfa"), (32, "sqfa")):
    S, scale = bench.make_feature_scatters(1000, 784 if K <= 16 else 2048, K, model, dev, torch.float64)
    line(f"BASELINE generator K={K} {model}", S, scale)
for m in (16, 17, 32):
    for cond in (1e2, 1e4):
        for spread in (0.2, 0.6):
            line(f"shared Sbar cond {cond:g} spread {spread}", shared_structure(1000 if m < 32 else 600, m, cond, spread, m))
for m in (16, 32):
    line("unrelated classes, cond 1e3", unrelated(600, m, 1e3, m))
