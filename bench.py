#!/usr/bin/env python3
"""Benchmark of the SQFA hot path on MI355X: SPD pairwise-distance loss+grad evaluations/s.

One *step* = one M1 evaluation: given the C feature scatters S (C,m,m)
resident in HBM, produce the scalar loss -mean_{i>j} d(S_i,S_j) and dloss/dS.
Workload (default) = BASELINE.json configs[2] "c3": C=1000 classes, n_dim=784, n_filters=16,
float32, SecondMomentsSQFA (m=16), synthetic Gaussians of SURVEY.md 8(d).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: spawns one rank per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank holds the same S, evaluates the tile shard (bi+bj) % N == rank and the
partial loss / flags / gradient are summed with one RCCL all-reduce per step ("strong" scaling:
the problem is fixed, `value` is whole-job evaluations per second).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (pair tile kernel; its duration is measured live with
HIP events on the launch stream in a short pass right AFTER the timed region, which itself runs with profiling
off), "cpu_baseline" (the oracle's torch-CPU port of the reference op sequence: ONE timed full-size evaluation on every
core the process is entitled to, plus a one-thread figure on a bounded sample), "prewarm" (what ran before the headline's own warm-up steps), "c3_sqfa" (the same configuration with the
reference's default model, sqfa.model.SQFA: m = K+1 = 17, own roofline), "scaling_c4_pairs" (BASELINE config 4's
pair stage, m=32, same sharding) and "scaling_c4_closure" (config 4 end to end at every N: class-sharded
(C,2048,2048) statistics, projection + all-gather + pair shard + all-reduce + backward + gradient all-reduce
as four captured graphs around the three collectives); N=1 adds "closure" (metric M2 on the headline workload).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector == f32 MFMA dense peak
FP64_PEAK_TFLOPS = 78.6    # MI355X public specification (FP64 vector = half the FP32 vector rate); the guide's table has no f64 row
HBM_PEAK_GBS = 8000.0
PMC_FILE = "r4_pmc_c3.json"   # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary the `traffic` fields quote

WORKLOADS = {
    # name: (C, D, K, model)   model: "smsqfa" -> m=K, "sqfa" -> m=K+1 (Calvo-Oller embedding, scale 1/2)
    "c2": (100, 784, 8, "smsqfa"),
    "c3": (1000, 784, 16, "smsqfa"),
    "c3-sqfa": (1000, 784, 16, "sqfa"),
    "c4": (1000, 2048, 32, "smsqfa"),
    "c5": (100, 3072, 16, "sqfa"),
}


def make_feature_scatters(C, D, K, model, device, dtype=torch.float32, seed=1234, feature_noise=0.01):
    """Synthetic classes of SURVEY.md 8(d): Sigma_c = A_c A_c^T + 0.05 I, A_c ~ N(0,1)^{D x R}/sqrt(R),
    R = min(D,128); mu_c ~ 0.1 N(0,1); filters F ~ N(0,1)^{K x D} (seed 7) on the sphere.
    Returns the kernel input S (C,m,m): F Psi_c F^T + noise*I (smsqfa) or the Calvo-Oller
    embedding of (F mu_c, F Sigma_c F^T + noise*I) (sqfa).  The (C,D,D) tensor is never formed:
    F Sigma_c F^T = (F A_c)(F A_c)^T + 0.05 F F^T."""
    R = min(D, 128)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    F = torch.randn(K, D, generator=torch.Generator(device="cpu").manual_seed(7), dtype=torch.float64)
    F = F / F.norm(dim=1, keepdim=True)
    F = F.to(device)
    eye = torch.eye(K, dtype=torch.float64, device=device)
    cov = torch.empty(C, K, K, dtype=torch.float64, device=device)
    fmu = torch.empty(C, K, dtype=torch.float64, device=device)
    chunk = 50
    for c0 in range(0, C, chunk):
        n = min(chunk, C - c0)
        A = torch.randn(n, D, R, generator=gen, dtype=torch.float32).to(device, torch.float64) / R ** 0.5
        mu = 0.1 * torch.randn(n, D, generator=gen, dtype=torch.float32).to(device, torch.float64)
        FA = F[None] @ A
        cov[c0:c0 + n] = FA @ FA.transpose(1, 2) + 0.05 * (F @ F.T)[None] + feature_noise * eye[None]
        fmu[c0:c0 + n] = mu @ F.T
    if model == "smsqfa":
        S = cov + fmu[:, :, None] * fmu[:, None, :]
        scale = 1.0
    else:
        from sqfa_amd.distances import embed_gaussian
        S = embed_gaussian({"means": fmu, "covariances": cov})
        scale = 0.5
    return S.to(dtype).contiguous(), scale


def capture_closure(closure, params):
    """(replay, captured): `closure` (forward + backward into params' .grad) captured once in a HIP graph; if the capture
    fails the eager closure itself is returned and `captured` is False."""
    for p in params:
        p.grad = None
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            closure()
        return g.replay, True
    except Exception as err:
        sys.stderr.write(f"bench.py: HIP graph capture of the closure failed ({err}); timing eager closures\n")
        torch.cuda.synchronize()
        return closure, False


def closure_benchmark(C, D, K, model_name, device, steps, lib, with_fit=False):
    """Secondary measurement (metric M2 of SURVEY.md 8d, N=1 only): one full closure = projection
    of the (C,D,D) scatters through the current filters (streaming HIP kernel), fused pairwise
    loss+grad, backward to the raw filter parameter.  Reports closures/s and the HBM roofline of
    the projection kernel (algorithmic bytes 4*C*D^2 per launch)."""
    import ctypes
    import sqfa_amd
    gen = torch.Generator(device="cpu").manual_seed(1234)
    R = min(D, 128)
    cov = torch.empty(C, D, D, device=device)
    mu = torch.empty(C, D, device=device)
    for c0 in range(0, C, 50):
        n = min(50, C - c0)
        A = (torch.randn(n, D, R, generator=gen) / R ** 0.5).to(device)
        cov[c0:c0 + n] = A @ A.transpose(1, 2) + 0.05 * torch.eye(D, device=device)
        mu[c0:c0 + n] = 0.1 * torch.randn(n, D, generator=gen).to(device)
    torch.manual_seed(7)
    if model_name == "sqfa":
        model = sqfa_amd.model.SQFA(n_dim=D, n_filters=K, feature_noise=0.01).to(device)
        prepared = model._prepare_statistics({"means": mu, "covariances": cov})
    else:
        model = sqfa_amd.model.SecondMomentsSQFA(n_dim=D, n_filters=K, feature_noise=0.01).to(device)
        cov += mu[:, :, None] * mu[:, None, :]
        prepared = model._prepare_statistics(cov)

    def closure():
        model.zero_grad()
        loss, flags = model._fused_closure_loss(prepared)
        loss.backward()
        return loss

    for _ in range(max(3, min(25, steps))):   # the statistics were just generated on an idle GPU: reach the sustained clock
        closure()
    torch.cuda.synchronize()
    # the timed closures are replayed from a captured HIP graph -- what fit() does after three eager closures
    # (sqfa_amd/_optim.py) -- so that the number is the GPU's, not the box's host (eager: ~10 launches + the autograd
    # engine per closure; the c2-sized closure read 0.20-0.32 ms depending on the box)
    replay, graphed = capture_closure(closure, [model.parametrizations.filters.original])
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):                    # timed region: no profiling events
        replay()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loss = closure()
    lib.sqfa_airm_profile(1)                  # separate short pass with HIP events around the kernels (plain launches)
    for _ in range(max(5, min(30, steps))):
        closure()
    torch.cuda.synchronize()
    lib.sqfa_airm_profile(0)
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_project_profile_read(ctypes.byref(ms), ctypes.byref(n))
    ms2, n2 = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_airm_profile_read(ctypes.byref(ms2), ctypes.byref(n2))
    k_ms = ms.value / max(n.value, 1)
    # the projection streams the block-triangular packed statistics when _prepare_statistics packed them (symmetric float32,
    # K <= 16, many classes: sqfa_project_scatters_packed); its roofline is quoted against the bytes THAT kernel must read
    from sqfa_amd import _native
    scat = prepared["covariances"] if isinstance(prepared, dict) else prepared
    packed = _native.packed_for(scat, K)
    full_bytes = 4.0 * C * D * D
    byts = 4.0 * packed.numel() if packed is not None else full_bytes
    gbs = byts / (k_ms * 1e-3) / 1e9

    result = {
        "value": steps / elapsed,
        "unit": "closures/s",
        "ms_per_closure": elapsed / steps * 1e3,
        "what": f"projection F Psi_c F^T of (C={C},D={D},D) scatters + fused pairwise loss+grad + backward to the raw filters ({model_name}, K={K})",
        "replayed_from_hip_graph": graphed,
        "pair_kernel_ms": ms2.value / max(n2.value, 1),
        "loss": loss.item(),
        "roofline": {
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": None, "kernel": "project_packed_kernel" if packed is not None else "project_kernel", "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": byts, "packed_statistics": packed is not None, "full_tensor_bytes": full_bytes,
            "full_tensor_equivalent_GBs": full_bytes / (k_ms * 1e-3) / 1e9,
        },
    }
    if not with_fit:
        return result

    # tertiary measurement (metric M3, --fit): fit() wall-clock to the reference stopping rule from
    # the fit_pca initialisation, on the same statistics.  Opt-in, so that the kernel statistics of
    # the default command only contain launches on the benchmark's own input.
    stats_for_fit = prepared
    evals = [0]
    fused = model._fused_closure_loss
    replay = torch.cuda.CUDAGraph.replay

    def counted_fused(p):
        evals[0] += 1
        return fused(p)

    def counted_replay(self):
        evals[0] += 1
        return replay(self)

    model._fused_closure_loss = counted_fused
    torch.cuda.CUDAGraph.replay = counted_replay
    try:
        with contextlib.redirect_stdout(sys.stderr):
            model.fit_pca(data_statistics=stats_for_fit)
            model.fit(data_statistics=stats_for_fit, max_epochs=2, show_progress=False)  # untimed: first-use costs
        model.fit_pca(data_statistics=stats_for_fit)
        evals[0] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):  # the stopping message must not join the JSON line
            fit_loss, _ = model.fit(data_statistics=stats_for_fit, max_epochs=300, show_progress=False, return_loss=True)
        torch.cuda.synchronize()
        fit_seconds = time.perf_counter() - t0
    finally:
        torch.cuda.CUDAGraph.replay = replay
        model._fused_closure_loss = fused
    fit = {
        "seconds": fit_seconds, "epochs": int(len(fit_loss)), "closures": evals[0],
        "ms_per_closure": fit_seconds / max(evals[0], 1) * 1e3, "final_loss": float(fit_loss[-1]),
        "what": "model.fit() from the fit_pca initialisation to the reference's stopping rule (|dloss| < 1e-6 for 3 epochs), float32",
    }
    result["fit"] = fit
    return result


def make_class_shard_statistics(C, D, lo, hi, device, seed=1234):
    """Second-moment matrices Psi_c = Sigma_c + mu_c mu_c^T of classes [lo, hi) of the SURVEY.md 8(d) synthetic set,
    generated ON the device that will hold them (float32).  Each chunk of 50 classes has its own generator seed, so a
    rank produces exactly its own class shard without drawing the other ranks' classes, and the union over ranks
    is the same data set for every N."""
    R = min(D, 128)
    out = torch.empty(hi - lo, D, D, device=device)
    eye = torch.eye(D, device=device)
    for k in range(lo // 50, (hi + 49) // 50):
        gen = torch.Generator(device="cpu").manual_seed(seed + 1000003 * (k + 1))
        c0, c1 = 50 * k, min(50 * k + 50, C)
        A = (torch.randn(c1 - c0, D, R, generator=gen) / R ** 0.5)
        mu = 0.1 * torch.randn(c1 - c0, D, generator=gen)
        a, b = max(lo, c0), min(hi, c1)
        if a >= b:
            continue
        A, mu = A[a - c0:b - c0].to(device), mu[a - c0:b - c0].to(device)
        out[a - lo:b - lo] = A @ A.transpose(1, 2) + 0.05 * eye + mu[:, :, None] * mu[:, None, :]
    return out


def c4_closure_leg(world, rank, device, pair_shard, steps, lib, fence, dist):
    """BASELINE config 4 as written: C=1000 classes, n_dim=2048, n_filters=32, the (C,D,D) statistics
    CLASS-sharded over the ranks (each rank generates and keeps only its own C/N classes: 16.8 GB / N), one full
    closure = sphere -> projection of the local classes -> all-gather of the (C_r,m,m) slices -> pair-tile shard ->
    all-reduce [loss, flags, dL/dS] -> backward of the local classes -> all-reduce of dL/dF, run as four captured
    graphs around the three collectives (sqfa_amd._optim.ShardedClosure).  N=1: the plain single-GPU closure.
    Reports closures/s (max time over ranks) and the per-rank projection kernel against the HBM roofline."""
    import ctypes
    import sqfa_amd
    from sqfa_amd._optim import ShardedClosure
    from sqfa_amd.parallel import ClassShard
    C = int(os.environ.get("SQFA_BENCH_C4_CLASSES", "1000"))   # rehearsals on a shared GPU shrink the class count
    _, D, K, _ = WORKLOADS["c4"]
    lo, hi = rank * C // world, (rank + 1) * C // world
    t_gen = time.perf_counter()
    local = make_class_shard_statistics(C, D, lo, hi, device)
    torch.manual_seed(7)                                        # identical initial filters on every rank
    model = sqfa_amd.model.SecondMomentsSQFA(n_dim=D, n_filters=K, feature_noise=0.01).to(device)
    if world > 1:
        model.pair_shard = pair_shard
        model.class_shard = ClassShard(hi - lo)
    prepared = model._prepare_statistics(local)
    if world > 1:
        assert ShardedClosure.supported(model, prepared), "the graphs-around-collectives closure does not apply"
        sharded = ShardedClosure(model, prepared)

        def closure(eager=False):
            packed, _grad = sharded.run(eager)
            return packed
    else:
        def closure(eager=False):
            model.zero_grad()
            loss, flags = model._fused_closure_loss(prepared)
            loss.backward()
            return torch.cat([loss.detach().reshape(1), flags.to(loss.dtype)])
    for _ in range(6):          # three eager evaluations, the capture, two replays
        packed = closure()
    fence()
    timed, graphed_single = closure, False
    if world == 1:              # one GPU: replay the closure from a captured graph as fit() does (N > 1: ShardedClosure does)
        replay, graphed_single = capture_closure(closure, [model.parametrizations.filters.original])
        if graphed_single:
            timed = replay
        for _ in range(2):
            timed()
        fence()
    gen_seconds = time.perf_counter() - t_gen
    t0 = time.perf_counter()
    for _ in range(steps):
        timed()
    fence()
    seconds = time.perf_counter() - t0
    packed = closure() if world == 1 else packed
    lib.sqfa_airm_profile(1)    # separate profiled pass (HIP events around project_kernel / pair_tile_kernel),
    for _ in range(max(3, min(10, steps))):   # as plain launches: a graph replay does not repeat the event records
        closure(True)
    fence()
    lib.sqfa_airm_profile(0)
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_project_profile_read(ctypes.byref(ms), ctypes.byref(n))
    ms2, n2 = ctypes.c_double(0), ctypes.c_int(0)
    lib.sqfa_airm_profile_read(ctypes.byref(ms2), ctypes.byref(n2))
    proj_ms, pair_ms = ms.value / max(n.value, 1), ms2.value / max(n2.value, 1)
    head = packed[:3].double()
    if world > 1:
        both = torch.tensor([seconds, proj_ms, pair_ms], dtype=torch.float64, device=device)
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        seconds, proj_ms, pair_ms = both.tolist()
    loss, n_nan, n_inf = head.tolist()
    assert n_nan == 0 and n_inf == 0, "non-finite distances in the c4 closure"
    byts = 4.0 * (hi - lo) * D * D
    gbs = byts / (proj_ms * 1e-3) / 1e9 if proj_ms > 0 else 0.0
    return {
        "workload": f"c4 closure: C={C} classes, n_dim={D}, n_filters={K} (m={K}), statistics class-sharded "
                    f"({hi - lo} classes = {byts / 1e9:.2f} GB on this rank), pair tiles sharded",
        "value": steps / seconds, "unit": "closures/s", "n_gpus": world, "steps": steps, "warmup": 6,
        "ms_per_closure": seconds / steps * 1e3, "scaling": "strong", "loss": loss,
        "collectives_per_closure": 3 if world > 1 else 0,
        "graphs_per_closure": (4 if sharded.state == "on" else 0) if world > 1 else (1 if graphed_single else 0),
        "pair_kernel_ms": pair_ms,
        "projection": {"bound": "hbm", "kernel": "project_kernel", "kernel_ms": proj_ms, "achieved": gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s per rank", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts},
        "setup_seconds": gen_seconds,
    }


def pmc_traffic(kernel, workload, dtype):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (collected in separate
    FETCH_SIZE / WRITE_SIZE passes as MI355X_MICROARCH.md prescribes; see the file's `correction`).
    Only quoted for the workload it was measured on."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    if workload != "c3" or dtype != "f32" or not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)["kernels"].get(kernel, {}).get("hbm_bytes_per_launch")


def _entitled_threads():
    """CPUs this process may use: the cgroup quota when there is one, not every hardware thread of the host."""
    threads = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return threads


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(S_cpu, scale, C_full):
    """The oracle's torch-CPU port of the reference algorithm (all ordered pairs, eigh whitening, batched eigvalsh,
    autograd: oracle/reference_path.py) timed on this box's host cores, SURVEY.md 8(d):
      * ONE evaluation of the FULL workload (all C classes) on every core the process is entitled to -- `value`;
      * a one-thread figure on a bounded sample (the first C_s classes; cost ~ C^2, scaled and labelled as such)."""
    from oracle import reference_path
    threads = _entitled_threads()
    torch.set_num_threads(threads)
    m = S_cpu.shape[-1]
    t0 = time.perf_counter()
    reference_path.pairwise_loss_and_grad(S_cpu[:60].clone(), scale=scale)  # warm-up (MKL init, thread pool)
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    reference_path.pairwise_loss_and_grad(S_cpu.clone(), scale=scale)       # the full workload, once
    t_full = time.perf_counter() - t0
    # one thread: a sample sized for ~15-20 s (the full evaluation would take minutes on one core)
    C_s = min(C_full, 300 if m <= 17 else 150)
    torch.set_num_threads(1)
    sample = S_cpu[:C_s].clone()
    reference_path.pairwise_loss_and_grad(sample[:40], scale=scale)
    t0 = time.perf_counter()
    reference_path.pairwise_loss_and_grad(sample, scale=scale)
    t_one = time.perf_counter() - t0
    torch.set_num_threads(threads)
    est_one = t_one * (C_full / C_s) ** 2
    return {
        "value": 1.0 / t_full,
        "unit": "evals/s",
        "cores": threads,
        "kind": "port",
        "full_size": True,
        "cpu_model": _cpu_model(),
        "sample": f"oracle/reference_path.py (torch CPU, {threads} threads = every core this process is entitled to, {_cpu_model()}): "
                  f"ONE timed evaluation of the full workload, all {C_full} classes ({C_full * C_full} ordered pairs, m={m}): "
                  f"{t_full:.2f} s; warm-up {t_warm:.1f} s on 60 classes",
        "seconds_per_eval": t_full,
        "one_thread": {
            "value": 1.0 / est_one, "unit": "evals/s", "cores": 1, "full_size": False,
            "sample": f"1 thread on the first {C_s} of {C_full} classes ({C_s * C_s} ordered pairs): {t_one:.2f} s, scaled by "
                      f"(C/C_s)^2 to the full workload (= {est_one:.0f} s/eval)",
            "sample_seconds_per_eval": t_one,
        },
    }


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without torch.distributed.run: start one child process per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, exactly what torch.distributed.run
    would set), relay rank 0's JSON line, return the worst exit code.  Runs BEFORE anything in this
    process touches the GPU, and never replaces a process: children are plain subprocesses."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread; meanwhile watch every child: if one dies with an error the
    # others would wait for it in a collective forever, so the remaining ranks are stopped (exact PIDs)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while procs[0].poll() is None:
        for r, p in enumerate(procs[1:], start=1):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, rc)
        if failed is not None:
            break
        time.sleep(0.2)
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}; stopping the other ranks\n")
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:  # rank 0 is gone: a rank still alive is stuck in a collective
            p.kill()
            codes.append(p.wait())
    reader.join(timeout=10)
    out = b"".join(c for c in chunks if c)
    for line in out.decode().splitlines():   # stdout carries the JSON line only (libraries' chatter -> stderr)
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-closure", action="store_true", help="skip the secondary full-closure measurement (N=1)")
    ap.add_argument("--no-c4-pairs", action="store_true", help="skip the second (m=32) pair workload of the scaling curve")
    ap.add_argument("--no-c4-closure", action="store_true", help="skip the class-sharded c4 closure leg of the scaling curve")
    ap.add_argument("--no-c3-sqfa", action="store_true", help="skip the leg on the reference's default model (SQFA: m = K+1)")
    ap.add_argument("--fit", action="store_true", help="also time model.fit() (metric M3) inside the closure object")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launcher_selftest:  # CPU check of the launcher: no GPU, no process group
        if os.environ.get("SQFA_BENCH_SELFTEST_FAIL_RANK") == str(rank):
            raise SystemExit(7)          # a rank that dies: the launcher must stop the others and report it
        if os.environ.get("SQFA_BENCH_SELFTEST_FAIL_RANK") is not None:
            time.sleep(60)               # ... while they would wait for it in a collective
        if rank == 0:
            print(json.dumps({"selftest": True, "world": world, "rank": rank, "local_rank": local_rank,
                              "master": os.environ.get("MASTER_ADDR"), "port": int(os.environ.get("MASTER_PORT", "0"))}))
        return
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal aid: SQFA_BENCH_REHEARSAL=1 runs all ranks on cuda:0 over gloo (one-GPU box)
    rehearsal = os.environ.get("SQFA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from sqfa_amd import _lib, _native
    from sqfa_amd.parallel import PairShard

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        shard = PairShard()
    else:
        shard = PairShard(rank=0, world_size=1)

    C, D, K, model = WORKLOADS[args.workload]
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    lib = _lib.load()
    import ctypes

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graphs_used = []   # N > 1: whether each pair workload's launches were replayed from a captured graph

    def time_pair_workload(S, scale, warmup, steps):
        """W untimed + K timed loss+grad evaluations of the pair workload S; returns
        (seconds for the K steps [max over ranks], pair kernel ms per launch [max over ranks], last outputs)."""
        P_ = S.shape[0] * (S.shape[0] - 1) // 2
        weight = -1.0 / P_

        fused = torch.empty(S.numel() + 3, dtype=S.dtype, device=device) if world > 1 else None

        def launch_shard():
            """This rank's tile shard into the fused buffer [loss, nan, inf, dL/dS] (what PairwiseLoss.forward does)."""
            out = _native.hip_pair_backend(S, None, scale=scale, eps=_native.EPSILON, sqrt_mode=True, weights=None,
                                           uniform_weight=weight, shard=shard.shard, want_loss=True, want_grad=True,
                                           want_dist=False, want_eig=False, out_loss=fused[0],
                                           out_gradA=fused[3:].view(S.shape))
            fused[1:3].copy_(out["nonfinite"])   # int32 -> real in the copy itself

        graph = {"g": None}

        def step(eager=False):
            if world > 1:
                # N > 1: the kernel launches of a step are replayed from a captured HIP graph, the all-reduce stays
                # eager -- exactly how a sharded fit runs its closure (sqfa_amd._optim.ShardedClosure).  At 8 ranks a
                # rank's GPU work is 0.17 ms; launched eagerly from Python the step was host-bound at 0.29 ms
                # (tools/time_shard.py, profiles/r3_shard_timings.txt).
                if graph["g"] is not None and not eager:
                    graph["g"].replay()
                else:
                    launch_shard()
                dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=shard.group)
                return fused[0], fused[1:3].to(torch.int32), fused[3:].view(S.shape)
            out = _native.hip_pair_backend(S, None, scale=scale, eps=_native.EPSILON, sqrt_mode=True, weights=None,
                                           uniform_weight=weight, shard=shard.shard, want_loss=True, want_grad=True,
                                           want_dist=False, want_eig=False)
            return out["loss"], out["nonfinite"], out["gradA"]

        if world > 1:
            for _ in range(2):
                step()
            fence()
            try:
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread must not invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    launch_shard()
                graph["g"] = g
            except Exception as err:   # the capture is an optimisation: keep measuring, eagerly, and say so
                sys.stderr.write(f"bench.py: HIP graph capture of the shard launches failed ({err}); running them eagerly\n")
                graph["g"] = None
            graphs_used.append(graph["g"] is not None)
            fence()
        for _ in range(warmup):
            res = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):    # the timed region: exactly K steps, profiling off
            res = step()
        fence()
        seconds = time.perf_counter() - t0
        # kernel duration: a SEPARATE short pass with HIP events around the pair kernel on its launch stream
        # (asynchronous records; plain launches: a graph replay does not repeat them), right after the timed
        # region -- same inputs, same clock state
        lib.sqfa_airm_profile(1)
        for _ in range(max(5, min(30, steps))):
            step(True)
        fence()
        lib.sqfa_airm_profile(0)
        ms_total, launches = ctypes.c_double(0), ctypes.c_int(0)
        lib.sqfa_airm_profile_read(ctypes.byref(ms_total), ctypes.byref(launches))
        k_ms = ms_total.value / max(launches.value, 1)
        if world > 1:
            both = torch.tensor([seconds, k_ms], dtype=torch.float64, device=device)
            dist.all_reduce(both, op=dist.ReduceOp.MAX)
            seconds, k_ms = both.tolist()
        return seconds, k_ms, res

    # second pair workload for the scaling curve (measured FIRST: half a second of the heavier kernel also
    # brings the chip to its sustained clock before the headline's own warm-up): BASELINE config 4's pair stage
    # (C=1000, m=32), 8x the
    # work per pair of c3, so that the per-evaluation fixed costs (launches, one all-reduce) stay small
    # against the kernel also at 8 ranks.  Same sharding, same step; every rank takes part.
    c4_pairs = None
    prewarm = None
    S, scale = make_feature_scatters(C, D, K, model, device, dtype)   # both inputs first: no idle gap between the legs
    if args.workload != "c4" and not args.no_c4_pairs and dtype == torch.float32:
        C4, D4, K4, model4 = WORKLOADS["c4"]
        S4, scale4 = make_feature_scatters(C4, D4, K4, model4, device, dtype)
        steps4, warm4 = max(5, min(40, args.steps)), max(2, min(10, args.warmup))
        t_pre = time.perf_counter()
        sec4, kms4, (loss4, flags4, _g4) = time_pair_workload(S4, scale4, warm4, steps4)
        prewarm = {
            "what": "the scaling_c4_pairs leg (C=1000, m=32 pair kernel) runs BEFORE the headline's own --warmup steps, so "
                    "the headline is measured at the chip's sustained clock (what a fit sees), not on the ramp from idle",
            "launches": warm4 + steps4 + max(5, min(30, steps4)) + (2 if world > 1 else 0), "seconds": time.perf_counter() - t_pre,
        }
        assert flags4.tolist() == [0, 0]
        c4_pairs = {
            "workload": f"c4 pair stage: C={C4} classes, n_filters={K4} (m={S4.shape[1]}), {C4 * (C4 - 1) // 2} unordered pairs per eval",
            "value": steps4 / sec4, "unit": "evals/s", "n_gpus": world, "steps": steps4, "warmup": warm4,
            "ms_per_step": sec4 / steps4 * 1e3, "scaling": "strong", "pair_kernel_ms": kms4, "loss": loss4.item(),
            "roofline_frac": 8.0 * C4 * (C4 - 1) * S4.shape[1] ** 3 / world / (kms4 * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
        }
        del S4, _g4

    m = S.shape[1]
    P = C * (C - 1) // 2
    elapsed, kernel_ms, (loss, flags, grad) = time_pair_workload(S, scale, args.warmup, args.steps)
    assert flags.tolist() == [0, 0], f"non-finite distances: {flags.tolist()}"

    # the reference's DEFAULT model (sqfa.model.SQFA, README flow: Calvo-Oller embedding, m = K+1) on the same configuration,
    # same step, same sharding -- its own roofline next to the headline's
    c3_sqfa = None
    if args.workload == "c3" and not args.no_c3_sqfa:
        S17, scale17 = make_feature_scatters(C, D, K, "sqfa", device, dtype)
        steps17, warm17 = max(5, min(100, args.steps)), max(2, min(20, args.warmup))
        sec17, kms17, (loss17, flags17, _g17) = time_pair_workload(S17, scale17, warm17, steps17)
        assert flags17.tolist() == [0, 0]
        m17 = S17.shape[1]
        fl17 = 8.0 * C * (C - 1) * m17 ** 3 / world
        peak17 = FP32_PEAK_TFLOPS if dtype == torch.float32 else FP64_PEAK_TFLOPS
        c3_sqfa = {
            "workload": f"c3-sqfa: C={C} classes, n_dim={D}, n_filters={K}, sqfa (m={m17}: Calvo-Oller embedding, scale 1/2), "
                        f"{C * (C - 1) // 2} unordered pairs per eval",
            "value": steps17 / sec17, "unit": "evals/s", "n_gpus": world, "steps": steps17, "warmup": warm17,
            "ms_per_step": sec17 / steps17 * 1e3, "scaling": "strong", "loss": loss17.item(),
            "roofline": {"bound": "valu", "achieved": fl17 / (kms17 * 1e-3) / 1e12, "peak": peak17, "unit": "TFLOP/s",
                         "frac": fl17 / (kms17 * 1e-3) / 1e12 / peak17, "traffic": None, "kernel": "pair_tile_kernel",
                         "kernel_ms": kms17, "algorithmic_flops_per_launch": fl17},
        }
        del S17, _g17

    S_cpu = S.detach().cpu() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    # BASELINE config 4 end to end (class-sharded projection + pair shard + collectives), every rank takes part
    c4_closure = None
    if not args.no_c4_closure and dtype == torch.float32:
        del S, grad
        torch.cuda.empty_cache()
        c4_closure = c4_closure_leg(world, rank, device, shard, max(5, min(20, args.steps // 4)), lib, fence, dist)
        torch.cuda.empty_cache()
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        evals_per_s = args.steps / elapsed
        flops_eval = 8.0 * C * (C - 1) * m ** 3          # 16 m^3 per unordered pair
        flops_launch = flops_eval / world                # each rank's launch covers 1/world of the tiles
        achieved_tf = flops_launch / (kernel_ms * 1e-3) / 1e12
        peak_tf = FP32_PEAK_TFLOPS if dtype == torch.float32 else FP64_PEAK_TFLOPS
        esz = 4 if dtype == torch.float32 else 8
        bytes_eval = esz * (2 * C * m * m + 1)
        result = {
            "metric": "SPD pairwise-distance loss+grad evals/sec",
            "value": evals_per_s,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: C={C} classes, n_dim={D}, n_filters={K}, {model} (m={m}), "
                            f"{P} unordered pairs per eval, inputs resident in HBM",
                "parallelism": f"pair-tile shard over {world} GPU(s) + 1 all-reduce of [loss,flags,grad] per eval",
            },
            "pairs_per_s": evals_per_s * P,
            "loss": loss.item(),
            "rccl_ranks": dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0),
            "shard_launches_from_hip_graph": bool(graphs_used and all(graphs_used)),
            "roofline": {
                "bound": "valu",
                "achieved": achieved_tf,
                "peak": peak_tf,
                "unit": "TFLOP/s",
                "frac": achieved_tf / peak_tf,
                "traffic": pmc_traffic("pair_tile_kernel", args.workload, args.dtype) if world == 1 else None,
                "traffic_unit": f"bytes per launch (profiles/{PMC_FILE})",
                "kernel": "pair_tile_kernel",
                "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": flops_launch,
                "note": ("FP32 compute-bound kernel: the f32 VALU peak equals the f32 MFMA dense peak (157.3 TF); " if dtype == torch.float32 else
                         "FP64 compute-bound kernel: peak = the f64 vector rate, 78.6 TF (public specification: half the f32 vector rate); ") +
                        "algorithmic flops = 16*m^3 per unordered pair (SURVEY.md 8d). Algorithmic HBM traffic is "
                        f"{bytes_eval / 1e6:.2f} MB per eval = {bytes_eval / (kernel_ms * 1e-3) / 1e9:.1f} GB/s, "
                        f"{bytes_eval / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS * 100:.3f}% of the 8 TB/s HBM roofline",
            },
        }
        result["prewarm"] = prewarm
        if c3_sqfa is not None:
            result["c3_sqfa"] = c3_sqfa
        if c4_pairs is not None:
            result["scaling_c4_pairs"] = c4_pairs
        if c4_closure is not None:
            result["scaling_c4_closure"] = c4_closure
        if not args.no_closure and world == 1 and dtype == torch.float32:
            result["closure"] = closure_benchmark(C, D, K, model, device, max(10, min(100, args.steps // 2)), lib, with_fit=args.fit)
            result["closure"]["roofline"]["traffic"] = pmc_traffic(result["closure"]["roofline"]["kernel"], args.workload, args.dtype)
            result["closure"]["roofline"]["traffic_unit"] = f"bytes per launch (profiles/{PMC_FILE})"
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(S_cpu, scale, C)
            result["speedup_vs_cpu_baseline"] = evals_per_s / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
