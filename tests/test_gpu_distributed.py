"""Two ranks sharing cuda:0 over gloo (a one-GPU rehearsal of the multi-GPU path with the REAL
HIP backend): tile shards + the fused all-reduce buffer written by the kernel (PairShard),
and the class-sharded projection (ClassShard).  On a multi-GPU node the same code runs with
backend "nccl" (RCCL) and one device per rank (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import model_cases as mc
        from sqfa_amd import _native, distances
        from sqfa_amd.parallel import ClassShard, PairShard
        dev = torch.device("cuda:0")
        G1 = load_golden("g1_airm_self.npz")
        shard = PairShard()
        out = {}
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            S = torch.tensor(G1["C37_m16_S"], dtype=dtype, device=dev, requires_grad=True)
            loss, flags = _native.PairwiseLoss.apply(S, 1.0, distances.EPSILON, True, -1.0 / 666, shard.shard, shard.reduce)
            loss.backward()
            out[tag] = (loss.item(), S.grad.cpu().numpy(), flags.tolist())
        stats = mc.fit_stats("syn", torch.float64, dev)
        lo, hi = (0, 10) if rank == 0 else (10, 20)
        local = {k: v[lo:hi].clone() for k, v in stats.items()}
        model = mc.make_model("sqfa", 50, 2, 1e-3, "sphere", torch.float64, dev)
        model.pair_shard = shard
        model.class_shard = ClassShard(hi - lo)
        model.fit_pca(data_statistics=local)
        fl, _ = model.fit(data_statistics=local, max_epochs=3, show_progress=False, return_loss=True)
        q.put((rank, out, fl.numpy(), model.filters.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    G1 = load_golden("g1_airm_self.npz")
    G4 = load_golden("g4_fit.npz")
    (_, o0, fl0, F0), (_, o1, fl1, F1) = results
    for tag, ltol, gtol in (("f64", 1e-11, 1e-8), ("f32", 1e-5, 5e-5)):
        l0, g0, f0 = o0[tag]
        l1, g1, f1 = o1[tag]
        assert l0 == l1 and np.array_equal(g0, g1) and f0 == f1 == [0, 0]        # identical on both ranks
        ref_l = float(G1["C37_m16_loss_f64"])
        assert abs(l0 - ref_l) < ltol * abs(ref_l)
        assert np.linalg.norm(g0 - G1["C37_m16_grad_f64"]) < gtol * np.linalg.norm(G1["C37_m16_grad_f64"])
    assert np.array_equal(F0, F1)
    assert np.abs(fl0 - G4["syn_sqfa_K2_e3_loss"]).max() < 1e-6
    assert np.linalg.norm(F0 - G4["syn_sqfa_K2_e3_filters"]) < 1e-7 * np.linalg.norm(F0)


def _rccl_worker(port, q):
    """World of ONE rank on backend "nccl" (= RCCL): the collective calls of the multi-GPU path
    (fused-buffer all-reduce, all-gather of class slices, gradient all-reduce, barrier) run
    through the real library, with the same tensors, as far as a one-GPU box allows."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import model_cases as mc
        from sqfa_amd import _native, distances
        from sqfa_amd.parallel import ClassShard, PairShard
        G1 = load_golden("g1_airm_self.npz")
        shard = PairShard()
        S = torch.tensor(G1["C37_m16_S"], dtype=torch.float32, device=dev)
        # the fused buffer exactly as PairwiseLoss.forward / bench.py build it for world > 1
        fused = torch.empty(S.numel() + 3, dtype=S.dtype, device=dev)
        out = _native.hip_pair_backend(S, None, scale=1.0, eps=distances.EPSILON, sqrt_mode=True, weights=None,
                                       uniform_weight=-1.0 / 666, shard=(0, 1), want_loss=True, want_grad=True,
                                       want_dist=False, want_eig=False, out_loss=fused[0],
                                       out_gradA=fused[3:].view(S.shape))
        loss, flags, grad = shard.reduce_fused(fused, out["nonfinite"], S.shape)
        dist.barrier()
        stats = mc.fit_stats("syn", torch.float64, dev)
        model = mc.make_model("sqfa", 50, 2, 1e-3, "sphere", torch.float64, dev)
        model.pair_shard = shard
        model.class_shard = ClassShard(20)
        model.fit_pca(data_statistics=stats)
        fl, _ = model.fit(data_statistics=stats, max_epochs=3, show_progress=False, return_loss=True)
        q.put((loss.item(), flags.tolist(), grad.cpu().numpy(), fl.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_backend_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29800 + os.getpid() % 2000, q))
    p.start()
    loss, flags, grad, fl = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0
    G1 = load_golden("g1_airm_self.npz")
    G4 = load_golden("g4_fit.npz")
    ref_l = float(G1["C37_m16_loss_f64"])
    assert flags == [0, 0]
    assert abs(loss - ref_l) < 1e-5 * abs(ref_l)
    assert np.linalg.norm(grad - G1["C37_m16_grad_f64"]) < 5e-5 * np.linalg.norm(G1["C37_m16_grad_f64"])
    assert np.abs(fl - G4["syn_sqfa_K2_e3_loss"]).max() < 1e-6


def _rccl_graph_worker(port, q):
    """Graph captures while an RCCL communicator (and its watchdog thread) is alive: what every rank of
    `bench.py --gpus N` and of a sharded fit does.  One rank is all a one-GPU box allows; the sharded closure is
    built directly (ShardedClosure.supported asks for more than one rank)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import model_cases as mc
        from sqfa_amd._optim import ShardedClosure
        from sqfa_amd.parallel import ClassShard, PairShard
        warm = torch.ones(8, device=dev)
        dist.all_reduce(warm)                                   # the communicator exists from here on
        stats = {k: v.to(dev) for k, v in mc.c2_statistics(C=24, D=96).items()}
        results = {}
        for class_sharded in (False, True):
            model = mc.make_model("sqfa", 96, 4, 0.01, "sphere", torch.float64, dev)
            model.fit_pca(data_statistics=stats)
            model.pair_shard = PairShard()
            if class_sharded:
                model.class_shard = ClassShard(24)
            prepared = model._prepare_statistics(stats)
            closure = ShardedClosure(model, prepared)
            outs = []
            for _ in range(7):                                   # three eager, the capture, replays
                packed, grad = closure.run()
                outs.append(packed.detach().cpu().numpy().copy())
            eager, _ = closure.run(eager=True)
            # replay -> eager -> replay: after an eager pass the replays must still return the tensors the graphs
            # write; with the filters changed in between, a stale eager tensor would show the OLD filters' gradient
            raw = model.parametrizations.filters.original
            with torch.no_grad():
                raw.mul_(1.0 + 0.05 * torch.linspace(-1, 1, raw.numel(), dtype=raw.dtype, device=dev).view_as(raw))
            after, after_grad = closure.run()
            after = after.detach().cpu().numpy().copy()
            after_grad = after_grad.detach().cpu().numpy().copy()
            fresh, _ = ShardedClosure(model, prepared).run()      # an eager evaluation of a new object: the reference
            results[class_sharded] = (closure.state, [g is not None for g in closure.graphs or []], outs, eager.cpu().numpy(),
                                      after, after_grad, fresh.detach().cpu().numpy())
        q.put(results)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_graph_capture_with_live_rccl_communicator():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_graph_worker, args=(27800 + os.getpid() % 2000, q))
    p.start()
    results = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0
    for class_sharded, (state, graphs, outs, eager, after, after_grad, fresh) in results.items():
        assert state == "on", f"capture failed with a live RCCL communicator (class_sharded={class_sharded})"
        assert sum(graphs) == 4 and len(graphs) == 7           # four captured stages, three eager collectives
        for o in outs:
            assert np.array_equal(o, outs[0])                    # replays are bit-identical to the eager evaluations
        assert np.array_equal(eager, outs[0])
        assert np.isfinite(outs[0]).all() and outs[0][1] == 0 and outs[0][2] == 0
        # replay after an eager pass, on changed filters: the graph's own tensors, not the eager pass's stale ones
        assert not np.array_equal(after, outs[0])
        assert np.array_equal(after, fresh)
        assert np.array_equal(after_grad, fresh[3:].reshape(after_grad.shape))


def _split_graph_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import model_cases as mc
        import sqfa_amd._optim as opt
        from sqfa_amd.parallel import PairShard
        dev = torch.device("cuda:0")
        stats = {k: v.to(dev) for k, v in mc.c2_statistics(C=24, D=96).items()}
        counts = {"replay": 0, "cpu": 0}
        original_replay, original_cpu = torch.cuda.CUDAGraph.replay, torch.Tensor.cpu

        def replay(self):
            counts["replay"] += 1
            return original_replay(self)

        def cpu(self, *a, **k):
            if self.is_cuda:
                counts["cpu"] += 1
            return original_cpu(self, *a, **k)

        torch.cuda.CUDAGraph.replay = replay
        torch.Tensor.cpu = cpu
        runs = {}
        from sqfa_amd.parallel import ClassShard
        lo, hi = (0, 10) if rank == 0 else (10, 24)          # uneven class shards
        local = {k: v[lo:hi].contiguous() for k, v in stats.items()}
        for name, use_graph, sharded, class_sharded in (("eager", False, True, False), ("split", True, True, False),
                                                        ("single", True, False, False), ("eager_cs", False, True, True),
                                                        ("split_cs", True, True, True)):
            opt.GRAPH_CLOSURE = use_graph
            model = mc.make_model("sqfa", 96, 4, 0.01, "sphere", torch.float64, dev)
            if sharded:
                model.pair_shard = PairShard()
            if class_sharded:
                model.class_shard = ClassShard(hi - lo)
            data = local if class_sharded else stats
            model.fit_pca(data_statistics=data)
            counts["replay"] = counts["cpu"] = 0
            loss, _ = model.fit(data_statistics=data, max_epochs=6, show_progress=False, return_loss=True)
            runs[name] = (loss.numpy(), original_cpu(model.filters.detach()).numpy(), dict(counts))
        q.put((rank, runs))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_closure_runs_as_two_graphs_around_one_all_reduce():
    """the pair-sharded closure is captured as graph A (parametrization ->
    projection -> pair kernels -> fused buffer) and graph B (backward -> packed result) with the single
    all-reduce between them: identical filters to the eager sharded fit and to the single-process fit,
    two replays and ONE device-to-host read per closure."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30300 + os.getpid() % 2000
    procs = [ctx.Process(target=_split_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, r0), (_, r1) = results
    for name in ("eager", "split", "single", "eager_cs", "split_cs"):
        assert np.array_equal(r0[name][0], r1[name][0]) and np.array_equal(r0[name][1], r1[name][1])   # ranks agree bitwise
    le, Fe, ce = r0["eager"]
    ls, Fs, cs = r0["split"]
    l1, F1, c1 = r0["single"]
    assert ce["replay"] == 0
    assert cs["replay"] > 40, "the split-graph path was not taken"
    closures = cs["replay"] // 2 + 3                      # + the eager warm-up closures
    assert cs["cpu"] <= closures + 2, (cs, closures)      # one read-back per closure
    assert np.abs(ls - le).max() < 1e-12 and np.linalg.norm(Fs - Fe) < 1e-10 * np.linalg.norm(Fe)
    assert np.abs(ls - l1).max() < 1e-9 and np.linalg.norm(Fs - F1) < 1e-8 * np.linalg.norm(F1)
    # class-sharded statistics (uneven shards): four graphs around all-gather, all-reduce, gradient all-reduce
    lce, Fce, cce = r0["eager_cs"]
    lcs, Fcs, ccs = r0["split_cs"]
    assert cce["replay"] == 0 and ccs["replay"] > 80
    assert ccs["cpu"] <= ccs["replay"] // 4 + 3 + 2
    assert np.abs(lcs - lce).max() < 1e-11 and np.linalg.norm(Fcs - Fce) < 1e-9 * np.linalg.norm(Fce)
    assert np.abs(lcs - l1).max() < 1e-9 and np.linalg.norm(Fcs - F1) < 1e-8 * np.linalg.norm(F1)


@pytest.mark.timeout(600)
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` by itself: the launcher starts both ranks; on a
    one-GPU box they share cuda:0 over gloo (SQFA_BENCH_REHEARSAL=1).  One JSON line, n_gpus = 2, the same
    loss as the single-process run, the c4 pair leg and the class-sharded c4 closure leg present."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SQFA_BENCH_REHEARSAL"] = "1"
    env["SQFA_BENCH_C4_CLASSES"] = "60"        # the c4 closure leg (D=2048, K=32) on 60 instead of 1000 classes
    args = ["--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--workload", "c2"]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env,
                         capture_output=True, text=True, timeout=500)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    r2 = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-closure"] + args, env=env,
                         capture_output=True, text=True, timeout=500)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert r2["n_gpus"] == 2 and r2["scaling"] == "strong" and r2["unit"] == "evals/s" and r2["value"] > 0
    assert abs(r2["loss"] - r1["loss"]) < 1e-5 * abs(r1["loss"])
    assert r2["scaling_c4_pairs"]["n_gpus"] == 2 and r2["scaling_c4_pairs"]["value"] > 0
    assert r2["roofline"]["bound"] == "valu" and 0 < r2["roofline"]["frac"] < 1
    assert "launches" in r2["prewarm"] and r2["prewarm"]["seconds"] > 0           # what ran before the headline's warm-up
    # BASELINE config 4 end to end: class-sharded projection + pair shard, four graphs around three collectives;
    # the union of the ranks' class shards is the same data set as the single-process run's
    c2_, c1_ = r2["scaling_c4_closure"], r1["scaling_c4_closure"]
    assert c2_["n_gpus"] == 2 and c2_["graphs_per_closure"] == 4 and c2_["collectives_per_closure"] == 3
    assert c1_["n_gpus"] == 1 and c1_["collectives_per_closure"] == 0
    assert c2_["value"] > 0 and c2_["projection"]["kernel_ms"] > 0 and 0 < c2_["projection"]["frac"] < 1
    assert abs(c2_["loss"] - c1_["loss"]) < 2e-5 * abs(c1_["loss"])
