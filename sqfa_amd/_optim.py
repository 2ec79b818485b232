"""LBFGS driver for the SQFA models (reference: src/sqfa/_optim.py).

The optimizer is torch.optim.LBFGS' algorithm with identical control flow and state (either
torch's class itself or the vectorised CompactLBFGS subclass, sqfa_amd/_lbfgs.py), so epoch
semantics are the reference's (the loss recorded for an epoch is that of the first closure
call of the step).  What changes is the closure body: when the model's ``distance_fun``
is one of the native affine-invariant operators the closure issues ONE fused
loss+gradient launch (no (C,C) matrix, no tril gather, validity flag read with the loss)
instead of the reference's distance-matrix -> guard -> gather -> mean chain.
"""
import contextlib
import gc
import time
import warnings

import torch
from tqdm import tqdm

__all__ = ["fitting_loop"]


def __dir__():
    return __all__


# LBFGS' two-loop recursion is ~200 tiny vector operations per iteration (history_size=100).
# On a GPU every one of them is a kernel launch (~5-10 us), which for the small parameter of
# SQFA (K x D <= 16 x 3072 floats) is far more than the closure itself at small C (6.6 ms per
# closure at C=10 against 0.8 ms of closure work).  With this switch on, torch.optim.LBFGS
# runs -- unmodified -- on HOST copies of the parameters; each closure call pushes the current
# host values into the device parameters, evaluates loss and gradient on the GPU, and pulls the
# gradient back (two small copies).  Same arithmetic, same stopping rule.  Only used while the
# parameter vector stays below torch's intra-op parallel grain (32768 elements): beyond it
# the host vector ops fan out over every hardware thread the process can see, which is far
# slower than the GPU launches on an oversubscribed host.
HOST_SIDE_LBFGS = True
HOST_SIDE_LBFGS_MAX_NUMEL = 32768
# CompactLBFGS without a line search keeps its state on the DEVICE at every size since late round 2 (native
# push / direction / statistics kernels, one synchronisation per iteration: c2-SQFA 0.47 -> 0.31 ms per closure,
# c1 0.35 -> 0.32; tools/ab_host_vs_device_lbfgs.py); with a line search (torch's own step) small parameters
# stay on the host as before.
HOST_SIDE_LBFGS_MAX_NUMEL_COMPACT = 0
HOST_SIDE_LBFGS_MAX_NUMEL_LINE_SEARCH = 8192
HOST_SIDE_LBFGS_THREADS = 4  # intra-op threads while the optimizer state lives on the host
# Device-side optimizer state: the closure hands [loss, nan, inf] to CompactLBFGS on the device (one host
# synchronisation per LBFGS iteration instead of three, see _lbfgs.CompactLBFGS.speculate_descent_test).
DEFERRED_CLOSURE = True

# Use sqfa_amd._lbfgs.CompactLBFGS (torch.optim.LBFGS with the two-loop recursion evaluated as
# two triangular solves: ~15 instead of ~400 vector operations per iteration).  False selects
# torch.optim.LBFGS itself.
COMPACT_LBFGS = True

# Closure latency.  After GRAPH_WARMUP_CLOSURES eager evaluations the
# fused closure (parametrization -> projection -> pairwise loss+gradient -> backward to the raw
# filters -> packing of [loss, nan, inf, grad]) is captured once in a HIP graph; every later
# closure is ONE graph launch plus ONE device-to-host copy instead of ~40 launches and a trip
# through the autograd engine.  Shapes are static within a fitting_loop call; data-dependent
# errors are reported through the flags, which are read on the host after the replay.  Only
# the single-process fused path is captured (no collectives inside the graph).  If the capture
# fails the loop warns once and continues eagerly -- same arithmetic either way.
GRAPH_CLOSURE = True
GRAPH_WARMUP_CLOSURES = 3

_NAN_MSG = "Some distances between classes are NaN. Try using float64 or a different regularization parameter."
_INF_MSG = "Some distances between classes are inf. Try using float64 or a different regularization parameter."


def check_distances_valid(distances):
    """Raise ValueError when the distance matrix holds NaN or inf.  Like the reference
    (src/sqfa/_optim.py:16-30) the whole matrix is inspected."""
    if torch.isnan(distances).any():
        raise ValueError(_NAN_MSG)
    if torch.isinf(distances).any():
        raise ValueError(_INF_MSG)


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is capturing: a collection that finalises device tensors or an
    older CUDAGraph in the middle of a capture aborts the process (seen in the GPU test suite: 'Fatal Python
    error: Aborted ... Garbage-collecting' inside the captured closure)."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def raise_on_flags(flags):
    """flags: int tensor {nan_count, inf_count} produced by the fused kernel."""
    n_nan, n_inf = (int(v) for v in flags.tolist())
    if n_nan:
        raise ValueError(_NAN_MSG)
    if n_inf:
        raise ValueError(_INF_MSG)


def _n_classes(data_statistics):
    if isinstance(data_statistics, dict):
        return data_statistics["means"].shape[0]
    return data_statistics.shape[0]


# Sharded fits: once per epoch the ranks compare a checksum of their parameters (parallel.replicas_agree: one
# all-reduce of two integers).  If they ever differ -- an all-reduce that did not return the same bits to every
# rank -- rank 0's parameters are broadcast again and every rank restarts its LBFGS history, with a warning.
REPLICA_CHECK = True


def _sharded_group(model):
    shard = getattr(model, "pair_shard", None) or getattr(model, "class_shard", None)
    if shard is None or getattr(shard, "world_size", 1) <= 1:
        return None, False
    return getattr(shard, "group", None), True


def _broadcast_replicated_state(model):
    """Multi-GPU fits assume bit-identical filters on every rank (each rank evaluates its tile
    shard of the SAME feature scatters, and every rank repeats the same LBFGS update on the
    all-reduced gradient).  Rank 0's parameters and buffers are therefore broadcast once per
    fitting_loop call: ranks that were initialised from different seeds (the default
    ``torch.randn`` filters) would otherwise mix inconsistent shards without any error."""
    shard = getattr(model, "pair_shard", None) or getattr(model, "class_shard", None)
    if shard is None or getattr(shard, "world_size", 1) <= 1:
        return
    import torch.distributed as dist
    group = getattr(shard, "group", None)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class ShardedClosure:
    """One closure evaluation of a sharded (multi-GPU) fit as captured HIP graphs around the collectives.

    pair tiles over ranks (PairShard):  graph A = parametrization -> projection -> pair kernels -> fused buffer
      [loss, nan, inf, dL/dS];  ONE all-reduce (eager: RCCL and gloo alike);  graph B = backward -> packed
      [loss, nan, inf, grad];  the caller makes ONE device-to-host copy per closure.
    + class-sharded statistics (ClassShard, large D: BASELINE config 4):  graph A1 = parametrization -> projection
      of the LOCAL classes into their slice of the (C,m,m) batch;  all-gather of the slices;  graph A2 = pair
      kernels -> fused buffer;  all-reduce;  graph B = backward of the local classes;  all-reduce of the filter
      gradient;  graph C = packing.
    Needs the single-node closure's conditions (model._single_node_inputs); other sharded fits stay eager.
    Used by fitting_loop and, directly, by bench.py's N > 1 closure leg."""

    @staticmethod
    def supported(model, prepared):
        params = list(model.parameters())
        return bool(GRAPH_CLOSURE and len(params) == 1 and params[0].is_cuda
                    and getattr(model, "pair_shard", None) is not None and model.pair_shard.world_size > 1
                    and hasattr(model, "_has_fused_closure") and model._has_fused_closure()
                    and hasattr(model, "_single_node_inputs")
                    and model._single_node_inputs(prepared, allow_class_shard=True) is not None
                    and model._noise_scalar() is not None)

    def __init__(self, model, prepared):
        self.state, self.calls, self.graphs = "warmup", 0, None
        self.stages, self.box = self._build(model, prepared)

    @staticmethod
    def _build(model, prepared):
        """Stage functions over static tensors, for eager warm-up and capture."""
        from . import _native, distances
        import torch.distributed as dist
        raw, scatters, means, sphere = model._single_node_inputs(prepared, allow_class_shard=True)
        _, scale, sqrt_mode = distances.fused_spec(model.distance_fun)
        noise = model._noise_scalar()
        shard = model.pair_shard
        cshard = getattr(model, "class_shard", None)
        C_loc, K = scatters.shape[0], raw.shape[0]
        C = cshard.n_classes if cshard is not None else C_loc
        offset = cshard.offset if cshard is not None else 0
        m = K + 1 if means is not None else K
        weight = -1.0 / (C * (C - 1) // 2)
        S_full = torch.empty((C, m, m), dtype=scatters.dtype, device=scatters.device)
        fused = torch.empty(C * m * m + 3, dtype=scatters.dtype, device=scatters.device)
        box = {}
        if cshard is not None:
            # every rank sends its slice padded to the largest shard (equal sizes: RCCL and gloo alike)
            n_max = max(cshard.counts)
            send = torch.zeros((n_max, m, m), dtype=scatters.dtype, device=scatters.device)
            recv = torch.empty((cshard.world_size, n_max, m, m), dtype=scatters.dtype, device=scatters.device)
            local_out = send[:C_loc]
        else:
            local_out = S_full

        def stage_project():
            box["st"] = _native.closure_stage_project(raw, scatters, means, noise, sphere, out_S=local_out)

        def gather():
            if cshard is not None:
                dist.all_gather([recv[r] for r in range(cshard.world_size)], send, group=cshard.group)

        def stage_pairs():
            if cshard is not None:
                start = 0
                for r, n in enumerate(cshard.counts):
                    S_full[start:start + n].copy_(recv[r, :n])
                    start += n
            _native.closure_stage_pairs(S_full, scale, sqrt_mode, weight, shard.shard, fused)

        def reduce():
            dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=shard.group)

        def stage_backward():
            gS = fused[3:].view(C, m, m)[offset:offset + C_loc]
            box["grad"] = _native.closure_stage_backward(box["st"], gS, None)

        def reduce_grad():
            if cshard is not None:
                dist.all_reduce(box["grad"], op=dist.ReduceOp.SUM, group=cshard.group)

        def stage_pack():
            box["packed"] = torch.cat([fused[:3], box["grad"].reshape(-1)])

        return [stage_project, gather, stage_pairs, reduce, stage_backward, reduce_grad, stage_pack], box

    def _capture(self):
        graphs = []
        pool = None
        for i, stage in enumerate(self.stages):
            if i % 2 == 1:          # the collectives stay eager
                stage()
                graphs.append(None)
                continue
            g = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread polls its events while this thread captures; under
            # the default "global" mode such a call from ANOTHER thread invalidates the capture
            with _no_gc(), torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                stage()
            pool = g.pool()
            graphs.append(g)
        self.graphs, self.state = graphs, "on"

    def run(self, eager=False):
        """Enqueue one evaluation; returns (packed [loss, nan, inf, grad...], grad) as device tensors, no host sync.
        eager=True runs the stages as plain launches even after the capture (profiling passes: the library's HIP
        event records around its kernels happen at launch time, which a graph replay does not repeat)."""
        if eager:
            # The stage functions REBIND box["st"] / box["grad"] / box["packed"] to fresh tensors, while captured graphs keep
            # writing into the tensors bound at capture time (and the eager collectives between the graphs read the box).
            # An eager pass after the capture therefore works on its own bindings, which are dropped again afterwards:
            # the next replay must find the capture-time tensors (it all-reduced and returned the stale eager ones).
            saved = dict(self.box) if self.state == "on" else None
            for stage in self.stages:
                stage()
            out = self.box["packed"], self.box["grad"]
            if saved is not None:
                self.box.clear()
                self.box.update(saved)
            return out
        if self.state == "warmup" and self.calls >= GRAPH_WARMUP_CLOSURES:
            try:
                self._capture()
            except Exception as err:
                warnings.warn(f"sqfa_amd: HIP graph capture of the sharded closure failed ({err}); running eagerly")
                self.state = "eager"
        if self.state == "on":
            for stage, g in zip(self.stages, self.graphs):
                if g is None:
                    stage()
                else:
                    g.replay()
        else:
            self.calls += 1
            for stage in self.stages:
                stage()
        return self.box["packed"], self.box["grad"]


def fitting_loop(model, data_statistics, max_epochs=200, lr=0.1, atol=1e-6, show_progress=True,
                 return_loss=False, **kwargs):
    """Learn the filters with LBFGS.  Same arguments, stopping rule (|dloss| < atol for three
    consecutive epochs), messages and return value as the reference's fitting_loop
    (src/sqfa/_optim.py:33-145); extra keyword arguments go to torch.optim.LBFGS."""
    _broadcast_replicated_state(model)
    device_params = list(model.parameters())
    # the compact form is four (history x n) matrix-vector products per iteration: on the host
    # only while they are a fraction of a millisecond, otherwise on the device
    if not COMPACT_LBFGS:
        host_limit = HOST_SIDE_LBFGS_MAX_NUMEL
    elif kwargs.get("line_search_fn") is not None:
        host_limit = HOST_SIDE_LBFGS_MAX_NUMEL_LINE_SEARCH
    else:
        host_limit = HOST_SIDE_LBFGS_MAX_NUMEL_COMPACT
    use_host = (HOST_SIDE_LBFGS and len(device_params) > 0 and all(p.is_cuda for p in device_params)
                and sum(p.numel() for p in device_params) <= host_limit)
    if use_host:
        opt_params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in device_params]
    else:
        opt_params = device_params
    def new_optimizer():
        if COMPACT_LBFGS:
            from ._lbfgs import CompactLBFGS
            return CompactLBFGS(opt_params, lr=lr, **kwargs)
        return torch.optim.LBFGS(opt_params, lr=lr, **kwargs)

    optimizer = new_optimizer()
    shard_group, sharded = _sharded_group(model)

    def push_parameters():
        if use_host:
            with torch.no_grad():
                for p, h in zip(device_params, opt_params):
                    p.copy_(h, non_blocking=True)
    prepared = model._prepare_statistics(data_statistics)
    n_classes = model._n_classes_total(prepared) if hasattr(model, "_n_classes_total") else _n_classes(data_statistics)
    if n_classes < 2:
        raise ValueError("At least two classes are needed to fit the filters.")
    rows, cols = torch.tril_indices(n_classes, n_classes, offset=-1)

    def evaluate():
        """Enqueue loss and gradient on the device: (loss, flags or None); no host sync."""
        fused = model._fused_closure_loss(prepared)
        if fused is not None:
            loss, flags = fused
        else:
            if getattr(model, "class_shard", None) is not None:
                raise NotImplementedError("class-sharded statistics need one of the native distance operators")
            distances = model.get_class_distances(prepared, regularized=True)
            check_distances_valid(distances)
            loss = -distances[rows.to(distances.device), cols.to(distances.device)].mean()
            flags = None
        loss.backward()
        if hasattr(model, "_sync_gradients"):
            model._sync_gradients()
        return loss.detach(), flags

    def pack(loss, flags):
        """[loss, nan, inf, grad...] in one device tensor: a single copy brings it to the host."""
        dtype = loss.dtype
        head = [loss.reshape(1), flags.to(dtype) if flags is not None else loss.new_zeros(2)]
        return torch.cat(head + [p.grad.reshape(-1).to(dtype) for p in device_params])

    def unpack_to_host(packed):
        host = packed.cpu()  # the only synchronisation of the closure
        raise_on_flags(host[1:3].round().to(torch.int32))
        offset = 3
        for p, h in zip(device_params, opt_params):
            n = p.numel()
            h.grad = host[offset:offset + n].view_as(h).to(h.dtype)
            offset += n
        return host[0]

    # ---- sharded fits: captured graphs around the collectives (ShardedClosure below) -------------------
    split = ShardedClosure(model, prepared) if (len(device_params) == 1 and ShardedClosure.supported(model, prepared)) else None

    def split_closure(defer=False):
        push_parameters()
        packed, grad = split.run()
        device_params[0].grad = grad
        if use_host:
            return unpack_to_host(packed)
        if defer:
            return packed[:3]
        head = packed[:3].cpu()
        raise_on_flags(head[1:3].round().to(torch.int32))
        return head[0]

    graph = {"state": "off", "calls": 0, "graph": None, "packed": None, "grads": None}
    if (GRAPH_CLOSURE and len(device_params) > 0 and all(p.is_cuda for p in device_params)
            and getattr(model, "pair_shard", None) is None and getattr(model, "class_shard", None) is None
            and hasattr(model, "_has_fused_closure") and model._has_fused_closure()):
        graph["state"] = "warmup"

    def capture():
        for p in device_params:
            p.grad = None
        g = torch.cuda.CUDAGraph()
        with _no_gc(), torch.cuda.graph(g):
            loss, flags = evaluate()
            packed = pack(loss, flags)
        graph.update(graph=g, packed=packed, grads=[p.grad for p in device_params], state="on")

    def closure(defer=False):
        """defer=True (device-side optimizer state only): enqueue everything and return [loss, nan, inf] as a
        DEVICE tensor without synchronising -- CompactLBFGS reads it back together with its own decision
        scalars and then calls closure.check_flags (one host synchronisation per LBFGS iteration)."""
        if split is not None:
            return split_closure(defer)
        push_parameters()
        if graph["state"] == "warmup" and graph["calls"] >= GRAPH_WARMUP_CLOSURES:
            try:
                capture()
            except Exception as err:  # capture is an optimisation: report and continue eagerly
                warnings.warn(f"sqfa_amd: HIP graph capture of the closure failed ({err}); running eagerly")
                graph["state"] = "off"
        if graph["state"] == "on":
            graph["graph"].replay()
            packed = graph["packed"]
            for p, grad in zip(device_params, graph["grads"]):
                p.grad = grad  # the static gradient tensors the graph writes
            if use_host:
                return unpack_to_host(packed)
            if defer:
                return packed[:3]
            head = packed[:3].cpu()  # loss and flags in one read-back; the gradient stays on the device
            raise_on_flags(head[1:3].round().to(torch.int32))
            return head[0]
        graph["calls"] += 1
        for p in device_params:
            p.grad = None
        loss, flags = evaluate()
        if use_host:
            return unpack_to_host(pack(loss, flags))
        if defer:
            return torch.cat([loss.reshape(1), flags.to(loss.dtype) if flags is not None else loss.new_zeros(2)])
        if flags is not None:
            raise_on_flags(flags)
        return loss

    if DEFERRED_CLOSURE and not use_host and len(device_params) > 0 and all(p.is_cuda for p in device_params):
        closure.deferred = lambda: closure(True)
        closure.check_flags = lambda n_nan, n_inf: raise_on_flags(torch.tensor([round(n_nan), round(n_inf)], dtype=torch.int32))

    losses, times = [], []
    start = time.time()
    previous = 0.0
    streak = 0
    saved_threads = torch.get_num_threads()
    if use_host:
        torch.set_num_threads(min(saved_threads, HOST_SIDE_LBFGS_THREADS))
    try:
        for epoch in tqdm(range(max_epochs), desc="Epochs", unit="epoch", disable=not show_progress):
            value = optimizer.step(closure).item()
            times.append(time.time() - start)
            losses.append(value)
            if sharded and REPLICA_CHECK:
                from .parallel import replicas_agree
                push_parameters()   # host-side optimizer state: compare (and, if need be, re-broadcast) the device copies
                if not replicas_agree(device_params, shard_group):
                    warnings.warn("sqfa_amd: the ranks of this sharded fit no longer hold identical filters (the "
                                  "all-reduce returned different bits to different ranks); re-broadcasting rank 0's "
                                  "filters and restarting the LBFGS history on every rank")
                    _broadcast_replicated_state(model)
                    if use_host:
                        with torch.no_grad():
                            for p_dev, h in zip(device_params, opt_params):
                                h.copy_(p_dev.detach().cpu())
                    optimizer = new_optimizer()
            streak = streak + 1 if abs(previous - value) < atol else 0
            previous = value
            if streak >= 3:
                tqdm.write(
                    f"Loss change below {atol} for 3 consecutive epochs. "
                    f"Stopping training at epoch {epoch + 1}/{max_epochs}."
                )
                break
        else:
            print(
                f"Reached max_epochs ({max_epochs}) without meeting stopping criteria."
                + "Consider increasing max_epochs, changing initialization or using dtype=torch.float64."
            )
    finally:
        if use_host:
            torch.set_num_threads(saved_threads)
    push_parameters()  # the values LBFGS ended on
    if return_loss:
        return torch.tensor(losses), torch.tensor(times)
    return None
