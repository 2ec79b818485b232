"""L-BFGS with a vectorised two-loop recursion.

Same algorithm, defaults, state handling and stopping rules as ``torch.optim.LBFGS`` (the
optimizer the reference uses, src/sqfa/_optim.py:78-82) for ``line_search_fn=None``; the only
change is HOW the search direction ``d = -H g`` is evaluated.  torch walks the history with two
Python loops -- about ``4 * history_size`` tiny vector operations per iteration, which is
launch-bound on a GPU and dispatch-bound on the host for the small parameter of SQFA.  Here the
history lives in two ``(h, n)`` buffers ``S`` (steps) and ``Y`` (gradient differences) plus the
small matrix ``SY[i, j] = s_i . y_j``, and the two loops become two triangular solves:

    first loop   (newest -> oldest)  al_i = rho_i s_i.(q0 - sum_{j>i} al_j y_j)
                 <=>  triu(SY) al = S q0                      (diag(SY) = 1/rho)
    second loop  (oldest -> newest)  c_i = al_i - rho_i y_i.(r0 + sum_{j<i} c_j s_j)
                 <=>  tril(SY^T) c = diag(SY) al - Y r0
    d = r0 + S^T c,   q = q0 - Y^T al,   r0 = H_diag q,   q0 = -g

i.e. four (h, n) matrix-vector products and two h x h solves per iteration: identical in exact
arithmetic, different only in floating-point summation order.  With a line search the class
defers to ``torch.optim.LBFGS.step``.
"""
import torch


class _History:
    """Ring buffers for the (s, y) pairs; `slots` lists the ring rows in chronological order."""

    # device-resident history: push / direction through the library's six-launch compact form
    # (sqfa_lbfgs_push / sqfa_lbfgs_direction) instead of ~35 torch launches and an index upload
    native = True

    def __init__(self, size, like):
        n = like.numel()
        self.size = size
        self.S = like.new_zeros(size, n)
        self.Y = like.new_zeros(size, n)
        self.SY = like.new_zeros(size, size)
        self.slots = []
        self._lib = None
        if self.native and like.is_cuda and like.dtype in (torch.float32, torch.float64):
            from . import _lib
            lib = _lib.load()
            if size <= lib.sqfa_lbfgs_max_history():
                self._lib = lib
                self._check = _lib.check
                self._code = _lib.SQFA_F32 if like.dtype == torch.float32 else _lib.SQFA_F64
                self._work = like.new_empty(lib.sqfa_lbfgs_work_elems(size, n))

    def _stream(self):
        import ctypes
        return ctypes.c_void_p(torch.cuda.current_stream(self.S.device).cuda_stream)

    def push(self, y, s):
        if len(self.slots) == self.size:
            slot = self.slots.pop(0)  # the oldest pair is overwritten
        else:
            slot = len(self.slots)
        self.slots.append(slot)
        if self._lib is not None:
            s, y = s.contiguous(), y.contiguous()
            with torch.cuda.device(self.S.device):
                self._check(self._lib.sqfa_lbfgs_push(self.S.data_ptr(), self.Y.data_ptr(), self.SY.data_ptr(), self.size,
                                                      self.S.shape[1], slot, s.data_ptr(), y.data_ptr(), self._work.data_ptr(),
                                                      self._code, self._stream()), "sqfa_lbfgs_push")
            return
        self.S[slot] = s
        self.Y[slot] = y
        self.SY[slot, :] = self.Y @ s   # s_new . y_j
        self.SY[:, slot] = self.S @ y   # s_i . y_new

    def step_stats(self, flat_grad, prev_flat_grad, d, t):
        """(y, s, scalars) with y = g - g_prev, s = t d and the DEVICE vector scalars = [max|g|, max|s|, y.s, y.y,
        y.s / y.y] in two launches (sqfa_lbfgs_step_stats); None without the native library."""
        if self._lib is None:
            return None
        g, gp, dd = flat_grad.contiguous(), prev_flat_grad.contiguous(), d.contiguous()
        y, s = torch.empty_like(g), torch.empty_like(g)
        scal = g.new_empty(5)
        with torch.cuda.device(g.device):
            self._check(self._lib.sqfa_lbfgs_step_stats(g.data_ptr(), gp.data_ptr(), dd.data_ptr(), float(t), g.numel(),
                                                        y.data_ptr(), s.data_ptr(), scal.data_ptr(), self._work.data_ptr(),
                                                        self._code, self._stream()), "sqfa_lbfgs_step_stats")
        return y, s, scal

    def direction(self, flat_grad, H_diag):
        q0 = flat_grad.neg()
        k = len(self.slots)
        if k == 0:
            return q0 * H_diag
        if self._lib is not None:
            import ctypes
            g = flat_grad.contiguous()
            d = torch.empty_like(g)
            H = H_diag if torch.is_tensor(H_diag) else (None if H_diag == 1 else g.new_tensor(float(H_diag)))
            if H is not None:
                H = H.to(g.dtype).reshape(1).contiguous()
            slots = (ctypes.c_int * k)(*self.slots)
            with torch.cuda.device(g.device):
                self._check(self._lib.sqfa_lbfgs_direction(self.S.data_ptr(), self.Y.data_ptr(), self.SY.data_ptr(), self.size,
                                                           g.numel(), slots, k, g.data_ptr(),
                                                           H.data_ptr() if H is not None else None, d.data_ptr(),
                                                           self._work.data_ptr(), self._code, self._stream()),
                            "sqfa_lbfgs_direction")
            return d
        idx = torch.as_tensor(self.slots, device=flat_grad.device)
        SY = self.SY.index_select(0, idx).index_select(1, idx)  # chronological k x k
        b0 = (self.S @ q0).index_select(0, idx)
        al = torch.linalg.solve_triangular(torch.triu(SY), b0.unsqueeze(1), upper=True).squeeze(1)
        al_ring = torch.zeros(self.size, dtype=al.dtype, device=al.device).index_copy_(0, idx, al)
        q = q0 - self.Y.t() @ al_ring
        r0 = q * H_diag
        rhs = torch.diagonal(SY) * al - (self.Y @ r0).index_select(0, idx)
        c = torch.linalg.solve_triangular(torch.tril(SY.t()), rhs.unsqueeze(1), upper=False).squeeze(1)
        c_ring = torch.zeros(self.size, dtype=c.dtype, device=c.device).index_copy_(0, idx, c)
        return r0 + self.S.t() @ c_ring


class CompactLBFGS(torch.optim.LBFGS):
    """Drop-in for torch.optim.LBFGS (same constructor)."""

    # None: gather the per-iteration decision scalars in one device-to-host copy when the
    # parameters live on a GPU; True/False force the choice (tests run the fused path on the CPU)
    fuse_readback = None
    # With the fused read-back, torch's test "g.d > -tolerance_change: stop before the step" is evaluated
    # AFTER the step and its closure, in the same copy (it only fires at convergence): when it does, the
    # parameters are restored and the extra evaluation is discarded -- same iterates, same counters, one host
    # synchronisation per iteration instead of two.  A closure with a `deferred` attribute (sqfa_amd/_optim.py)
    # hands its [loss, nan, inf] over on the device, so that copy is also the closure's: ONE per iteration.
    speculate_descent_test = True

    @torch.no_grad()
    def step(self, closure):
        group = self.param_groups[0]
        if group["line_search_fn"] is not None:
            return super().step(closure)
        raw_closure = closure
        closure = torch.enable_grad()(closure)
        lr = float(group["lr"])
        max_iter = group["max_iter"]
        max_eval = group["max_eval"]
        tolerance_grad = group["tolerance_grad"]
        tolerance_change = group["tolerance_change"]
        history_size = group["history_size"]

        state = self.state[self._params[0]]
        state.setdefault("func_evals", 0)
        state.setdefault("n_iter", 0)

        orig_loss = closure()
        loss = float(orig_loss.detach())
        current_evals = 1
        state["func_evals"] += 1

        flat_grad = self._gather_flat_grad()
        opt_cond = flat_grad.abs().max() <= tolerance_grad
        if opt_cond:
            return orig_loss

        d = state.get("d")
        t = state.get("t")
        hist = state.get("history")
        H_diag = state.get("H_diag")
        prev_flat_grad = state.get("prev_flat_grad")
        prev_loss = state.get("prev_loss")

        n_iter = 0
        ahead = None
        while n_iter < max_iter:
            n_iter += 1
            state["n_iter"] += 1

            if state["n_iter"] == 1:
                d = flat_grad.neg()
                hist = _History(history_size, flat_grad)
                H_diag = 1
            else:
                H_next = None
                if ahead is not None:
                    y, s, ys, H_next = ahead  # formed (and ys read back) together with the stopping-rule scalars
                    ahead = None
                else:
                    y = flat_grad.sub(prev_flat_grad)
                    s = d.mul(t)
                    ys = y.dot(s)
                if ys > 1e-10:
                    hist.push(y, s)
                    H_diag = H_next if H_next is not None else ys / y.dot(y)
                d = hist.direction(flat_grad, H_diag)

            if prev_flat_grad is None:
                prev_flat_grad = flat_grad.clone(memory_format=torch.contiguous_format)
            else:
                prev_flat_grad.copy_(flat_grad)
            prev_loss = loss

            if state["n_iter"] == 1:
                t = min(1.0, 1.0 / flat_grad.abs().sum()) * lr
            else:
                t = lr

            gtd = flat_grad.dot(d)
            fused = flat_grad.is_cuda if self.fuse_readback is None else self.fuse_readback
            # first iteration of a fit: t was just read back anyway; last iteration of a step: no closure follows
            speculate = fused and self.speculate_descent_test and n_iter != max_iter and state["n_iter"] != 1
            if not speculate:
                if gtd > -tolerance_change:
                    break
            else:
                backup = self._clone_param()

            ls_func_evals = 0
            self._add_grad(t, d)
            step_max = None
            if n_iter != max_iter:
                deferred = getattr(raw_closure, "deferred", None) if fused else None
                head = None
                with torch.enable_grad():
                    if deferred is not None:
                        head = deferred()  # [loss, nan, inf] on the device, no synchronisation
                    else:
                        loss_t = closure().detach()
                flat_grad = self._gather_flat_grad()
                if fused:
                    # one read-back for everything the stopping rules need (instead of three
                    # synchronisations): loss, max |g|, max |t d|
                    # ... and s.y of the NEXT iteration, whose sign decides the history update
                    native = hist.step_stats(flat_grad, prev_flat_grad, d, t) if flat_grad.is_cuda else None
                    if native is not None:
                        y_next, s_next, scal = native
                        H_next = scal[4]   # y.s / y.y, on the device: torch's H_diag if the pair is accepted
                        packed = torch.cat([scal[:3], gtd.reshape(1) if speculate else scal.new_zeros(1)])
                    else:
                        y_next, s_next, H_next = flat_grad.sub(prev_flat_grad), d.mul(t), None
                        parts = [flat_grad.abs().max(), s_next.abs().max(), y_next.dot(s_next),
                                 gtd if speculate else flat_grad.new_zeros(())]
                        packed = torch.stack(parts)
                    if head is not None:
                        packed = torch.cat([packed, head.detach().to(flat_grad.dtype).reshape(3)])
                    elif loss_t.device == flat_grad.device:
                        packed = torch.cat([packed, loss_t.to(flat_grad.dtype).reshape(1)])
                    vals = packed.tolist()
                    if speculate and vals[3] > -tolerance_change:
                        # torch stops BEFORE this step: undo it, forget the extra evaluation -- including its
                        # validity flags (torch.optim.LBFGS and the reference never evaluate that point, so a NaN
                        # there must not raise)
                        self._set_param(backup)
                        flat_grad = prev_flat_grad.clone(memory_format=torch.contiguous_format)
                        loss = prev_loss
                        break
                    if head is not None:
                        raw_closure.check_flags(vals[5], vals[6])  # raises like the synchronous closure would have
                    g_max, step_max = vals[0], vals[1]
                    ahead = (y_next, s_next, vals[2], H_next)
                    loss = vals[4] if len(vals) > 4 else float(loss_t)
                    opt_cond = g_max <= tolerance_grad
                else:
                    loss = float(loss_t)
                    opt_cond = flat_grad.abs().max() <= tolerance_grad
                ls_func_evals = 1

            current_evals += ls_func_evals
            state["func_evals"] += ls_func_evals

            if n_iter == max_iter:
                break
            if current_evals >= max_eval:
                break
            if opt_cond:
                break
            if (step_max if step_max is not None else d.mul(t).abs().max()) <= tolerance_change:
                break
            if abs(loss - prev_loss) < tolerance_change:
                break

        state["d"] = d
        state["t"] = t
        state["history"] = hist
        state["H_diag"] = H_diag
        state["prev_flat_grad"] = prev_flat_grad
        state["prev_loss"] = prev_loss
        return orig_loss
