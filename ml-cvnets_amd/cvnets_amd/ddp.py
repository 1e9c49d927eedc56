"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI on a communicator of this package's OWN
(cvnets_amd/comm.py -> cvh_comm_* in the C ABI: ncclGetUniqueId on rank 0, the 128 bytes carried by the launcher's TCP store,
ncclCommInitRank on every rank) — every collective of the path (bucket all-reduce, parameter / buffer broadcast, CLIP feature gather) is
enqueued on a HIP stream through it.  Replaces torch.nn.parallel.DistributedDataParallel at main_train.py:90-96 and the rendezvous of
utils/ddp_utils.py:47-89 (same contract: ``.module``, parameters/buffers broadcast from rank 0, gradients averaged).
torch.distributed remains the control plane the reference's engine already uses off this path (barriers, metric reductions), the carrier
of the unique id, and the data plane of the CPU tests (gloo) — and the fall-back if the communicator cannot be brought up (comm.init_default
says so on stderr; CVH_OWN_COMM=0 forces it).

Design for the 8-GPU xGMI mesh (7 links x ~153 GB/s per GPU, point-to-point):
  * every parameter gradient lives in a FLAT fp32 bucket (``p.grad`` is a view), so a bucket is one contiguous RCCL
    message — few, large collectives.  Bucket sizes follow what the xGMI mesh needs, not torch's 25 MB: an 8 MB message already runs
    at link bandwidth (~80 us over 7 x 153 GB/s links against ~20 us of launch latency), and the FIRST bucket is capped at 1 MB so that
    the exchange starts as soon as the classifier's gradients exist (MobileViT-S, 22.3 MB: 4 buckets of 1 / 8 / 8 / 5.3 MB; ViT-B: 45);
  * buckets are filled in reverse-registration order (≈ autograd order); when the last gradient of a bucket has been
    accumulated its all-reduce is enqueued on a SIDE HIP stream behind an event, overlapping the rest of backward — also inside a
    hipGraph capture (bench.py: the hooks run while the step is captured, so the fork / join around every bucket are graph edges and the
    replayed step overlaps by construction); ``overlap_report()`` says how many buckets started before the end of backward;
  * ``finish`` (queued as an autograd end-of-backward callback) makes the compute stream wait for the side stream; the mean is taken
    by the collective itself (``ReduceOp.AVG`` on RCCL; gloo, which has no AVG, divides afterwards — CPU tests only);
  * with hipGraph-captured steps (bench.py) hooks do not fire on replay, so ``allreduce_flat`` runs the same buckets
    right after the replay on the side stream (MobileViT-S: one 22 MB message behind a >= 20 ms step);
  * gradients must stay views of the buckets: a training loop that drops them (``optimizer.zero_grad(set_to_none=True)``, the default
    of torch >= 2 and of the reference's engine) gets them copied back in and re-pointed before every reduction;
  * float buffers (BatchNorm running statistics) are made views of ONE flat tensor at construction, so the per-forward rank-0
    broadcast of main_train.py's DDP (``broadcast_buffers``) is a single collective with no gather / scatter copies.

ORDER REQUIREMENT: wrap the model BEFORE anything records raw pointers into it — ``optim.EMABuffers``, a hipGraph capture, the fused
AdamW plan — because the constructor re-points gradient and float-buffer storage into the flat tensors (cvnets_amd/launch.py wraps right
after ``model.to(device)``, as main_train.py does).  A later ``module.to()`` / ``.float()`` detaches the buffer views; ``forward`` checks
that and falls back to per-buffer broadcasts.
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


def distributed_init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> int:
    """utils/ddp_utils.py:47-89: env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher),
    then one dummy all-reduce to create the communicator (ddp_utils.py:84-85).  Returns the rank."""
    if dist.is_initialized():
        return dist.get_rank()
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "30786")
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = device
    dist.init_process_group(backend=backend, init_method="env://", **kwargs)
    if device is not None and device.type == "cuda":
        from . import comm as _comm
        with torch.cuda.device(device):
            own = _comm.init_default(device)  # unique id through the store of the rendezvous above; self-tested (ddp_utils.py:84-85's dummy all-reduce)
        if own is not None:
            return dist.get_rank()
    t = torch.zeros(1, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    return dist.get_rank()


class _StreamOrdered:
    """what `_launch` stores in `bucket.work` for a collective issued through cvnets_amd.comm: it is ordered by the stream it was enqueued
    on (the side stream, which `finish` joins) — there is no host-side handle to wait for"""

    @staticmethod
    def wait():
        return True


_STREAM_ORDERED = _StreamOrdered()


class _BoundaryPreHook:
    """forward pre-hook of a top-level child (boundary-driven overlap).  A plain picklable object instead of a closure: copies of the module
    (torch.save(model), EMA's deepcopy) carry an INERT hook — no reference to the live wrapper, nothing to call."""

    def __init__(self, ddp, ci: int):
        import weakref
        self._ddp = weakref.ref(ddp)
        self.ci = ci

    def __call__(self, mod, args):
        ddp = self._ddp() if self._ddp is not None else None
        return None if ddp is None else ddp._pre_forward(self.ci, args)

    def __reduce__(self):
        return (_inert_boundary_hook, (self.ci,))

    def __deepcopy__(self, memo):
        return _inert_boundary_hook(self.ci)


class _BoundaryEndHook:
    """forward hook of the root module: the forward is over, its candidate order is known.  Inert on copies, like _BoundaryPreHook."""

    def __init__(self, ddp):
        import weakref
        self._ddp = weakref.ref(ddp)

    def __call__(self, mod, args, output):
        ddp = self._ddp() if self._ddp is not None else None
        if ddp is not None:
            ddp._post_forward()
        return None

    def __reduce__(self):
        return (_inert_end_hook, ())

    def __deepcopy__(self, memo):
        return _inert_end_hook()


def _inert_end_hook() -> "_BoundaryEndHook":
    h = _BoundaryEndHook.__new__(_BoundaryEndHook)
    h._ddp = None
    return h


def _inert_boundary_hook(ci: int) -> "_BoundaryPreHook":
    h = _BoundaryPreHook.__new__(_BoundaryPreHook)
    h._ddp, h.ci = None, ci
    return h


class _Bucket:
    def __init__(self, params: List[nn.Parameter], device):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.views = []
        off = 0
        for p in params:
            v = self.flat[off: off + p.numel()].view_as(p)
            self.views.append(v)
            p.grad = v
            off += p.numel()
        self.pending = len(params)
        self.work = None

    def adopt_stray_grads(self) -> None:
        """re-establish ``p.grad is a view of flat`` (a zero_grad(set_to_none=True) / fresh autograd allocation breaks it)"""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
                p.grad = v
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
                p.grad = v


_LIVE = weakref.WeakSet()  # the wrappers of this process (cvnets_amd.optim asks them before it updates parameters)


def exchange_pending() -> None:
    """called by the optimizer step: every wrapper whose last hooked forward has not been followed by a gradient exchange runs it now"""
    for d in list(_LIVE):
        d.assert_exchanged()


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, bucket_cap_mb: float = 8.0, overlap: bool = True, broadcast_buffers: bool = True,
                 process_group=None, force_collectives: Optional[bool] = None, first_bucket_mb: float = 1.0,
                 boundary_overlap: bool = False):
        super().__init__()
        _LIVE.add(self)
        self._expect_exchange = False
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        # a single-rank group normally skips every collective; `force_collectives` (or CVH_DDP_FORCE_COLLECTIVES=1) issues them anyway —
        # a world-size-1 RCCL all-reduce is an identity that still runs the communicator, the side stream and the event ordering
        # (tests/test_rccl_gpu.py executes the whole GPU side of this file on one GPU that way)
        if force_collectives is None:
            force_collectives = os.environ.get("CVH_DDP_FORCE_COLLECTIVES", "0") == "1"
        self.active = dist.is_initialized() and (self.world > 1 or bool(force_collectives))
        self.comm = None  # the package's own RCCL communicator (GPU runs); None: torch.distributed carries the data (gloo CPU tests, fall-back)
        self.overlap = overlap
        self.broadcast_buffers = broadcast_buffers
        params = [p for p in module.parameters() if p.requires_grad]
        self.device = params[0].device
        self.use_side_stream = self.device.type == "cuda"
        self.side_stream = torch.cuda.Stream(device=self.device) if self.use_side_stream else None
        if self.active and self.device.type == "cuda" and process_group is None:
            from . import comm as _comm
            with torch.cuda.device(self.device):
                self.comm = _comm.init_default(self.device)
        # float buffers -> views of one flat tensor (one broadcast message per forward, no torch.cat / copy-back)
        fbufs = [b for b in module.buffers() if b.dtype.is_floating_point and b.device == self.device]
        self.flat_buffers = None
        if fbufs:
            self.flat_buffers = torch.empty(sum(b.numel() for b in fbufs), dtype=fbufs[0].dtype, device=self.device) \
                if all(b.dtype == fbufs[0].dtype for b in fbufs) else None
        if self.flat_buffers is not None:
            off = 0
            for b in fbufs:
                self.flat_buffers[off: off + b.numel()].copy_(b.reshape(-1))
                b.data = self.flat_buffers[off: off + b.numel()].view_as(b)
                off += b.numel()
        # parameters + buffers start identical on every rank (DDP ctor broadcast)
        if self.active:
            for t in list(module.parameters()) + [b for b in module.buffers()]:
                if self.comm is not None and t.data.is_contiguous():
                    self.comm.broadcast(t.data, 0)
                else:
                    dist.broadcast(t.data, src=0, group=self.pg)
        self._avg_op = dist.ReduceOp.AVG if (dist.is_initialized() and dist.get_backend(self.pg) == "nccl") else None
        # buckets in reverse registration order: the last layers' gradients are ready first
        cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        first_cap = min(cap, int(first_bucket_mb * 1024 * 1024 / 4))
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in reversed(params):  # a bucket closes once it HOLDS its cap (a lone bias in front of a large weight is not a message)
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= (cap if self.buckets else first_cap):
                self.buckets.append(_Bucket(cur, self.device))
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, self.device))
        self._bucket_of = {}
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
                p.register_post_accumulate_grad_hook(self._hook)
        self._callback_task = None  # autograd graph-task id whose end-of-backward `finish` is queued (a dropped callback cannot go stale)
        self.hooks_enabled = True
        self.early_launches = 0   # buckets whose all-reduce started from a hook, i.e. before the end of backward (overlapped)
        self.late_launches = 0    # buckets that `finish` / `allreduce_flat` had to launch (no overlap: e.g. a parameter without gradient)
        self._warned_no_overlap = False
        self.finish_count = 0     # completed end-of-backward exchanges (hook- or boundary-driven)
        # Overlap WITHOUT per-parameter hooks (`boundary_overlap`): with in-place parameter gradients (ops.set_inplace_param_grads — the kernels add
        # straight into the bucket views, autograd never sees those gradients and post-accumulate hooks do not fire) a bucket is known to be
        # complete when backward has passed the INPUT of the earliest top-level child that owns one of its parameters: a tensor hook on that
        # input (attached by a forward pre-hook) flushes the deferred dW reductions queued so far and starts the bucket's all-reduce on the side
        # stream.  Requires the top-level children to run in registration order, each consuming the previous one's output (MobileViT /
        # MobileViTv2: conv_1, layer_1..5, conv_1x1_exp, classifier) and no parameter shared across children; the order is verified on every
        # forward and the feature switches itself off (with a warning) if it does not hold.  Works inside a hipGraph capture: the forks and
        # the join at `finish` become graph edges (bench.py).
        self.boundary_overlap = False
        self._boundary_buckets = {}
        if boundary_overlap:
            self._setup_boundaries()
        self._buf_span = None
        if self.flat_buffers is not None and fbufs:
            self._buf_span = tuple(fbufs)  # every re-pointed buffer is checked before the flat broadcast (a partial .float() / re-registration)

    # ---- boundary-driven overlap (in-place parameter gradients / captured steps) ----------------
    def _setup_boundaries(self):
        """Boundary candidates = the model's top-level children, a plain container (Sequential / ModuleList) of >= 2 parameterised blocks
        standing for its blocks (MobileViT: layer_2's three InvertedResiduals; ViT / CLIP towers: the transformer blocks).  Their EXECUTION
        order is what matters, and it is learnt: every forward records the order in which the candidates run (first guess: registration
        order — right for MobileViT / MobileViTv2; ViT registers `pos_embed` after the blocks it precedes), a forward that deviates from
        the known order exchanges its buckets at the end of backward and teaches the new order.  A bucket is tied to the earliest-run
        candidate that owns one of its parameters; when backward reaches that candidate's input, everything that ran after it has its
        gradient kernels enqueued (the autograd engine works a device's ready nodes newest-first, so this also holds for the two towers
        of CLIP) and the bucket's all-reduce forks onto the side stream.  Parameters of the root module (`cls_token`), parameters shared
        between candidates and candidates that run twice in one forward fall back to the end of backward."""
        cands = []
        for ch in self.module.children():
            subs = list(ch.children()) if isinstance(ch, (nn.Sequential, nn.ModuleList)) else []
            if sum(1 for m in subs if next(m.parameters(), None) is not None) >= 2:
                cands.extend(subs)
            else:
                cands.append(ch)
        owner = {}
        for ci, m in enumerate(cands):
            for p in m.parameters():
                owner[p] = ci if p not in owner else -1  # shared between candidates: complete only at the end of backward
        self._cands = cands
        self._bucket_owners = [{owner.get(p, -1) for p in b.params} for b in self.buckets]
        self._seq, self._fid, self._tracking, self._dup = [], 0, False, False
        self._fwd_ok = {}
        self._learn_order(tuple(range(len(cands))))
        self._boundary_handles = [m.register_forward_pre_hook(_BoundaryPreHook(self, ci)) for ci, m in enumerate(cands)]
        self._boundary_handles.append(self.module.register_forward_pre_hook(_BoundaryPreHook(self, -1)))
        self._boundary_handles.append(self.module.register_forward_hook(_BoundaryEndHook(self)))
        self.boundary_overlap = True

    def _learn_order(self, order):
        self._order = tuple(order)
        rank = {ci: r for r, ci in enumerate(self._order)}
        self._boundary_buckets = {}
        for b, owners in zip(self.buckets, self._bucket_owners):
            if -1 in owners or any(ci not in rank for ci in owners):
                continue  # launched by `finish`
            self._boundary_buckets.setdefault(min(owners, key=rank.get), []).append(b)

    def _pre_forward(self, ci: int, args):
        if not self.boundary_overlap:
            return None
        if ci < 0:  # the root module starts a forward
            # (a forward INSIDE a backward pass is a checkpoint recomputation: it neither defines an order nor gets boundaries)
            self._tracking = not self._in_no_sync and torch._C._current_graph_task_id() < 0
            if self._tracking:
                self._fid += 1
                self._seq, self._dup = [], False
            return None
        if not self._tracking:
            return None
        if ci in self._seq:
            self._dup = True
        self._seq.append(ci)
        n = len(self._seq)
        if (not self._dup and tuple(self._seq) == self._order[:n] and ci in self._boundary_buckets and self.active and self.boundary_enabled
                and torch.is_grad_enabled() and args and isinstance(args[0], torch.Tensor) and args[0].requires_grad):
            args[0].register_hook(lambda g, ci=ci, fid=self._fid: self._boundary(ci, fid))
        return None

    def _post_forward(self):
        if not (self.boundary_overlap and self._tracking):
            return
        self._tracking = False
        seq = tuple(self._seq)
        ok = not self._dup and seq == self._order
        self._fwd_ok = {self._fid: ok}  # only the newest forward can have live boundary hooks worth honouring
        if not ok and not self._dup and len(set(seq)) == len(seq):
            self._learn_order(seq)  # this step exchanges at the end of backward; the next one uses the order just seen
        elif self._dup and not self._warned_dup:
            self._warned_dup = True
            import warnings
            warnings.warn("cvnets_amd.ddp: a boundary module ran twice in one forward: its buckets are exchanged at the end of backward")

    boundary_enabled = True
    _in_no_sync = False
    _warned_dup = False

    def _boundary(self, ci: int, fid: int):
        """backward has produced the gradient w.r.t. the input of candidate `ci`: every parameter of the candidates that ran at or after it has
        its gradient kernels (or deferred partial sums) enqueued"""
        if not (self.active and self.boundary_overlap and self.boundary_enabled) or self._in_no_sync:
            return None  # (no_sync: the micro-steps of a gradient accumulation exchange nothing — torch DDP's contract)
        # the end-of-backward exchange is queued by WHICHEVER hook of this backward fires first — also on the steps on which the boundary
        # order is still being learned (`_fwd_ok` false): with in-place parameter gradients no post-accumulate hook exists to do it
        tid = torch._C._current_graph_task_id()
        if tid >= 0 and self._callback_task != tid:
            if self._callback_task is not None:
                self._discard_stale_task()
            self._callback_task = tid
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        if not self._fwd_ok.get(fid, False):
            return None
        from . import ops
        ops.flush_deferred_reductions()  # the partial sums queued so far all belong to candidates that ran at or after this one
        for b in self._boundary_buckets.get(ci, ()):
            if b.work is None:
                self.early_launches += 1
                self._launch(b)
        return None

    def _discard_stale_task(self):
        """A new autograd graph task reached the hooks while `finish` of another one is still queued.  Normally that other backward raised
        and its callback never ran: drop its bucket state.  A NESTED backward over DDP parameters (torch.utils.checkpoint with
        use_reentrant=True, autograd.grad inside a custom backward) looks the same and is NOT supported — its gradients would be exchanged
        per inner task: say so instead of reducing incomplete buckets silently (ops.checkpoint uses use_reentrant=False)."""
        started = [b for b in self.buckets if b.work is not None or b.pending != len(b.params)]
        if started:
            import warnings
            warnings.warn("cvnets_amd.ddp: a backward pass started while the gradient exchange of another one was unfinished "
                          f"({len(started)} bucket(s) partly filled).  If the previous backward raised, this is the clean-up; nested / "
                          "re-entrant backward passes over DDP parameters are not supported (use use_reentrant=False).")
        for b in self.buckets:
            b.work = None
            b.pending = len(b.params)

    @contextlib.contextmanager
    def no_sync(self):
        """torch DDP's contract: backward passes inside the context accumulate locally, the first backward after it reduces the
        accumulated gradients (gradient accumulation, engine/training_engine.py:221,289: `accum_freq` micro-steps per update).
        Enter it OUTSIDE a backward pass (a backward already in progress has its exchange queued)."""
        if torch._C._current_graph_task_id() >= 0:
            raise RuntimeError("cvnets_amd.ddp.no_sync() must be entered outside a backward pass")
        prev = (self.hooks_enabled, self._in_no_sync)
        self.hooks_enabled, self._in_no_sync = False, True
        try:
            yield
        finally:
            self.hooks_enabled, self._in_no_sync = prev

    # ---- autograd-driven path (eager) --------------------------------------------------------
    def _hook(self, p: nn.Parameter):
        if not self.hooks_enabled or not self.active:
            return
        tid = torch._C._current_graph_task_id()
        if self._callback_task != tid:
            if self._callback_task is not None:  # the backward that queued `finish` died before it ran: start from a clean slate
                self._discard_stale_task()
            self._callback_task = tid
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0 and self.overlap:
            from . import ops
            if ops._INPLACE_PARAM_GRADS:
                # (the hook of a parameter whose gradient the kernels added in place fires with an undefined gradient on current PyTorch:)
                # the partial sums of its weight gradient may still sit in the deferred-reduction queue — reduce them before the message leaves
                ops.flush_deferred_reductions()
            self.early_launches += 1
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.adopt_stray_grads()
        op = self._avg_op if self._avg_op is not None else dist.ReduceOp.SUM
        if self.use_side_stream:
            self.side_stream.wait_stream(torch.cuda.current_stream(self.device))  # fork: everything enqueued so far precedes the message
            if self.comm is not None:
                self.comm.all_reduce(b.flat, average=True, stream=self.side_stream)  # ncclAllReduce(ncclAvg) on the side stream
                b.work = _STREAM_ORDERED
            else:
                with torch.cuda.stream(self.side_stream):
                    b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)

    def _average(self):
        if self._avg_op is None and self.comm is None:  # gloo (CPU tests): no AVG reduction
            for b in self.buckets:
                b.flat.div_(self.world)

    def finish(self):
        """end of backward: launch whatever was not launched, wait, average."""
        from . import ops
        # whichever end-of-backward callback runs first: no partial sum of THIS backward may still be queued
        ops.flush_deferred_reductions(self._callback_task)
        self.finish_count += 1
        late = 0
        for b in self.buckets:
            if b.work is None:
                late += 1
                self._launch(b)
        self.late_launches += late
        if late and self.overlap and len(self.buckets) > 1 and late == len(self.buckets) and not self._warned_no_overlap:
            # every bucket waited for the end of backward: some parameter of each bucket produced no gradient (the
            # find_unused_parameters case of main_train.py:95) — results are correct, nothing overlapped; say so once
            self._warned_no_overlap = True
            import warnings
            warnings.warn("cvnets_amd.ddp: no gradient bucket was complete before the end of backward (parameters without gradients?): "
                          "the all-reduce did not overlap with backward")
        for b in self.buckets:
            b.work.wait()
            b.work = None
            b.pending = len(b.params)
        if self.use_side_stream:
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
        self._average()
        self._callback_task = None
        self._expect_exchange = False

    def assert_exchanged(self):
        """Safety net of the in-place-gradient mode, called by cvnets_amd.optim.AdamW.step (exchange_pending below): a forward whose outputs
        were hooked must have been followed by a gradient exchange before the parameters are updated.  If the optimizer steps and no hook
        of that forward's backward has fired, the ranks would diverge silently: exchange now (correct, not overlapped) and say so once.
        (A forward that was never differentiated ends up here too: an extra all-reduce of unchanged buckets, harmless.)"""
        if not (self.active and getattr(self, "_expect_exchange", False)) or self._in_no_sync:
            return
        import warnings
        if not getattr(self, "_warned_late_exchange", False):
            self._warned_late_exchange = True
            warnings.warn("cvnets_amd.ddp: the optimizer steps although no gradient exchange has run since the last hooked forward (no hook on "
                          "the wrapped model's outputs fired during backward); exchanging the gradients now, without overlap")
        self.allreduce_flat()
        self._expect_exchange = False

    def overlap_report(self) -> dict:
        """how the gradient exchange has been scheduled so far: buckets, their sizes, launches before / at the end of backward"""
        return {"buckets": len(self.buckets), "bucket_mb": [round(b.numel * 4 / 2 ** 20, 2) for b in self.buckets],
                "launched_during_backward": self.early_launches, "launched_at_end_of_backward": self.late_launches}

    # ---- copies (EMA does deepcopy(model) on the WRAPPED model, cvnets/misc/averaging_utils.py:33; torch.save(model) pickles it) ----
    # A copy is a passive holder of a copy of `.module`: no process group, no side stream (HIP streams cannot be copied or pickled), no
    # buckets, no hooks — exactly what the reference needs from it (`.module`, parameters(), state_dict(), eval()).
    def _passive_copy(self, module: nn.Module) -> "DistributedDataParallel":
        new = DistributedDataParallel.__new__(DistributedDataParallel)
        nn.Module.__init__(new)
        new.module = module
        new.pg, new.world, new.active, new.overlap, new.broadcast_buffers = None, 1, False, False, False
        new.device = self.device
        new.use_side_stream, new.side_stream, new.flat_buffers, new._buf_span = False, None, None, None
        new._avg_op, new.buckets, new._bucket_of, new.comm = None, [], {}, None
        new._callback_task, new.hooks_enabled = None, False
        new.early_launches = new.late_launches = new.finish_count = 0
        new._warned_no_overlap = True
        new.boundary_overlap, new._boundary_buckets = False, {}
        new.training = self.training
        return new

    def __deepcopy__(self, memo):
        import copy
        new = self._passive_copy(copy.deepcopy(self.module, memo))
        memo[id(self)] = new
        return new

    def __reduce__(self):
        return (_rebuild_passive, (self.module, self.training))

    # ---- explicit path (after a hipGraph replay) ---------------------------------------------
    def allreduce_flat(self):
        if not self.active:
            return
        from . import ops
        ops.finish_backward()  # deferred dW reductions / side-stream joins of a backward whose callback was lost must not leak in
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            b.work.wait()
            b.work = None
        if self.use_side_stream:
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
        self._average()

    def zero_grad(self, set_to_none: bool = False):
        """gradients are views of the flat buckets: zero in place (never set to None)."""
        for b in self.buckets:
            b.adopt_stray_grads()
            b.flat.zero_()

    def grad_bytes(self) -> int:
        return sum(b.numel for b in self.buckets) * 4

    def _buffers_still_flat(self) -> bool:
        """the float buffers were re-pointed into `flat_buffers` at construction; a later `module.to()` / `.float()` detaches them"""
        if self._buf_span is None:
            return False
        lo, hi = self.flat_buffers.data_ptr(), self.flat_buffers.data_ptr() + self.flat_buffers.numel() * self.flat_buffers.element_size()
        # every float buffer the module holds NOW (a re-registered or converted buffer is a different tensor object than the one captured)
        n = 0
        for b in self.module.buffers():
            if b.dtype.is_floating_point:
                n += 1
                if not (lo <= b.data_ptr() < hi):
                    return False
        return n == len(self._buf_span)

    def forward(self, *args, **kwargs):
        if self.broadcast_buffers and self.active and self.training:
            if self.flat_buffers is not None and self._buffers_still_flat():
                if self.comm is not None:
                    self.comm.broadcast(self.flat_buffers, 0)  # the buffers ARE views of this tensor: one message on the compute stream
                else:
                    dist.broadcast(self.flat_buffers, src=0, group=self.pg)
            else:
                for b in self.module.buffers():
                    if b.dtype.is_floating_point:
                        if self.comm is not None and b.data.is_contiguous():
                            self.comm.broadcast(b.data, 0)
                        else:
                            dist.broadcast(b.data, src=0, group=self.pg)
        out = self.module(*args, **kwargs)
        if self.active and self.hooks_enabled and not self._in_no_sync and torch.is_grad_enabled():
            from . import ops
            if ops._INPLACE_PARAM_GRADS:
                # in-place parameter gradients never reach autograd, so no post-accumulate hook will queue `finish`: a hook on the output
                # does it when backward starts (buckets not started by a boundary are then exchanged at the end of backward)
                # EVERY output that requires grad gets the hook (a dict / tuple output whose first tensor is off the loss path — an augmented
                # input, an auxiliary head — must not decide whether the ranks exchange gradients); `_backward_started` is idempotent per backward
                hooked = 0
                for t in _tensors_of(out):
                    if t.requires_grad:
                        t.register_hook(self._backward_started)
                        hooked += 1
                self._expect_exchange = hooked > 0
        return out

    def _backward_started(self, grad):
        tid = torch._C._current_graph_task_id()
        if tid >= 0 and self._callback_task != tid and self.active and self.hooks_enabled and not self._in_no_sync:
            if self._callback_task is not None:
                self._discard_stale_task()
            self._callback_task = tid
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        return None


def _tensors_of(out):
    """every tensor of a (nested) tensor / dict / list / tuple output"""
    if isinstance(out, torch.Tensor):
        yield out
    elif isinstance(out, dict):
        for o in out.values():
            yield from _tensors_of(o)
    elif isinstance(out, (list, tuple)):
        for o in out:
            yield from _tensors_of(o)


def _first_tensor(out):
    for t in _tensors_of(out):
        return t
    return None


def _rebuild_passive(module: nn.Module, training: bool) -> DistributedDataParallel:
    new = DistributedDataParallel.__new__(DistributedDataParallel)
    nn.Module.__init__(new)
    new.device = next(module.parameters()).device
    new = DistributedDataParallel._passive_copy(new, module)
    new.training = training
    return new


def _own_comm(x: torch.Tensor, group):
    """the package's communicator for a collective over the DEFAULT group on GPU float tensors (None: torch.distributed carries it)"""
    if group is not None or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        return None
    from . import comm as _comm
    c = _comm.default()
    if c is None and dist.is_initialized():
        with torch.cuda.device(x.device):
            c = _comm.init_default(x.device)
    return c if (c is not None and c.world == dist.get_world_size()) else None


class _AllGatherWithGrad(torch.autograd.Function):
    """utils/tensor_utils.py:121-122 (gather_all_features -> torch.distributed.nn.all_gather): forward = all-gather along dim 0
    (one RCCL collective into a contiguous [W*N, d] buffer), backward = reduce-scatter(sum) of the gathered gradient — every rank
    receives the sum over ranks of the gradient slices that belong to its own rows."""

    @staticmethod
    def forward(ctx, x, group):
        x = x.contiguous()
        world = dist.get_world_size(group)
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        ctx.comm = _own_comm(x, group)
        if ctx.comm is not None:
            ctx.comm.all_gather(out, x)
        else:
            dist.all_gather_into_tensor(out, x, group=group)
        ctx.group = group
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        world, rank = dist.get_world_size(ctx.group), dist.get_rank(ctx.group)
        if dist.get_backend(ctx.group) == "gloo":  # gloo has no reduce-scatter: all-reduce, keep the own slice (CPU tests only)
            dist.all_reduce(g, group=ctx.group)
            return g[rank * ctx.n:(rank + 1) * ctx.n].clone(), None
        out = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        if ctx.comm is not None:
            ctx.comm.reduce_scatter(out, g)
        else:
            dist.reduce_scatter_tensor(out, g, group=ctx.group)
        return out, None


def gather_all_features(features: torch.Tensor, group=None, force: Optional[bool] = None) -> torch.Tensor:
    """[N, d] on every rank -> [W*N, d], differentiable (ContrastiveLossClip, contrastive_loss_clip.py:144-172).  `force` (or
    CVH_DDP_FORCE_COLLECTIVES=1) issues the collectives on a single-rank group too (identity; executes the RCCL path on one GPU)."""
    if force is None:
        force = os.environ.get("CVH_DDP_FORCE_COLLECTIVES", "0") == "1"
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return features
    return _AllGatherWithGrad.apply(features, group)
