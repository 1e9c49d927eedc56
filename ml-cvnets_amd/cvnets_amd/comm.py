"""The hot path's OWN communicator: RCCL over xGMI through the C ABI (cvh_comm_*, csrc/comm.hip), not torch.distributed's ProcessGroupNCCL.

What the reference reaches through torch.distributed on this path and what carries it here:

  utils/ddp_utils.py:63-89      init_process_group + dummy all_reduce   -> Communicator.from_store: rank 0 draws the RCCL unique id, the
                                                                            launcher's key-value store (the TCP store of the env://
                                                                            rendezvous) carries its 128 bytes, every rank joins
  main_train.py:91-96           DistributedDataParallel: gradient mean   -> Communicator.all_reduce(bucket, average=True) on ddp's side stream
  (its ctor / forward)          parameter + buffer broadcast from rank 0 -> Communicator.broadcast
  contrastive_loss_clip.py:144  gather_all_features (all_gather + grad)  -> Communicator.all_gather / reduce_scatter

Every collective is enqueued on a HIP stream (torch's current stream unless one is given): stream-ordered, no host wait, capturable into
the step's hipGraph.  torch.distributed stays what the reference's engine uses it for off this path (barriers, metric reductions); CPU
tests run the same host logic over gloo (`Communicator` is GPU-only — `default()` is None there and callers keep the torch path).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

_BYTES = 100  # cvh_comm dtype code for "opaque bytes" (broadcast / all-gather of tensors that are neither float32 nor bfloat16)
_ID_BYTES = 128


def available() -> bool:
    """librccl can be opened on this machine (does not create anything)"""
    try:
        return bool(_lib.load().cvh_comm_available())
    except Exception:
        return False


def _code(t: torch.Tensor, reduce: bool) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    if reduce:
        raise RuntimeError(f"cvnets_amd.comm reduces float32 / bfloat16 tensors only, not {t.dtype}")
    return _BYTES


def _check(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("cvnets_amd.comm moves device memory only (CPU tests use torch.distributed / gloo)")
    if not t.is_contiguous():
        raise RuntimeError("cvnets_amd.comm needs contiguous tensors (a collective is one message)")


class Communicator:
    """One RCCL communicator of one process (= one GPU)."""

    def __init__(self, world: int, rank: int, unique_id: bytes, device: Optional[torch.device] = None):
        if len(unique_id) != _ID_BYTES:
            raise ValueError("an RCCL unique id is 128 bytes")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.world, self.rank = int(world), int(rank)
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, _ID_BYTES)
        with torch.cuda.device(self.device):
            _lib.call("cvh_comm_init", ctypes.byref(handle), self.world, self.rank, buf)
        self._h = handle

    # ---- construction -----------------------------------------------------------------------------------------------------------------
    @staticmethod
    def new_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(_ID_BYTES)
        _lib.call("cvh_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_store(cls, store, world: int, rank: int, device=None, key: str = "cvnets_amd/comm/0") -> "Communicator":
        """utils/ddp_utils.py:63-89 without a torch process group on the data path: `store` is any torch.distributed.Store (the TCP store
        the env:// rendezvous opened, or one created for the purpose); rank 0 publishes the unique id under `key`, the others block on it."""
        return cls(world, rank, cls.exchange_unique_id(store, rank, key), device)

    @classmethod
    def exchange_unique_id(cls, store, rank: int, key: str = "cvnets_amd/comm/0") -> bytes:
        """the rendezvous half of `from_store` (host only: tests/test_ddp_cpu.py runs it on two gloo ranks): rank 0 draws the id and publishes
        it, every other rank blocks on the key (the store's own timeout applies)"""
        if rank == 0:
            uid = cls.new_unique_id()
            store.set(key, uid)
        else:
            uid = bytes(store.get(key))
        if len(uid) != _ID_BYTES:
            raise RuntimeError(f"rendezvous key {key!r} holds {len(uid)} bytes, not an RCCL unique id")
        return uid

    @classmethod
    def single(cls, device=None) -> "Communicator":
        """a world of one (every collective is an identity that still runs RCCL's kernels on the stream): tests, CVH_DDP_FORCE_COLLECTIVES"""
        return cls(1, 0, cls.new_unique_id(), device)

    # ---- collectives ------------------------------------------------------------------------------------------------------------------
    def _stream(self, stream) -> int:
        return (torch.cuda.current_stream(self.device) if stream is None else stream).cuda_stream

    def all_reduce(self, t: torch.Tensor, average: bool = False, stream=None) -> torch.Tensor:
        """in place: sum (or mean, formed by the collective: ncclAvg) over ranks"""
        _check(t)
        _lib.call("cvh_comm_allreduce", self._h, t.data_ptr(), t.numel(), _code(t, True), 1 if average else 0, self._stream(stream))
        return t

    def broadcast(self, t: torch.Tensor, root: int = 0, stream=None) -> torch.Tensor:
        _check(t)
        code = _code(t, False)
        n = t.numel() * t.element_size() if code == _BYTES else t.numel()
        _lib.call("cvh_comm_broadcast", self._h, t.data_ptr(), n, code, int(root), self._stream(stream))
        return t

    def all_gather(self, out: torch.Tensor, x: torch.Tensor, stream=None) -> torch.Tensor:
        """out[world * n] = every rank's x[n] in rank order (dim 0)"""
        _check(out), _check(x)
        if out.numel() != self.world * x.numel() or out.dtype != x.dtype:
            raise RuntimeError("all_gather: out must hold world x the elements of x, same dtype")
        code = _code(x, False)
        n = x.numel() * x.element_size() if code == _BYTES else x.numel()
        _lib.call("cvh_comm_allgather", self._h, x.data_ptr(), out.data_ptr(), n, code, self._stream(stream))
        return out

    def reduce_scatter(self, out: torch.Tensor, x: torch.Tensor, stream=None) -> torch.Tensor:
        """out[n] = this rank's slice of the sum over ranks of x[world * n]"""
        _check(out), _check(x)
        if x.numel() != self.world * out.numel() or out.dtype != x.dtype:
            raise RuntimeError("reduce_scatter: x must hold world x the elements of out, same dtype")
        _lib.call("cvh_comm_reducescatter", self._h, x.data_ptr(), out.data_ptr(), out.numel(), _code(x, True), self._stream(stream))
        return out

    def self_test(self) -> None:
        """one all-reduce and one broadcast with known answers (run once after creation: a communicator that came up with the wrong peers,
        or a stack on which the collectives do not complete, fails here and not inside a training step)"""
        t = torch.full((256,), float(self.rank + 1), dtype=torch.float32, device=self.device)
        self.all_reduce(t)
        b = torch.full((64,), float(self.rank + 7), dtype=torch.float32, device=self.device)
        self.broadcast(b, 0)
        torch.cuda.synchronize(self.device)
        want = self.world * (self.world + 1) / 2
        if not (bool((t == want).all()) and bool((b == 7.0).all())):
            raise RuntimeError(f"cvnets_amd.comm self-test failed on rank {self.rank} of {self.world}: all-reduce {float(t[0])} (expected {want}), "
                               f"broadcast {float(b[0])} (expected 7)")

    def destroy(self) -> None:
        if self._h is not None and self._h.value:
            _lib.call("cvh_comm_destroy", self._h)
        self._h = None

    def __reduce__(self):
        raise TypeError("a Communicator is process-local and cannot be pickled")


def counters(reset: bool = False):
    """(all-reduce, broadcast, all-gather, reduce-scatter) launches issued through this library so far"""
    out = (ctypes.c_longlong * 4)()
    _lib.call("cvh_comm_counters", 1 if reset else 0, out)
    return tuple(int(v) for v in out)


# ---- the process-wide default communicator (what ddp.DistributedDataParallel and gather_all_features use) ---------------------------------
_default: Optional[Communicator] = None
_generation = 0
_attempted = False  # init_default has made its (collective) decision for this process


def default() -> Optional[Communicator]:
    return _default


def init_default(device=None, store=None, world: Optional[int] = None, rank: Optional[int] = None) -> Optional[Communicator]:
    """Create (once) the default communicator of this process.  Rendezvous: `store`, else the store of an initialised torch.distributed
    default group (the launcher's env:// TCP store), else — a world of one — none at all.  Returns None (and leaves the callers on
    torch.distributed) when the device is not a GPU, librccl is missing, CVH_OWN_COMM=0, or bring-up / self-test fails: the failure is
    reported once on stderr, never silently."""
    global _default, _generation, _attempted
    if _default is not None:
        return _default
    if _attempted:  # the fall-back decision is made ONCE per process: a failed bring-up is not retried by every DDP constructor / CLIP gather
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        return None  # (a CPU process group has no use for it: every rank of such a job returns here alike)
    _attempted = True
    import torch.distributed as dist

    # CVH_OWN_COMM=0 / a missing librccl are LOCAL facts: they enter the collective vote below as "could not bring it up" instead of returning
    # before it (a rank that left early would leave the others blocked in the store exchange and the MIN all-reduce)
    local_ok = os.environ.get("CVH_OWN_COMM", "1") != "0" and available()

    if world is None or rank is None:
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
        else:
            world, rank = 1, 0
    import sys

    c = None
    try:
        if not local_ok:
            raise RuntimeError("switched off (CVH_OWN_COMM=0)" if os.environ.get("CVH_OWN_COMM", "1") == "0" else "librccl not found")
        if world == 1:
            c = Communicator.single(dev)
        else:
            if store is None:
                if not (dist.is_available() and dist.is_initialized()):
                    raise RuntimeError("no store for the unique-id exchange: initialise torch.distributed (env://) or pass one")
                store = dist.distributed_c10d._get_default_store()
            c = Communicator.from_store(store, world, rank, dev, key=f"cvnets_amd/comm/{_generation}")
        c.self_test()
    except Exception as e:  # never silent: the run goes on over torch.distributed, and says so
        sys.stderr.write(f"[cvnets_amd.comm] own RCCL communicator unavailable ({type(e).__name__}: {e}); falling back to torch.distributed\n")
        if c is not None:
            try:
                c.destroy()
            except Exception:
                pass
        c = None
    _generation += 1
    if world > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size() == world:
        # the decision is COLLECTIVE: a communicator that came up on some ranks only would leave the ranks on different data planes
        ok = torch.tensor([1 if c is not None else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and c is not None:
            sys.stderr.write("[cvnets_amd.comm] another rank could not bring its communicator up; every rank falls back to torch.distributed\n")
            c.destroy()
            c = None
    _default = c
    return c


def destroy_default() -> None:
    global _default, _attempted
    _attempted = False  # a later init_default may try again (tests; a re-initialised process group)
    if _default is not None:
        _default.destroy()
        _default = None
