"""Autograd operators of the hot path: thin torch.autograd.Function wrappers that enqueue the HIP kernels
of libcvnets_hip.so (through ctypes, on torch's current stream) for forward AND backward.

PyTorch is used here only for device memory (torch.empty -> caching allocator), stream identity and the
autograd graph; no ATen compute kernel is on the path (exceptions, all off the per-element path, are marked
"plumbing": zero-fills of parameter-gradient buffers and tiny constant vectors).

Tensor conventions
  * feature maps: logical [B, C, H, W] torch tensors whose memory is NHWC (channels_last strides), dtype =
    compute dtype (float32 or bfloat16), C % 8 == 0;
  * token matrices: contiguous [rows, C];
  * parameters / statistics / parameter gradients: float32 in torch's own layouts.
"""
from __future__ import annotations

import ctypes
import contextlib
import os

import itertools
import weakref
from typing import Optional, Tuple

import torch

from . import _lib

ACT_NONE, ACT_SILU, ACT_GELU, ACT_RELU = 0, 1, 2, 3
ACT_DERIV, ACT_GELU_D = 17, 18  # include/cvnets_hip.h: "multiply by the stored derivative" / "GELU forward that stores its derivative"
_COMPUTE_DTYPE: Optional[torch.dtype] = None


def set_compute_dtype(dtype: Optional[torch.dtype]) -> None:
    """float32 / bfloat16, or None = follow torch autocast (bf16 under autocast(bfloat16), else fp32) — the
    behaviour of the reference under engine/utils.py:19-36 autocast_fn."""
    global _COMPUTE_DTYPE
    assert dtype in (None, torch.float32, torch.bfloat16)
    _COMPUTE_DTYPE = dtype


def compute_dtype() -> torch.dtype:
    if _COMPUTE_DTYPE is not None:
        return _COMPUTE_DTYPE
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.bfloat16:
            return torch.bfloat16
        if dt == torch.float16:
            # The reference's YAMLs say `mixed_precision: true` and leave `common.mixed_precision_dtype` at its default, float16
            # (options/opts.py:126-131).  There are no float16 kernels here: such regions are computed with bfloat16 storage / fp32
            # accumulation (same exponent range as fp32, so the engine's GradScaler — whatever its scale — cannot overflow them; 8 instead of
            # 11 significand bits in stored activations).  Said once, not silently; CVH_STRICT_AUTOCAST=1 restores the refusal.
            if os.environ.get("CVH_STRICT_AUTOCAST", "0") == "1":
                raise RuntimeError("cvnets_amd has no float16 kernels (set common.mixed_precision_dtype=bfloat16, or unset CVH_STRICT_AUTOCAST)")
            global _WARNED_FP16
            if not _WARNED_FP16:
                _WARNED_FP16 = True
                import warnings
                warnings.warn("cvnets_amd: float16 autocast regions are computed in bfloat16 storage / float32 accumulation "
                              "(no float16 kernels on this path; set common.mixed_precision_dtype=bfloat16 to say so in the config)")
            return torch.bfloat16
        raise RuntimeError(f"cvnets_amd: unsupported autocast dtype {dt}")
    return torch.float32


_WARNED_FP16 = False


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f"unsupported activation dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_dev(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("cvnets_amd ops run on the GPU only (HIP kernels; there is no CPU fallback)")


def _f32(n, device, *shape):
    return torch.empty((n, *shape) if shape else (n,), dtype=torch.float32, device=device)


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------
def nhwc_empty(B, C, H, W, dtype, device) -> torch.Tensor:
    return torch.empty((B, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def is_nhwc(x: torch.Tensor) -> bool:
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


def as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Gradient tensors handed to us by autograd are normally already NHWC; otherwise repack (plumbing)."""
    if is_nhwc(x):
        return x
    return x.contiguous(memory_format=torch.channels_last)


def tokens_of(x: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] NHWC feature map -> its [B*H*W, C] token matrix (a view)."""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C)


def fmap_of(t: torch.Tensor, B, H, W) -> torch.Tensor:
    return t.view(B, H, W, t.shape[-1]).permute(0, 3, 1, 2)


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, dtype: torch.dtype):
        _check_dev(x)
        B, C, H, W = x.shape
        Cp = pad8(C)
        ctx.meta = (B, C, H, W, Cp)
        xin = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()  # plumbing
        out = nhwc_empty(B, Cp, H, W, dtype, x.device)
        _lib.call("cvh_nchw_to_nhwc", _dt(out), _p(xin), _p(out), B, C, H, W, Cp, _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, Cp = ctx.meta
        g = as_nhwc(g)
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
        _lib.call("cvh_nhwc_to_nchw", _dt(g), _p(g), _p(dx), B, C, H, W, Cp, _stream())
        return dx, None


def to_nhwc(x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Entry of the hot path: NCHW float32 image batch -> NHWC compute-dtype tensor (channels padded to 8)."""
    dtype = dtype or compute_dtype()
    if x.dtype == dtype and is_nhwc(x) and x.shape[1] % 8 == 0:
        return x
    return _ToNHWC.apply(x, dtype)


def nhwc_to_nchw_f32(x: torch.Tensor, C: Optional[int] = None) -> torch.Tensor:
    B, Cs, H, W = x.shape
    C = C or Cs
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _lib.call("cvh_nhwc_to_nchw", _dt(x), _p(x), _p(out), B, C, H, W, Cs, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# dropout state: one device-resident 64-bit seed per device; masks are f(seed, stream_id, element)
# ------------------------------------------------------------------------------------------------
class _Counter:
    """dropout stream ids: one per op call, in call order (a plain counter whose position `checkpoint` can pin and restore)"""

    def __init__(self, start: int = 1):
        self.value = start

    def __next__(self) -> int:
        v = self.value
        self.value += 1
        return v


_stream_ids = _Counter(1)
_seeds = {}      # device index -> the generator state (advanced once per training forward)
_seed_snap = {}  # device index -> snapshot of the state taken by the CURRENT forward


def _seed_state(device) -> torch.Tensor:
    key = torch.device(device).index or 0
    if key not in _seeds:
        _seeds[key] = torch.tensor([0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device) + torch.initial_seed()
    return _seeds[key]


def dropout_seed(device) -> torch.Tensor:
    """The seed tensor the dropout masks of the CURRENT forward are drawn from.  Ops save it for their backward, so a later forward
    (gradient accumulation over several forwards, a second model in train mode, a teacher) cannot change the mask a pending backward
    regenerates: every training forward snapshots its own copy (`advance_dropout_seed`)."""
    key = torch.device(device).index or 0
    snap = _seed_snap.get(key)
    return snap if snap is not None else _seed_state(device)


def release_capture_state() -> None:
    """Drop the module-level references to tensors that were allocated INSIDE a hipGraph capture (the per-forward dropout seed snapshot lives in
    the capturing graph's private memory pool).  Call it before destroying such a graph: a tensor that outlives the pool it was allocated
    from keeps a block of a dead pool alive (tests that capture and discard several graphs in one process)."""
    _seed_snap.clear()


def advance_dropout_seed(device) -> None:
    """Called once at the start of every training forward (captured into the step's hipGraph, so replays draw
    fresh masks): advances the generator state and snapshots it for this forward."""
    state = _seed_state(device)
    _lib.call("cvh_seed_advance", _p(state), _stream())
    _seed_snap[torch.device(device).index or 0] = state.clone()  # plumbing: 8 bytes


def next_stream_id() -> int:
    return next(_stream_ids) & 0xFFFFFFFF


# Debug export of the dropout draws (tests only: tests/test_zz_dropout_step_gpu.py feeds the oracle the masks the kernels drew).
_dropout_trace = None  # None, or a list that every dropout site of a forward appends (kind, stream id, p, shape of the masked tensor) to


def trace_dropout_sites(sink):
    """sink = a list: record every dropout draw of the following forwards in call order; None: stop recording"""
    global _dropout_trace
    _dropout_trace = sink


def _trace_site(kind: str, sid: int, p: float, shape) -> None:
    if _dropout_trace is not None and p > 0:
        _dropout_trace.append((kind, int(sid), float(p), tuple(int(v) for v in shape)))


def dropout_keep_scale(seed: torch.Tensor, stream_id: int, shape, p: float) -> torch.Tensor:
    """The fp32 factor (0 or 1 / (1 - p)) that the element-wise dropout with this seed / stream id applies to a dense tensor of `shape`,
    in MEMORY order: the standalone kernel run on ones (the GEMM epilogues index the same counter by row * N + column)."""
    ones = torch.ones(tuple(shape), dtype=torch.float32, device=seed.device)
    out = torch.empty_like(ones)
    _lib.call("cvh_dropout", 0, _p(ones), _p(out), ones.numel(), float(p), _p(seed), int(stream_id), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# weight packing
# ------------------------------------------------------------------------------------------------
# Packed weight images are owned by the model: a PackPlan (module.__dict__["_cvh_pack_plan"]) holds the flat buffer, and this table maps a
# live PARAMETER OBJECT to the views cut for it — id(parameter) -> (weakref to the parameter, {(mode, dtype): (view, version, epoch)}).  The
# entry is removed by a finalizer when the parameter dies, so nothing of a dead model (its pack buffer included) stays referenced.
_PACKED = {}
_PACK_EPOCH = 0  # bumped by everything that rewrites parameters through raw pointers (cvh_adamw_multi, EMA kernels): `_version` cannot see those


def invalidate_packed() -> None:
    """Every cached packed weight becomes stale (call after parameters were updated behind autograd's back: the fused optimizer / EMA
    kernels do it themselves when stepped eagerly; after hipGraph REPLAYS of a captured optimizer step call it before using layers outside
    a model-level forward — the model-level forwards re-pack unconditionally)."""
    global _PACK_EPOCH
    _PACK_EPOCH += 1


def _drop_packs(key: int, slot) -> None:
    if _PACKED.get(key) is slot:
        del _PACKED[key]


def _pack_slot(w: torch.Tensor, create: bool):
    slot = _PACKED.get(id(w))
    if slot is not None and slot[0]() is w:
        return slot[1]
    if not create:
        return None
    slot = (weakref.ref(w), {})
    _PACKED[id(w)] = slot  # (an entry left by a dead parameter whose id was reused is replaced; its own finalizer then finds another slot)
    weakref.finalize(w, _drop_packs, id(w), slot)
    return slot[1]


def _pack_numel(shape, mode: int) -> int:
    Cout, Cin = shape[0], shape[1]
    KHW = 1 if len(shape) == 2 else shape[2] * shape[3]
    if mode == 0:
        return Cout * KHW * pad8(Cin)
    if mode == 1:
        return Cin * KHW * pad8(Cout)
    if mode == 3:
        return KHW * pad8(Cin) * pad8(Cout)
    return KHW * Cout


class PackPlan:
    """All conv / linear weights of a module tree packed by ONE kernel launch per step (cvh_weight_pack_multi) into a flat
    compute-dtype buffer; `pack_weight` then hands out views.  Replaces ~140 per-layer pack launches per training step."""

    def __init__(self, module: torch.nn.Module, dtype: torch.dtype):
        entries = []
        for m in module.modules():
            if isinstance(m, torch.nn.Conv2d):
                if getattr(m, "_cvh_skip_pack", False):  # weight is consumed through a permuted copy (LinearSelfAttention.qkv_proj)
                    continue
                w = m.weight
                if m.groups == 1:
                    entries.append((w, 0))
                    if m.stride[0] == 1:
                        entries.append((w, 1))
                    elif m.stride[0] == m.kernel_size[0] and m.padding[0] == 0 and m.in_channels % 8 == 0:
                        entries.append((w, 3))
                elif m.groups == m.in_channels == m.out_channels:
                    entries.append((w, 2))
            elif hasattr(m, "weight") and isinstance(getattr(m, "weight", None), torch.nn.Parameter) and m.weight.dim() == 2 \
                    and m.__class__.__name__ == "LinearLayer":
                entries.append((m.weight, 0))
                entries.append((m.weight, 1))
        self.dtype = dtype
        self.entries = [(w, mode) for w, mode in entries if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()]
        self.ptrs = [w.data_ptr() for w, _ in self.entries]
        dev = self.entries[0][0].device if self.entries else None
        table, off, start = [], 0, 0
        self.offsets = []
        for w, mode in self.entries:
            n = _pack_numel(w.shape, mode)
            KHW = 1 if w.dim() == 2 else w.shape[2] * w.shape[3]
            table.append([w.data_ptr(), off, w.shape[0], w.shape[1], KHW, mode, start])
            self.offsets.append((off, n))
            off += (n + 7) // 8 * 8
            start += n
        self.total = start
        table.append([0, 0, 0, 0, 0, 0, start])
        if self.entries:
            self.table = torch.tensor(table, dtype=torch.int64, device=dev)
            self.flat = torch.empty(max(off, 8), dtype=dtype, device=dev)

    def valid(self) -> bool:
        return all(w.data_ptr() == p for (w, _), p in zip(self.entries, self.ptrs))

    def run(self) -> None:
        if not self.entries:
            return
        _lib.call("cvh_weight_pack_multi", _dt(self.flat), _p(self.table), len(self.entries), self.total, _p(self.flat), _stream())
        for (w, mode), (off, n) in zip(self.entries, self.offsets):
            _pack_slot(w, True)[(mode, self.dtype)] = (self.flat[off: off + n], w._version, _PACK_EPOCH)


def pack_all(module: torch.nn.Module, dtype: Optional[torch.dtype] = None) -> None:
    """Pack every weight under `module` for the coming step (call at the start of a training forward; it is captured into the
    step's hipGraph, so replays re-pack the freshly updated parameters)."""
    dtype = dtype or compute_dtype()
    plan = module.__dict__.get("_cvh_pack_plan")
    if plan is None or plan.dtype != dtype or not plan.valid():
        plan = PackPlan(module, dtype)
        module.__dict__["_cvh_pack_plan"] = plan
    plan.run()


_COUNTERS_BUMPED = False  # True while a model forward runs whose BatchNorm step counters were advanced by one batched launch


def bump_bn_counters(module: torch.nn.Module) -> None:
    """num_batches_tracked += 1 for every training BatchNorm of `module` in ONE multi-tensor launch (instead of one 1-block kernel per
    layer: 32 launches per MobileViT step); the per-layer increments are skipped until `end_bn_counters()`."""
    global _COUNTERS_BUMPED
    lst = module.__dict__.get("_cvh_bn_counters")
    if lst is None:
        lst = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        module.__dict__["_cvh_bn_counters"] = lst
    live = [m.num_batches_tracked for m in lst if m.training and m.track_running_stats and m.num_batches_tracked is not None]
    if live:
        torch._foreach_add_(live, 1)
    _COUNTERS_BUMPED = True


def end_bn_counters() -> None:
    global _COUNTERS_BUMPED
    _COUNTERS_BUMPED = False


def bn_counters_bumped() -> bool:
    return _COUNTERS_BUMPED


_INPLACE_PARAM_GRADS = False


def set_inplace_param_grads(flag: bool) -> None:
    """When on, backward kernels ADD parameter gradients straight into existing ``param.grad`` buffers (zeroed by the caller,
    e.g. one flat bucket) and hand autograd ``None`` — no per-parameter zero-fill / AccumulateGrad add kernels.  Autograd
    post-accumulate hooks do not fire for those parameters (bench.py's hipGraph step reduces the flat buckets explicitly)."""
    global _INPLACE_PARAM_GRADS
    _INPLACE_PARAM_GRADS = bool(flag)


def _grad_sink(param: Optional[torch.Tensor]):
    if not _INPLACE_PARAM_GRADS or param is None:
        return None
    if not param.is_leaf:
        return None
    g = param.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous():
        return None
    return g


def pack_weight(w: torch.Tensor, dtype: torch.dtype, mode: int) -> torch.Tensor:
    """mode 0: [Cout][KH*KW][pad8(Cin)] ; mode 1: [Cin][KH*KW][pad8(Cout)] (taps flipped) ; mode 2: depthwise [KH*KW][C]."""
    slot = _pack_slot(w, False)
    hit = slot.get((mode, dtype)) if slot is not None else None
    if hit is not None and hit[1] == w._version and hit[2] == _PACK_EPOCH:
        return hit[0]
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()  # plumbing (parameters are fp32 contiguous in practice)
    if wf.dim() == 2:
        Cout, Cin, KHW = wf.shape[0], wf.shape[1], 1
    else:
        Cout, Cin, KHW = wf.shape[0], wf.shape[1], wf.shape[2] * wf.shape[3]
    n = _pack_numel(wf.shape, mode)
    out = torch.empty(n, dtype=dtype, device=w.device)
    _lib.call("cvh_weight_pack", _dt(out), _p(wf), _p(out), Cout, Cin, KHW, mode, _stream())
    return out


# ------------------------------------------------------------------------------------------------
# low-level launch helpers
# ------------------------------------------------------------------------------------------------
def _conv_gemm(src1, src2, C1, C2, wp, out, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, bias=None, act=0, save_pre=None,
               actgrad_aux=None, actgrad_act=0, residual=None, drop_p=0.0, seed=None, stream_id=0, stats_part=None, wp_offset=0):
    wptr = wp.data_ptr() + wp_offset * wp.element_size()
    _lib.call("cvh_conv_gemm", _dt(out), _p(src1), _p(src2), C1, C2, wptr, _p(out), B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N,
              _p(bias), act, _p(save_pre), _p(actgrad_aux), actgrad_act, _p(residual), float(drop_p), _p(seed), stream_id,
              _p(stats_part), _stream())


_UNIT_COEFFS = {}


def _unit_coeffs(C, device):
    key = (int(C), str(device))
    if key not in _UNIT_COEFFS:  # tiny constant vectors, made once per width (not two fill kernels per call)
        _UNIT_COEFFS[key] = (torch.ones(C, dtype=torch.float32, device=device), torch.zeros(C, dtype=torch.float32, device=device))
    return _UNIT_COEFFS[key]


def _act_backward(pre: torch.Tensor, dout: torch.Tensor, act: int, rows: int, C: int) -> torch.Tensor:
    """dpre = dout * act'(pre)  (BatchNorm-backward apply kernel with identity normalisation)."""
    if C > 2048:  # purely elementwise with unit coefficients: view the matrix with narrower rows (ViT-B FFN: 3072 -> 2 x 1536)
        Cb = next(c for c in range(2048, 7, -8) if C % c == 0)
        rows, C = rows * (C // Cb), Cb
    ones, zeros = _unit_coeffs(C, pre.device)
    dx = torch.empty_like(dout)
    _lib.call("cvh_bn_bwd_apply", _dt(pre), _p(pre), _p(dout), _p(ones), _p(zeros), act, _p(ones), _p(zeros), _p(zeros), _p(dx), rows, C,
              _stream())
    return dx


def _colsum(x2d: torch.Tensor, rows: int, C: int, sink: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """column sums (bias gradients); with a sink the result is added into it and None is returned"""
    R = _lib.query("cvh_colreduce_rows", rows, C)
    side = _param_grad_stream(x2d.device) if sink is not None else None
    if side is not None:
        with torch.cuda.stream(side):
            part = _f32(R * 2 * C, x2d.device)
            x2d.record_stream(side)
            deferred = C <= 2048 and defer_reduce(part, sink, R, 2 * C, C)
            _lib.call("cvh_colsum", _dt(x2d), _p(x2d), rows, C, _p(part), None if deferred else _p(sink), 1.0, 1, _stream())
        return None
    part = _f32(R * 2 * C, x2d.device)
    out = sink if sink is not None else _f32(C, x2d.device)
    _lib.call("cvh_colsum", _dt(x2d), _p(x2d), rows, C, _p(part), _p(out), 1.0, 1 if sink is not None else 0, _stream())
    return None if sink is not None else out


# ------------------------------------------------------------------------------------------------
# parameter-gradient side stream
# ------------------------------------------------------------------------------------------------
# dW (and bias-gradient) kernels are off the backward critical path: nothing in the backward pass reads them, only the optimizer
# does.  With in-place parameter gradients they are launched on a SIDE HIP stream behind an event on the main stream, so the many
# under-filled dW / reduce launches of the small layers overlap with the dX chain; the main stream re-joins at the end of the
# backward pass (autograd end-of-backward callback).  Captured into a hipGraph this becomes two parallel branches per layer.
# Default OFF since round 3: with the faster dX chain the overlap stopped paying (92.3 ms with, 91.9 ms without, same box), and the fp32
# MobileViTv2 golden case that failed once in ~25 suite runs in round 2 reproduces ONLY with the side stream on (tools/stress_v2.py: 1 of 300
# runs at 2e-3 with CVH_ASYNC_DW=1, 0 of 300 with it off — a cross-stream ordering hole that single-stream execution cannot have).
_ASYNC_PARAM_GRADS = os.environ.get("CVH_ASYNC_DW", "0") == "1"
_side_streams = {}
# id of the autograd graph task (one per backward call) whose end-of-backward callback is queued, or None.  Tied to the task id rather
# than a bare flag: autograd DROPS queued callbacks when a backward raises (an OOM the engine skips, a kernel error), and a stale "already
# queued" flag would make every later backward skip its join + deferred reductions — training on with missing gradients.


def _param_grad_stream(device):
    """returns the side stream (already made to wait for everything queued so far on the current stream) or None"""
    if not (_ASYNC_PARAM_GRADS and _INPLACE_PARAM_GRADS):
        return None
    side = _side_streams.get(device)
    if side is None:
        side = _side_streams[device] = torch.cuda.Stream(device=device)
    if not _ensure_backward_callback():  # not inside a backward pass (direct Function.backward call in a test)
        return None
    side.wait_stream(torch.cuda.current_stream(device))
    return side


def _join_param_grad_stream():
    for dev, side in _side_streams.items():
        torch.cuda.current_stream(dev).wait_stream(side)


# Deferred reductions: "sum the partial rows" tails whose result only the optimizer reads (dW split partials, LayerNorm dgamma/dbeta,
# bias gradients, depthwise dW) are queued during backward and executed by cvh_reduce_multi at the end of it — one launch per 48 tensors
# instead of ~140 latency-bound launches per MobileViT-S step.  Only with in-place parameter gradients (the result must not be needed by
# autograd) and inside a backward pass (the flush rides on the engine's end-of-backward callback).
# The queue is kept PER autograd graph task: a nested / re-entrant backward (torch.utils.checkpoint(use_reentrant=True), autograd.grad
# inside a custom backward) gets its own task id while the outer task is still alive — its callback flushes its own entries only.
_DEFER_REDUCTIONS = os.environ.get("CVH_DEFER_REDUCE", "1") != "0"
_pending_by_task = {}  # graph-task id -> [(descriptor, partial buffer, destination)]; a key exists <=> that task's callback is queued


def _end_of_backward(tid: int) -> None:
    _join_param_grad_stream()
    _flush_deferred_reductions(tid)
    _pending_by_task.pop(tid, None)


def _ensure_backward_callback() -> bool:
    tid = torch._C._current_graph_task_id()
    if tid < 0:  # not inside a backward pass (direct Function.backward call in a test)
        return False
    if tid not in _pending_by_task:
        _pending_by_task[tid] = []
        torch.autograd.Variable._execution_engine.queue_callback(lambda tid=tid: _end_of_backward(tid))
    return True


def finish_backward() -> None:
    """Idempotent join of the parameter-gradient side stream + drop of what a failed backward left queued.  The end-of-backward callback
    normally joins and flushes; the consumers of .grad (fused AdamW, DDP.allreduce_flat) call this first so that state can never leak
    across steps."""
    if torch._C._current_graph_task_id() >= 0:
        return  # still inside a backward pass: its own callback will run
    _bwd_handover.clear()
    if _pending_by_task:
        # callbacks that never ran (backward raised): the queued reductions belong to void backward passes — drop them, re-join the stream
        _pending_by_task.clear()
        _join_param_grad_stream()


def defer_reduce(part, out, rows, row_stride, n_out, *, kind=0, N=0, Ktot=0, Cin=0, Cin_real=0, khw=1, scale=1.0, part_offset=0) -> bool:
    """queue  out += scale * sum_r part[part_offset + r*row_stride + j]  (see cvh_reduce_multi); False = caller must reduce now"""
    if not (_DEFER_REDUCTIONS and _INPLACE_PARAM_GRADS) or not _ensure_backward_callback():
        return False
    desc = _lib.ReduceDesc(part.data_ptr() + 4 * int(part_offset), out.data_ptr(), int(row_stride), int(n_out), int(rows), int(kind), int(N), int(Ktot),
                           int(Cin), int(Cin_real), int(khw), float(scale), 1, 0)
    part.record_stream(torch.cuda.current_stream(part.device))
    _pending_by_task[torch._C._current_graph_task_id()].append((desc, part, out))
    return True


def flush_deferred_reductions(task: Optional[int] = None) -> None:
    """launch the reductions a backward pass queued so far NOW (cvnets_amd.ddp: a gradient bucket is about to be all-reduced); later
    reductions of the same backward queue up again and are flushed by the end-of-backward callback.  `task` = the autograd graph-task id
    whose queue is meant (default: the one that is running; an end-of-backward callback passes the id it was registered under, since
    the order of the engine's final callbacks is the order in which they were queued, not ours to choose)"""
    tid = torch._C._current_graph_task_id() if task is None else task
    if tid >= 0:
        # the partial buffers (and, undeferred, the .grad sinks) may have been written on the parameter-gradient side stream (CVH_ASYNC_DW=1):
        # the reduction below — and the bucket all-reduce the caller is about to start — must come after those kernels
        _join_param_grad_stream()
        _flush_deferred_reductions(tid)


def _flush_deferred_reductions(tid: int) -> None:
    pending = _pending_by_task.get(tid)
    if not pending:
        return
    by_dev = {}
    for item in pending:
        by_dev.setdefault(item[1].device, []).append(item)
    pending.clear()
    for dev, items in by_dev.items():
        # cvh_reduce_multi adds into `out` without atomics, one workgroup set per descriptor: two descriptors with the SAME destination
        # (a shared weight, a module applied twice in one graph) must not share a launch — later duplicates go to later launches,
        # which the stream serialises
        rounds = []
        for it in items:
            key = it[2].data_ptr()
            for seen, lst in rounds:
                if key not in seen:
                    seen.add(key)
                    lst.append(it)
                    break
            else:
                rounds.append(({key}, [it]))
        with torch.cuda.device(dev):
            for _, lst in rounds:
                arr = (_lib.ReduceDesc * len(lst))(*[it[0] for it in lst])
                for it in lst:
                    it[1].record_stream(torch.cuda.current_stream(dev))
                _lib.call("cvh_reduce_multi", arr, len(lst), _stream())


_FOLD_BIAS = os.environ.get("CVH_FOLD_BIAS", "1") != "0"


def _weight_grad(dy, x, x2, C1, C2, weight, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real, bias_sink=None):
    """dW = dY^T x im2col(x): split over M into a scratch buffer + one reduce kernel (no atomics, no zero-fill).  Returns the
    gradient tensor, or None when it was added in place into weight.grad.

    With `bias_sink` (the .grad buffer of the same layer's bias) the return value is (dW, folded): folded == True means the bias
    gradient was produced by the dW kernel itself (cvh_gemm_dw_bias: the column sums of dY ride along, dY is read once) and is summed into
    bias_sink by the deferred reduction; False leaves it to the caller (cvh_colsum)."""
    sink = _grad_sink(weight)
    Ktot = KH * KW * (C1 + C2)
    M = B * Ho * Wo
    n_scr = _lib.query("cvh_gemm_dw_scratch_elems_conv", _dt(dy), B, H, W, Ho, Wo, C1, C2, KH, KW, stride, pad, dil, N,
                       1 if bias_sink is not None else 0)
    if sink is not None:
        # in-place parameter gradients: split partials now, their sum at the end of backward (cvh_reduce_multi) — or right here when deferral is
        # off; on the parameter-gradient side stream when that is enabled (CVH_ASYNC_DW=1), else on the current stream
        side = _param_grad_stream(dy.device)
        folded = False
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            scr = _f32(max(n_scr, 1), dy.device)
            if side is not None:
                for t in (dy, x, x2):
                    if t is not None:
                        t.record_stream(side)
            rows = n_scr // (N * Ktot)
            deferred = n_scr > 0 and defer_reduce(scr, sink, rows, N * Ktot, N * Ktot,
                                                  kind=0 if (KH * KW == 1 and Cin_real == Ktot) else 1, N=N, Ktot=Ktot, Cin=C1 + C2,
                                                  Cin_real=Cin_real, khw=KH * KW)
            bpart = None
            if (bias_sink is not None and deferred and _FOLD_BIAS and _lib.query("cvh_gemm_dw_folds_bias", _dt(dy), M, N, Ktot)):
                bpart = _f32(rows * N, dy.device)
                folded = defer_reduce(bpart, bias_sink, rows, N, N)
            _lib.call("cvh_gemm_dw_bias", _dt(dy), _p(dy), _p(x), _p(x2), C1, C2, None if deferred else _p(sink), _p(bpart) if folded else None,
                      B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real, _p(scr), n_scr, 1, _stream())
        return (None, folded) if bias_sink is not None else None
    dw = torch.empty(weight.shape, dtype=torch.float32, device=dy.device)
    scr = _f32(max(n_scr, 1), dy.device)
    _lib.call("cvh_gemm_dw", _dt(dy), _p(dy), _p(x), _p(x2), C1, C2, _p(dw), B, H, W, Ho, Wo, KH, KW, stride, pad, dil, N, Cin_real,
              _p(scr), n_scr, 0, _stream())
    return (dw, False) if bias_sink is not None else dw


def _bn_forward(y, rows, C, part, R, gamma, beta, rmean, rvar, training, momentum, eps):
    """returns stats [4][C] = (mean, invstd, scale, shift)."""
    stats = _f32(4, y.device, C)
    if training:
        _lib.call("cvh_bn_finalize", _p(part), R, C, float(rows), _p(gamma), _p(beta), _p(rmean), _p(rvar), float(momentum), float(eps),
                  _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]), _stream())
    else:
        _lib.call("cvh_bn_eval_coeff", _p(gamma), _p(beta), _p(rmean), _p(rvar), float(eps), C, _p(stats[0]), _p(stats[1]), _p(stats[2]),
                  _p(stats[3]), _stream())
    return stats


# ------------------------------------------------------------------------------------------------
# BatchNorm apply that also serves the NEXT fused InvertedResidual block (csrc/bngram.hip)
# ------------------------------------------------------------------------------------------------
# A fused InvertedResidual block starts from the Gram matrix G = x^T x and the column sums of its narrow input (the statistics of its
# expansion BatchNorm follow from them, fused.py).  The producer of that input — the BatchNorm apply of the previous block / the stem /
# a ConvLayer2d — can form both in the pass that writes the tensor.  Who consumes a layer's output is not known to the layer, so it is
# LEARNED: every such producer tags its output with the BatchNorm weight it used (`_cvh_src`); a fused block that receives a tagged input
# without a Gram matrix sets `_cvh_gram_wanted` on that parameter, and from the next forward on the producer's apply pass emits it
# (`_cvh_gram` on the output: (G | s) float32 [C*C + C], tensor version, address — dropped after any in-place modification).
_GRAM_OUT = os.environ.get("CVH_GRAM_OUT", "1") != "0"


def _bn_apply_out(y, stats, act, residual, out, rows, C, src, training):
    """cvh_bn_apply into `out`; returns the (G | s) tensor when the pass also formed it, else None"""
    if (_GRAM_OUT and training and src is not None and getattr(src, "_cvh_gram_wanted", False) and y.dtype == torch.bfloat16):
        R = _lib.query("cvh_bn_apply_gram_rows", int(rows), int(C))
        if R > 0:
            n = C * C + C
            part = _f32(R * n, y.device)
            gs = _f32(n, y.device)
            _lib.call("cvh_bn_apply_gram", _dt(y), _p(y), _p(stats[2]), _p(stats[3]), act, _p(residual), _p(out), rows, C, _p(part), R, _p(gs),
                      _stream())
            return gs
    _lib.call("cvh_bn_apply", _dt(y), _p(y), _p(stats[2]), _p(stats[3]), act, _p(residual), _p(out), rows, C, _stream())
    return None


def tag_producer(res, src):
    """res = `out` or (out, gs) as returned by a producer's autograd function: attach the tags, return out"""
    out, gs = res if isinstance(res, tuple) else (res, None)
    if src is not None and _GRAM_OUT:
        out._cvh_src = src
        if gs is not None:
            out._cvh_gram = (gs, out._version, out.data_ptr())
    return out


def gram_of_input(x, C, training):
    """the (G | s) tensor a producer attached to x, if still valid; otherwise ask the producer for it from the next step on"""
    if not (_GRAM_OUT and training):
        return None
    tag = getattr(x, "_cvh_gram", None)
    if tag is not None:
        gs, ver, ptr = tag
        if ver == x._version and ptr == x.data_ptr() and gs.numel() == C * C + C and gs.device == x.device:
            return gs
    src = getattr(x, "_cvh_src", None)
    if src is not None and x.dtype == torch.bfloat16:
        src._cvh_gram_wanted = True
    return None


# Gradient statistics handed BACKWARD from the kernel that produces a block's incoming gradient to that block's BatchNorm backward
# (CVH_BN_HANDOVER=0 switches it off): key = address of the gradient tensor -> (partial rows [R][2][C] of (sum dX, sum dX * x), R, C, rows, address and
# version of x at the producer's forward).  Entries are consumed by the first lookup and dropped at the end of every backward pass.
_BN_HANDOVER = os.environ.get("CVH_BN_HANDOVER", "1") == "1"
_bwd_handover = {}


def offer_grad_stats(dx, part, R, C, rows, x):
    if len(_bwd_handover) > 64:  # offers nobody took (their consumer is not a fused block): never more than a step's worth
        _bwd_handover.clear()
    _bwd_handover[dx.data_ptr()] = (part, int(R), int(C), int(rows), x.data_ptr(), x._version)


def take_grad_stats(dout, C, rows, out_ptr, out_version):
    h = _bwd_handover.pop(dout.data_ptr(), None)
    if h is None:
        return None
    part, R, hc, hrows, xptr, xver = h
    if hc != C or hrows != rows or xptr != out_ptr or xver != out_version:
        return None
    return part, R


def _bn_backward_coeffs(y, dout, stats, gamma, act, rows, C, training, beta=None, handed=None):
    """The statistics half of the BatchNorm backward: returns (coeff[3][C], dgamma, dbeta) with dy_raw = coeff[0] * (dout * act'(bn(y))) +
    coeff[1] * y + coeff[2] left to the consumer (cvh_bn_bwd_apply, or an operand load that forms it: csrc/ir_pb.hip).
    handed = (part, R): partial rows (sum dout, sum dout * OUT) formed by the kernel that wrote dout (cvh_ir_exp_bwd_s), OUT being this
    BatchNorm's own stored output — no pass over (dout, y) here."""
    dev = y.device
    if handed is None:
        R = _lib.query("cvh_colreduce_rows", rows, C)
        part = _f32(R * 2 * C, dev)
        _lib.call("cvh_bn_bwd_reduce", _dt(y), _p(y), _p(dout), _p(stats[2]), _p(stats[3]), _p(stats[0]), _p(stats[1]), act, rows, C, _p(part),
                  _stream())
    else:
        part, R = handed
    sg, sb = _grad_sink(gamma), _grad_sink(beta)
    inplace = sg is not None and sb is not None
    dgamma = sg if inplace else _f32(C, dev)
    dbeta = sb if inplace else _f32(C, dev)
    coeff = _f32(3, dev, C)
    if handed is None:
        _lib.call("cvh_bn_bwd_finalize", _p(part), R, C, float(rows), _p(gamma), _p(stats[0]), _p(stats[1]), 1 if training else 0,
                  1 if inplace else 0, _p(dgamma), _p(dbeta), _p(coeff[0]), _p(coeff[1]), _p(coeff[2]), _stream())
    else:
        _lib.call("cvh_bn_bwd_finalize_out", _p(part), R, C, float(rows), _p(gamma), _p(beta), _p(stats[0]), _p(stats[1]), 1 if training else 0,
                  1 if inplace else 0, _p(dgamma), _p(dbeta), _p(coeff[0]), _p(coeff[1]), _p(coeff[2]), _stream())
    if inplace:
        dgamma = dbeta = None
    return coeff, dgamma, dbeta


def _bn_backward(y, dout, stats, gamma, act, rows, C, training, beta=None):
    """returns (dy_raw, dgamma, dbeta); dgamma/dbeta are None when they were added in place into gamma.grad / beta.grad."""
    coeff, dgamma, dbeta = _bn_backward_coeffs(y, dout, stats, gamma, act, rows, C, training, beta=beta)
    dy = torch.empty_like(y)
    _lib.call("cvh_bn_bwd_apply", _dt(y), _p(y), _p(dout), _p(stats[2]), _p(stats[3]), act, _p(coeff[0]), _p(coeff[1]), _p(coeff[2]), _p(dy),
              rows, C, _stream())
    return dy, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# dense conv (+BatchNorm +activation +residual), optional channel-concat of two inputs
# ------------------------------------------------------------------------------------------------
class ConvBNAct(torch.autograd.Function):
    """ConvLayer2d.forward for groups == 1 (cvnets/layers/conv_layer.py:254-255): conv -> BN(train/eval) -> act,
    plus the residual add / channel concat that surround it in InvertedResidual / MobileViTBlock."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, gamma, beta, rmean, rvar, residual, cfg):
        stride, pad, dil, act, use_bn, training, momentum, eps = cfg
        _check_dev(x)
        B, C1, H, W = x.shape
        C2 = x2.shape[1] if x2 is not None else 0
        Cout, Cin_real, KH, KW = weight.shape
        if pad8(Cin_real) != C1 + C2:
            raise RuntimeError(f"conv input has {C1 + C2} (padded) channels, weight expects {Cin_real}")
        if Cout % 8:
            raise RuntimeError("cvnets_amd convs need out_channels % 8 == 0")
        Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
        M = B * Ho * Wo
        dev, dtype = x.device, x.dtype
        wp = pack_weight(weight, dtype, 0)
        y = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
        ctx.cfg = cfg
        ctx.shapes = (B, C1, C2, H, W, Ho, Wo, Cout, Cin_real, KH, KW)
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.params = (bias, beta)  # parameter handles for in-place gradient accumulation
        if use_bn:
            part, R = None, 0
            if training:
                R = _lib.query("cvh_conv_gemm_grid_rows", M, Cout)
                part = _f32(R * 2 * Cout, dev)
            _conv_gemm(x, x2, C1, C2, wp, y, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, Cout, bias=bias, stats_part=part)
            stats = _bn_forward(y, M, Cout, part, R, gamma, beta, rmean, rvar, training, momentum, eps)
            out = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
            gs = _bn_apply_out(y, stats, act, residual, out, M, Cout, gamma, training)
            ctx.save_for_backward(x, x2, weight, y, stats, gamma)
            if gs is not None:
                ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for the side output
                ctx.mark_non_differentiable(gs)
                return out, gs
            return out
        pre = nhwc_empty(B, Cout, Ho, Wo, dtype, dev) if act != ACT_NONE else None
        _conv_gemm(x, x2, C1, C2, wp, y, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, Cout, bias=bias, act=act, save_pre=pre, residual=residual)
        ctx.save_for_backward(x, x2, weight, pre, None, None)
        return y

    @staticmethod
    def backward(ctx, dout, *_unused):
        stride, pad, dil, act, use_bn, training, momentum, eps = ctx.cfg
        B, C1, C2, H, W, Ho, Wo, Cout, Cin_real, KH, KW = ctx.shapes
        x, x2, weight, y, stats, gamma = ctx.saved_tensors
        dout = as_nhwc(dout)
        M = B * Ho * Wo
        dev, dtype = dout.device, dout.dtype
        dgamma = dbeta = dbias = None
        bias_p, beta_p = ctx.params
        if use_bn:
            dy, dgamma, dbeta = _bn_backward(y, dout, stats, gamma, act, M, Cout, training, beta=beta_p)
        elif act != ACT_NONE:
            dy = _act_backward(y, dout, act, M, Cout)
        else:
            dy = dout
        bsink = _grad_sink(bias_p) if ctx.has_bias else None
        if bsink is not None:  # the bias gradient rides along in the dW kernel where it can
            dw_ret, folded = _weight_grad(dy, x, x2, C1, C2, weight, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, Cout, Cin_real, bias_sink=bsink)
            if not folded:
                _colsum(dy, M, Cout, bsink)
        else:
            if ctx.has_bias:
                dbias = _colsum(dy, M, Cout, None)
            dw_ret = _weight_grad(dy, x, x2, C1, C2, weight, B, H, W, Ho, Wo, KH, KW, stride, pad, dil, Cout, Cin_real)
        dx = dx2 = None
        need1, need2 = ctx.needs_input_grad[0], (x2 is not None and ctx.needs_input_grad[1])
        if need1 or need2:
            if stride != 1:
                # non-overlapping patch convs (kernel == stride, pad 0: the ViT stem) -> one GEMM with a scatter epilogue
                if not (KH == KW == stride and pad == 0 and dil == 1 and x2 is None):
                    raise NotImplementedError("dX of an overlapping strided dense conv is not on the hot path (only MobileViT's stem, which needs none)")
                wp3 = pack_weight(weight, dtype, 3)  # [(kh,kw,c)][Cout]
                dx = nhwc_empty(B, C1, H, W, dtype, dev)
                _lib.call("cvh_conv_dx_patch", _dt(dy), _p(dy), _p(wp3), _p(dx), B, Ho, Wo, Cout, KH, KW, stride, C1, H, W, _stream())
                dres = dout if ctx.has_res else None
                return dx, None, dw_ret, dbias, dgamma, dbeta, None, None, dres, None
            wpt = pack_weight(weight, dtype, 1)  # [Cin][KH*KW][Cout]
            pad_t = dil * (KH - 1) - pad
            kk = KH * KW * Cout
            if need1:
                dx = nhwc_empty(B, C1, H, W, dtype, dev)
                _conv_gemm(dy, None, Cout, 0, wpt, dx, B, Ho, Wo, H, W, KH, KW, 1, pad_t, dil, C1)
            if need2:
                dx2 = nhwc_empty(B, C2, H, W, dtype, dev)
                _conv_gemm(dy, None, Cout, 0, wpt, dx2, B, Ho, Wo, H, W, KH, KW, 1, pad_t, dil, C2, wp_offset=C1 * kk)
        dres = dout if ctx.has_res else None
        return dx, dx2, dw_ret, dbias, dgamma, dbeta, None, None, dres, None


def conv_bn_act(x, weight, bias=None, gamma=None, beta=None, rmean=None, rvar=None, *, stride=1, pad=0, dil=1, act=ACT_NONE,
                use_bn=False, training=True, momentum=0.1, eps=1e-5, residual=None, x2=None):
    cfg = (int(stride), int(pad), int(dil), int(act), bool(use_bn), bool(training), float(momentum), float(eps))
    return tag_producer(ConvBNAct.apply(x, x2, weight, bias, gamma, beta, rmean, rvar, residual, cfg), gamma if use_bn else None)


# ------------------------------------------------------------------------------------------------
# the stem: 3x3 stride-2 conv of the NCHW image batch (+BatchNorm +activation), image planes read once
# ------------------------------------------------------------------------------------------------
def stem_eligible(x, weight, bias, stride, pad, dil, use_bn, residual=None, x2=None) -> bool:
    """conv_1 of MobileViT (cvnets/models/classification/mobilevit.py:62-72) fed with the raw NCHW batch: 3 -> 16 / 32 channels, 3x3,
    stride 2, pad 1, BatchNorm behind it, bf16 compute, no gradient wanted for the image."""
    return (_STEM_KERNEL and use_bn and bias is None and residual is None and x2 is None and x.dim() == 4 and x.shape[1] == 3
            and tuple(weight.shape[1:]) == (3, 3, 3) and weight.shape[0] in (16, 32) and (stride, pad, dil) == (2, 1, 1)
            and x.shape[3] % 4 == 0 and x.shape[2] >= 2 and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous()
            and not x.requires_grad and compute_dtype() == torch.bfloat16 and weight.dtype in (torch.float32, torch.bfloat16)
            # the kernels read the weight (and add its gradient) as dense OIHW: a model moved to channels_last (launch.py does that for
            # `common.channels_last`) has conv_1.weight strided (27, 1, 9, 3) — such a layer takes the repack + implicit-GEMM path, which
            # goes through pack_weight's contiguous copy
            and weight.is_contiguous() and (weight.grad is None or weight.grad.is_contiguous()))


_STEM_KERNEL = os.environ.get("CVH_STEM_KERNEL", "1") != "0"


class StemConvBNAct(torch.autograd.Function):
    """ConvLayer2d.forward (cvnets/layers/conv_layer.py:254-255) for the stem; x stays NCHW and is never repacked."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, rmean, rvar, cfg):
        act, training, momentum, eps = cfg
        _check_dev(x)
        B, _, H, W = x.shape
        Cout = weight.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        M = B * Ho * Wo
        dev, dtype = x.device, torch.bfloat16
        y = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
        part, R = None, 0
        if training:
            R = _lib.query("cvh_stem_rows", B, H, W, Cout)
            part = _f32(R * 2 * Cout, dev)
        _lib.call("cvh_stem_conv_fwd", _dt(x), _p(x), _dt(weight), _p(weight), _p(y), _p(part), B, H, W, Cout, _stream())
        stats = _bn_forward(y, M, Cout, part, R, gamma, beta, rmean, rvar, training, momentum, eps)
        out = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
        gs = _bn_apply_out(y, stats, act, None, out, M, Cout, gamma, training)
        ctx.cfg = cfg
        ctx.beta = beta
        ctx.save_for_backward(x, weight, y, stats, gamma)
        if gs is not None:
            ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for the side output
            ctx.mark_non_differentiable(gs)
            return out, gs
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        act, training, momentum, eps = ctx.cfg
        x, weight, y, stats, gamma = ctx.saved_tensors
        B, _, H, W = x.shape
        Cout = weight.shape[0]
        M = y.shape[0] * y.shape[2] * y.shape[3]
        dout = as_nhwc(dout)
        dy, dgamma, dbeta = _bn_backward(y, dout, stats, gamma, act, M, Cout, training, beta=ctx.beta)
        R = _lib.query("cvh_stem_rows", B, H, W, Cout)
        part = _f32(R * Cout * 72, dy.device)
        _lib.call("cvh_stem_conv_dw", _dt(x), _p(x), _p(dy), _p(part), B, H, W, Cout, _stream())
        sink = _grad_sink(weight)
        kw = dict(kind=1, N=Cout, Ktot=72, Cin=8, Cin_real=3, khw=9)
        if sink is not None and defer_reduce(part, sink, R, Cout * 72, Cout * 72, **kw):
            return None, None, dgamma, dbeta, None, None, None
        dw = sink if sink is not None else torch.zeros(weight.shape, dtype=torch.float32, device=dy.device)
        desc = _lib.ReduceDesc(part.data_ptr(), dw.data_ptr(), Cout * 72, Cout * 72, R, 1, Cout, 72, 8, 3, 9, 1.0, 1, 0)
        _lib.call("cvh_reduce_multi", (_lib.ReduceDesc * 1)(desc), 1, _stream())
        return None, (None if sink is not None else dw), dgamma, dbeta, None, None, None


def stem_conv_bn_act(x, weight, gamma, beta, rmean, rvar, *, act=ACT_NONE, training=True, momentum=0.1, eps=1e-5):
    return tag_producer(StemConvBNAct.apply(x, weight, gamma, beta, rmean, rvar, (int(act), bool(training), float(momentum), float(eps))), gamma)


# ------------------------------------------------------------------------------------------------
# depthwise conv (+BatchNorm +activation)
# ------------------------------------------------------------------------------------------------
class DWConvBNAct(torch.autograd.Function):
    """ConvLayer2d.forward with groups == C (InvertedResidual.conv_3x3, cvnets/modules/mobilenetv2.py:194-207)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, rmean, rvar, cfg):
        stride, pad, dil, act, use_bn, training, momentum, eps = cfg
        _check_dev(x)
        B, C, H, W = x.shape
        K = weight.shape[-1]
        Ho = (H + 2 * pad - dil * (K - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (K - 1) - 1) // stride + 1
        M = B * Ho * Wo
        dev, dtype = x.device, x.dtype
        wp = pack_weight(weight, dtype, 2)
        y = nhwc_empty(B, C, Ho, Wo, dtype, dev)
        ctx.cfg = cfg
        ctx.shapes = (B, C, H, W, Ho, Wo, K)
        ctx.beta = beta
        part, R = None, 0
        if use_bn and training:
            R = _lib.query("cvh_dwconv_rows", B, Ho, Wo, C, K, stride, pad, dil)
            part = _f32(R * 2 * C, dev)
        _lib.call("cvh_dwconv_fwd", _dt(x), _p(x), _p(wp), _p(y), B, H, W, Ho, Wo, C, K, stride, pad, dil, _p(part), _stream())
        if not use_bn:
            if act != ACT_NONE:
                raise NotImplementedError("depthwise conv + activation without normalisation")
            ctx.save_for_backward(x, weight, None, None, None)
            return y
        stats = _bn_forward(y, M, C, part, R, gamma, beta, rmean, rvar, training, momentum, eps)
        out = nhwc_empty(B, C, Ho, Wo, dtype, dev)
        _lib.call("cvh_bn_apply", _dt(y), _p(y), _p(stats[2]), _p(stats[3]), act, None, _p(out), M, C, _stream())
        ctx.save_for_backward(x, weight, y, stats, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        stride, pad, dil, act, use_bn, training, momentum, eps = ctx.cfg
        B, C, H, W, Ho, Wo, K = ctx.shapes
        x, weight, y, stats, gamma = ctx.saved_tensors
        dout = as_nhwc(dout)
        dev, dtype = dout.device, dout.dtype
        M = B * Ho * Wo
        dgamma = dbeta = None
        if use_bn:
            dy, dgamma, dbeta = _bn_backward(y, dout, stats, gamma, act, M, C, training, beta=ctx.beta)
        else:
            dy = dout
        R = _lib.query("cvh_dwconv_bwd_w_rows", B, Ho, Wo, C, K, stride, pad, dil)
        part = _f32(R * C * K * K, dev)
        _lib.call("cvh_dwconv_bwd_w", _dt(x), _p(x), _p(dy), _p(part), B, H, W, Ho, Wo, C, K, stride, pad, dil, _stream())
        sink = _grad_sink(weight)
        dw = None if sink is not None else torch.empty(weight.shape, dtype=torch.float32, device=dev)
        if not (sink is not None and defer_reduce(part, sink, R, C * K * K, C * K * K)):
            _lib.call("cvh_sum_partials", _p(part), R, C * K * K, C * K * K, _p(sink if sink is not None else dw), 1.0,
                      1 if sink is not None else 0, _stream())
        dx = None
        if ctx.needs_input_grad[0]:
            wp = pack_weight(weight, dtype, 2)
            dx = nhwc_empty(B, C, H, W, dtype, dev)
            _lib.call("cvh_dwconv_bwd_x", _dt(dy), _p(dy), _p(wp), _p(dx), B, H, W, Ho, Wo, C, K, stride, pad, dil, _stream())
        return dx, dw, dgamma, dbeta, None, None, None


def dwconv_bn_act(x, weight, gamma=None, beta=None, rmean=None, rvar=None, *, stride=1, pad=1, dil=1, act=ACT_NONE, use_bn=True,
                  training=True, momentum=0.1, eps=1e-5):
    cfg = (int(stride), int(pad), int(dil), int(act), bool(use_bn), bool(training), float(momentum), float(eps))
    return DWConvBNAct.apply(x, weight, gamma, beta, rmean, rvar, cfg)


# ------------------------------------------------------------------------------------------------
# standalone BatchNorm (+act): used when a norm layer is called outside a ConvLayer2d
# ------------------------------------------------------------------------------------------------
class BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, cfg):
        act, training, momentum, eps = cfg
        _check_dev(x)
        B, C, H, W = x.shape
        M = B * H * W
        part, R = None, 0
        if training:
            R = _lib.query("cvh_colreduce_rows", M, C)
            part = _f32(R * 2 * C, x.device)
            _lib.call("cvh_bn_stats", _dt(x), _p(x), M, C, _p(part), _stream())
        stats = _bn_forward(x, M, C, part, R, gamma, beta, rmean, rvar, training, momentum, eps)
        out = torch.empty_like(x)
        _lib.call("cvh_bn_apply", _dt(x), _p(x), _p(stats[2]), _p(stats[3]), act, None, _p(out), M, C, _stream())
        ctx.cfg = cfg
        ctx.beta = beta
        ctx.save_for_backward(x, stats, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        act, training, momentum, eps = ctx.cfg
        x, stats, gamma = ctx.saved_tensors
        B, C, H, W = x.shape
        dx, dgamma, dbeta = _bn_backward(x, as_nhwc(dout), stats, gamma, act, B * H * W, C, training, beta=ctx.beta)
        return dx, dgamma, dbeta, None, None, None


# ------------------------------------------------------------------------------------------------
# linear on token matrices (+bias +act +dropout +residual)
# ------------------------------------------------------------------------------------------------
# Dropout backward without a pass of its own (csrc/layernorm.hip, DROP): x = res + Dropout(linear(h)) is followed by a LayerNorm in every
# TransformerEncoder; the LayerNorm backward kernel that produces d(x) also stores d(x) * keep-mask, which is what that linear's dW / dX GEMMs
# consume.  The hand-over: `linear` tags its output with the draw (p, stream id, seed snapshot); the LayerNorm functions remember the tag of
# their input; their backward leaves the masked gradient in a one-entry slot keyed by the address of the gradient they return, and the
# linear's backward takes it when address, draw and shape match (anything else — a summed gradient, a hook, another consumer — misses the
# slot and runs cvh_dropout as before).
# OFF by default (CVH_LN_DROP=1 switches it on): measured in the 1024-image MobileViT-S step (profiles/r06_ab_runs.txt) the LayerNorm backward
# kernels grow by 1.18 ms (18 launches, one more write stream each) while the 20 dropout launches they replace cost 0.75 ms — those re-read a
# tensor the previous kernel has just written, largely out of the 256 MB memory-side cache, so the "saved" read was cheap to begin with.
_LN_DROP = os.environ.get("CVH_LN_DROP", "0") != "0"
_dropped_slot = None


def _drop_tag_of(x):
    tag = getattr(x, "_cvh_drop", None) if _LN_DROP else None
    if tag is None or tag[3] != x._version or tag[4] != x.data_ptr():
        return None
    return tag[:3]


def _ln_backward_launch(x, dout, gamma, mr, dx, part, rows, C, dres, drop):
    """cvh_layernorm_bwd_res; with `drop` = (p, stream id, seed) of the Dropout in front of this LayerNorm also the masked copy of dx"""
    global _dropped_slot
    if drop is not None and _lib.query("cvh_ln_bwd_drop_ok", C):
        p, sid, seed = drop
        dxd = torch.empty_like(dx)
        _lib.call("cvh_layernorm_bwd_res_drop", _dt(x), _p(x), _p(dout), _p(gamma), _p(mr[0]), _p(mr[1]), _p(dx), _p(part), rows, C, _p(dres),
                  _p(dxd), float(p), _p(seed), int(sid), _stream())
        _dropped_slot = (dx.data_ptr(), tuple(dx.shape), dx.dtype, float(p), int(sid), seed.data_ptr(), dxd)
        return
    _lib.call("cvh_layernorm_bwd_res", _dt(x), _p(x), _p(dout), _p(gamma), _p(mr[0]), _p(mr[1]), _p(dx), _p(part), rows, C, _p(dres), _stream())


def _take_dropped(dout, p, sid, seed):
    global _dropped_slot
    slot, _dropped_slot = _dropped_slot, None
    if slot is None or seed is None:
        return None
    ptr, shape, dtype, sp, ssid, sseed, dxd = slot
    if ptr == dout.data_ptr() and shape == tuple(dout.shape) and dtype == dout.dtype and sp == float(p) and ssid == int(sid) and sseed == seed.data_ptr():
        return dxd
    return None


class LinearAct(torch.autograd.Function):
    """LinearLayer.forward = F.linear (cvnets/layers/linear_layer.py:74-91) with the activation / Dropout / residual
    add that follow it in TransformerEncoder (cvnets/modules/transformer.py:140-155) fused into the GEMM epilogue.

    FFN pairing (fc1 -> act -> fc2): with ``expose_pre`` fc1 also returns its pre-activation tensor; fc2 takes it as ``in_pre`` and
    its dX GEMM multiplies by act'(pre) in the epilogue, handing fc1 the gradient of the PRE-activation directly — the separate
    activation-backward pass over the 4x-wide hidden tensor (2 reads + 1 write) disappears."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, in_pre, cfg):
        act, drop_p, stream_id, expose_pre, in_act = cfg
        _check_dev(x)
        rows, K = x.shape
        N = weight.shape[0]
        if K % 8 or N % 8:
            raise RuntimeError("cvnets_amd linear layers need in/out features % 8 == 0")
        dev, dtype = x.device, x.dtype
        wp = pack_weight(weight, dtype, 0)
        out = torch.empty((rows, N), dtype=dtype, device=dev)
        pre = torch.empty((rows, N), dtype=dtype, device=dev) if act != ACT_NONE else None
        seed = dropout_seed(dev) if drop_p > 0 else None
        _conv_gemm(x, None, K, 0, wp, out, rows, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, bias=bias, act=act, save_pre=pre, residual=residual,
                   drop_p=drop_p, seed=seed, stream_id=stream_id)
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.bias = bias
        ctx.seed = seed  # this forward's snapshot: the backward regenerates exactly this mask
        ctx.save_for_backward(x, weight, pre, in_pre)
        if expose_pre:
            if pre is None:
                raise RuntimeError("expose_pre needs an activation")
            ctx.set_materialize_grads(False)
            return out, pre
        return out

    @staticmethod
    def backward(ctx, dout, dpre=None):
        act, drop_p, stream_id, expose_pre, in_act = ctx.cfg
        x, weight, pre, in_pre = ctx.saved_tensors
        rows, K = x.shape
        N = weight.shape[0]
        if expose_pre and dout is None:
            dy = dpre.contiguous()  # already the gradient of the pre-activation (fused into the consumer's dX GEMM)
            dev, dtype = dy.device, dy.dtype
        else:
            dev, dtype = dout.device, dout.dtype
            dout = dout.contiguous()
            dy = dout
            if drop_p > 0:
                dy = _take_dropped(dout, drop_p, stream_id, ctx.seed)  # formed by the LayerNorm backward that produced dout
                if dy is None:
                    dy = torch.empty_like(dout)
                    _lib.call("cvh_dropout", _dt(dout), _p(dout), _p(dy), rows * N, float(drop_p), _p(ctx.seed), stream_id, _stream())
            if act != ACT_NONE:  # (ACT_GELU_D: `pre` holds GELU'(pre-activation) - one multiply)
                dy = _act_backward(pre, dy, ACT_DERIV if act == ACT_GELU_D else act, rows, N)
            if expose_pre and dpre is not None:
                dy = add(dy, dpre.contiguous())
        dbias = None
        bsink = _grad_sink(ctx.bias) if ctx.has_bias else None
        if bsink is not None:  # the bias gradient rides along in the dW kernel where it can
            dw_ret, folded = _weight_grad(dy, x, None, K, 0, weight, rows, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, K, bias_sink=bsink)
            if not folded:
                _colsum(dy, rows, N, bsink)
        else:
            if ctx.has_bias:
                dbias = _colsum(dy, rows, N, None)
            dw_ret = _weight_grad(dy, x, None, K, 0, weight, rows, 1, 1, 1, 1, 1, 1, 1, 0, 1, N, K)
        dx = d_in_pre = None
        if ctx.needs_input_grad[0] or (in_pre is not None and ctx.needs_input_grad[4]):
            wpt = pack_weight(weight, dtype, 1)  # [K][N]
            dx = torch.empty((rows, K), dtype=dtype, device=dev)
            if in_pre is not None:  # x = act(in_pre): hand the producer d(in_pre) = (dy W) * act'(in_pre)
                _conv_gemm(dy, None, N, 0, wpt, dx, rows, 1, 1, 1, 1, 1, 1, 1, 0, 1, K, actgrad_aux=in_pre, actgrad_act=in_act)
                d_in_pre, dx = dx, None
            else:
                _conv_gemm(dy, None, N, 0, wpt, dx, rows, 1, 1, 1, 1, 1, 1, 1, 0, 1, K)
        dres = dout if (ctx.has_res and dout is not None) else None
        return dx, dw_ret, dbias, dres, d_in_pre, None


def ffn_stores_derivative(x2d, weight, act) -> bool:
    """transformer-sized FFN (ViT-B / CLIP: 768 -> 3072): fc1 stores GELU'(pre) instead of the pre-activation and fc2's dX GEMM multiplies by
    it in its epilogue (ACT_GELU_D / ACT_DERIV) - where the large-tile kernel with that epilogue takes the product"""
    if act != ACT_GELU or x2d.dtype != torch.bfloat16 or not x2d.is_cuda:
        return False
    return _lib.query("cvh_conv_gemm_takes_gelu_d", x2d.shape[0], x2d.shape[1], weight.shape[0]) == 1


def linear(x2d, weight, bias=None, *, act=ACT_NONE, drop_p=0.0, residual=None, expose_pre=False, in_pre=None, in_act=ACT_NONE):
    sid = next_stream_id() if drop_p > 0 else 0
    _trace_site("linear", sid, drop_p, (x2d.shape[0], weight.shape[0]))
    res = LinearAct.apply(x2d, weight, bias, residual, in_pre, (int(act), float(drop_p), sid, bool(expose_pre), int(in_act)))
    if drop_p > 0 and _LN_DROP and isinstance(res, torch.Tensor):
        res._cvh_drop = (float(drop_p), sid, dropout_seed(x2d.device), res._version, res.data_ptr())
    return res


# ------------------------------------------------------------------------------------------------
# reference quirk: channel-first LayerNorm branch hit by [B', S, C] token tensors with S == C
# ------------------------------------------------------------------------------------------------
class LayerNormSeqFn(torch.autograd.Function):
    """cvnets/layers/normalization/layer_norm.py:53-66 as the reference evaluates it on a token tensor whose sequence length equals
    its channel count: (x - mean_over_tokens) / (std_over_tokens + eps), weight / bias indexed by the TOKEN position."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, seqmap):
        _check_dev(x)
        nseq, S, ph, pw, n_w, H, W = seqmap
        rows, C = x.shape
        y = torch.empty_like(x)
        stats = _f32(nseq * 2 * C, x.device)
        _lib.call("cvh_ln_seq_fwd", _dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(stats), nseq, S, C, ph, pw, n_w, H, W, float(eps), _stream())
        ctx.save_for_backward(x, gamma, stats)
        ctx.cfg = (float(eps), tuple(seqmap))
        return y

    @staticmethod
    def backward(ctx, dout):
        x, gamma, stats = ctx.saved_tensors
        eps, seqmap = ctx.cfg
        nseq, S, ph, pw, n_w, H, W = seqmap
        C = x.shape[1]
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        part = _f32(nseq * 2 * S, x.device)
        _lib.call("cvh_ln_seq_bwd", _dt(x), _p(x), _p(dout), _p(gamma), _p(stats), _p(dx), _p(part), nseq, S, C, ph, pw, n_w, H, W, eps, _stream())
        dgb = _f32(2 * S, x.device)
        _lib.call("cvh_sum_partials", _p(part), nseq, 2 * S, 2 * S, _p(dgb), 1.0, 0, _stream())
        return dx, dgb[:S], dgb[S:], None, None


class LayerNormChannelFirstFn(torch.autograd.Function):
    """cvnets/layers/normalization/layer_norm.py:53-66 on a genuine feature map: per pixel, over its channels,
    (x - mean) / (biased std + eps) * weight[c] + bias[c].  `x` is the [pixels, C] token matrix of the NHWC map."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _check_dev(x)
        rows, C = x.shape
        y = torch.empty_like(x)
        mr = _f32(2, x.device, rows)
        _lib.call("cvh_layernorm_cf_fwd", _dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(mr[0]), _p(mr[1]), rows, C, float(eps), _stream())
        ctx.eps = float(eps)
        ctx.save_for_backward(x, gamma, mr)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, gamma, mr = ctx.saved_tensors
        rows, C = x.shape
        dout = dout.contiguous()
        R = _lib.query("cvh_ln_bwd_rows", rows)
        part = _f32(R * 2 * C, x.device)
        dx = torch.empty_like(x)
        _lib.call("cvh_layernorm_cf_bwd", _dt(x), _p(x), _p(dout), _p(gamma), _p(mr[0]), _p(mr[1]), _p(dx), _p(part), rows, C, ctx.eps, _stream())
        dgb = _f32(2 * C, x.device)
        _lib.call("cvh_sum_partials", _p(part), R, 2 * C, 2 * C, _p(dgb), 1.0, 0, _stream())
        return dx, dgb[:C], dgb[C:], None


def layer_norm_channel_first(x2d, gamma, beta, eps):
    return LayerNormChannelFirstFn.apply(x2d.contiguous(), gamma, beta, float(eps))


def layer_norm_tokens(x2d, ln, seqmap):
    """LayerNorm of a token matrix whose sequences are described by `seqmap`, reproducing WHICH branch the reference's LayerNorm
    takes for the equivalent [B', S, C] tensor: channel-last F.layer_norm normally, the channel-first formula when S == C
    (layer_norm.py:51-68).  `ln.reference_quirk = False` opts out (always the documented channel-last LayerNorm)."""
    S, C = seqmap[1], x2d.shape[1]
    if S == C and getattr(ln, "reference_quirk", True):
        return LayerNormSeqFn.apply(x2d, ln.weight, ln.bias, float(ln.eps), tuple(int(v) for v in seqmap))
    return layer_norm(x2d, ln.weight, ln.bias, ln.eps)


# ------------------------------------------------------------------------------------------------
# LayerNorm over the last dim of a token matrix
# ------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, drop=None):
        _check_dev(x)
        ctx.drop = drop
        rows, C = x.shape
        y = torch.empty_like(x)
        mr = _f32(2, x.device, rows)
        _lib.call("cvh_layernorm_fwd", _dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(mr[0]), _p(mr[1]), rows, C, float(eps), _stream())
        ctx.beta = beta
        ctx.save_for_backward(x, gamma, mr)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, gamma, mr = ctx.saved_tensors
        rows, C = x.shape
        dout = dout.contiguous()
        R = _lib.query("cvh_ln_bwd_rows", rows)
        part = _f32(R * 2 * C, x.device)
        dx = torch.empty_like(x)
        _ln_backward_launch(x, dout, gamma, mr, dx, part, rows, C, None, ctx.drop)
        sg, sb = _grad_sink(gamma), _grad_sink(ctx.beta)
        if sg is not None and sb is not None:
            if not (defer_reduce(part, sg, R, 2 * C, C) and defer_reduce(part, sb, R, 2 * C, C, part_offset=C)):
                _lib.call("cvh_sum_partials", _p(part), R, 2 * C, C, _p(sg), 1.0, 1, _stream())
                _lib.call("cvh_sum_partials", part.data_ptr() + 4 * C, R, 2 * C, C, _p(sb), 1.0, 1, _stream())
            return dx, None, None, None, None
        dgb = _f32(2 * C, x.device)
        _lib.call("cvh_sum_partials", _p(part), R, 2 * C, 2 * C, _p(dgb), 1.0, 0, _stream())
        return dx, dgb[:C], dgb[C:], None, None


def layer_norm(x2d, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x2d, gamma, beta, float(eps), _drop_tag_of(x2d))


_LN_FORK = os.environ.get("CVH_LN_FORK", "1") != "0"


class LayerNormForkFn(torch.autograd.Function):
    """x -> (x, LayerNorm(x)) as ONE autograd node: the pre-norm residual fork of TransformerEncoder (cvnets/modules/transformer.py:139-155,
    `res = x; x = norm(x); ...; x = x + res`).  With two separate consumers of x autograd sums their gradients in an elementwise add
    kernel (one read of each + one write per fork, 18 forks per MobileViT-S step); here the gradient of the pass-through output joins
    inside the LayerNorm backward kernel (cvh_layernorm_bwd_res)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, drop=None):
        _check_dev(x)
        rows, C = x.shape
        y = torch.empty_like(x)
        mr = _f32(2, x.device, rows)
        _lib.call("cvh_layernorm_fwd", _dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(mr[0]), _p(mr[1]), rows, C, float(eps), _stream())
        ctx.beta = beta
        ctx.drop = drop
        ctx.save_for_backward(x, gamma, mr)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dout):
        x, gamma, mr = ctx.saved_tensors
        rows, C = x.shape
        if dout is None:  # only the pass-through was used
            return dres, None, None, None, None
        dout = dout.contiguous()
        if dres is not None:
            dres = dres.contiguous()
            if dres.dtype != x.dtype:
                dres = dres.to(x.dtype)
        R = _lib.query("cvh_ln_bwd_rows", rows)
        part = _f32(R * 2 * C, x.device)
        dx = torch.empty_like(x)
        _ln_backward_launch(x, dout, gamma, mr, dx, part, rows, C, dres, ctx.drop)
        sg, sb = _grad_sink(gamma), _grad_sink(ctx.beta)
        if sg is not None and sb is not None:
            if not (defer_reduce(part, sg, R, 2 * C, C) and defer_reduce(part, sb, R, 2 * C, C, part_offset=C)):
                _lib.call("cvh_sum_partials", _p(part), R, 2 * C, C, _p(sg), 1.0, 1, _stream())
                _lib.call("cvh_sum_partials", part.data_ptr() + 4 * C, R, 2 * C, C, _p(sb), 1.0, 1, _stream())
            return dx, None, None, None, None
        dgb = _f32(2 * C, x.device)
        _lib.call("cvh_sum_partials", _p(part), R, 2 * C, 2 * C, _p(dgb), 1.0, 0, _stream())
        return dx, dgb[:C], dgb[C:], None, None


def layer_norm_fork(x2d, ln, seqmap):
    """returns (x, LayerNorm(x)); falls back to two consumers of x where the fused node does not apply (the reference's S == C quirk branch)"""
    S, C = seqmap[1], x2d.shape[1]
    if (S == C and getattr(ln, "reference_quirk", True)) or not torch.is_grad_enabled() or not x2d.requires_grad or not _LN_FORK:
        return x2d, layer_norm_tokens(x2d, ln, seqmap)
    return LayerNormForkFn.apply(x2d, ln.weight, ln.bias, float(ln.eps), _drop_tag_of(x2d))


# ------------------------------------------------------------------------------------------------
# fused multi-head self-attention on the qkv token matrix
# ------------------------------------------------------------------------------------------------
class AttentionFn(torch.autograd.Function):
    """MultiHeadAttention.forward_default lines 148-233 (cvnets/layers/multi_head_attention.py) between the two
    projections; `seqmap` = (nseq, S, ph, pw, n_w, H, W) realises MobileViTBlock.unfolding/folding by addressing."""

    @staticmethod
    def forward(ctx, qkv, kpm, bias, cfg):
        heads, seqmap, causal, drop_p, stream_id = cfg
        nseq, S, ph, pw, n_w, H, W = seqmap
        _check_dev(qkv)
        rows, d3 = qkv.shape
        d = d3 // 3
        c = d // heads
        if c > 64:
            raise NotImplementedError("head_dim > 64")
        out = torch.empty((rows, d), dtype=qkv.dtype, device=qkv.device)
        lse = _f32(nseq * heads * S, qkv.device)
        seed = dropout_seed(qkv.device) if drop_p > 0 else None  # the backward regenerates the keep mask from the same seed / stream id
        bstride = 0
        if bias is not None:  # additive mask [S, S] (shared) or [nseq, S, S], float32 (multi_head_attention.py:197-208)
            if bias.dtype != torch.float32 or not bias.is_contiguous() or tuple(bias.shape) not in ((S, S), (nseq, S, S)):
                raise RuntimeError(f"attention bias must be contiguous float32 [{S}, {S}] or [{nseq}, {S}, {S}]")
            bstride = S * S if bias.dim() == 3 else 0
        _lib.call("cvh_attn_fwd_mask", _dt(qkv), _p(qkv), _p(out), _p(lse), _p(kpm), _p(bias), bstride, nseq, S, heads, c, ph, pw, n_w, H, W,
                  float(c) ** -0.5, 1 if causal else 0, float(drop_p), _p(seed), stream_id, _stream())
        ctx.cfg = cfg
        ctx.seed = seed
        ctx.bstride = bstride
        ctx.save_for_backward(qkv, out, lse, kpm, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        heads, seqmap, causal, drop_p, stream_id = ctx.cfg
        nseq, S, ph, pw, n_w, H, W = seqmap
        qkv, out, lse, kpm, bias = ctx.saved_tensors
        d = qkv.shape[1] // 3
        c = d // heads
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        dsum = _f32(nseq * heads * S, qkv.device)
        _lib.call("cvh_attn_bwd_mask", _dt(qkv), _p(qkv), _p(out), _p(dout), _p(dqkv), _p(lse), _p(dsum), _p(kpm), _p(bias), ctx.bstride, nseq, S,
                  heads, c, ph, pw, n_w, H, W, float(c) ** -0.5, 1 if causal else 0, float(drop_p), _p(ctx.seed), stream_id, _stream())
        return dqkv, None, None, None


def attention(qkv2d, heads: int, seqmap: Tuple[int, ...], causal: bool = False, key_padding_mask: Optional[torch.Tensor] = None,
              drop_p: float = 0.0, attn_bias: Optional[torch.Tensor] = None):
    """drop_p > 0: dropout on the attention probabilities (the caller passes it in training mode only), mask regenerated in backward.
    attn_bias: additive mask [S, S] or [nseq, S, S] (added to the scaled scores before the softmax; a constant: no gradient)."""
    kpm = None
    if key_padding_mask is not None:
        kpm = (key_padding_mask != 0).to(torch.uint8).contiguous()  # plumbing; any non-zero entry (True, 1, -inf) masks the key, as .to(torch.bool) does in the reference
    bias = None
    if attn_bias is not None:
        bias = attn_bias.detach().to(torch.float32).contiguous()  # plumbing
    sid = next_stream_id() if drop_p > 0 else 0
    _trace_site("attention", sid, drop_p, qkv2d.shape)
    return AttentionFn.apply(qkv2d, kpm, bias, (int(heads), tuple(int(v) for v in seqmap), bool(causal), float(drop_p), sid))


# ------------------------------------------------------------------------------------------------
# global average pool
# ------------------------------------------------------------------------------------------------
class AddFn(torch.autograd.Function):
    """y = a + b for same-layout activations (residual adds that cannot ride in a GEMM epilogue, e.g. after a dropout)."""

    @staticmethod
    def forward(ctx, a, b):
        _check_dev(a)
        if a.shape != b.shape or a.stride() != b.stride() or a.dtype != b.dtype:
            raise RuntimeError("add: operands must share shape, layout and dtype")
        y = torch.empty_like(a)
        _lib.call("cvh_add", _dt(a), _p(a), _p(b), _p(y), a.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class Fork2(torch.autograd.Function):
    """x -> (x, x) for a feature map with two consumers inside one module (MobileViTBlock: local_rep and the concat in front of the fusion
    conv, cvnets/modules/mobilevit_block.py:270-287).  Forward is free (two aliases); backward sums the two gradients with cvh_add —
    otherwise autograd's own accumulation does it with an ATen kernel inside the captured step."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        if g1.shape != g2.shape or g1.stride() != g2.stride() or g1.dtype != g2.dtype or not g1.is_cuda:
            return g1 + g2  # plumbing (foreign layouts)
        y = torch.empty_like(g1)
        _lib.call("cvh_add", _dt(g1), _p(g1), _p(g2), _p(y), g1.numel(), _stream())
        return y


def fork2(x):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    return Fork2.apply(x)


def add(a, b):
    return AddFn.apply(a, b)


class GroupNorm1Fn(torch.autograd.Function):
    """nn.GroupNorm(num_groups=1) == LayerNorm2D_NCHW (cvnets/layers/normalization/layer_norm.py:75-108) on an NHWC map."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _check_dev(x)
        B, C, H, W = x.shape
        chunks = _lib.query("cvh_gn_chunks", B, H * W, C)
        y = torch.empty_like(x)
        stats = _f32(B * 2, x.device)
        part = _f32(B * chunks * 2, x.device)
        _lib.call("cvh_gn_fwd", _dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(part), B, H * W, C, float(eps), _stream())
        ctx.save_for_backward(x, gamma, stats)
        ctx.params = (gamma, beta)
        ctx.chunks = chunks
        return y

    @staticmethod
    def backward(ctx, dout):
        x, gamma, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        dout = as_nhwc(dout)
        chunks = ctx.chunks
        part = _f32(B * chunks * 2 * C, x.device)
        coeff = _f32(B * 2, x.device)
        dx = torch.empty_like(x)
        _lib.call("cvh_gn_bwd", _dt(x), _p(x), _p(dout), _p(stats), _p(gamma), _p(dx), _p(part), _p(coeff), B, H * W, C, _stream())
        gp, bp = ctx.params
        sg, sb = _grad_sink(gp), _grad_sink(bp)
        if sg is not None and sb is not None:
            _lib.call("cvh_sum_partials", _p(part), B * chunks, 2 * C, C, _p(sb), 1.0, 1, _stream())
            _lib.call("cvh_sum_partials", part.data_ptr() + 4 * C, B * chunks, 2 * C, C, _p(sg), 1.0, 1, _stream())
            return dx, None, None, None
        dgb = _f32(2 * C, x.device)
        _lib.call("cvh_sum_partials", _p(part), B * chunks, 2 * C, 2 * C, _p(dgb), 1.0, 0, _stream())
        return dx, dgb[C:], dgb[:C], None


def group_norm1(x, gamma, beta, eps=1e-5):
    return GroupNorm1Fn.apply(x, gamma, beta, float(eps))


class LinearAttnFn(torch.autograd.Function):
    """LinearSelfAttention core (cvnets/layers/linear_attention.py:147-162) between qkv_proj and out_proj, on the un-unfolded NHWC
    map: kvq [B, 2C+8, H, W] (key | value | query | zero pad) -> relu(value) * context_vector [B, C, H, W]."""

    @staticmethod
    def forward(ctx, kvq, C, ph, pw):
        _check_dev(kvq)
        B, LD, H, W = kvq.shape
        if LD != 2 * C + 8:
            raise RuntimeError("kvq tensor must have 2C+8 channels")
        out = nhwc_empty(B, C, H, W, kvq.dtype, kvq.device)
        cv = _f32(B * ph * pw * C, kvq.device)
        _lib.call("cvh_linattn_fwd", _dt(kvq), _p(kvq), _p(out), _p(cv), B, H, W, ph, pw, C, _stream())
        ctx.save_for_backward(kvq, cv)
        ctx.geom = (C, ph, pw)
        return out

    @staticmethod
    def backward(ctx, dout):
        kvq, cv = ctx.saved_tensors
        C, ph, pw = ctx.geom
        B, LD, H, W = kvq.shape
        dout = as_nhwc(dout)
        dkvq = torch.empty_like(kvq)
        _lib.call("cvh_linattn_bwd", _dt(kvq), _p(kvq), _p(cv), _p(dout), _p(dkvq), B, H, W, ph, pw, C, _stream())
        return dkvq, None, None, None


def linear_attention(kvq, C: int, ph: int, pw: int):
    return LinearAttnFn.apply(kvq, int(C), int(ph), int(pw))


class GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _check_dev(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C), dtype=x.dtype, device=x.device)
        _lib.call("cvh_pool_fwd", _dt(x), _p(x), _p(y), B, H * W, C, _stream())
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dy.contiguous()
        dx = nhwc_empty(B, C, H, W, dy.dtype, dy.device)
        _lib.call("cvh_pool_bwd", _dt(dy), _p(dy), _p(dx), B, H * W, C, _stream())
        return dx


class AdaptiveAvgPool(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(OS) on an NHWC map -> [B, C, OS, OS] (cvnets/modules/pspnet_module.py:73-88; OS = 1: ASPP image pooling)"""

    @staticmethod
    def forward(ctx, x, OS):
        _check_dev(x)
        B, C, H, W = x.shape
        y = nhwc_empty(B, C, OS, OS, x.dtype, x.device)
        _lib.call("cvh_adaptive_pool_fwd", _dt(x), _p(x), _p(y), B, H, W, C, OS, _stream())
        ctx.shape = (B, C, H, W, OS)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, OS = ctx.shape
        dy = as_nhwc(dy)
        dx = nhwc_empty(B, C, H, W, dy.dtype, dy.device)
        _lib.call("cvh_adaptive_pool_bwd", _dt(dy), _p(dy), _p(dx), B, H, W, C, OS, _stream())
        return dx, None


def adaptive_avg_pool(x, OS: int):
    return AdaptiveAvgPool.apply(to_nhwc(x), int(OS))


class ResizeBilinear(torch.autograd.Function):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) on an NHWC map (mobilevit_block.py:191-200, 260-266)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        _check_dev(x)
        B, C, H, W = x.shape
        y = nhwc_empty(B, C, Ho, Wo, x.dtype, x.device)
        _lib.call("cvh_resize_bilinear_fwd", _dt(x), _p(x), _p(y), B, H, W, Ho, Wo, C, align, _stream())
        ctx.shape = (B, C, H, W, Ho, Wo, align)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, Ho, Wo, align = ctx.shape
        dy = as_nhwc(dy)
        dx = nhwc_empty(B, C, H, W, dy.dtype, dy.device)
        _lib.call("cvh_resize_bilinear_bwd", _dt(dy), _p(dy), _p(dx), B, H, W, Ho, Wo, C, align, _stream())
        return dx, None, None, None


def resize_bilinear(x, Ho: int, Wo: int, align_corners: bool = False):
    return ResizeBilinear.apply(x, int(Ho), int(Wo), 1 if align_corners else 0)


class VitEmbed(torch.autograd.Function):
    """patch tokens + positional embedding (+ class token)  ->  [B*(1+N), E]   (vit.py:480-509).  pos [N,E] / cls [E] are fp32."""

    @staticmethod
    def forward(ctx, patch, pos, cls, B):
        _check_dev(patch)
        N, E = pos.shape
        S = N + (1 if cls is not None else 0)
        out = torch.empty((B * S, E), dtype=patch.dtype, device=patch.device)
        _lib.call("cvh_vit_embed_fwd", _dt(patch), _p(patch), _p(pos), _p(cls), _p(out), B, N, E, _stream())
        ctx.meta = (B, N, E, cls is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, E, has_cls = ctx.meta
        S = N + (1 if has_cls else 0)
        dout = dout.contiguous()
        dpatch = torch.empty((B * N, E), dtype=dout.dtype, device=dout.device)
        _lib.call("cvh_vit_embed_bwd", _dt(dout), _p(dout), _p(dpatch), B, N, E, 1 if has_cls else 0, _stream())
        dsum = _f32(S * E, dout.device)
        _lib.call("cvh_batch_sum", _dt(dout), _p(dout), _p(dsum), B, S * E, 0, _stream())
        if has_cls:
            return dpatch, dsum[E:].view(N, E), dsum[:E], None
        return dpatch, dsum.view(N, E), None, None


class RowsGather(torch.autograd.Function):
    """x [B*S, E] -> rows b*S + idx (one per sequence), e.g. the class-token embeddings (vit.py:562-565)."""

    @staticmethod
    def forward(ctx, x, B, S, idx):
        _check_dev(x)
        E = x.shape[1]
        y = torch.empty((B, E), dtype=x.dtype, device=x.device)
        _lib.call("cvh_rows_copy", _dt(x), x.data_ptr() + idx * E * x.element_size(), _p(y), B, E, S * E, E, _stream())
        ctx.meta = (B, S, E, idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, S, E, idx = ctx.meta
        dy = dy.contiguous()
        dx = torch.zeros((B * S, E), dtype=dy.dtype, device=dy.device)  # plumbing: every other row has zero gradient
        _lib.call("cvh_rows_copy", _dt(dy), _p(dy), dx.data_ptr() + idx * E * dx.element_size(), B, E, E, S * E, _stream())
        return dx, None, None, None


class EmbedLookup(torch.autograd.Function):
    """token embedding + positional embedding for the text tower (cvnets/text_encoders/transformer.py:321-341):
    tokens [B, S] int64, table [V, E] fp32, pos [S, E] fp32 or None -> [B*S, E] in the compute dtype."""

    @staticmethod
    def forward(ctx, tokens, table, pos, padding_idx, dtype):
        _check_dev(table)
        B, S = tokens.shape
        V, E = table.shape
        tokens = tokens.contiguous()
        out = torch.empty((B * S, E), dtype=dtype, device=table.device)
        _lib.call("cvh_embed_lookup_fwd", _dt(out), _p(tokens), _p(table), _p(pos), _p(out), B * S, S, E, _stream())
        ctx.save_for_backward(tokens)
        ctx.meta = (B, S, V, E, -1 if padding_idx is None else int(padding_idx), pos is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        B, S, V, E, pad, has_pos = ctx.meta
        dout = dout.contiguous()
        dtable = torch.zeros((V, E), dtype=torch.float32, device=dout.device)  # plumbing: scatter-add target
        _lib.call("cvh_embed_lookup_bwd", _dt(dout), _p(tokens), _p(dout), _p(dtable), B * S, E, pad, _stream())
        dpos = None
        if has_pos:
            dpos = _f32(S * E, dout.device).view(S, E)
            _lib.call("cvh_batch_sum", _dt(dout), _p(dout), _p(dpos), B, S * E, 0, _stream())
        return None, dtable, dpos, None, None


class RowsGatherIdx(torch.autograd.Function):
    """x [R_src, C], rows int64 [R] -> x[rows]  (EOT-token embeddings, text_encoders/transformer.py:413-421)."""

    @staticmethod
    def forward(ctx, x, rows):
        _check_dev(x)
        R, C = rows.shape[0], x.shape[1]
        y = torch.empty((R, C), dtype=x.dtype, device=x.device)
        _lib.call("cvh_rows_gather_idx", _dt(x), _p(x), _p(rows), _p(y), R, C, 0, _stream())
        ctx.save_for_backward(rows)
        ctx.n_src = x.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.zeros((ctx.n_src, dy.shape[1]), dtype=dy.dtype, device=dy.device)  # plumbing: all other rows have zero gradient
        _lib.call("cvh_rows_gather_idx", _dt(dy), _p(dy), _p(rows), _p(dx), rows.shape[0], dy.shape[1], 1, _stream())
        return dx, None


class L2Normalize(torch.autograd.Function):
    """F.normalize(x, dim=-1) on [R, C]."""

    @staticmethod
    def forward(ctx, x, eps):
        _check_dev(x)
        x = x.contiguous()
        R, C = x.shape
        y = torch.empty_like(x)
        inv = _f32(R, x.device)
        _lib.call("cvh_l2norm_fwd", _dt(x), _p(x), _p(y), _p(inv), R, C, float(eps), _stream())
        ctx.save_for_backward(y, inv)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.call("cvh_l2norm_bwd", _dt(y), _p(y), _p(dy), _p(inv), _p(dx), y.shape[0], y.shape[1], ctx.eps, _stream())
        return dx, None


def l2_normalize(x, eps: float = 1e-12):
    return L2Normalize.apply(x, eps)


class ScaledCrossEntropy(torch.autograd.Function):
    """mean_i CE(scale * logits[i], i + label_offset)  (contrastive_loss_clip.py:77-94).  logits [N, M]; scale: 0-d fp32 tensor."""

    @staticmethod
    def forward(ctx, logits, scale, label_offset):
        _check_dev(logits)
        logits = logits.contiguous()
        N, M = logits.shape
        scale = scale.detach().float().reshape(1)  # plumbing: scalar
        rows = _f32(N, logits.device)
        lse = _f32(N, logits.device)
        _lib.call("cvh_scaled_ce_fwd", _dt(logits), _p(logits), _p(scale), _p(rows), _p(lse), N, M, int(label_offset), _stream())
        ctx.save_for_backward(logits, scale, lse)
        ctx.label_offset = int(label_offset)
        return rows.mean()  # plumbing: N-element reduction

    @staticmethod
    def backward(ctx, g):
        logits, scale, lse = ctx.saved_tensors
        N, M = logits.shape
        gout = (g.float() / N).reshape(1)  # plumbing: scalar
        dlogits = torch.empty_like(logits)
        ds_rows = _f32(N, logits.device)
        _lib.call("cvh_scaled_ce_bwd", _dt(logits), _p(logits), _p(scale), _p(lse), _p(gout), _p(dlogits), _p(ds_rows), N, M, ctx.label_offset,
                  _stream())
        return dlogits, ds_rows.sum().reshape(()), None


class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits [N, M], labels [N], ignore_index, label_smoothing), mean over the non-ignored rows
    (loss_fn/classification/cross_entropy.py:65-92)."""

    @staticmethod
    def forward(ctx, logits, labels, label_smoothing, ignore_index):
        _check_dev(logits)
        logits = logits.contiguous()
        labels = labels.contiguous()
        N, M = logits.shape
        rows, lse = _f32(N, logits.device), _f32(N, logits.device)
        _lib.call("cvh_ce_fwd", _dt(logits), _p(logits), _p(labels), float(label_smoothing), int(ignore_index), _p(rows), _p(lse), N, M, _stream())
        ctx.cfg = (float(label_smoothing), int(ignore_index))
        if N <= (1 << 20):  # classification batches: mean over the non-ignored rows and its reciprocal count in one launch
            out2 = _f32(2, logits.device)
            _lib.call("cvh_ce_mean", _p(rows), _p(labels), int(ignore_index), _p(out2), N, _stream())
            ctx.save_for_backward(logits, labels, lse, out2[1:2])
            ctx.inv = True
            return out2[0]
        n_valid = (labels != ignore_index).sum().clamp_(min=1).float()  # plumbing: label bookkeeping (dense-prediction sized label maps)
        ctx.save_for_backward(logits, labels, lse, n_valid)
        ctx.inv = False
        return rows.sum() / n_valid  # plumbing: N-element reduction

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse, n_valid = ctx.saved_tensors
        eps, ignore = ctx.cfg
        N, M = logits.shape
        gout = (g.float() * n_valid if ctx.inv else g.float() / n_valid).reshape(1)  # plumbing: scalar (n_valid holds 1 / count when ctx.inv)
        dlogits = torch.empty_like(logits)
        _lib.call("cvh_ce_bwd", _dt(logits), _p(logits), _p(labels), _p(lse), _p(gout), eps, ignore, _p(dlogits), N, M, _stream())
        return dlogits, None, None, None


class SoftCrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits [N, M], class probabilities [N, M], label_smoothing): the targets RandomMixup / RandomCutmix produce"""

    @staticmethod
    def forward(ctx, logits, target, label_smoothing):
        _check_dev(logits)
        logits = logits.contiguous()
        target = target.float().contiguous()  # plumbing
        N, M = logits.shape
        rows, lse, tsum = _f32(N, logits.device), _f32(N, logits.device), _f32(N, logits.device)
        _lib.call("cvh_ce_soft_fwd", _dt(logits), _p(logits), _p(target), float(label_smoothing), _p(rows), _p(lse), _p(tsum), N, M, _stream())
        ctx.save_for_backward(logits, target, lse, tsum)
        ctx.eps = float(label_smoothing)
        return rows.sum() / N  # plumbing: N-element reduction

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, tsum = ctx.saved_tensors
        N, M = logits.shape
        gout = (g.float() / N).reshape(1)  # plumbing: scalar
        dlogits = torch.empty_like(logits)
        _lib.call("cvh_ce_soft_bwd", _dt(logits), _p(logits), _p(target), _p(lse), _p(tsum), _p(gout), ctx.eps, _p(dlogits), N, M, _stream())
        return dlogits, None, None


def cross_entropy(logits, labels, label_smoothing: float = 0.0, ignore_index: int = -1):
    if labels.dim() == 2:  # class probabilities (mixup / cutmix)
        return SoftCrossEntropyFn.apply(logits, labels, float(label_smoothing))
    return CrossEntropyFn.apply(logits, labels, float(label_smoothing), int(ignore_index))


def scaled_cross_entropy(logits, scale, label_offset: int = 0):
    return ScaledCrossEntropy.apply(logits, scale, int(label_offset))


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, stream_id):
        _check_dev(x)
        if not (x.is_contiguous() or is_nhwc(x)):
            x = x.contiguous()
        y = torch.empty_like(x)  # same (dense) layout as x: the mask is a function of the MEMORY index, NHWC maps stay NHWC
        seed = dropout_seed(x.device)
        _lib.call("cvh_dropout", _dt(x), _p(x), _p(y), x.numel(), float(p), _p(seed), stream_id, _stream())
        ctx.seed = seed
        ctx.cfg = (p, stream_id, is_nhwc(x) and not x.is_contiguous())
        return y

    @staticmethod
    def backward(ctx, dy):
        p, stream_id, nhwc = ctx.cfg
        dy = as_nhwc(dy) if nhwc else dy.contiguous()  # the gradient must be walked in the layout the forward mask was drawn in
        dx = torch.empty_like(dy)
        _lib.call("cvh_dropout", _dt(dy), _p(dy), _p(dx), dy.numel(), float(p), _p(ctx.seed), stream_id, _stream())
        return dx, None, None


def dropout(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    sid = next_stream_id()
    _trace_site("dropout", sid, p, x.shape)
    return DropoutFn.apply(x, float(p), sid)


class Dropout2dFn(torch.autograd.Function):
    """nn.Dropout2d on an NHWC map (cvnets/layers/dropout.py:32-50): whole channels of a sample are zeroed, the rest scaled by 1/(1-p);
    the keep mask is a function of (seed, stream id, sample, channel) and is regenerated in backward."""

    @staticmethod
    def forward(ctx, x, p, stream_id):
        _check_dev(x)
        B, C, H, W = x.shape
        y = nhwc_empty(B, C, H, W, x.dtype, x.device)
        seed = dropout_seed(x.device)
        _lib.call("cvh_dropout2d", _dt(x), _p(x), _p(y), B, H * W, C, float(p), _p(seed), stream_id, _stream())
        ctx.seed = seed
        ctx.cfg = (float(p), stream_id)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, stream_id = ctx.cfg
        dy = as_nhwc(dy)
        B, C, H, W = dy.shape
        dx = nhwc_empty(B, C, H, W, dy.dtype, dy.device)
        _lib.call("cvh_dropout2d", _dt(dy), _p(dy), _p(dx), B, H * W, C, p, _p(ctx.seed), stream_id, _stream())
        return dx, None, None


def dropout2d(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    return Dropout2dFn.apply(to_nhwc(x), float(p), next_stream_id())


class CatChannelsFn(torch.autograd.Function):
    """torch.cat(tensors, dim=1) of NHWC maps with channel counts that are multiples of 8 (ASPP branches, cvnets/modules/aspp_block.py:118-121);
    backward splits the gradient back (the same kernel, inverse direction)."""

    @staticmethod
    def forward(ctx, *xs):
        for x in xs:
            _check_dev(x)
        B, _, H, W = xs[0].shape
        cs = [int(x.shape[1]) for x in xs]
        y = nhwc_empty(B, sum(cs), H, W, xs[0].dtype, xs[0].device)
        ptrs = (ctypes.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        chans = (ctypes.c_int * len(xs))(*cs)
        _lib.call("cvh_cat_channels", _dt(y), ptrs, chans, len(xs), _p(y), B * H * W, 0, _stream())
        ctx.meta = (B, H, W, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, cs = ctx.meta
        dy = as_nhwc(dy)
        outs = [nhwc_empty(B, c, H, W, dy.dtype, dy.device) for c in cs]
        ptrs = (ctypes.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        chans = (ctypes.c_int * len(outs))(*cs)
        _lib.call("cvh_cat_channels", _dt(dy), ptrs, chans, len(outs), _p(dy), B * H * W, 1, _stream())
        return tuple(outs)


def cat_channels(xs):
    xs = [to_nhwc(x) for x in xs]
    if len(xs) > 8 or any(x.shape[1] % 8 for x in xs):
        raise NotImplementedError("channel concat of more than 8 tensors / channel counts that are not multiples of 8")
    return CatChannelsFn.apply(*xs)


# ------------------------------------------------------------------------------------------------
# StochasticDepth ("row" mode) fused with the residual add
# ------------------------------------------------------------------------------------------------
class DropPathFn(torch.autograd.Function):
    """y = res + x * keep(sample) / (1 - p)  (cvnets/layers/stochastic_depth.py:10-18 = torchvision StochasticDepth(mode="row") followed by
    the residual add of cvnets/modules/transformer.py:140-155); the sample of a token row follows `seqmap` as in AttentionFn."""

    @staticmethod
    def forward(ctx, x, res, p, stream_id, seqmap):
        _check_dev(x)
        nseq, S, ph, pw, n_w, H, W = seqmap
        rows, C = x.shape
        y = torch.empty_like(x)
        seed = dropout_seed(x.device)
        _lib.call("cvh_drop_path", _dt(x), _p(x), _p(res), _p(y), rows, C, ph, pw, H, W, float(p), _p(seed), stream_id, _stream())
        ctx.cfg = (float(p), stream_id, (ph, pw, H, W), res is not None)
        ctx.seed = seed
        return y

    @staticmethod
    def backward(ctx, dy):
        p, stream_id, (ph, pw, H, W), has_res = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.call("cvh_drop_path", _dt(dy), _p(dy), None, _p(dx), dy.shape[0], dy.shape[1], ph, pw, H, W, p, _p(ctx.seed), stream_id, _stream())
        return dx, (dy if has_res else None), None, None, None


def drop_path(x2d, res, p: float, training: bool, seqmap):
    """x2d / res: [rows, C] token matrices; returns res + StochasticDepth(x2d) (res may be None)"""
    if not training or p <= 0.0:
        return x2d if res is None else add(x2d, res)
    return DropPathFn.apply(x2d, res, float(p), next_stream_id(), tuple(int(v) for v in seqmap))


# ------------------------------------------------------------------------------------------------
# activation checkpointing that keeps the counter-based dropout streams aligned
# ------------------------------------------------------------------------------------------------
def checkpoint(fn, *inputs):
    """torch.utils.checkpoint (base_image_encoder.py:196-204, vit.py:533 `gradient_checkpoint_fn`) for HIP layers: dropout / drop-path
    masks are functions of (per-forward seed snapshot, per-op stream id); the recomputation in backward must draw the SAME ids and read
    the SAME snapshot as the original forward, so both are pinned around `fn`."""
    import torch.utils.checkpoint as _cp

    dev = next((t.device for t in inputs if isinstance(t, torch.Tensor)), None)
    key = (dev.index or 0) if dev is not None and dev.type == "cuda" else None
    start = _stream_ids.value                      # the ids fn's ops will take (nothing is consumed here: a checkpointed and a plain
    snap = _seed_snap.get(key) if key is not None else None  # forward of the same model draw identical masks)

    def run(*args):
        saved_id, saved_snap = _stream_ids.value, (_seed_snap.get(key) if key is not None else None)
        _stream_ids.value = start
        if snap is not None:
            _seed_snap[key] = snap
        try:
            out = fn(*args)
        finally:
            if snap is not None and saved_snap is not None:
                _seed_snap[key] = saved_snap
            # original forward: saved_id == start, continue after the ids consumed here; recomputation in backward: restore the caller's
            _stream_ids.value = max(_stream_ids.value, saved_id)
        return out

    return _cp.checkpoint(run, *inputs, use_reentrant=False, preserve_rng_state=False)


# ------------------------------------------------------------------------------------------------
# device-side input stage: mixup / cutmix fused with the layout + dtype conversion
# ------------------------------------------------------------------------------------------------
def mix_batch(x: torch.Tensor, lam: float, box=None, *, to_nhwc_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """x: NCHW float32 batch on the GPU.  box = None: mixup, out[b] = lam*x[b] + (1-lam)*x[b-1];  box = (x1, y1, x2, y2): cutmix, the box is
    pasted from x[b-1] (lam is ignored inside and 1 outside).  `to_nhwc_dtype` set: returns the NHWC compute-dtype tensor the models
    consume (channels padded to 8) — mixing, layout change and cast in ONE pass; otherwise NCHW float32 like the reference."""
    _check_dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()  # plumbing
    B, C, H, W = x.shape
    x1, y1, x2, y2 = (0, 0, 0, 0) if box is None else (int(v) for v in box)
    lam_out = float(lam) if box is None else 1.0
    if to_nhwc_dtype is not None:
        Cp = pad8(C)
        out = nhwc_empty(B, Cp, H, W, to_nhwc_dtype, x.device)
        _lib.call("cvh_mix_batch", _dt(out), _p(x), _p(out), B, C, H, W, Cp, lam_out, x1, y1, x2, y2, _stream())
        return out
    out = torch.empty_like(x)
    _lib.call("cvh_mix_batch", 0, _p(x), _p(out), B, C, H, W, 0, lam_out, x1, y1, x2, y2, _stream())
    return out
