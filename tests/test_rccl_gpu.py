"""The GPU side of cvnets_amd/ddp.py and cvnets_amd/comm.py executed on ONE GPU (SURVEY.md §8 rows a15 / e): a single-rank world whose
collectives are issued anyway (`force_collectives`) — through the package's OWN RCCL communicator (cvh_comm_* in the C ABI: unique id,
ncclCommInitRank, ncclAllReduce(ncclAvg) / ncclBroadcast / ncclAllGather / ncclReduceScatter on HIP streams), with torch.distributed
("nccl" group bound to the device) left as the control plane.  An all-reduce over one rank is an identity — which is exactly
what makes it checkable: every result must equal the no-DDP result bit for bit — but it runs the communicator, `ReduceOp.AVG`, the side
HIP stream with its event ordering, the autograd hooks, `finish`, hipGraph capture of RCCL kernels next to the RCCL watchdog thread,
and the all-gather / reduce-scatter pair of the contrastive loss.  Multi-GPU runs are the driver's (replaces main_train.py:90-96,
utils/ddp_utils.py:47-89, utils/tensor_utils.py:121-122 of the reference)."""
import argparse
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_world1():
    from cvnets_amd import launch
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    opts = argparse.Namespace()
    for k, v in {"ddp.dist_url": None, "ddp.dist_port": port, "ddp.rank": 0, "ddp.world_size": 1, "ddp.backend": "nccl",
                 "dev.device": torch.device("cuda:0")}.items():
        setattr(opts, k, v)
    torch.cuda.set_device(0)
    assert not dist.is_initialized()
    from cvnets_amd import comm
    rank = launch.distributed_init(opts)          # the launcher's replacement of utils/ddp_utils.py:47-89
    assert rank == 0 and dist.get_backend() == "nccl" and getattr(opts, "ddp.dist_url") == f"tcp://127.0.0.1:{port}"
    assert comm.default() is not None and comm.default().world == 1   # the hot path's own communicator came up (and passed its self-test)
    yield opts
    comm.destroy_default()
    dist.destroy_process_group()


def _model(seed=0):
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from oracle.weights import seeded_state_dict
    opts = default_opts(**{"model.classification.mit.mode": "xx_small", "model.classification.mit.dropout": 0.0,
                           "model.classification.classifier_dropout": 0.0})
    m = cvnets_amd.MobileViT(opts)
    m.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed))
    return m.cuda().train()


def _batch(B=4, res=64, seed=5):
    from oracle.weights import seeded_input, seeded_labels
    return seeded_input((B, 3, res, res), seed=seed).cuda(), seeded_labels(B, 1000, seed=seed).cuda()


class _Count:
    """counts the bucket all-reduces: `n` = launches through the package's communicator (tallied inside the library, cvh_comm_counters),
    `torch_n` = calls that reached torch.distributed instead (must stay 0 on the GPU path)"""

    def __init__(self, monkeypatch):
        from cvnets_amd import comm
        self._comm = comm
        self.torch_n = 0
        self._base = comm.counters()[0]
        orig = dist.all_reduce

        def wrapped(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
            self.torch_n += 1
            return orig(t, op=op, group=group, async_op=async_op)
        monkeypatch.setattr(dist, "all_reduce", wrapped)

    @property
    def n(self):
        return self._comm.counters()[0] - self._base

    @n.setter
    def n(self, v):
        assert v == 0
        self._base = self._comm.counters()[0]


def test_own_communicator_collectives(rccl_world1):
    """cvnets_amd.comm.Communicator in a world of one: every collective runs RCCL's kernel on the given stream and is an identity (all-gather
    / reduce-scatter: a copy), float32 and bfloat16; broadcast also moves opaque bytes (int64 BatchNorm counters)."""
    from cvnets_amd import comm
    c = comm.default()
    assert comm.available() and c.rank == 0
    before = comm.counters()
    side = torch.cuda.Stream()
    for dtype in (torch.float32, torch.bfloat16):
        t = torch.randn(3001, device="cuda").to(dtype)
        want = t.clone()
        c.all_reduce(t)
        c.all_reduce(t, average=True)
        side.wait_stream(torch.cuda.current_stream())
        c.all_reduce(t, average=True, stream=side)      # the side-stream form ddp uses
        torch.cuda.current_stream().wait_stream(side)
        c.broadcast(t, 0)
        out = torch.empty_like(t)
        c.all_gather(out, t)
        rs = torch.empty_like(t)
        c.reduce_scatter(rs, out)
        torch.cuda.synchronize()
        assert torch.equal(t, want) and torch.equal(out, want) and torch.equal(rs, want)
    n = torch.arange(17, device="cuda", dtype=torch.int64)
    c.broadcast(n, 0)
    torch.cuda.synchronize()
    assert torch.equal(n, torch.arange(17, device="cuda"))
    after = comm.counters()
    assert tuple(a - b for a, b in zip(after, before)) == (6, 3, 2, 2)
    with pytest.raises(RuntimeError):
        c.all_reduce(n)                                   # integer reductions are not on the path
    with pytest.raises(RuntimeError):
        c.all_reduce(torch.zeros(4, 4, device="cuda").t())  # one message = one contiguous buffer
    c.self_test()


def test_eager_hooks_side_stream_allreduce_avg(rccl_world1, monkeypatch):
    """eager backward: post-accumulate hooks -> bucket all_reduce(AVG) on the side stream behind an event -> finish() at the end of backward"""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.ddp import DistributedDataParallel
    x, y = _batch()
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        ref = _model()
        ops.cross_entropy(ref(x), y, 0.1).backward()
        want = {k: p.grad.clone() for k, p in ref.named_parameters()}
        m = _model()
        ddp = DistributedDataParallel(m, bucket_cap_mb=1.0, force_collectives=True)   # 1 MB cap: several buckets, launched as they fill
        assert ddp.active and len(ddp.buckets) >= 3 and ddp.comm is not None
        cnt = _Count(monkeypatch)
        for _ in range(2):  # twice: bucket bookkeeping resets in finish()
            ddp.zero_grad()
            ops.cross_entropy(ddp(x), y, 0.1).backward()
            torch.cuda.synchronize()
            lo_hi = [(b.flat.data_ptr(), b.flat.data_ptr() + 4 * b.numel) for b in ddp.buckets]
            for k, p in m.named_parameters():
                assert any(lo <= p.grad.data_ptr() < hi for lo, hi in lo_hi), k   # gradients are views of the flat buckets
                # AVG over one rank is the identity: equal to the plain backward up to the run-to-run order of the fp32 atomics
                assert torch.allclose(p.grad, want[k], rtol=1e-4, atol=1e-6 + 1e-5 * float(want[k].abs().max())), k
        assert cnt.n == 2 * len(ddp.buckets) and cnt.torch_n == 0   # every bucket through cvh_comm_allreduce (ncclAvg), none through torch
        # gradient accumulation: nothing is reduced inside no_sync()
        cnt.n = 0
        ddp.zero_grad()
        with ddp.no_sync():
            ops.cross_entropy(ddp(x), y, 0.1).backward()
        assert cnt.n == 0
        ops.cross_entropy(ddp(x), y, 0.1).backward()
        torch.cuda.synchronize()
        assert cnt.n == len(ddp.buckets)
        for k, p in m.named_parameters():
            assert torch.allclose(p.grad, 2 * want[k], rtol=1e-3, atol=1e-6 + 1e-4 * float(want[k].abs().max())), k
    finally:
        cvnets_amd.set_compute_dtype(None)


@pytest.mark.parametrize("in_graph", [False, True, "overlap"])
def test_captured_step_with_rccl_allreduce(rccl_world1, in_graph):
    """bench.py's N > 1 path: hipGraph capture (capture_error_mode="thread_local", the RCCL watchdog thread is alive) of zero-grad + forward
    + loss + backward with in-place bucket gradients [+ the bucket all-reduces and the fused AdamW as graph nodes], replayed; against
    the same steps run eagerly without any process group involvement."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.ddp import DistributedDataParallel
    x, y = _batch()
    cvnets_amd.set_compute_dtype(torch.float32)
    ops.set_inplace_param_grads(True)
    try:
        overlap = in_graph == "overlap"

        def make():
            m = _model()
            # "overlap": bench.py's N > 1 configuration — the buckets fork onto the side stream INSIDE the captured backward, driven by the
            # child-input boundaries (in-place gradients never reach autograd hooks); xx_small = 5.1 MB -> 4 buckets of >= 0.25 / 1 MB
            ddp = DistributedDataParallel(m, bucket_cap_mb=1.0, first_bucket_mb=0.25, broadcast_buffers=False, force_collectives=True,
                                          boundary_overlap=overlap)
            ddp.hooks_enabled = False
            ddp.boundary_enabled = False
            opt = cvnets_amd.optim.AdamW(list(m.parameters()), lr=1e-3, weight_decay=0.01)
            return m, ddp, opt

        def eager_step(m, ddp, opt):
            ddp.zero_grad()
            loss = ops.cross_entropy(m(x), y, 0.1)
            loss.backward()
            opt.step()
            return loss

        # reference trajectory: 1 warm-up + 3 steps, eager, collectives off
        m0, d0, o0 = make()
        d0.active = False
        ref_losses = [float(eager_step(m0, d0, o0)) for _ in range(4)]

        m, ddp, opt = make()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ddp.zero_grad()
            l0 = ops.cross_entropy(m(x), y, 0.1)
            l0.backward()
            ddp.allreduce_flat()
            opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert abs(float(l0) - ref_losses[0]) < 1e-5
        g = torch.cuda.CUDAGraph()
        early0, fin0 = ddp.early_launches, ddp.finish_count
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            ddp.zero_grad()
            ddp.boundary_enabled = overlap
            static_loss = ops.cross_entropy(m(x), y, 0.1)
            static_loss.backward()
            ddp.boundary_enabled = False
            if overlap:
                # at least two all-reduce nodes were forked before the last backward kernel was captured; `finish` joined them and
                # launched the first child's bucket at the end of backward
                assert ddp.boundary_overlap and ddp.finish_count == fin0 + 1, (ddp.boundary_overlap, ddp.finish_count)
                assert ddp.early_launches - early0 >= 2, ddp.overlap_report()
                opt.step(sync_hyperparameters=False)
            elif in_graph:
                ddp.allreduce_flat()
                opt.step(sync_hyperparameters=False)
        losses = []
        for _ in range(3):
            g.replay()
            if not in_graph:
                ddp.allreduce_flat()
                opt.step(sync_hyperparameters=False)
            losses.append(float(static_loss))
        torch.cuda.synchronize()
        for a, b in zip(losses, ref_losses[1:]):
            assert abs(a - b) < 2e-4, (losses, ref_losses)   # fp32, different summation orders of the statistics (LDS atomics)
        # AdamW normalises every gradient: compare the accumulated UPDATE of all parameters, not element by element
        from oracle.weights import seeded_state_dict
        sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
        num = den = 0.0
        for (k, p), q in zip(m.named_parameters(), m0.parameters()):
            d1, d0 = (p.detach().cpu() - sd[k]).double(), (q.detach().cpu() - sd[k]).double()
            num += float((d1 - d0).pow(2).sum())
            den += float(d0.pow(2).sum())
        assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5
    finally:
        ops.set_inplace_param_grads(False)
        cvnets_amd.set_compute_dtype(None)


def test_gather_all_features_allgather_reducescatter(rccl_world1):
    """contrastive_loss_clip.py:144-172 / utils/tensor_utils.py:121-122: all_gather_into_tensor forward, reduce_scatter_tensor backward"""
    from cvnets_amd.ddp import gather_all_features
    f = torch.randn(16, 512, device="cuda", requires_grad=True)
    out = gather_all_features(f, force=True)
    assert out.shape == f.shape and out.data_ptr() != f.data_ptr() and torch.equal(out, f)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    assert torch.equal(f.grad, w)
    # the two-tower contrastive loss through the gathered features == the single-process loss
    from cvnets_amd.losses import ContrastiveLossClip
    img = torch.nn.functional.normalize(torch.randn(8, 64, device="cuda"), dim=-1)
    txt = torch.nn.functional.normalize(torch.randn(8, 64, device="cuda"), dim=-1)
    scale = torch.tensor(10.0, device="cuda")
    res = []
    for distributed in (True, False):
        o = argparse.Namespace()
        setattr(o, "ddp.rank", 0)
        setattr(o, "ddp.use_distributed", distributed)
        crit = ContrastiveLossClip(o).train()
        a, b = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
        if distributed:
            os.environ["CVH_DDP_FORCE_COLLECTIVES"] = "1"
        try:
            loss = crit(None, {"image": a, "text": b, "logit_scale": scale})["total_loss"]
            loss.backward()
        finally:
            os.environ.pop("CVH_DDP_FORCE_COLLECTIVES", None)
        res.append((loss.detach(), a.grad.clone(), b.grad.clone()))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-6)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-7) and torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-7)


def test_wrapper_deepcopy_and_ema_update_on_gpu(rccl_world1):
    """the reference's EMA deep-copies the DDP-wrapped model (cvnets/misc/averaging_utils.py:33); the GPU wrapper holds
    a HIP stream, which cannot be copied — the copy must be a passive holder of a copy of the module, and the EMA update must run."""
    import copy
    from cvnets_amd.ddp import DistributedDataParallel
    m = _model()
    ddp = DistributedDataParallel(m, broadcast_buffers=False, force_collectives=True)
    assert ddp.side_stream is not None
    ema = copy.deepcopy(ddp)
    ema.eval()
    assert not ema.active and ema.side_stream is None and hasattr(ema, "module")
    msd = ddp.state_dict()
    with torch.no_grad():
        for k, v in ema.state_dict().items():  # averaging_utils.py:43-55
            if v.is_floating_point():
                v.copy_(v * (1.0 - 0.0005) + 0.0005 * msd[k].detach())
    x, _ = _batch()
    import cvnets_amd
    cvnets_amd.set_compute_dtype(torch.float32)
    try:
        with torch.no_grad():
            out = ema(x)
    finally:
        cvnets_amd.set_compute_dtype(None)
    assert out.shape == (4, 1000) and torch.isfinite(out).all()
