"""CPU / gloo / world_size 2 test of the data-parallel gradient exchange (cvnets_amd/ddp.py): flat buckets, post-accumulate
hooks, end-of-backward join and the explicit post-graph-replay path must all produce the mean gradient on every rank, and
parameters/buffers must start identical (rank-0 broadcast)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, explicit, q):
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cvnets_amd.ddp import DistributedDataParallel, distributed_init

    assert distributed_init("gloo", torch.device("cpu")) == rank
    torch.manual_seed(100 + rank)  # different init per rank: the ctor broadcast must fix that
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    boundary = explicit == "boundary"
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001, first_bucket_mb=0.0001, overlap=True, boundary_overlap=boundary)  # tiny cap -> several buckets
    assert len(ddp.buckets) >= 3
    if boundary:  # no per-parameter hooks (the in-place-gradient regime of bench.py): the exchange is driven by the child-input boundaries
        assert ddp.boundary_overlap
        ddp.hooks_enabled = False
        explicit = False
    w0 = [p.detach().clone() for p in net.parameters()]
    gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
    dist.all_gather(gathered, w0[0])
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    torch.manual_seed(7 + rank)
    x = torch.randn(12, 16)
    ddp.zero_grad()
    if explicit:
        ddp.hooks_enabled = False
        ddp(x).square().mean().backward()
        ddp.allreduce_flat()
    else:
        ddp(x).square().mean().backward()
    got = [p.grad.clone() for p in net.parameters()]
    # reference: gradients of the same replica on every rank's batch, averaged
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        torch.manual_seed(7 + r)
        xr = torch.randn(12, 16)
        net2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
        net2.load_state_dict({k: v for k, v in zip(net.state_dict().keys(), [t.clone() for t in net.state_dict().values()])})
        for p2, w in zip(net2.parameters(), w0):
            p2.data.copy_(w)
        net2[1].reset_running_stats()
        net2(xr).square().mean().backward()
        for a, p2 in zip(ref, net2.parameters()):
            a += p2.grad / world
    err = max(float((a - b).abs().max()) for a, b in zip(got, ref))
    if boundary:  # buckets of the later children started inside backward; the first child's (its input needs no gradient) at the end
        rep = ddp.overlap_report()
        assert rep["launched_during_backward"] >= 2 and rep["launched_at_end_of_backward"] >= 1, rep
    # a second step must work too (bucket counters reset)
    ddp.zero_grad()
    ddp.hooks_enabled = not explicit and not boundary
    ddp(x).square().mean().backward()
    if explicit:
        ddp.allreduce_flat()
    q.put((rank, err, all(p.grad.data_ptr() != 0 for p in net.parameters())))
    dist.destroy_process_group()


@pytest.mark.parametrize("explicit", [False, True, "boundary"])
def test_ddp_gloo_world2(explicit):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, explicit, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    for rank, err, ok in res:
        assert err < 1e-6 and ok, (rank, err)


def _gather_worker(rank, world, port, q):
    """contrastive loss over 2 ranks (gloo): the differentiable all-gather must reproduce the single-process loss / gradients of
    the oracle's ContrastiveLossClip restatement evaluated on the concatenated batch (contrastive_loss_clip.py:35-103, 144-172)."""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.nn.functional as F
    from cvnets_amd.ddp import distributed_init, gather_all_features
    from oracle import mobilevit_oracle as orc

    distributed_init("gloo", torch.device("cpu"))
    n, d = 4, 16
    g = torch.Generator().manual_seed(3)
    img_all = F.normalize(torch.randn(world * n, d, generator=g), dim=-1)
    txt_all = F.normalize(torch.randn(world * n, d, generator=g), dim=-1)
    scale = torch.tensor(10.0)
    img = img_all[rank * n:(rank + 1) * n].clone().requires_grad_()
    txt = txt_all[rank * n:(rank + 1) * n].clone().requires_grad_()
    gi, gt = gather_all_features(img), gather_all_features(txt)
    assert torch.equal(gi.detach(), img_all) and torch.equal(gt.detach(), txt_all)
    loss, _, _ = orc.contrastive_loss_clip(img, txt, scale, gi, gt, rank=rank)
    loss.backward()
    # single-process reference: the mean over ranks of the per-rank losses == loss of the full batch; d(sum of rank losses)/d(local)
    ia, ta = img_all.clone().requires_grad_(), txt_all.clone().requires_grad_()
    total = sum(orc.contrastive_loss_clip(ia[r * n:(r + 1) * n], ta[r * n:(r + 1) * n], scale, ia, ta, rank=r)[0] for r in range(world))
    total.backward()
    full, _, _ = orc.contrastive_loss_clip(img_all, txt_all, scale)
    err = max(float((img.grad - ia.grad[rank * n:(rank + 1) * n]).abs().max()), float((txt.grad - ta.grad[rank * n:(rank + 1) * n]).abs().max()))
    q.put((rank, err, abs(float(total) / world - float(full))))
    dist.destroy_process_group()


def test_gather_all_features_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, err, dl in sorted(q.get(timeout=5) for _ in range(2)):
        assert err < 1e-6 and dl < 1e-6, (rank, err, dl)


def _stray_worker(rank, world, port, q):
    """a training loop that drops gradients (optimizer.zero_grad(set_to_none=True), the default of torch >= 2 and of the reference's
    engine) must still get the cross-rank mean: the wrapper re-adopts freshly allocated .grad tensors into its flat buckets"""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cvnets_amd.ddp import DistributedDataParallel, distributed_init

    distributed_init("gloo", torch.device("cpu"))
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    ok = True
    for it in range(3):
        opt.zero_grad()  # set_to_none=True: p.grad = None, autograd allocates new tensors outside the buckets
        torch.manual_seed(10 * it + rank)
        x = torch.randn(4, 6)
        ddp(x).square().mean().backward()
        g_local_mean = [p.grad.clone() for p in net.parameters()]
        gathered = [[torch.zeros_like(g) for _ in range(world)] for g in g_local_mean]
        for g, gl in zip(g_local_mean, gathered):
            dist.all_gather(gl, g)
        ok = ok and all(all(torch.allclose(a, gl[0], atol=1e-7) for a in gl) for gl in gathered)  # identical on every rank = reduced
        ok = ok and all(p.grad.data_ptr() == v.data_ptr() for b in ddp.buckets for p, v in zip(b.params, b.views))
        opt.step()
    w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    q.put((rank, bool(ok and torch.allclose(ws[0], ws[1], atol=1e-7))))
    dist.destroy_process_group()


def test_ddp_set_to_none_gradients_are_readopted():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stray_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok in sorted(q.get(timeout=5) for _ in range(2)):
        assert ok, rank


def test_ddp_wrapper_can_be_copied_and_pickled():
    """EMA deep-copies the WRAPPED model (cvnets/misc/averaging_utils.py:33) and torch.save pickles it: the copy is a passive holder of a copy
    of `.module` (no stream, no process group, no buckets, no hooks) with the same state_dict keys."""
    import copy
    import pickle
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    from cvnets_amd.ddp import DistributedDataParallel
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001)
    import threading
    ddp.side_stream = threading.Lock()  # stands in for the HIP stream the GPU wrapper holds: neither copyable nor picklable
    for c in (copy.deepcopy(ddp), pickle.loads(pickle.dumps(ddp))):
        assert isinstance(c, DistributedDataParallel) and not c.active and c.side_stream is None and not c.buckets
        assert list(c.state_dict().keys()) == list(ddp.state_dict().keys()) and c.module is not ddp.module
        c.eval()
        assert c(torch.randn(4, 16)).shape == (4, 8)
        for a, b in zip(c.parameters(), ddp.parameters()):  # the reference's EMA update (averaging_utils.py:43-55)
            a.detach().mul_(0.5).add_(b.detach(), alpha=0.5)


class _RootParamNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 8)
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 8))  # root-level parameter

    def forward(self, x):
        return self.b(torch.relu(self.a(x))) + self.cls_token


def test_boundary_overlap_wrapper_copy_pickle_no_sync_and_root_parameters():
    """(a) with boundary_overlap the top-level children carry forward pre-hooks — a copy / pickle of the wrapper must carry
    INERT hooks (no closure to pickle, no reference to the live wrapper whose order bookkeeping a forward of the copy could disturb);
    (b) no_sync() silences the boundary-driven exchange as well (torch DDP's contract: nothing is communicated inside it);
    (c) a parameter registered on the ROOT module (ViT / CLIP cls_token) no longer switches the feature off: its bucket belongs to no
    boundary and is launched at the end of backward."""
    import copy
    import pickle
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    from cvnets_amd.ddp import DistributedDataParallel

    Net = _RootParamNet
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", init_method="env://")
    try:
        net = Net()
        ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001, first_bucket_mb=0.0001, boundary_overlap=True, force_collectives=True)
        assert ddp.boundary_overlap and ddp.active                                  # (c): not switched off by cls_token
        owned = [b for lst in ddp._boundary_buckets.values() for b in lst]
        root_bucket = ddp._bucket_of[net.cls_token]
        assert root_bucket not in owned and len(owned) >= 2
        ddp.hooks_enabled = False                                                   # in-place-gradient regime: boundaries only
        x = torch.randn(4, 16).requires_grad_(True)
        e0, l0 = ddp.early_launches, ddp.late_launches
        ddp.zero_grad()
        ddp(x).square().mean().backward()
        assert ddp.early_launches > e0 and ddp.late_launches > l0                   # boundary buckets early, the root-level one in finish()
        want = [p.grad.clone() for p in net.parameters()]
        # (b) no_sync: no launch of any kind, gradients accumulate locally
        e1, l1, f1 = ddp.early_launches, ddp.late_launches, ddp.finish_count
        with ddp.no_sync():
            ddp(x).square().mean().backward()
        assert (ddp.early_launches, ddp.late_launches, ddp.finish_count) == (e1, l1, f1)
        for p, w in zip(net.parameters(), want):
            assert torch.allclose(p.grad, 2 * w)
        # (a) copies
        seen = (ddp._fid, ddp._order)
        for c in (copy.deepcopy(ddp), pickle.loads(pickle.dumps(ddp))):
            assert not c.active and not c.boundary_overlap
            c(torch.randn(2, 16))                                                   # the copy's (inert) hooks run
            assert (ddp._fid, ddp._order) == seen and ddp.boundary_overlap           # ... and did not touch the live wrapper
    finally:
        dist.destroy_process_group()


class _InplaceLinearFn(torch.autograd.Function):
    """a layer in the style of the HIP ops under ops.set_inplace_param_grads(True): the weight gradient is ADDED into weight.grad by the
    backward itself and autograd is handed None — no post-accumulate hook ever fires for the parameter"""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.detach().t()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        ctx.w.grad.add_(g.t() @ x)
        return g @ ctx.w.detach(), None


class _InplaceNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w1 = torch.nn.Parameter(torch.randn(32, 16) * 0.1)
        self.w2 = torch.nn.Parameter(torch.randn(8, 32) * 0.1)

    def forward(self, x):
        return _InplaceLinearFn.apply(torch.relu(_InplaceLinearFn.apply(x, self.w1)), self.w2)


def test_inplace_gradients_still_get_exchanged_at_the_end_of_backward():
    """the engine-driven path of cvnets_amd/launch.py with the fused optimizer: gradients are added in place, no per-parameter hook fires —
    the wrapper's forward puts a hook on the OUTPUT that queues `finish`, so every bucket is exchanged at the end of backward"""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    from cvnets_amd import ops
    from cvnets_amd.ddp import DistributedDataParallel
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", init_method="env://")
    ops.set_inplace_param_grads(True)
    try:
        net = _InplaceNet()
        ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001, first_bucket_mb=0.0001, force_collectives=True)
        x = torch.randn(4, 16).requires_grad_(True)
        ddp.zero_grad()
        f0, n0 = ddp.finish_count, ddp.late_launches + ddp.early_launches
        ddp(x).square().mean().backward()
        # every bucket exactly once — from `finish`, or earlier where this PyTorch fires the accumulate hook for an in-place gradient
        assert ddp.finish_count == f0 + 1 and ddp.late_launches + ddp.early_launches - n0 == len(ddp.buckets)
        assert float(net.w1.grad.abs().sum()) > 0 and float(net.w2.grad.abs().sum()) > 0
        with ddp.no_sync():                                      # accumulation micro-step: nothing queued
            ddp(x).square().mean().backward()
        assert ddp.finish_count == f0 + 1
    finally:
        ops.set_inplace_param_grads(False)
        dist.destroy_process_group()


def _uid_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cvnets_amd import comm
    from cvnets_amd.ddp import distributed_init

    assert distributed_init("gloo", torch.device("cpu")) == rank
    assert comm.default() is None and comm.init_default(torch.device("cpu")) is None   # CPU ranks stay on torch.distributed (gloo)
    store = dist.distributed_c10d._get_default_store()
    uid = comm.Communicator.exchange_unique_id(store, rank, key="cvnets_amd/comm/test")
    q.put((rank, uid))
    dist.barrier()
    dist.destroy_process_group()


def test_unique_id_rendezvous_through_the_tcp_store_world2():
    """the host half of the own-communicator bring-up (utils/ddp_utils.py:63-89 rewired): rank 0 draws the RCCL unique id (ncclGetUniqueId
    needs no GPU), the TCP store of the env:// rendezvous carries its 128 bytes, every rank ends up with the same id.  ncclCommInitRank and
    the collectives themselves need GPUs: tests/test_rccl_gpu.py (one GPU, world of one); N > 1 runs are the driver's."""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    from cvnets_amd import comm
    if not comm.available():
        pytest.skip("librccl not installed on this box")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_uid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert len(got[0]) == 128 and got[0] == got[1] and any(got[0])


class _OutOfOrderNet(torch.nn.Module):
    """children registered a, norm, blocks, head, pos — executed a, pos, blocks (a container of three blocks), norm, head: the shape of the
    reference's VisionTransformer (patch_emb, post_transformer_norm, transformer, classifier, pos_embed + a root-level cls_token)"""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.norm = torch.nn.LayerNorm(32)
        self.blocks = torch.nn.Sequential(torch.nn.Linear(32, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 32))
        self.head = torch.nn.Linear(32, 8)
        self.pos = torch.nn.Linear(32, 32)
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 32))

    def forward(self, x):
        x = self.pos(self.a(x)) + self.cls_token
        return self.head(self.norm(self.blocks(x)))


def test_boundary_overlap_learns_the_execution_order():
    """ddp.py round 5: the boundary candidates' EXECUTION order is learnt (first guess: registration order).  A model whose children run in
    another order (ViT: `pos_embed` is registered after the blocks it precedes) exchanges at the end of backward on its first step (the end-of-backward callback is queued by the first boundary hook that
    fires, whether or not its order is trusted yet) and overlaps from the second one on, with the blocks of a Sequential container as boundaries of their own; gradients equal the plain
    backward's in both steps; a checkpoint-style forward inside backward does not disturb the learnt order."""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    from cvnets_amd.ddp import DistributedDataParallel
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", init_method="env://")
    try:
        torch.manual_seed(0)
        net = _OutOfOrderNet()
        x = torch.randn(4, 16).requires_grad_(True)
        net(x).square().mean().backward()
        want = [p.grad.clone() for p in net.parameters()]
        net.zero_grad()
        ddp = DistributedDataParallel(net, bucket_cap_mb=0.0001, first_bucket_mb=0.0001, boundary_overlap=True, force_collectives=True)
        ddp.hooks_enabled = False   # the in-place-gradient regime: boundaries are the only driver besides `finish`
        assert ddp.boundary_overlap and len(ddp._cands) == 7   # a, norm, blocks.0-2, head, pos
        reg_order = ddp._order
        steps = []
        for _ in range(3):
            ddp.zero_grad()
            e0, l0, f0 = ddp.early_launches, ddp.late_launches, ddp.finish_count
            ddp(x).square().mean().backward()
            assert ddp.finish_count == f0 + 1  # also on the first step (wrong guess, no boundary launches anything): the first boundary hook of a backward queues the end-of-backward exchange
            steps.append((ddp.early_launches - e0, ddp.late_launches - l0))
            for p, w in zip(net.parameters(), want):
                assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7)
        assert ddp._order == (0, 6, 2, 3, 4, 1, 5) and ddp._order != reg_order          # a, pos, blocks.0-2, norm, head
        assert steps[0] == (0, len(ddp.buckets)) and steps[1][0] >= 3 and steps[2] == steps[1], steps   # first step (wrong guess): everything at the end of backward; then overlapped
        assert all(e + l == len(ddp.buckets) for e, l in steps[1:]), (steps, len(ddp.buckets))
        # the bucket of the root-level parameter never belongs to a boundary
        assert all(ddp._bucket_of[net.cls_token] not in lst for lst in ddp._boundary_buckets.values())
    finally:
        dist.destroy_process_group()


def _fallback_worker(rank, world, port, fail_rank, q):
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cvnets_amd import comm
    dist.init_process_group("gloo", init_method="env://")

    class _Fake:  # stands in for a communicator that came up (no GPU here): what matters is the agreement protocol around it
        destroyed = False

        def self_test(self):
            pass

        def destroy(self):
            _Fake.destroyed = True

    def fake_from_store(store, world_, rank_, dev, key):
        uid = comm.Communicator.exchange_unique_id(store, rank_, key)   # the real rendezvous
        assert len(uid) == 128
        if rank_ == fail_rank:
            raise RuntimeError("simulated: ncclCommInitRank failed on this rank")
        return _Fake()

    comm.Communicator.from_store = staticmethod(fake_from_store)
    comm.available = lambda: True
    torch.cuda.is_available = lambda: True
    got = comm.init_default(torch.device("cuda", 0))
    q.put((rank, got is None, _Fake.destroyed, comm.default() is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [1, -1])
def test_own_communicator_fallback_is_a_collective_decision(fail_rank):
    """comm.init_default on two ranks: if ONE rank cannot bring its communicator up, every rank must fall back to torch.distributed (a
    communicator that exists on some ranks only would leave the ranks on different data planes) and the rank that did come up destroys its
    communicator; when all succeed, all keep it.  The RCCL calls are stubbed (no GPU here); the rendezvous and the MIN all-reduce of the
    success flag on the control plane are the real ones."""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    from cvnets_amd import comm
    if not comm.available():
        pytest.skip("librccl not installed on this box")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fallback_worker, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=180) for _ in range(2))}
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    if fail_rank >= 0:
        assert got[0] == [True, True, True] and got[1][0] and got[1][2], got    # both None; rank 0's communicator was destroyed
    else:
        assert got[0] == [False, False, False] and got[1] == [False, False, False], got


def test_inplace_gradient_mode_hooks_every_output_and_the_optimizer_safety_net(monkeypatch):
    """with in-place parameter gradients no post-accumulate hook exists; the exchange is queued by a hook on the model
    OUTPUT.  A dict output whose first tensor is off the loss path (an augmented input next to the logits) must not leave the backward without
    its exchange: every output that requires grad carries the hook.  And if no hook fires at all, the optimizer step runs the exchange itself
    (cvnets_amd.ddp.exchange_pending) instead of letting the ranks diverge."""
    sys.path.insert(0, os.path.join(REPO, "ml-cvnets_amd"))
    import warnings
    from cvnets_amd import ddp as ddp_mod
    from cvnets_amd import ops
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", init_method="env://")
    try:
        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.fc = torch.nn.Linear(8, 4)

            def forward(self, x):
                return {"augmented_tensor": x, "logits": self.fc(x), "aux": [self.fc(x) * 2.0]}

        monkeypatch.setattr(ops, "_INPLACE_PARAM_GRADS", True)
        d = ddp_mod.DistributedDataParallel(Net(), force_collectives=True)
        x = torch.randn(3, 8)  # requires no gradient: the FIRST tensor of the output is off the loss path
        f0 = d.finish_count
        out = d(x)
        assert d._expect_exchange
        out["logits"].square().mean().backward()
        assert d.finish_count == f0 + 1 and not d._expect_exchange
        # both hooked outputs on the loss path: still one exchange per backward
        out = d(x)
        (out["logits"].sum() + out["aux"][0].sum()).backward()
        assert d.finish_count == f0 + 2
        # no hook fires (the loss does not depend on the wrapped model's outputs): the optimizer step exchanges
        d(x)
        assert d._expect_exchange
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ddp_mod.exchange_pending()
        assert not d._expect_exchange and any("without overlap" in str(m.message) for m in w)
    finally:
        dist.destroy_process_group()
