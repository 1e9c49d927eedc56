"""The BENCHMARKED configuration has dropout on (mit.dropout = 0.1, classifier_dropout = 0.1: config/classification/imagenet/mobilevit.yaml,
bench.py): whole-step parity with the oracle FED THE KEEP MASKS THE KERNELS DREW.

  * cvnets_amd.ops.trace_dropout_sites records every dropout draw of a forward (kind, stream id, p, shape) — the out-projection / second FFN
    linear epilogues of every TransformerEncoder and the classifier Dropout; ops.dropout_keep_scale regenerates the factor tensor (0 or
    1 / (1 - p)) of a draw from (seed snapshot, stream id) with the standalone kernel (the GEMM epilogues index the same counter);
  * the token matrices of the HIP path are NHWC maps, the oracle's are [B * patch_area, patches, C] (mobilevit_block.py:186-231): the masks
    go through the oracle's own `unfolding`;
  * oracle.mobilevit_oracle.train_step(drop=...) applies them where the reference applies its Dropout layers — pinned against the reference
    itself by tests/test_oracle_golden.py::test_oracle_with_dropout_factors_matches_reference_fixture.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(mode, p):
    import cvnets_amd
    from cvnets_amd.layers import default_opts
    from oracle.weights import seeded_state_dict

    opts = default_opts(**{"model.classification.mit.mode": mode, "model.classification.mit.dropout": p,
                           "model.classification.classifier_dropout": p})
    model = cvnets_amd.MobileViT(opts)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd)
    return model.cuda().train(), sd


def _oracle_masks(sites, seed, B, mode):
    """HIP draws (call order) -> the oracle's `drop` dict"""
    from cvnets_amd import ops
    from oracle import mobilevit_oracle as orc

    cfg = orc.mobilevit_config(mode)
    keys = []
    for li, name in ((3, "layer3"), (4, "layer4"), (5, "layer5")):
        for i in range(cfg[name]["nblk"]):
            keys += [f"layer_{li}.1.global_rep.{i}.mha", f"layer_{li}.1.global_rep.{i}.ffn"]
    keys.append("classifier")
    assert len(sites) == len(keys), (len(sites), len(keys))
    drop = {}
    for key, (kind, sid, p, shape) in zip(keys, sites):
        f = ops.dropout_keep_scale(seed, sid, shape, p).cpu()
        if key == "classifier":
            assert kind == "dropout"
            drop[key] = f.reshape(B, -1)
        else:
            assert kind == "linear"
            rows, C = shape
            hw = int(round((rows // B) ** 0.5))
            assert B * hw * hw == rows
            fm = f.view(B, hw, hw, C).permute(0, 3, 1, 2).contiguous()   # NHWC token matrix -> the NCHW map the oracle unfolds
            drop[key] = orc.unfolding(fm, 2, 2)[0]
    return drop


def _errs(model, logits, sd, o_logits, o_grads):
    num = den = 0.0
    for k, q in model.named_parameters():
        num += float((q.grad.detach().float().cpu().double() - o_grads[k].double()).pow(2).sum())
        den += float(o_grads[k].double().pow(2).sum())
    lg = float((logits.detach().float().cpu().double() - o_logits.double()).norm() / o_logits.double().norm())
    return lg, (num / den) ** 0.5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_step_with_dropout_matches_oracle_fed_the_hip_masks(dtype):
    import cvnets_amd
    from cvnets_amd import ops
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels

    mode, B, res, p = "xx_small", 8, 64, 0.1
    model, sd = _build(mode, p)
    x, y = seeded_input((B, 3, res, res), seed=7), seeded_labels(B, 1000, seed=7)
    xg, yg = x.cuda(), y.cuda()
    cvnets_amd.set_compute_dtype(dtype)
    try:
        sites = []
        ops.trace_dropout_sites(sites)
        logits = model(xg)
        ops.trace_dropout_sites(None)
        seed = ops.dropout_seed(xg.device).clone()
        ops.cross_entropy(logits, yg, 0.1).backward()
        torch.cuda.synchronize()
        drop = _oracle_masks(sites, seed, B, mode)
        dropped = sum(float((f == 0).sum()) for f in drop.values()) / sum(f.numel() for f in drop.values())
        assert 0.08 < dropped < 0.12, dropped
        o_logits, _, o_grads, _ = orc.train_step(sd, x, y, mode=mode, drop=drop)
        lg, gg = _errs(model, logits, sd, o_logits, o_grads)
        n_logits, _, n_grads, _ = orc.train_step(sd, x, y, mode=mode)  # the same oracle WITHOUT the masks: far away
        lg0, gg0 = _errs(model, logits, sd, n_logits, n_grads)
        print(f"[dropout step {dtype}] logits {lg:.2e} grads {gg:.2e}   (oracle without the masks: {lg0:.2e} {gg0:.2e})")
        assert lg0 > 5e-2 and gg0 > 1e-1
        if dtype == torch.float32:
            assert lg < 1e-4 and gg < 1e-3, (lg, gg)
        else:
            # bf16: the storage noise of this 8-image 64x64 case with p = 0 is what the dropout run may show, not more
            for m in model.modules():
                if hasattr(m, "p") and type(m).__name__ == "Dropout":
                    m.p = 0.0
            model.zero_grad()
            l0 = model(xg)
            ops.cross_entropy(l0, yg, 0.1).backward()
            torch.cuda.synchronize()
            lgp0, ggp0 = _errs(model, l0, sd, n_logits, n_grads)
            print(f"[dropout step bf16] the same model at p = 0 against the plain oracle: logits {lgp0:.2e} grads {ggp0:.2e}")
            assert lg < 1.5 * lgp0 + 5e-3 and gg < 1.5 * ggp0 + 1e-2, (lg, gg, lgp0, ggp0)
    finally:
        ops.trace_dropout_sites(None)
        cvnets_amd.set_compute_dtype(None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_benchmarked_configuration_with_dropout_follows_the_oracle(dtype):
    """bench.py's timed path with its dropout ON: hipGraph replay of zero-grad + forward + CE + backward (gradients accumulated in place into
    flat buckets) + cvh_adamw_multi, 4 steps; every replay draws new masks (the seed advance is part of the graph) and the oracle +
    torch.optim.AdamW trajectory is fed each step's masks."""
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.ddp import DistributedDataParallel
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels

    mode, B, res, steps, p = "xx_small", 8, 64, 4, 0.1
    model, sd = _build(mode, p)
    x, y = seeded_input((B, 3, res, res), seed=3), seeded_labels(B, 1000, seed=3)
    xg, yg = x.cuda(), y.cuda()
    cvnets_amd.set_compute_dtype(dtype)
    ops.set_inplace_param_grads(True)
    try:
        ddp = DistributedDataParallel(model, bucket_cap_mb=25.0, broadcast_buffers=False)
        ddp.hooks_enabled = False
        opt = cvnets_amd.optim.AdamW([q for q in model.parameters()], lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)

        def fwd_bwd():
            loss = ops.cross_entropy(model(xg), yg, 0.1)
            loss.backward()
            return loss

        def reset(snapshot):
            with torch.no_grad():
                for k, v in model.state_dict().items():
                    v.copy_(snapshot[k])
                opt._plan["m"].zero_()
                opt._plan["v"].zero_()
                opt._plan["step"].zero_()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        snapshot = {k: v.detach().clone() for k, v in model.state_dict().items()}
        with torch.cuda.stream(side):  # one eager step creates the lazily built tensors
            ddp.zero_grad()
            fwd_bwd()
            opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        reset(snapshot)
        sites = []
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ddp.zero_grad()
            ops.trace_dropout_sites(sites)
            static_loss = fwd_bwd()
            ops.trace_dropout_sites(None)
            opt.step(sync_hyperparameters=False)
        seed_in_graph = ops.dropout_seed(xg.device)  # the snapshot tensor the captured forward wrote (graph memory: rewritten by every replay)
        reset(snapshot)
        losses, seeds = [], []
        for _ in range(steps):
            g.replay()
            torch.cuda.synchronize()
            losses.append(float(static_loss))
            seeds.append(seed_in_graph.clone())
        assert len({int(s.item()) for s in seeds}) == steps  # new masks every step

        names = [k for k, _ in model.named_parameters()]
        ref_list = {k: torch.nn.Parameter(sd[k].clone()) for k in names}
        ref_opt = torch.optim.AdamW(list(ref_list.values()), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        state = {k: v.clone() for k, v in sd.items()}
        ref_losses = []
        for s in range(steps):
            for k, q in ref_list.items():
                state[k] = q.detach().clone()
            drop = _oracle_masks(sites, seeds[s], B, mode)
            _, o_loss, o_grads, o_running = orc.train_step(state, x, y, mode=mode, drop=drop)
            ref_losses.append(float(o_loss))
            for k, v in o_running.items():
                state[k] = v.clone()
            for k, q in ref_list.items():
                q.grad = o_grads[k].clone()
            ref_opt.step()
        fp32 = dtype == torch.float32
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) < (2e-3 if fp32 else 8e-2), (losses, ref_losses)
        num = den = 0.0
        for k, q in model.named_parameters():
            d_hip = (q.detach().float().cpu() - sd[k]).double()
            d_ref = (ref_list[k].detach() - sd[k]).double()
            num += float((d_hip - d_ref).pow(2).sum())
            den += float(d_ref.pow(2).sum())
        rel = (num / den) ** 0.5
        print(f"[captured trajectory with dropout {dtype}] losses {losses} oracle {ref_losses}; update rel-L2 {rel:.3e}")
        assert rel < (0.08 if fp32 else 0.35), rel
    finally:
        ops.trace_dropout_sites(None)
        ops.set_inplace_param_grads(False)
        cvnets_amd.set_compute_dtype(None)
        ops.release_capture_state()  # the seed snapshot was allocated in the captured graph's memory pool, which dies with this test
