"""CPU tests of the host-side dispatch logic added with the round-3 kernels (no GPU: only host-only size / eligibility queries of the C ABI
and the Python predicates in front of them are called)."""
import torch


def test_stem_dispatch_predicate_and_rows():
    """ops.stem_eligible: the stem kernels take conv_1 of MobileViT / MobileViTv2 on the raw NCHW batch (cvnets/models/classification/mobilevit.py:62-72) and
    nothing else; cvh_stem_rows = workgroups = partial rows, 0 < rows <= 2048, one per 8 x 64 output tile when there are fewer."""
    from cvnets_amd import _lib, ops
    ops.set_compute_dtype(torch.bfloat16)
    try:
        x = torch.zeros(2, 3, 64, 64)
        w16, w32, w24 = torch.zeros(16, 3, 3, 3), torch.zeros(32, 3, 3, 3), torch.zeros(24, 3, 3, 3)
        ok = lambda **kw: ops.stem_eligible(kw.get("x", x), kw.get("w", w16), kw.get("bias"), kw.get("s", 2), kw.get("p", 1), kw.get("d", 1),
                                            kw.get("bn", True), kw.get("res"), kw.get("x2"))
        assert ok() and ok(w=w32) and ok(x=x.bfloat16())
        assert not ok(w=w24)                                   # other widths: generic path
        assert not ok(s=1) and not ok(p=0) and not ok(d=2)     # other geometry
        assert not ok(bn=False) and not ok(bias=torch.zeros(16)) and not ok(res=x)
        assert not ok(x=torch.zeros(2, 3, 64, 62))             # W % 4 != 0
        assert not ok(x=torch.zeros(2, 8, 64, 64))             # not the 3-channel image
        assert not ok(x=x.clone().requires_grad_(True))        # an image that wants a gradient: the stem has no dX
        assert not ok(x=x.double())
        assert not ok(x=x.permute(0, 1, 3, 2))                 # not contiguous NCHW
        ops.set_compute_dtype(torch.float32)
        assert not ok()                                        # fp32 compute: generic fp32 kernels
    finally:
        ops.set_compute_dtype(None)
    assert _lib.query("cvh_stem_rows", 1024, 256, 256, 16) == 2048
    assert _lib.query("cvh_stem_rows", 2, 64, 64, 32) == 2 * 4 * 1
    assert _lib.load().cvh_stem_rows(2, 64, 62, 16) < 0 and _lib.load().cvh_stem_rows(2, 64, 64, 24) < 0


def test_streaming_ir_kernels_cover_exactly_the_big_blocks():
    """cvh_ir_exp_bwd_rows / cvh_ir_red_fwd_rows: > 0 for the InvertedResidual shapes of MobileViT-S layer_1 .. layer_3 at batch >= 64 k rows,
    0 (= use the generic GEMMs) for small inputs and for the widths whose weights do not fit LDS beside the tiles."""
    from cvnets_amd import _lib
    M = 1024 * 64 * 64
    assert _lib.query("cvh_ir_exp_bwd_rows", 1024 * 128 * 128, 64, 16) == 1024
    assert _lib.query("cvh_ir_exp_bwd_rows", 1024 * 128 * 128, 128, 32) == 512
    assert _lib.query("cvh_ir_exp_bwd_rows", M, 256, 64) == 256
    assert _lib.query("cvh_ir_exp_bwd_rows", M, 384, 96) == 0 and _lib.query("cvh_ir_exp_bwd_rows", M, 512, 128) == 0
    assert _lib.query("cvh_ir_exp_bwd_rows", 4096, 256, 64) == 0
    assert _lib.query("cvh_ir_exp_bwd_rows", 70000, 256, 64) == 256 and _lib.query("cvh_ir_exp_bwd_rows", 65536, 64, 16) == 1024
    for hid, n, rows in ((64, 32, 1024), (128, 64, 512), (256, 64, 256), (256, 96, 256)):
        assert _lib.query("cvh_ir_red_fwd_rows", M, hid, n) == rows
    assert _lib.query("cvh_ir_red_fwd_rows", M, 384, 128) == 0 and _lib.query("cvh_ir_red_fwd_rows", M, 256, 32) == 0
    assert _lib.query("cvh_ir_red_fwd_rows", 1000, 256, 64) == 0


def test_conv_dw_scratch_query_switches_kernels_by_geometry():
    """cvh_gemm_dw_scratch_elems_conv: a whole number of [N][KH*KW*Cin] partial rows; the 3x3 stride-1 convs of the MobileViT blocks get
    conv3x3_dw_kernel's row count (512 workgroups shared between the channel slabs), everything else the generic planner's."""
    from cvnets_amd import _lib
    B, H = 1024, 32
    for C1, C2, N, slabs in ((96, 0, 96, 2), (96, 96, 96, 4), (128, 128, 128, 8), (160, 0, 160, 5)):
        Hh = {96: 32, 128: 16, 160: 8}[N]
        n = _lib.query("cvh_gemm_dw_scratch_elems_conv", 1, B, Hh, Hh, Hh, Hh, C1, C2, 3, 3, 1, 1, 1, N, 0)
        row = N * 9 * (C1 + C2)
        assert n % row == 0 and n // row == 512 // slabs, (N, n // row)
    M = B * H * H
    generic = _lib.query("cvh_gemm_dw_scratch_elems", M, 96, 9 * 96)
    assert _lib.query("cvh_gemm_dw_scratch_elems_conv", 0, B, H, H, H, H, 96, 0, 3, 3, 1, 1, 1, 96, 0) == generic       # fp32: im2col kernel
    assert _lib.query("cvh_gemm_dw_scratch_elems_conv", 1, B, H, H, H, H, 96, 0, 3, 3, 1, 1, 1, 96, 1) == generic       # bias folded: im2col kernel
    assert _lib.query("cvh_gemm_dw_scratch_elems_conv", 1, B, H, H, H, H, 96, 0, 3, 3, 1, 1, 1, 64, 0) == _lib.query("cvh_gemm_dw_scratch_elems", M, 64, 864)
    assert _lib.query("cvh_gemm_dw_scratch_elems_conv", 1, M, 1, 1, 1, 1, 144, 0, 1, 1, 1, 0, 1, 432, 0) == _lib.query("cvh_gemm_dw_scratch_elems", M, 432, 144)


def test_round4_size_queries_and_eligibility():
    """cvh_ir_pb_rows (one-pass projection backward) and cvh_dwx_rows (y1-recomputing depthwise kernels): host-only planners — > 0 exactly for the
    shapes the kernels are instantiated for; fused._dwx_eligible mirrors the C side's refusals."""
    from cvnets_amd import _lib, fused, ops
    M = 1024 * 64 * 64
    assert _lib.query("cvh_ir_pb_rows", M, 256, 64) == 768 // 4          # ~3 workgroups per CU, split over the 4 channel chunks
    assert _lib.query("cvh_ir_pb_rows", 1024 * 128 * 128, 64, 32) == 768
    assert _lib.query("cvh_ir_pb_rows", 1024 * 8 * 8, 512, 160) == 768 // 8
    assert _lib.query("cvh_ir_pb_rows", 5000, 512, 160) == (5000 + 63) // 64     # fewer tiles than workgroup slots: one tile each
    for M_, hid, cout in ((1000, 256, 64), (M, 72, 32), (M, 256, 48), (M, 256, 192), (M, 256, 24)):   # tiny, hid % 64, Cout % 32, > 160, < 32
        assert _lib.query("cvh_ir_pb_rows", M_, hid, cout) == 0
    # dwx: 2048 / chunks workgroups (<= 1024, >= 32), never more than there are tiles
    assert _lib.query("cvh_dwx_rows", 1024, 64, 64, 256, 1) == 512
    assert _lib.query("cvh_dwx_rows", 1024, 128, 128, 64, 1) == 1024
    assert _lib.query("cvh_dwx_rows", 1, 8, 16, 64, 1) == 1
    w = lambda hid, cin: torch.zeros(hid, cin, 1, 1)
    ok = lambda dt, cin, hid, s, act: fused._dwx_eligible(dt, cin, w(hid, cin), hid, s, act)
    assert ok(torch.bfloat16, 64, 256, 1, ops.ACT_SILU) and ok(torch.bfloat16, 128, 512, 2, ops.ACT_SILU) and ok(torch.bfloat16, 16, 64, 1, ops.ACT_SILU)
    assert not ok(torch.float32, 64, 256, 1, ops.ACT_SILU)        # fp32: the y1-storing kernels
    assert not ok(torch.bfloat16, 96, 384, 1, ops.ACT_SILU)       # stride 1 is instantiated up to 64 input channels
    assert not ok(torch.bfloat16, 24, 96, 2, ops.ACT_SILU)        # widths of the xx_small model
    assert not ok(torch.bfloat16, 64, 256, 1, ops.ACT_RELU)       # SiLU-only kernels
    assert not fused._dwx_eligible(torch.bfloat16, 64, torch.zeros(256, 60, 1, 1), 256, 1, ops.ACT_SILU)  # padded input channels


def test_dropout_trace_hook_and_per_task_queue_are_inert_on_cpu():
    """ops.trace_dropout_sites: a plain Python recorder (p = 0 draws are not recorded); the deferred-reduction queue is per autograd graph
    task and empty outside a backward pass (finish_backward is idempotent)."""
    from cvnets_amd import ops
    sink = []
    ops.trace_dropout_sites(sink)
    try:
        ops._trace_site("linear", 7, 0.1, (4, 8))
        ops._trace_site("dropout", 8, 0.0, (4, 8))
    finally:
        ops.trace_dropout_sites(None)
    ops._trace_site("linear", 9, 0.5, (1,))
    assert sink == [("linear", 7, 0.1, (4, 8))]
    assert ops._pending_by_task == {} and not ops._ensure_backward_callback()   # no graph task is running
    ops.finish_backward()
    ops.finish_backward()
    ops.flush_deferred_reductions()            # outside backward: nothing to do, no error
    ops.flush_deferred_reductions(task=12345)  # an id that never queued anything


def test_bf16_rounding_points_oracle_rounds_where_it_says():
    """oracle/bf16_points.py (test infrastructure): conv / linear inputs, weights and outputs, activations, LayerNorm, P.V and residual sums are
    rounded to bf16, gradients through the same points too; scores stay fp32.  On a tiny MobileViT the emulation sits at bf16 distance from
    the fp32 oracle — not at zero (it rounds) and not far (it rounds only)."""
    import json
    import os
    from oracle import bf16_points, mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict
    gold = os.path.join(os.path.dirname(__file__), "golden")
    sd = seeded_state_dict(json.load(open(os.path.join(gold, "mobilevit_xx_small_keys.json"))), seed=0)
    x, y = seeded_input((2, 3, 64, 64), seed=11), seeded_labels(2, 1000, seed=11)
    m = bf16_points.Bf16Points()
    with m:
        out = torch.nn.functional.linear(torch.full((1, 3), 1.0 + 2 ** -10), torch.eye(3))
    assert torch.equal(out, torch.ones(1, 3)) and m.counts == {"linear": 1}     # 1 + 2^-10 rounds to 1 in bf16
    l32, loss32, g32, _ = orc.train_step(sd, x, y, mode="xx_small")
    l16, loss16, g16, _ = bf16_points.train_step(sd, x, y, mode="xx_small")
    rel = float((l16 - l32).norm() / l32.norm())
    assert 1e-4 < rel < 3e-1, rel
    assert abs(float(loss16) - float(loss32)) < 1e-1
    k = "classifier.fc.weight"
    assert 1e-4 < float((g16[k] - g32[k]).norm() / g32[k].norm()) < 5e-1


def test_bf16_rounded_evaluation_is_chaotic_at_the_noise_level():
    """Why tests/test_bf16_parity_gpu.py bounds the HIP path by the rounding-points emulation's own NOISE instead of asking for ~1e-3 agreement
    with it: an fp32-ulp perturbation of the input (1e-6 relative — a stand-in for another fp32 summation order) leaves the fp32 oracle where
    it was (< 1e-4) and moves the bf16-rounded evaluation by about as much as bf16 moves it away from fp32 (> 3e-3 here; tools/bf16_sensitivity.py
    on MobileViT-S 256^2 b16: 1.5e-2 logits / 5.1e-2 gradients against 1.4e-6 / 4.8e-6)."""
    import json
    import os
    from oracle import bf16_points, mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_labels, seeded_state_dict
    gold = os.path.join(os.path.dirname(__file__), "golden")
    sd = seeded_state_dict(json.load(open(os.path.join(gold, "mobilevit_xx_small_keys.json"))), seed=0)
    x, y = seeded_input((4, 3, 64, 64), seed=1), seeded_labels(4, 1000, seed=1)
    xp = x * (1 + 1e-6 * torch.randn(x.shape, generator=torch.Generator().manual_seed(5)))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    l32, _, _, _ = orc.train_step(sd, x, y, mode="xx_small")
    l32p, _, _, _ = orc.train_step(sd, xp, y, mode="xx_small")
    l16, _, _, _ = bf16_points.train_step(sd, x, y, mode="xx_small")
    l16p, _, _, _ = bf16_points.train_step(sd, xp, y, mode="xx_small")
    assert rel(l32p, l32) < 1e-4
    assert rel(l16p, l16) > 3e-3 and rel(l16p, l16) > 0.2 * rel(l16, l32)


def test_temporal_block_host_composition_against_the_oracle(monkeypatch):
    """MobileViTBlock.forward((x, x_prev)) (cvnets/modules/mobilevit_block.py:289-326) is HOST composition over the hot path's ops: patch
    order, which tensors feed query / key / value, the padded cross-attention, where the residuals and the final LayerNorm sit.  With torch
    stand-ins for the kernels (test doubles, CPU) the composition must reproduce the oracle — outputs and the gradients of x and x_prev — for
    a first frame (x_prev = None), equal patch counts, and foreign patch counts on either side (T < N and T > N)."""
    import torch.nn.functional as F

    from cvnets_amd import ops
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlock
    from oracle import mobilevit_oracle as orc
    from oracle.weights import seeded_input, seeded_state_dict

    def linear(x, w, b=None, *, act=ops.ACT_NONE, drop_p=0.0, residual=None, expose_pre=False, in_pre=None, in_act=ops.ACT_NONE):
        pre = F.linear(x, w, b)
        y = F.silu(pre) if act == ops.ACT_SILU else pre
        assert act in (ops.ACT_NONE, ops.ACT_SILU) and drop_p == 0.0
        if residual is not None:
            y = y + residual
        return (y, pre) if expose_pre else y  # (in_pre / in_act only steer the fused backward: the input is already the activated tensor)

    def attention(qkv, heads, seqmap, causal=False, key_padding_mask=None, drop_p=0.0, attn_bias=None):
        b, L = seqmap[0], seqmap[1]
        assert tuple(seqmap[2:]) == (1, 1, L, 1, L) and not causal and drop_p == 0.0
        d = qkv.shape[1] // 3
        q, k, v = qkv.view(b, L, 3, heads, d // heads).permute(2, 0, 3, 1, 4)
        a = (q * (d // heads) ** -0.5) @ k.transpose(-1, -2)
        if key_padding_mask is not None:
            a = a.masked_fill(key_padding_mask.bool()[:, None, None, :], float("-inf"))
        if attn_bias is not None:
            a = a + (attn_bias if attn_bias.dim() == 2 else attn_bias[:, None])
        return (torch.softmax(a, -1) @ v).transpose(1, 2).reshape(b * L, d)

    class TorchConv(torch.nn.Module):  # ConvLayer2d stand-in on NCHW tensors, same parameters
        def __init__(self, layer):
            super().__init__()
            self.layer = layer

        def forward(self, x, residual=None, x2=None):
            blk = self.layer.block
            if x2 is not None:
                x = torch.cat((x, x2), dim=1)
            k = blk.conv.weight.shape[-1]
            y = F.conv2d(x, blk.conv.weight, blk.conv.bias, padding=(k - 1) // 2)
            if "norm" in blk._modules:
                y = F.batch_norm(y, None, None, blk.norm.weight, blk.norm.bias, training=True, eps=blk.norm.eps)
            return F.silu(y) if "act" in blk._modules else y

    monkeypatch.setattr(ops, "linear", linear)
    monkeypatch.setattr(ops, "attention", attention)
    monkeypatch.setattr(ops, "compute_dtype", lambda: torch.float32)
    monkeypatch.setattr(ops, "to_nhwc", lambda x, dtype=None: x)
    monkeypatch.setattr(ops, "layer_norm_tokens", lambda x, ln, seqmap: F.layer_norm(x, (x.shape[1],), ln.weight, ln.bias, ln.eps))
    monkeypatch.setattr(ops, "layer_norm_fork", lambda x, ln, seqmap: (x, ops.layer_norm_tokens(x, ln, seqmap)))
    monkeypatch.setattr(ops, "add", lambda a, b: a + b)
    monkeypatch.setattr(ops, "resize_bilinear", lambda x, h, w: F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False))

    b, cin, d, ffn, blocks, hd, patch = 2, 16, 32, 64, 2, 8, 2
    block = MobileViTBlock(default_opts(), in_channels=cin, transformer_dim=d, ffn_dim=ffn, n_transformer_blocks=blocks, head_dim=hd,
                           patch_h=patch, patch_w=patch).train()
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in block.state_dict().items()}, seed=23)
    block.load_state_dict(sd)
    block.local_rep.conv_3x3, block.local_rep.conv_1x1 = TorchConv(block.local_rep.conv_3x3), TorchConv(block.local_rep.conv_1x1)
    block.conv_proj, block.fusion = TorchConv(block.conv_proj), TorchConv(block.fusion)
    for (H, W), T in (((8, 8), None), ((8, 8), 16), ((8, 6), 5), ((7, 9), 40)):  # (7, 9): resized to 8 x 10 and back
        x = seeded_input((b, cin, H, W), seed=51).requires_grad_(True)
        xo = x.detach().clone().requires_grad_(True)
        xp = xpo = None
        if T is not None:
            xp = seeded_input((b * patch * patch, T, d), seed=52).requires_grad_(True)
            xpo = xp.detach().clone().requires_grad_(True)
        fm, p = block((x, xp))
        fm_o, p_o = orc.mobilevit_block_temporal(sd, "", xo, xpo, blocks, d // hd, True, None, patch, patch)
        assert fm.shape == fm_o.shape and p.shape == p_o.shape
        assert torch.allclose(fm, fm_o, rtol=1e-4, atol=1e-5) and torch.allclose(p, p_o, rtol=1e-4, atol=1e-5), (H, W, T)
        g, gp = seeded_input(tuple(fm.shape), seed=53), seeded_input(tuple(p.shape), seed=54)
        ins, ins_o = ([x], [xo]) if T is None else ([x, xp], [xo, xpo])
        got = torch.autograd.grad((fm * g).sum() + (p * gp).sum(), ins)
        want = torch.autograd.grad((fm_o * g).sum() + (p_o * gp).sum(), ins_o)
        for a, w in zip(got, want):
            assert torch.allclose(a, w, rtol=1e-3, atol=1e-5), (H, W, T)


def test_gradient_statistics_handover_is_keyed_and_validated():
    """ops.offer_grad_stats / take_grad_stats (the backward hand-over of DESIGN.md section 4a'''): statistics are taken only for the very
    gradient tensor they were formed on, only against the output tensor (address and version) they were formed with, and only once."""
    import torch
    from cvnets_amd import ops

    ops._bwd_handover.clear()
    dx, x, part = torch.zeros(8, 4), torch.zeros(8, 4), torch.zeros(3 * 2 * 4)
    ops.offer_grad_stats(dx, part, 3, 4, 8, x)
    assert ops.take_grad_stats(torch.zeros(8, 4), 4, 8, x.data_ptr(), x._version) is None          # another gradient tensor
    ops.offer_grad_stats(dx, part, 3, 4, 8, x)
    assert ops.take_grad_stats(dx, 4, 8, torch.zeros(8, 4).data_ptr(), 0) is None                  # formed against another output tensor
    assert ops.take_grad_stats(dx, 4, 8, x.data_ptr(), x._version) is None                         # ... and an offer is consumed by the first lookup
    ops.offer_grad_stats(dx, part, 3, 4, 8, x)
    x.add_(1.0)                                                                                     # the output was modified in place since
    assert ops.take_grad_stats(dx, 4, 8, x.data_ptr(), x._version) is None
    ops.offer_grad_stats(dx, part, 3, 4, 8, x)
    assert ops.take_grad_stats(dx, 8, 8, x.data_ptr(), x._version) is None                         # another width
    ops.offer_grad_stats(dx, part, 3, 4, 8, x)
    got = ops.take_grad_stats(dx, 4, 8, x.data_ptr(), x._version)
    assert got is not None and got[0] is part and got[1] == 3
    for i in range(70):                                                                             # offers nobody takes do not pile up
        ops.offer_grad_stats(torch.zeros(1), part, 3, 4, 8, x)
    assert len(ops._bwd_handover) <= 65
    ops.finish_backward()
    assert not ops._bwd_handover


def test_mobilevitv2_patch_fold_and_unfold_follow_the_reference_layout():
    """MobileViTBlockv2._unfold / _fold (the [B, C, P, N] patches the temporal path returns and takes) against the reference's
    unfolding_pytorch / folding_pytorch formulas (cvnets/modules/mobilevit_block.py:526-555: F.unfold / F.fold with kernel = stride = patch)."""
    import torch
    import torch.nn.functional as F
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import MobileViTBlockv2

    for ph, pw, H, W in ((2, 2, 8, 12), (4, 2, 8, 6)):
        blk = MobileViTBlockv2(default_opts(), in_channels=16, attn_unit_dim=32, n_attn_blocks=1, patch_h=ph, patch_w=pw)
        fm = torch.randn(2, 32, H, W)
        ref = F.unfold(fm, kernel_size=(ph, pw), stride=(ph, pw)).reshape(2, 32, ph * pw, -1)
        assert torch.equal(blk._unfold(fm), ref)
        back = F.fold(ref.reshape(2, 32 * ph * pw, -1), output_size=(H, W), kernel_size=(ph, pw), stride=(ph, pw))
        assert torch.equal(blk._fold(ref, H, W), back) and torch.equal(back, fm)
