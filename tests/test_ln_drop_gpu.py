"""Dropout backward formed inside the LayerNorm backward kernel (csrc/layernorm.hip, DROP; cvnets_amd.ops._ln_backward_launch — off by default,
CVH_LN_DROP=1): x = res + Dropout(linear(h)) followed by LayerNorm (cvnets/modules/transformer.py:140-155).  The masked gradient the kernel
stores must be what cvh_dropout makes of the stored dx — bit for bit — and a TransformerEncoder step must give identical gradients on both paths."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,C,res", [(4099, 144, True), (1000, 192, False), (777, 240, True), (513, 96, True)])
def test_kernel_matches_dropout_of_dx(rows, C, res):
    from cvnets_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=DEV).manual_seed(rows + C)
    x = torch.randn(rows, C, device=DEV, generator=g).bfloat16()
    dy = torch.randn(rows, C, device=DEV, generator=g).bfloat16()
    dres = torch.randn(rows, C, device=DEV, generator=g).bfloat16() if res else None
    gamma = torch.rand(C, device=DEV, generator=g) + 0.5
    mean = x.float().mean(1).contiguous()
    rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    seed = torch.tensor([0x1234567], dtype=torch.int64, device=DEV)
    R = _lib.query("cvh_ln_bwd_rows", rows)
    assert _lib.query("cvh_ln_bwd_drop_ok", C) == 1
    outs = []
    for fused in (False, True):
        dx = torch.empty_like(x)
        dxd = torch.empty_like(x)
        part = torch.empty(R * 2 * C, device=DEV)
        if fused:
            _lib.call("cvh_layernorm_bwd_res_drop", 1, x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                      part.data_ptr(), rows, C, dres.data_ptr() if res else None, dxd.data_ptr(), 0.1, seed.data_ptr(), 77, st)
        else:
            _lib.call("cvh_layernorm_bwd_res", 1, x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                      part.data_ptr(), rows, C, dres.data_ptr() if res else None, st)
            _lib.call("cvh_dropout", 1, dx.data_ptr(), dxd.data_ptr(), rows * C, 0.1, seed.data_ptr(), 77, st)
        torch.cuda.synchronize()
        outs.append((dx.clone(), dxd.clone(), part.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
    assert 0.05 < float((outs[1][1] == 0).float().mean()) < 0.15


def test_encoder_step_is_identical_on_both_paths(monkeypatch):
    import cvnets_amd
    from cvnets_amd import ops
    from cvnets_amd.layers import default_opts
    from cvnets_amd.modules import TransformerEncoder

    cvnets_amd.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(1)
        enc = torch.nn.Sequential(TransformerEncoder(default_opts(), 144, 288, num_heads=4, dropout=0.1),
                                  TransformerEncoder(default_opts(), 144, 288, num_heads=4, dropout=0.1)).to(DEV).train()
        x = torch.randn(8, 64, 144, device=DEV)
        grads = []
        for on in (False, True):
            monkeypatch.setattr(ops, "_LN_DROP", on)
            monkeypatch.setattr(ops, "_seed_snap", {})
            monkeypatch.setattr(ops, "_seeds", {})
            torch.manual_seed(7)
            enc.zero_grad(set_to_none=True)
            ops.advance_dropout_seed(torch.device(DEV))
            ops._stream_ids.value = 1  # (the counter object itself stays: other tests reset it the same way)
            y = enc(x.clone().requires_grad_(True))
            y.float().square().mean().backward()
            ops.finish_backward()
            torch.cuda.synchronize()
            grads.append({k: p.grad.detach().clone() for k, p in enc.named_parameters()})
        for k in grads[0]:
            assert torch.equal(grads[0][k], grads[1][k]), k
    finally:
        cvnets_amd.set_compute_dtype(None)
