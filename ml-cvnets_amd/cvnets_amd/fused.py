"""Block-level autograd functions built on BatchNorm links (csrc/bnlink.hpp): conv -> BN -> act chains whose BatchNorm passes live
inside the neighbouring conv kernels.

``InvertedResidualFn`` is InvertedResidual.forward (cvnets/modules/mobilenetv2.py:231-235) — exp 1x1-BN-act -> depthwise 3x3-BN-act
-> red 1x1-BN (+x) — as ONE autograd node.  HBM traffic of the 4x-expanded tensors (fused vs the per-layer path of ops.py):

    forward   write y1, read y1, write y2, read y2                                   (4 wide passes instead of 8)
    backward  dX3: read y2, write g2 | dW3: read y2 | depthwise: read g2, y2, y1, write g1 | dW1, dX1: read g1, y1 each
                                                                                      (11 wide passes instead of 18)

where y = raw conv outputs and g = dz * act'(bn(y)).  Normalised activations and BatchNorm input gradients are formed on load by the
consumer (cvh_operand_xf); statistics leave the producer's epilogue as per-workgroup partial rows and are finalised by the tiny
cvh_bn_finalize / cvh_bn_bwd_finalize launches.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib, ops
from .ops import ACT_NONE, _dt, _f32, _p, _stream

_IR_EXP_FUSED = __import__("os").environ.get("CVH_IR_EXP_FUSED", "1") != "0"
_IR_RED_FUSED = __import__("os").environ.get("CVH_IR_RED_FUSED", "1") != "0"
# expansion + depthwise as ONE kernel per direction with y1 recomputed from the narrow input (csrc/dwx.hip): 0 = off, 1 = forward and
# backward, "fwd" = forward only (the backward then rebuilds y1 with the expansion GEMM and runs cvh_dwconv_bn_bwd: A/B runs, tests)
_IR_X = __import__("os").environ.get("CVH_IR_X", "1")
# projection backward (dW3, g2, statistics) from ONE pass over y2, with the BatchNorm-backward apply of the block output folded into its
# operand load (csrc/ir_pb.hip); 0 = the dW GEMM + dX GEMM pair
_IR_PB = __import__("os").environ.get("CVH_IR_PB", "1") != "0"
DW_SHAPE_LOG = None  # set to a list to record the plain dW GEMMs launched from this module (bench.py)


def _xf(mode: int = 0, src2=None, c0=None, c1=None, c2=None, act: int = 0):
    if mode == 0:
        return None
    return ctypes.byref(_lib.OperandXf(mode, _p(src2), _p(c0), _p(c1), _p(c2), int(act)))


def _pw_gemm(a, a_xf, K, wp, out, M, N, *, residual=None, e_mode=0, e_aux=None, e_stats=None, e_act=0, want_stats=False):
    """returns (partial statistics rows, R) of the epilogue (or (None, 0))"""
    part, R = None, 0
    if want_stats:
        R = _lib.query("cvh_conv_gemm_grid_rows", int(M), int(N))
        part = _f32(R * 2 * N, out.device)
    _lib.call("cvh_pw_gemm_bn", _dt(out), _p(a), a_xf, int(K), _p(wp), _p(out), int(M), int(N), _p(residual), int(e_mode), _p(e_aux),
              _p(e_stats), int(e_act), _p(part), _stream())
    return part, R


def _bwd_finalize(part, R, C, count, gamma, stats, pg, pb, training):
    """(sum g, sum g*xhat) partial rows -> dgamma, dbeta (in place when the parameters have gradient sinks) and coef[3][C]"""
    dev = part.device
    sg, sb = ops._grad_sink(pg), ops._grad_sink(pb)
    inplace = sg is not None and sb is not None
    dgamma = sg if inplace else _f32(C, dev)
    dbeta = sb if inplace else _f32(C, dev)
    coef = _f32(3, dev, C)
    _lib.call("cvh_bn_bwd_finalize", _p(part), R, C, float(count), _p(gamma), _p(stats[0]), _p(stats[1]), 1 if training else 0,
              1 if inplace else 0, _p(dgamma), _p(dbeta), _p(coef[0]), _p(coef[1]), _p(coef[2]), _stream())
    if inplace:
        return coef, None, None
    return coef, dgamma, dbeta


def _pw_weight_grad(dy, dy_xf_args, x, x_xf_args, weight, M, N, K):
    """dW of a pointwise conv with both operands transformed on load; returns the gradient or None (added in place, side stream)."""
    sink = ops._grad_sink(weight)
    n_scr = _lib.query("cvh_gemm_dw_scratch_elems", int(M), int(N), int(K))
    Cin_real = weight.shape[1]

    def launch(dw, accumulate):
        scr = _f32(max(n_scr, 1), dy.device)
        deferred = accumulate and n_scr > 0 and ops.defer_reduce(scr, dw, n_scr // (N * K), N * K, N * K, kind=0 if Cin_real == K else 1, N=N,
                                                                 Ktot=K, Cin=K, Cin_real=Cin_real, khw=1)
        _lib.call("cvh_pw_gemm_dw_bn", _dt(dy), _p(dy), _xf(*dy_xf_args), _p(x), _xf(*x_xf_args), None if deferred else _p(dw), int(M), int(N),
                  int(K), int(Cin_real), _p(scr), n_scr, accumulate, _stream())

    side = ops._param_grad_stream(dy.device) if sink is not None else None
    if side is not None:
        with torch.cuda.stream(side):
            for t in (dy, x) + tuple(dy_xf_args[1:]) + tuple(x_xf_args[1:]):  # everything the side-stream launch reads
                if isinstance(t, torch.Tensor):
                    t.record_stream(side)
            launch(sink, 1)
        return None
    if sink is not None:
        launch(sink, 1)
        return None
    dw = torch.empty(weight.shape, dtype=torch.float32, device=dy.device)
    launch(dw, 0)
    return dw


def _dwx_eligible(dt, Cin, w1, hid, stride, act1) -> bool:
    """shapes cvh_dwx_fwd / cvh_dwx_bwd cover (bf16; SiLU; unpadded input channels; stride 1 up to 64 input channels)"""
    return (_IR_X != "0" and dt == torch.bfloat16 and act1 == ops.ACT_SILU and w1.shape[1] == Cin and Cin in (16, 32, 64, 96, 128)
            and not (stride == 1 and Cin > 64) and hid % 8 == 0)


def _gram(x, M, Kp, s_known=None):
    """G = x^T x [Kp][Kp] and s = 1^T x [Kp] of a narrow [M][Kp] tensor (float32): one cvh_gemm_dw + one cvh_colsum (the latter skipped
    when the producer of x already knows its column sums: `s_known`)"""
    dev = x.device
    if DW_SHAPE_LOG is not None:
        DW_SHAPE_LOG.append((int(M), 1, 1, 1, 1, int(Kp), 0, 1, 1, 1, 0, 1, int(Kp), int(Kp)))
    G = torch.empty(Kp * Kp, dtype=torch.float32, device=dev)
    n_scr = _lib.query("cvh_gemm_dw_scratch_elems", int(M), int(Kp), int(Kp))
    scr = _f32(max(n_scr, 1), dev)
    _lib.call("cvh_gemm_dw", _dt(x), _p(x), _p(x), None, Kp, 0, _p(G), int(M), 1, 1, 1, 1, 1, 1, 1, 0, 1, int(Kp), int(Kp), _p(scr), n_scr, 0,
              _stream())
    if s_known is not None:
        return G, s_known
    R = _lib.query("cvh_colreduce_rows", int(M), int(Kp))
    part = _f32(R * 2 * Kp, dev)
    s_ = _f32(Kp, dev)
    _lib.call("cvh_colsum", _dt(x), _p(x), int(M), int(Kp), _p(part), _p(s_), 1.0, 0, _stream())
    return G, s_


def _linear_bn_weight_grad(g, x, weight, coef, M, N, Kp, P_ready=None, gram=None):
    """dW of a 1x1 conv y = x W^T that sits in front of a train-mode BatchNorm, from g = dz * act'(bn(y)) and the backward coefficients
    coef[3][N] of that BatchNorm:  dW = diag(ca) (g^T x) + diag(cb) W (x^T x) + cc (1^T x)  — y is not read (csrc/bnlink.hip)."""
    sink = ops._grad_sink(weight)
    Kr = weight.shape[1]
    dev = g.device

    def launch(dw, accumulate):
        def gemm_dw(dy, n_cols, cin_real):
            if DW_SHAPE_LOG is not None:  # bench.py's kernel probe: (B, H, W, Ho, Wo, C1, C2, KH, KW, stride, pad, dil, N, Cin_real)
                DW_SHAPE_LOG.append((int(M), 1, 1, 1, 1, int(Kp), 0, 1, 1, 1, 0, 1, int(n_cols), int(cin_real)))
            out = torch.empty(n_cols * cin_real, dtype=torch.float32, device=dev)
            n_scr = _lib.query("cvh_gemm_dw_scratch_elems", int(M), int(n_cols), int(Kp))
            scr = _f32(max(n_scr, 1), dev)
            _lib.call("cvh_gemm_dw", _dt(dy), _p(dy), _p(x), None, Kp, 0, _p(out), int(M), 1, 1, 1, 1, 1, 1, 1, 0, 1, int(n_cols), int(cin_real),
                      _p(scr), n_scr, 0, _stream())
            return out
        P = P_ready if P_ready is not None else gemm_dw(g, N, Kr)      # g^T x   [N][Kr]
        if gram is not None:       # the forward pass already formed them for the BatchNorm statistics (csrc/dwx.hip)
            G, s_ = gram
        else:
            G = gemm_dw(x, Kp, Kp)     # x^T x   [Kp][Kp]
            R = _lib.query("cvh_colreduce_rows", int(M), int(Kp))
            part = _f32(R * 2 * Kp, dev)
            s_ = _f32(Kp, dev)
            _lib.call("cvh_colsum", _dt(x), _p(x), int(M), int(Kp), _p(part), _p(s_), 1.0, 0, _stream())
        _lib.call("cvh_bn_dw_combine", _p(P), _p(weight), _p(G), _p(s_), _p(coef), _p(dw), int(N), int(Kr), accumulate, _stream())

    side = ops._param_grad_stream(dev) if (sink is not None and P_ready is None) else None
    if side is not None:
        with torch.cuda.stream(side):
            for t in (g, x, coef) + (tuple(gram) if gram is not None else ()):
                t.record_stream(side)
            launch(sink, 1)
        return None
    if sink is not None:
        launch(sink, 1)
        return None
    dw = torch.empty(weight.shape, dtype=torch.float32, device=dev)
    launch(dw, 0)
    return dw


class InvertedResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, wd, g2, b2, rm2, rv2, w3, g3, b3, rm3, rv3, cfg, xsum=None, gram_in=None):
        stride, use_res, training, act1, act2, mom, eps = cfg
        ops._check_dev(x)
        B, Cin, H, W = x.shape
        hid, Cout = w1.shape[0], w3.shape[0]
        if ops.pad8(w1.shape[1]) != Cin or hid % 8 or Cout % 8:
            raise RuntimeError("InvertedResidualFn: channel counts must be multiples of 8")
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        M1, M2 = B * H * W, B * Ho * Wo
        dev, dt = x.device, x.dtype
        wp1, wpd, wp3 = ops.pack_weight(w1, dt, 0), ops.pack_weight(wd, dt, 2), ops.pack_weight(w3, dt, 0)
        if not training:
            sts = []
            for g, b, rm, rv, e, C in ((g1, b1, rm1, rv1, eps[0], hid), (g2, b2, rm2, rv2, eps[1], hid), (g3, b3, rm3, rv3, eps[2], Cout)):
                st = _f32(4, dev, C)
                _lib.call("cvh_bn_eval_coeff", _p(g), _p(b), _p(rm), _p(rv), float(e), C, _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]),
                          _stream())
                sts.append(st)
            st1, st2, st3 = sts
        use_x = _dwx_eligible(dt, Cin, w1, hid, stride, act1)
        y2 = ops.nhwc_empty(B, hid, Ho, Wo, dt, dev)
        gram = None
        if use_x:
            # y1 = x W1^T never exists in HBM: BatchNorm statistics of the expansion from the Gram matrix of the narrow input (y1 is linear
            # in x), expansion + BN + act + depthwise conv in one kernel (csrc/dwx.hip)
            y1 = None
            if training:
                # G = x^T x, 1^T x: formed by the producer of x in its BatchNorm-apply pass (ops._bn_apply_out) or by a pass over x here
                gram = (gram_in[:Cin * Cin], gram_in[Cin * Cin:]) if gram_in is not None else _gram(x, M1, Cin, s_known=xsum)
                part = _f32(2 * hid, dev)
                _lib.call("cvh_gram_bn_stats", _p(gram[0]), _p(gram[1]), _p(wp1), _p(part), hid, Cin, Cin, _stream())
                st1 = ops._bn_forward(x, M1, hid, part, 1, g1, b1, rm1, rv1, True, mom[0], eps[0])
            part, R = None, 0
            if training:
                R = _lib.query("cvh_dwx_fwd_rows", B, H, W, Cin, hid, stride)
                part = _f32(R * 2 * hid, dev)
            _lib.call("cvh_dwx_fwd", _dt(x), _p(x), _p(wp1), _p(st1[2]), _p(st1[3]), act1, _p(wpd), _p(y2), _p(part), B, H, W, Ho, Wo, Cin, hid,
                      stride, _stream())
        else:
            y1 = ops.nhwc_empty(B, hid, H, W, dt, dev)
            part, R = _pw_gemm(x, None, Cin, wp1, y1, M1, hid, want_stats=training)
            if training:
                st1 = ops._bn_forward(y1, M1, hid, part, R, g1, b1, rm1, rv1, True, mom[0], eps[0])
            part, R = None, 0
            if training:
                R = _lib.query("cvh_dwconv_bn_rows", B, Ho, Wo, hid, stride)
                part = _f32(R * 2 * hid, dev)
            _lib.call("cvh_dwconv_bn_fwd", _dt(x), _p(y1), _xf(1, None, st1[2], st1[3], None, act1), _p(wpd), _p(y2), B, H, W, Ho, Wo, hid,
                      stride, _p(part), _stream())
        if training:
            st2 = ops._bn_forward(y2, M2, hid, part, R, g2, b2, rm2, rv2, True, mom[1], eps[1])
        y3 = ops.nhwc_empty(B, Cout, Ho, Wo, dt, dev)
        R3 = _lib.query("cvh_ir_red_fwd_rows", M2, hid, Cout) if (_IR_RED_FUSED and dt == torch.bfloat16) else 0
        if R3 > 0:  # projection as a read-dominated stream: BN2 + act on the way into LDS, W3 resident (csrc/ir_fwd.hip)
            part, R = (_f32(R3 * 2 * Cout, dev), R3) if training else (None, 0)
            _lib.call("cvh_ir_red_fwd", _dt(y2), _p(y2), _p(st2[2]), _p(st2[3]), act2, _p(wp3), _p(y3), _p(part), M2, hid, Cout, _stream())
        else:
            part, R = _pw_gemm(y2, _xf(1, None, st2[2], st2[3], None, act2), hid, wp3, y3, M2, Cout, want_stats=training)
        if training:
            st3 = ops._bn_forward(y3, M2, Cout, part, R, g3, b3, rm3, rv3, True, mom[2], eps[2])
        out = ops.nhwc_empty(B, Cout, Ho, Wo, dt, dev)
        gs = ops._bn_apply_out(y3, st3, ACT_NONE, x if use_res else None, out, M2, Cout, g3, training)
        ctx.cfg = cfg
        ctx.geom = (B, Cin, H, W, Ho, Wo, hid, Cout)
        ctx.params = (g1, b1, g2, b2, g3, b3)
        # column sums of the block OUTPUT for the next block's statistics, without a pass over it: the output is a train-mode BatchNorm
        # (+ the input on the residual path), so 1^T out = rows * beta3 (+ 1^T x).  (bf16 rounding of the stored output is zero-mean noise.)
        osum = None
        if gs is None and use_x and training and (not use_res or gram is not None):
            osum = _f32(Cout, dev)
            _lib.call("cvh_axpb", _p(b3), float(M2), _p(gram[1]) if use_res else None, _p(osum), Cout, _stream())
        ctx.has_osum = osum is not None
        ctx.use_x = use_x
        ctx.out_id = (out.data_ptr(), out._version)  # to recognise statistics handed back by the consumer of exactly this tensor
        ctx.save_for_backward(x, w1, wd, w3, g1, g2, g3, y1, y2, y3, st1, st2, st3, *(gram if gram is not None else (None, None)))
        # (the side outputs carry no gradient: without this the engine hands backward a freshly zero-filled tensor for each of them —
        # an ATen fill kernel per block inside the captured step)
        ctx.set_materialize_grads(False)
        if gs is not None:
            ctx.mark_non_differentiable(gs)
            return out, None, gs
        if osum is not None:
            ctx.mark_non_differentiable(osum)
            return out, osum
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        stride, use_res, training, act1, act2, mom, eps = ctx.cfg
        B, Cin, H, W, Ho, Wo, hid, Cout = ctx.geom
        x, w1, wd, w3, g1, g2, g3, y1, y2, y3, st1, st2, st3, gram_G, gram_s = ctx.saved_tensors
        gram = (gram_G, gram_s) if gram_G is not None else None
        pg1, pb1, pg2, pb2, pg3, pb3 = ctx.params
        dout = ops.as_nhwc(dout)
        dev, dt = dout.device, dout.dtype
        M1, M2 = B * H * W, B * Ho * Wo
        Rp = 0
        if _IR_PB and dt == torch.bfloat16 and act2 == ops.ACT_SILU and w3.shape[1] == hid:
            Rp = _lib.query("cvh_ir_pb_rows", M2, hid, Cout)
        g2t = torch.empty_like(y2)
        if Rp > 0:
            # BatchNorm of the projection conv: statistics pass over the narrow (dout, y3) only — dy3 = ca dout + cb y3 + cc is formed by the
            # consumer on load; then dW3, g2 and (sum g2, sum g2*xhat2) from ONE pass over y2 (csrc/ir_pb.hip)
            handed = None
            if ops._BN_HANDOVER and training and not use_res and pb3 is not None and dt == torch.bfloat16:
                # the next block's expansion backward wrote dout AND its column sums against this block's output (cvh_ir_exp_bwd_s)
                handed = ops.take_grad_stats(dout, Cout, M2, *ctx.out_id)
            coef3, dg3, db3 = ops._bn_backward_coeffs(y3, dout, st3, g3, ACT_NONE, M2, Cout, training, beta=pb3, handed=handed)
            part = _f32(Rp * 2 * hid, dev)
            dw_part = _f32(Rp * Cout * hid, dev)
            wp3t = ops.pack_weight(w3, dt, 1)  # alive across the launch (uncached packs are temporaries)
            _lib.call("cvh_ir_pb", _dt(y2), _p(dout), _p(y3), _p(coef3), _p(y2), _p(st2), act2, _p(wp3t), _p(g2t), _p(part), _p(dw_part), M2, hid,
                      Cout, _stream())
            sink3 = ops._grad_sink(w3)
            dw3 = None if sink3 is not None else torch.empty(w3.shape, dtype=torch.float32, device=dev)
            n3 = Cout * hid
            if not (sink3 is not None and ops.defer_reduce(dw_part, sink3, Rp, n3, n3)):
                _lib.call("cvh_sum_partials", _p(dw_part), Rp, n3, n3, _p(sink3 if sink3 is not None else dw3), 1.0, 1 if sink3 is not None else 0,
                          _stream())
            R = Rp
        else:
            # BatchNorm of the projection conv: narrow tensor, standalone passes
            dy3, dg3, db3 = ops._bn_backward(y3, dout, st3, g3, ACT_NONE, M2, Cout, training, beta=pb3)
            # dW3 = dy3^T x act(bn2(y2))
            dw3 = _pw_weight_grad(dy3, (0,), y2, (1, None, st2[2], st2[3], None, act2), w3, M2, Cout, hid)
            # g2 = (dy3 W3) * act2'(bn2(y2)) with (sum g2, sum g2*xhat2) from the same epilogue
            part, R = _pw_gemm(dy3, None, Cout, ops.pack_weight(w3, dt, 1), g2t, M2, hid, e_mode=1, e_aux=y2, e_stats=st2, e_act=act2,
                               want_stats=True)
        coef2, dg2, db2 = _bwd_finalize(part, R, hid, M2, g2, st2, pg2, pb2, training)
        # depthwise backward in one pass: dy2 formed on load, g1 out, dW of the depthwise conv, statistics of g1
        g1t = ops.nhwc_empty(B, hid, H, W, dt, dev)
        if ctx.use_x and _IR_X != "fwd":
            # y1 recomputed from x at the tile's own pixels; dX, dW and the statistics on the matrix pipe (csrc/dwx.hip)
            R = _lib.query("cvh_dwx_rows", B, Ho, Wo, hid, stride)
            part = _f32(R * 2 * hid, dev)
            dw_part = _f32(R * hid * 9, dev)
            wp1, wpd = ops.pack_weight(w1, dt, 0), ops.pack_weight(wd, dt, 2)  # both alive across the launch (uncached packs are temporaries)
            _lib.call("cvh_dwx_bwd", _dt(g2t), _p(x), _p(wp1), _p(st1), act1, _p(g2t), _p(y2), _p(coef2[0]), _p(coef2[1]), _p(coef2[2]),
                      _p(wpd), _p(g1t), _p(part), _p(dw_part), B, H, W, Ho, Wo, Cin, hid, stride, _stream())
        else:
            if y1 is None:  # forward ran without y1 (CVH_IR_X=fwd): rebuild it with the expansion GEMM
                y1 = ops.nhwc_empty(B, hid, H, W, dt, dev)
                _pw_gemm(x, None, Cin, ops.pack_weight(w1, dt, 0), y1, M1, hid)
            R = _lib.query("cvh_dwconv_bn_rows", B, Ho, Wo, hid, stride)
            part = _f32(R * 2 * hid, dev)
            dw_part = _f32(R * hid * 9, dev)
            wpd = ops.pack_weight(wd, dt, 2)  # a named reference: an uncached pack is a temporary, and `_p(temporary)` frees it before the launch
            _lib.call("cvh_dwconv_bn_bwd", _dt(g2t), _p(g2t), _xf(2, y2, coef2[0], coef2[1], coef2[2]), _p(y1), _p(st1), act1,
                      _p(wpd), _p(g1t), _p(part), _p(dw_part), B, H, W, Ho, Wo, hid, stride, _stream())
        coef1, dg1, db1 = _bwd_finalize(part, R, hid, M1, g1, st1, pg1, pb1, training)
        sink = ops._grad_sink(wd)
        dwd = None if sink is not None else torch.empty(wd.shape, dtype=torch.float32, device=dev)
        if not (sink is not None and ops.defer_reduce(dw_part, sink, R, hid * 9, hid * 9)):
            _lib.call("cvh_sum_partials", _p(dw_part), R, hid * 9, hid * 9, _p(sink if sink is not None else dwd), 1.0,
                      1 if sink is not None else 0, _stream())
        # expansion conv (LINEAR in x): dy1 = ca*g1 + cb*y1 + cc is never formed and y1 is never re-read — dX1 is one plain GEMM over
        # the channel-concat [g1 | x] with a small derived weight, dW1 the plain dW GEMM on g1 plus K x K glue (csrc/bnlink.hip)
        dx = None
        P1 = None
        if ctx.needs_input_grad[0]:
            wcat = torch.empty(Cin * (hid + Cin), dtype=dt, device=dev)
            bias = _f32(Cin, dev)
            _lib.call("cvh_bn_dx_weights", _dt(g1t), _p(w1), _p(coef1), _p(wcat), _p(bias), hid, w1.shape[1], _stream())
            dx = ops.nhwc_empty(B, Cin, H, W, dt, dev)
            R1 = _lib.query("cvh_ir_exp_bwd_rows", M1, hid, Cin) if (_IR_EXP_FUSED and dt == torch.bfloat16 and w1.shape[1] == Cin) else 0
            if R1 > 0:  # dX1 and the raw dW1 product g1^T x from ONE pass over g1 (csrc/ir_bwd.hip)
                ppart = _f32(R1 * hid * Cin, dev)
                spart = _f32(R1 * 2 * Cin, dev) if ops._BN_HANDOVER else None
                _lib.call("cvh_ir_exp_bwd_s", _dt(g1t), _p(g1t), _p(x), _p(wcat), _p(bias), _p(dout if use_res else None), _p(dx), _p(ppart),
                          _p(spart), M1, hid, Cin, _stream())
                if spart is not None:  # (sum dX, sum dX * x) for the BatchNorm backward of the block whose output x is
                    ops.offer_grad_stats(dx, spart, R1, Cin, M1, x)
                P1 = _f32(hid * Cin, dev)
                _lib.call("cvh_sum_partials", _p(ppart), R1, hid * Cin, hid * Cin, _p(P1), 1.0, 0, _stream())
            else:
                ops._conv_gemm(g1t, x, hid, Cin, wcat, dx, M1, 1, 1, 1, 1, 1, 1, 1, 0, 1, Cin, bias=bias, residual=dout if use_res else None)
        elif use_res:
            dx = dout
        dw1 = _linear_bn_weight_grad(g1t, x, w1, coef1, M1, hid, Cin, P_ready=P1, gram=gram)
        return (dx, dw1, dg1, db1, None, None, dwd, dg2, db2, None, None, dw3, dg3, db3, None, None, None, None, None)


def inverted_residual(x, exp, dw, red, *, stride: int, use_res: bool):
    """exp / dw / red: (conv, norm, act_code) triples of the three ConvLayer2d blocks."""
    (c1, n1, a1), (cd, n2, a2), (c3, n3, _) = exp, dw, red
    training = n1.training or not n1.track_running_stats
    cfg = (int(stride), bool(use_res), bool(training), int(a1), int(a2), (n1.momentum, n2.momentum, n3.momentum), (n1.eps, n2.eps, n3.eps))
    # `_cvh_colsum`: column sums a previous fused block attached to ITS output tensor, with the tensor's version counter and address at that
    # moment: an in-place modification between the blocks (a user hook, .add_ / .mul_, in-place dropout or stochastic depth) bumps the
    # version and the analytic sums are dropped — the next block then forms its column sums from the data (gram_bn_stats)
    xsum = None
    tag = getattr(x, "_cvh_colsum", None)
    if tag is not None:
        osum0, ver0, ptr0 = tag
        if ver0 == x._version and ptr0 == x.data_ptr() and osum0.numel() == x.shape[1] and osum0.device == x.device:
            xsum = osum0
    gram_in = None
    if x.is_cuda and _dwx_eligible(x.dtype, x.shape[1], c1.weight, c1.weight.shape[0], int(stride), int(a1)):
        gram_in = ops.gram_of_input(x, x.shape[1], training)
    res = InvertedResidualFn.apply(x, c1.weight, n1.weight, n1.bias, n1.running_mean, n1.running_var, cd.weight, n2.weight, n2.bias,
                                   n2.running_mean, n2.running_var, c3.weight, n3.weight, n3.bias, n3.running_mean, n3.running_var, cfg, xsum,
                                   gram_in)
    if isinstance(res, tuple) and len(res) == 3:  # (out, None, G | s): the apply pass of the block output formed the next block's Gram matrix
        return ops.tag_producer((res[0], res[2]), n3.weight)
    if isinstance(res, tuple):
        out, osum = res
        out._cvh_colsum = (osum, out._version, out.data_ptr())
        return ops.tag_producer(out, n3.weight)
    return ops.tag_producer(res, n3.weight)
