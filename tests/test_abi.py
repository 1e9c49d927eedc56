"""CPU test: libcvnets_hip.so loads and exports every symbol declared in include/cvnets_hip.h, and the ctypes
prototypes in cvnets_amd/_lib.py agree with the header argument-for-argument (no compute calls — no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "cvnets_hip.h")

CT = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
      "unsigned int": ctypes.c_uint}


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|long long)\s+(cvh_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2)
        types = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a == "void":  # f(void)
                continue
            if "*" in a:
                types.append(ctypes.c_void_p)
            else:
                t = " ".join(a.split(" ")[:-1])
                types.append(CT[t])
        protos[name] = types
    return protos


def test_header_matches_ctypes_table():
    from cvnets_amd import _lib
    protos = header_prototypes()
    assert len(protos) >= 30
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, types in protos.items():
        assert types == _lib.SIGNATURES[name], name


def test_library_exports_every_declared_symbol():
    from cvnets_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        import sys
        subprocess.check_call([sys.executable, os.path.join(REPO, "ml-cvnets_amd", "build.py")])
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_prototypes():
        assert hasattr(lib, name), name
    # size-query entry points are host-only and safe to call without a GPU
    assert _lib.query("cvh_conv_gemm_grid_rows", 128 * 128 * 64, 64) == 512
    assert _lib.query("cvh_colreduce_rows", 1000, 144) > 0
    assert _lib.query("cvh_ln_bwd_rows", 1000) == 63
    with pytest.raises(RuntimeError):
        _lib.query("cvh_colreduce_rows", 1000, 20)  # C % 8 != 0 is rejected


def test_dw_split_planner_fills_whole_rounds():
    """host-side planning of the transformer-sized dW GEMM (csrc/gemm.hip tn_plan, no GPU needed): scratch = splits * N * K floats;
    the split count must make (output tiles x splits) land just under a multiple of the workgroup slots instead of just over.  Round 5: the
    direct-to-LDS kernel takes 256 x 256 tiles (one 8-wave workgroup per CU: 256 slots) wherever that moves fewer operand columns per row
    than 128 x 128 tiles (two workgroups per CU: 512 slots) and both dimensions are >= 512 (ViT-B / CLIP) — csrc/gemm_big.hip gemm_tn256_shape."""
    from cvnets_amd import _lib

    def tiling(N, K):
        c128 = -(-N // 128) * -(-K // 128) * 256
        c256 = -(-N // 256) * -(-K // 256) * 512
        return (-(-N // 256) * -(-K // 256), 256) if (c256 < c128 and min(N, K) >= 512) else (-(-N // 128) * -(-K // 128), 512)

    M = 128 * 197  # ViT-B tokens at batch 128
    for N, K in ((3072, 768), (768, 768), (2304, 768), (768, 3072), (144, 144), (432, 144), (576, 192), (720, 240), (96, 96)):
        tiles, slots = tiling(N, K)
        for rows in (M, 1 << 20):
            splits = _lib.query("cvh_gemm_dw_scratch_elems", rows, N, K) // (N * K)
            rounds = tiles * splits / float(slots)
            assert 1 <= splits <= 512 and (N * K * splits) == _lib.query("cvh_gemm_dw_scratch_elems", rows, N, K)
            assert rounds <= 1.0 or (rounds % 1.0) >= 0.85 or (rounds % 1.0) == 0.0, (N, K, rows, splits, rounds)
    assert tiling(144, 144) == (4, 512) and tiling(720, 240) == (12, 512) and tiling(512, 512) == (4, 256) and tiling(3072, 768) == (36, 256)
    # small conv-style problems keep the many-split plan (tall-skinny dW: one 128x128 tile, reduction over 2M pixels)
    assert _lib.query("cvh_gemm_dw_scratch_elems", 128 * 128 * 128, 128, 32) // (128 * 32) >= 256


def test_ops_refuse_to_run_without_gpu():
    """the product path has no CPU fallback: calling an op on CPU tensors raises instead of silently computing."""
    import torch
    from cvnets_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(8, 8), torch.zeros(8, 8))
    with pytest.raises(RuntimeError):
        ops.to_nhwc(torch.zeros(1, 3, 4, 4))
    with pytest.raises(RuntimeError):
        ops.cross_entropy(torch.zeros(4, 10), torch.zeros(4, dtype=torch.long), 0.1)
    from cvnets_amd.optim import AdamW
    w = torch.nn.Parameter(torch.zeros(8))
    w.grad = torch.zeros(8)
    with pytest.raises(RuntimeError):
        AdamW([w], lr=1e-3).step()
