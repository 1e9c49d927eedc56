"""Guards a property of the COMPILED dW kernels that no numerical test can see: while a wave has direct-to-LDS loads
outstanding, hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS transpose read issued through the builtin, which turns
`request step st + 1 -> read + MFMA step st` into `request -> wait for it -> compute` (no load / compute overlap).  The kernels issue those reads from
inline asm (csrc/common.hpp: tr_frag_raw / tr_wait); this test compiles the two files to gfx950 assembly and checks that no transpose read of the
dW kernels has such a drain in front of it.  CPU only (hipcc cross-compiles); skipped where hipcc is missing."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "ml-cvnets_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


def _kernels(src, tmp_path, extra=()):
    out = str(tmp_path / (src + ".s"))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", f"-I{CSRC}", "-S", "--cuda-device-only", *extra,
                    os.path.join(CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    res = {}
    for body in re.split(r"\n(?=_Z[\w]+:)", text):
        name = body.split(":")[0]
        if name.startswith("_Z"):
            res[name] = [l.strip() for l in body.split("s_endpgm")[0].splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    return res


def _drained_reads(lines):
    """transpose reads with an `s_waitcnt vmcnt(0)` among the three instructions in front of them"""
    bad = 0
    for i, l in enumerate(lines):
        if l.startswith("ds_read_b64_tr_b16") and not lines[i - 1].startswith("ds_read_b64_tr_b16"):
            if any(x.startswith("s_waitcnt") and "vmcnt(0)" in x for x in lines[max(0, i - 3):i]):
                bad += 1
    return bad


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
@pytest.mark.parametrize("src,pattern", [("gemm_big.hip", "gemm_tn128_kernel"), ("gemm_big.hip", "gemm_tn256_kernel"), ("gemm_rows.hip", "gemm_tn_rows_kernel")])
def test_no_drain_in_front_of_transpose_reads(src, pattern, tmp_path):
    # (the 4 x 4 rectangle of gemm_tn_rows keeps the builtin reads: the asm form spills there - csrc/gemm_rows.hip)
    ks = {n: l for n, l in _kernels(src, tmp_path).items() if pattern in n and "gemm_tn_rows_kernelILi4ELi4E" not in n}
    assert ks, f"no kernel matching {pattern} in {src}"
    for name, lines in ks.items():
        reads = sum(l.startswith("ds_read_b64_tr_b16") for l in lines)
        assert reads > 0, name
        assert any(l.startswith("global_load_lds") for l in lines), name  # the kernels this is about keep direct-to-LDS loads in flight
        assert _drained_reads(lines) == 0, (name, _drained_reads(lines))


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
def test_the_builtin_form_does_draw_the_drain(tmp_path):
    """the reason the asm form exists: built with -DCVH_TR_ASM=0 (the builtin) the same kernel has the wait in front of its reads"""
    ks = {n: l for n, l in _kernels("gemm_big.hip", tmp_path, extra=("-DCVH_TR_ASM=0",)).items() if "gemm_tn256_kernel" in n}
    assert ks and all(_drained_reads(l) > 0 for l in ks.values())
