"""Ordered launch list of the last full training step in a rocprofv3 kernel trace: one line per launch with its start offset in
the step, duration, grid, registers, LDS and stream — the table the per-layer kernel work is planned from.

usage: python tools/step_trace.py <rocprof dir> [step marker kernel substring, default stem_fwd_kernel]"""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:56]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    marker = sys.argv[2] if len(sys.argv) > 2 else "stem_fwd_kernel"
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    step = rows[idx[-2]:idx[-1]]
    t0 = int(step[0]["Start_Timestamp"])
    streams = {}
    print(f"{len(step)} launches, window {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms")
    # rocprofv3's VGPR_Count / Accum_VGPR_Count columns are HALF the per-lane allocation the compiler reports (`.vgpr_count` of the code
    # object rounded up to the 8-register granule: dwf_bwd_kernel 244 -> 248 -> trace 124; gemm_nt256_kernel 189 -> 192 -> trace 96); the
    # columns below are the trace values x 2 = allocated registers per lane, the number occupancy follows from (512 / alloc waves per SIMD)
    print(f"{'#':>4s} {'start ms':>9s} {'us':>8s} {'grid':>12s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scr':>4s} q kernel")
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = max(int(r.get("Workgroup_Size_X", 1) or 1), 1)
        gx = int(r.get("Grid_Size_X", 0) or 0) // wg
        gy = r.get("Grid_Size_Y", "1")
        q = streams.setdefault(r.get("Queue_Id", r.get("Stream_Id", "0")), len(streams))
        vg, ag = (str(2 * int(r[k])) if (r.get(k) or "").isdigit() else "" for k in ("VGPR_Count", "Accum_VGPR_Count"))
        print(f"{i:4d} {(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {f'{gx}x{gy}':>12s} {vg:>5s} {ag:>5s} "
              f"{r.get('SGPR_Count', ''):>5s} {r.get('LDS_Block_Size', ''):>7s} {r.get('Scratch_Size', ''):>4s} {q} {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
