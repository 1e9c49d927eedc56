cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_advice_gpu.py tests/test_bf16_parity_gpu.py -q -m gpu -s 2>&1 | grep -E "bf16 points|passed|failed|Error|assert|FAILED" | tee $O/tests.log
for b in 128 1024; do timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, d['ms_per_step'], d['value'])"; done | tee $O/b128.log
CVH_DDP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>&1 | tail -1 | cut -c1-1500 | tee $O/b128_rccl.json
