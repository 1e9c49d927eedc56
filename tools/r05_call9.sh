cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O
timeout 600 python -m pytest tests/test_advice_gpu.py tests/test_kernels_gpu.py tests/test_rccl_gpu.py tests/test_launch_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | head -12
# bench.py's N > 1 code path on one GPU: single-rank world, collectives issued anyway, through the own communicator, inside the captured graph
(CVH_DDP_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>&1 | grep "^{" | tail -1) > $O/bench_force_collectives.json; python -c "
import json; d=json.load(open('$O/bench_force_collectives.json')); print(d['value'], d['ms_per_step'], {k: d['config'].get(k) for k in ('allreduce','communicator','allreduce_buckets','allreduce_buckets_started_inside_backward','hipgraph_error')})"
(timeout 1200 python tools/bench_models.py --models vit_base,vit_base_ckpt,clip,clip_ckpt,mobilevitv2,mobilevitv2_vbs --batch vit_base=512,clip=256,mobilevitv2=128 2>/dev/null | grep "^{") > $O/bench_models.jsonl; cut -c1-200 $O/bench_models.jsonl
(timeout 600 python tools/bench_models.py --models mobilevitv2_vbs --steps 200 --warmup 5 2>/dev/null | grep "^{") >> $O/bench_models.jsonl; tail -1 $O/bench_models.jsonl | cut -c1-300
