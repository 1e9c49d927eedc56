cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03m
(CVH_ASYNC_DW=1 timeout 900 python tools/stress_v2.py 300; CVH_ASYNC_DW=0 timeout 900 python tools/stress_v2.py 300) > gpurun_out/r03m/stress_v2.txt 2>&1; cat gpurun_out/r03m/stress_v2.txt | grep -v amdgpu.ids
timeout 1500 python tools/bench_models.py --models vit_base,vit_base_ckpt,clip,clip_ckpt,mobilevitv2,mobilevitv2_vbs --batch vit_base=512,clip=256,mobilevitv2=128 --steps 8 --warmup 2 > gpurun_out/r03m/bench_models.jsonl 2>gpurun_out/r03m/bench_models.err; cut -c1-330 gpurun_out/r03m/bench_models.jsonl
