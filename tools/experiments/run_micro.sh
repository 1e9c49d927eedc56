cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03info
timeout 300 python tools/experiments/bench_fx.py > gpurun_out/r03info/bench_fx.txt 2>&1
timeout 300 python tools/experiments/bench_dwf.py 1024 > gpurun_out/r03info/bench_dwf.txt 2>&1
