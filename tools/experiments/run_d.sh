cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r03d/t.log 2>&1; tail -3 gpurun_out/r03d/t.log
bash tools/experiments/run_prof.sh r03p4
