cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03d/t.log 2>&1; tail -3 gpurun_out/r03d/t.log
bash tools/ab.sh r03d libcvnets_hip_base.so libcvnets_hip.so
