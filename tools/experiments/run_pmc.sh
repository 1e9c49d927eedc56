cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=${1:-1024}
for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-probe --batch $B > gpurun_out/pmc_$c.log 2>&1; done
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary_b$B.txt 2>&1; cat gpurun_out/pmc_summary_b$B.txt
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete
