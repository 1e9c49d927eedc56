cd $GRAFT_REPO_ROOT
for d in "" _w0 "" _w0; do echo "== lib $d"; CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/libcvnets_hip$d.so python tools/bench_dwx.py --only new --reps 10 2>&1 | grep -v amdgpu | sed 's/dwx fwd.*bwd/bwd/' | cut -c1-70 | head -6; done
