cd $GRAFT_REPO_ROOT
for lib in lib_f4.so lib_f8.so lib_f4.so lib_f8.so; do echo $lib; CVNETS_HIP_LIB=$GRAFT_REPO_ROOT/ml-cvnets_amd/lib/$lib timeout 300 python tools/bench_dwx.py --only new --reps 10 2>&1 | grep -v amdgpu.ids | grep "s2" | cut -c1-100; done
bash tools/ab.sh abf lib_f4.so lib_f8.so
