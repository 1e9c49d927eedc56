cd $GRAFT_REPO_ROOT
for t in "" "2=1024" "2=256" "1=1"; do echo "== CVH_TUNE=$t"; CVH_TUNE=$t timeout 120 python tools/experiments/bench_tn.py 2>&1 | grep "^M="; done
