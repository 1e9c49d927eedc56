cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-kernel-probe > /tmp/b.log 2>&1 &
BP=$!
for i in $(seq 1 60); do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/GPU\[0\]\s*: //' | tr '\n' ' '; echo; if ! kill -0 $BP 2>/dev/null; then break; fi; done
wait $BP
tail -1 /tmp/b.log | cut -c1-200
