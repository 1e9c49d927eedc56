# Round evidence in ONE gpurun call: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats + per-step timeline of the bench
# command, the two PMC traffic passes, profiles/step_profile.json (kernel table + traffic, stamped with the source hash), bench line again
# (now quoting the fresh profile).      usage: bash tools/run_evidence.sh r03x
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests_gpu_full.log 2>&1; grep -E "^FAILED|^ERROR| passed| failed|Memory access" $O/tests_gpu_full.log > $O/tests_gpu.log; rm -f $O/tests_gpu_full.log; cat $O/tests_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep "smoke") > $O/smoke.log; cat $O/smoke.log
rm -rf $O/prof; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-probe > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-160
python tools/prof_summary.py $O/prof 60 40 > $O/prof_summary.txt 2>&1; python tools/timeline.py $O/prof > $O/timeline.txt 2>&1; head -3 $O/timeline.txt
python tools/step_trace.py $O/prof > $O/step_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-kernel-probe --no-cpu-baseline > $O/pmc_$c.log 2>&1; done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_summary.txt 2>&1; head -3 $O/pmc_summary.txt
python tools/make_step_profile.py $O/prof $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 1024 $TAG > $O/step_profile.json 2>$O/step_profile.err; head -12 $O/step_profile.json
cp $O/step_profile.json profiles/step_profile.json   # so that the bench line below quotes THIS build's profile (copied back into the repo by hand afterwards)
(timeout 900 python bench.py 2>&1 | tail -1) > $O/bench.json; cut -c1-600 $O/bench.json
cp $O/prof/bench_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; du -sh $O
# extra lines of the round: the 128-image (8-GPU strong-scaling shard) step on one GPU, the engine-driven eager loop next to the replayed step
(timeout 300 python bench.py --batch 128 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1) > $O/bench_b128.json; cut -c1-200 $O/bench_b128.json
(timeout 600 python tools/bench_engine.py --batch 128,1024 --graph-compare 2>/dev/null | grep "^{") > $O/bench_engine.jsonl; cut -c1-300 $O/bench_engine.jsonl
# derived utilisation counters per kernel family (one more pass of the bench command; counters only, no other trace domain)
rm -rf $O/pmc_util; timeout 900 rocprofv3 --kernel-trace --pmc VALUBusy MfmaUtil LdsUtil MeanOccupancyPerCU --output-format csv -d $O/pmc_util -o pmc -- python bench.py --steps 2 --warmup 1 --no-kernel-probe --no-cpu-baseline > $O/pmc_util.log 2>&1
python tools/pmc_util.py $O/pmc_util > $O/pmc_util.txt 2>&1; head -14 $O/pmc_util.txt | cut -c1-160
# LDS bank-conflict share of the depthwise pair and the streaming GEMM (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
rm -rf $O/pmc_lds; timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_lds -o pmc -- python bench.py --steps 2 --warmup 1 --no-kernel-probe --no-cpu-baseline > $O/pmc_lds.log 2>&1
python tools/pmc_util.py $O/pmc_lds > $O/pmc_lds.txt 2>&1; head -10 $O/pmc_lds.txt | cut -c1-160
# bench.py's N > 1 code path on one GPU (single-rank world, collectives issued anyway through the own communicator, inside the captured graph)
(CVH_DDP_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-probe 2>&1 | grep "^{" | tail -1) > $O/bench_force_collectives.json; cut -c1-200 $O/bench_force_collectives.json
find $O -name "*counter_collection.csv" -size +8M -delete; du -sh $O
# BASELINE configs 3-5 (ViT-B/16, CLIP ViT-B/16, MobileViTv2-1.0 at 384 x 384): the JSON lines and, per model, a kernel trace + the two traffic passes
(timeout 900 python tools/bench_models.py --models vit_base,clip,mobilevitv2 --batch vit_base=512,clip=256,mobilevitv2=128 --steps 10 --warmup 3 2>/dev/null | grep "^{") > $O/bench_models.jsonl; cut -c1-260 $O/bench_models.jsonl
for m in vit_base:512 clip:256 mobilevitv2:128; do
  M=${m%%:*}; B=${m##*:}
  rm -rf $O/mprof_$M; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mprof_$M -o bench -- python tools/bench_models.py --models $M --batch $M=$B --steps 4 --warmup 2 > $O/mprof_$M.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/mpmc_${M}_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/mpmc_${M}_$c -o pmc -- python tools/bench_models.py --models $M --batch $M=$B --steps 2 --warmup 1 > $O/mpmc_${M}_$c.log 2>&1; done
  python tools/model_prof_summary.py $M $B $O/mprof_$M $O/mpmc_${M}_FETCH_SIZE $O/mpmc_${M}_WRITE_SIZE > $O/${M}_prof_summary.txt 2>&1; head -8 $O/${M}_prof_summary.txt | cut -c1-200
  cp $O/mprof_$M/bench_kernel_stats.csv $O/${M}_kernel_stats.csv 2>/dev/null
done
find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; du -sh $O
