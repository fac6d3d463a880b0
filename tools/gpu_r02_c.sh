#!/usr/bin/env bash
# Round-2 GPU pass C: launch-shape variants, bench lines (env / rollout / train), rocprofv3 kernel trace + PMC traffic.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/c; export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
O="$R/gpurun_out/c"
echo "== pytest trainer"; timeout 600 python -m pytest tests/test_gpu_trainer.py -m gpu -q > $O/pytest_trainer.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_trainer.log
echo "== ablate"; timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids > $O/ablate.txt; echo "rc=$?"; cat $O/ablate.txt
echo "== bench env"; timeout 600 python bench.py --steps 1000 --warmup 100 > $O/bench_env.json 2> $O/bench.err; echo "rc=$?"; cut -c1-1500 $O/bench_env.json
echo "== bench rollout (fused, graph)"; timeout 600 python bench.py --mode rollout --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_rollout.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-900 $O/bench_rollout.json
echo "== bench rollout (fused, no graph)"; timeout 600 python bench.py --mode rollout --steps 200 --warmup 20 --no-cpu-baseline --no-graph > $O/bench_rollout_nograph.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_rollout_nograph.json
echo "== bench rollout (stock, no graph)"; timeout 600 python bench.py --mode rollout --steps 200 --warmup 20 --no-cpu-baseline --no-graph --policy-path stock > $O/bench_rollout_stock.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_rollout_stock.json
echo "== bench train"; timeout 900 python bench.py --mode train --steps 256 --warmup 0 --no-cpu-baseline > $O/bench_train.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-400 $O/bench_train.json
echo "== bench stage2"; timeout 600 python bench.py --scenario stage2 --worlds 187 --steps 500 --warmup 50 --no-cpu-baseline --no-extra > $O/bench_stage2.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-400 $O/bench_stage2.json
tail -5 $O/bench.err
echo "== big world"; timeout 600 python tools/bigworld_bench.py 500 5000 50000 2>&1 | grep -v amdgpu.ids > $O/bigworld.jsonl; cat $O/bigworld.jsonl
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o trace -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extra > "$O/prof_trace.log" 2>&1; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/prof" -o pmc_$C -- python "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extra > "$O/prof_pmc_$C.log" 2>&1; echo "pmc $C rc=$?"
done
cd "$R"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" && cp "$f" $O/env_kernel_stats.csv
python tools/pmc_summary.py $O/prof > $O/pmc_summary.txt 2>&1; grep mrca $O/pmc_summary.txt
python tools/pmc_summary.py $O/prof --traffic-json $O/pmc_traffic.json 4096 stage1
rm -rf $O/prof/*/*kernel_trace.csv $O/prof/*/*agent_info.csv 2>/dev/null
du -sh $O
