#!/usr/bin/env bash
# Round-2 GPU pass E: full validation of the state to be judged -- smoke, every GPU test, launch shapes, bench lines,
# rocprofv3 kernel trace + PMC traffic + SQ counters, exhaustive reciprocal check, circle test of the committed checkpoint.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/e; export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
O="$R/gpurun_out/e"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?"; tail -1 $O/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pytest_gpu.log | tail -25 | cut -c1-400
echo "== ablate"; timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids > $O/ablate.txt; echo "rc=$?"; cat $O/ablate.txt
echo "== bench env"; timeout 600 python bench.py --steps 1000 --warmup 100 > $O/bench_env.json 2> $O/bench.err; echo "rc=$?"; cut -c1-1800 $O/bench_env.json
echo "== bench rollout"; timeout 600 python bench.py --mode rollout --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_rollout.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_rollout.json
echo "== bench rollout no graph"; timeout 600 python bench.py --mode rollout --steps 400 --warmup 40 --no-cpu-baseline --no-graph > $O/bench_rollout_nograph.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_rollout_nograph.json
echo "== bench train"; timeout 900 python bench.py --mode train --steps 256 --warmup 0 --no-cpu-baseline > $O/bench_train.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_train.json
echo "== bench stage2"; timeout 600 python bench.py --scenario stage2 --worlds 187 --steps 500 --warmup 50 --no-cpu-baseline --no-extra > $O/bench_stage2.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_stage2.json
grep -v amdgpu.ids $O/bench.err | tail -5
echo "== circle test, committed checkpoint"
P="$R/rl-collision-avoidance_amd/mrca/data/policy_r02_stage2_circles.pth"
for C in 1 200 1000; do timeout 300 python -m mrca.evaluate --circles $C --policy $P --max-ticks 2000 2>/dev/null | tail -1 | tee -a $O/circle_eval.jsonl | cut -c1-400; done
for SPEC in "10 8" "20 12" "30 16" "40 20"; do set -- $SPEC; timeout 300 python -m mrca.evaluate --circles 20 --robots $1 --radius $2 --policy $P --max-ticks 2000 2>/dev/null | tail -1 | tee -a $O/circle_eval.jsonl | cut -c1-300; done
echo "== reciprocal check"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/check_rcp.hip -o /tmp/check_rcp 2>/dev/null && timeout 300 /tmp/check_rcp | tee $O/check_rcp.txt
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o trace -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extra > "$O/prof_trace.log" 2>&1; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/prof" -o pmc_$C -- python "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extra > "$O/prof_pmc_$C.log" 2>&1; echo "pmc $C rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_rollout" -o trace -- python "$R/bench.py" --mode rollout --steps 100 --warmup 20 --no-cpu-baseline --no-graph > "$O/prof_rollout.log" 2>&1; echo "rollout trace rc=$?"
cd "$R"
f=$(find $O/prof -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-200 && cp "$f" $O/env_kernel_stats.csv
f=$(find $O/prof_rollout -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220 && cp "$f" $O/rollout_kernel_stats.csv
python tools/pmc_summary.py $O/prof > $O/pmc_summary.txt 2>&1; grep mrca $O/pmc_summary.txt | cut -c1-200
python tools/pmc_summary.py $O/prof --traffic-json $O/pmc_traffic.json 4096 stage1 | cut -c1-400
rm -rf $O/prof $O/prof_rollout
echo "== SQ counters"; TAG=r02e timeout 900 bash tools/pmc_profile.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_r02e/summary.txt $O/pmc_sq_summary.txt 2>/dev/null; grep -E "raycast|move_kernel" $O/pmc_sq_summary.txt | head -60 | cut -c1-200
du -sh $O
