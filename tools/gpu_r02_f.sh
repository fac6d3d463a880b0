#!/usr/bin/env bash
# Round-2 GPU pass F: re-validation after a kernel change (parity + big worlds + policy ops), launch shapes, env bench,
# kernel trace, PMC traffic (bench.py refuses counters measured on other kernel sources), SQ counters.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; export TMPDIR=/tmp; export PYTHONPATH="$R/rl-collision-avoidance_amd"
O="$R/gpurun_out/${TAG:-f}"; mkdir -p $O
echo "== pytest parity/bigworld/circle"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigworld.py tests/test_gpu_circle.py -m gpu -q > $O/pytest_parity.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pytest_parity.log | tail -4 | cut -c1-300
echo "== ablate"; timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids > $O/ablate.txt; echo "rc=$?"; grep -E "flags=(0|1|2|3|56|512) " $O/ablate.txt
echo "== bench env"; timeout 600 python bench.py --steps 1000 --warmup 100 > $O/bench_env.json 2> $O/bench.err; echo "rc=$?"; cut -c1-400 $O/bench_env.json
echo "== bench stage2"; timeout 600 python bench.py --scenario stage2 --worlds 187 --steps 500 --warmup 50 --no-cpu-baseline --no-extra > $O/bench_stage2.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench_stage2.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o trace -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extra > "$O/prof_trace.log" 2>&1; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/prof" -o pmc_$C -- python "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extra > "$O/prof_pmc_$C.log" 2>&1; echo "pmc $C rc=$?"
done
cd "$R"
f=$(find $O/prof -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -3 "$f" | cut -c1-200 && cp "$f" $O/env_kernel_stats.csv
python tools/pmc_summary.py $O/prof > $O/pmc_summary.txt 2>&1
python tools/pmc_summary.py $O/prof --traffic-json $O/pmc_traffic.json 4096 stage1 | cut -c1-300
rm -rf $O/prof
TAG=sq_${TAG:-f} timeout 900 bash tools/pmc_profile.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq_${TAG:-f}/summary.txt $O/pmc_sq_summary.txt 2>/dev/null; grep -E "SQ_INSTS_VALU|SQ_WAVES |SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY" $O/pmc_sq_summary.txt | cut -c1-160
