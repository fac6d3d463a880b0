#!/usr/bin/env bash
# One GPU-box pass: smoke, GPU parity tests, bench (env / rollout / train), rocprofv3 kernel trace
# and PMC passes.  Logs -> gpurun_out/ (copy what should be judged into profiles/).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
TAG="${TAG:-r01}"
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; rocm-smi --showproductname 2>/dev/null | head -20 >> gpurun_out/host.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
fi
echo "== bench env"; timeout 600 python bench.py ${BENCH_ARGS:---steps 1000 --warmup 100} > gpurun_out/bench_env.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench_env.json; tail -3 gpurun_out/bench.err
if [ "${SKIP_MODES:-0}" != "1" ]; then
  echo "== bench rollout"; timeout 600 python bench.py --mode rollout --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_rollout.json 2>> gpurun_out/bench.err; echo "rc=$?"; cat gpurun_out/bench_rollout.json
  echo "== bench train"; timeout 900 python bench.py --mode train --steps 256 --warmup 0 --no-cpu-baseline > gpurun_out/bench_train.json 2>> gpurun_out/bench.err; echo "rc=$?"; cat gpurun_out/bench_train.json
  echo "== bench stage2"; timeout 600 python bench.py --scenario stage2 --worlds 187 --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/bench_stage2.json 2>> gpurun_out/bench.err; echo "rc=$?"; cat gpurun_out/bench_stage2.json
  tail -5 gpurun_out/bench.err
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3 kernel trace"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$TAG" -o trace -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extra > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/gpurun_out/prof_$TAG" -o pmc_$C -- python "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extra > "$R/gpurun_out/prof_pmc_$C.log" 2>&1; echo "pmc $C rc=$?"
  done
  cd "$R"
  find gpurun_out/prof_$TAG -type f | head -20
  f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f"
  python tools/pmc_summary.py gpurun_out/prof_$TAG > gpurun_out/pmc_summary_$TAG.txt 2>&1; cat gpurun_out/pmc_summary_$TAG.txt
fi
