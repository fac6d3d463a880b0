#!/usr/bin/env bash
# One GPU-box pass: smoke, GPU parity tests, bench, rocprofv3 kernel trace.  Logs -> gpurun_out/.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; rocm-smi --showproductname 2>/dev/null | head -20 >> gpurun_out/host.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py ${BENCH_ARGS:---steps 500 --warmup 50} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o r1 -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline > "$R/gpurun_out/prof_bench.log" 2>&1; echo "prof rc=$?"
  cd "$R"
  find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -12 "$f"
fi
