#!/usr/bin/env bash
# Round-2 final check of the committed state: smoke, every GPU test, the default bench line.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; export TMPDIR=/tmp; export PYTHONPATH="$R/rl-collision-avoidance_amd"
O="$R/gpurun_out/final"; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest_gpu.log | grep -E "passed|failed|robot circles|trained checkpoint" | cut -c1-250
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-2500 $O/bench_default.json
