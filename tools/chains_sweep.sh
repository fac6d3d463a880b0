#!/usr/bin/env bash
# bench.py env mode under every schedule (one chain / world ranges half a tick apart, replayed as hipGraphs / launched eagerly),
# alternating, on ONE box: gpurun_out/$TAG/chains_sweep.txt.   TAG=r05_b bash tools/chains_sweep.sh [REPS] [bench.py args ...]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
TAG="${TAG:-pass}"; O="$R/gpurun_out/$TAG"; mkdir -p "$O"
reps="${1:-2}"; [ $# -gt 0 ] && shift
line() {
  python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('chains %s %-6s steps %5d  value %7.2f M  tick %6.2f us  (kernels over all worlds: ray %5.2f + move %5.2f us)' % ('$1', '$2', d['steps'], d['value'] / 1e6,
      d['ms_per_step'] * 1e3, r['kernel_avg_us'], r['move_kernel_avg_us']))"
}
for _ in $(seq "$reps"); do
  for C in 1 2 3 4 8; do
    python bench.py --steps 2000 --warmup 200 --chains $C --schedule graph --no-cpu-baseline --no-extra "$@" 2>>"$O/chains_sweep.err" | line $C graph
    python bench.py --steps 2000 --warmup 200 --chains $C --no-graph --no-cpu-baseline --no-extra "$@" 2>>"$O/chains_sweep.err" | line $C eager
    python bench.py --steps 20 --warmup 5 --chains $C --no-cpu-baseline --no-extra "$@" 2>>"$O/chains_sweep.err" | line $C graph
  done
done 2>&1 | tee "$O/chains_sweep.txt"
