#!/usr/bin/env bash
# bench.py --mode rollout of THIS tree against a second checkout of the repo (OLD=<dir inside the repo>, sources + built
# libraries: `git archive <rev> | tar -x -C <dir>` and the .so files of that revision), alternating on one box.
#   TAG=r05_y OLD=_ab_old bash tools/rollout_ab.sh
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
TAG="${TAG:-rollout_ab}"; O="$R/gpurun_out/$TAG"; mkdir -p "$O"
OLD="${OLD:-_ab_old}"
export TMPDIR=/tmp
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.3f M  %.2f us/tick' % (d['value']/1e6, d['ms_per_step']*1e3))" "$1"; }
for rep in 1 2 3; do
  for side in old new; do
    if [[ $side == old ]]; then dir="$R/$OLD"; else dir="$R"; fi
    for mode in graph eager; do
      [[ $mode == eager && $rep != 1 ]] && continue
      flag=""; [[ $mode == eager ]] && flag="--no-graph"
      ( cd "$dir" && PYTHONPATH="$dir/rl-collision-avoidance_amd" timeout 300 python bench.py --mode rollout --steps 400 --warmup 40 --no-cpu-baseline $flag \
          > "$O/rollout_${side}_${mode}_${rep}.json" 2>> "$O/err.txt" )
      echo "$side $mode rep $rep: $(val "$O/rollout_${side}_${mode}_${rep}.json")"
    done
  done
done | tee "$O/rollout_ab.txt"
grep -v amdgpu.ids "$O/err.txt" | tail -5
