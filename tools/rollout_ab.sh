#!/usr/bin/env bash
# bench.py --mode rollout | train of THIS tree against a second checkout of the repo (OLD=<dir inside the repo>, sources +
# built libraries: `git archive <rev> | tar -x -C <dir>` and the .so files of that revision), alternating on one box.
#   TAG=r05_y OLD=_ab_old [MODE=train STEPS=256 WARMUP=0 REPS=3] bash tools/rollout_ab.sh
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
TAG="${TAG:-rollout_ab}"; O="$R/gpurun_out/$TAG"; mkdir -p "$O"
OLD="${OLD:-_ab_old}"; MODE="${MODE:-rollout}"; STEPS="${STEPS:-400}"; WARMUP="${WARMUP:-40}"; REPS="${REPS:-3}"
export TMPDIR=/tmp
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.3f M  %.2f us/tick' % (d['value']/1e6, d['ms_per_step']*1e3))" "$1"; }
for rep in $(seq 1 "$REPS"); do
  for side in old new; do
    if [[ $side == old ]]; then dir="$R/$OLD"; else dir="$R"; fi
    for how in graph eager; do
      [[ $how == eager && ( $rep != 1 || "${EAGER:-1}" == 0 ) ]] && continue
      flag=""; [[ $how == eager ]] && flag="--no-graph"
      ( cd "$dir" && PYTHONPATH="$dir/rl-collision-avoidance_amd" timeout 600 python bench.py --mode "$MODE" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline $flag \
          > "$O/${MODE}_${side}_${how}_${rep}.json" 2>> "$O/err.txt" )
      echo "$MODE $side $how rep $rep: $(val "$O/${MODE}_${side}_${how}_${rep}.json")"
    done
  done
done | tee "$O/${MODE}_ab.txt"
grep -v amdgpu.ids "$O/err.txt" | tail -5
