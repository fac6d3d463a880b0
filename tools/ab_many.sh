#!/usr/bin/env bash
# Many short alternations of the committed library against variant libraries (tools/ab_variant.sh build ...), replayed schedule
# only: for effects of a per cent or two, below the run-to-run spread of three alternations.
#   ON THE BOX:  bash tools/ab_many.sh REPS NAME [NAME ...]      -> gpurun_out/ab_many.txt (median / min tick per library)
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
reps="$1"; shift
mkdir -p gpurun_out; : > gpurun_out/ab_many_raw.txt
for i in $(seq "$reps"); do
  for which in base "$@"; do
    if [ "$which" = base ]; then unset MRCA_ENV_LIB; else export MRCA_ENV_LIB="$R/tools/_build/$which/rl-collision-avoidance_amd/mrca/libmrca_env.so"; fi
    python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$which', d['ms_per_step'] * 1e3, r['kernel_avg_us'], r['move_kernel_avg_us'])" >> gpurun_out/ab_many_raw.txt
  done
done
python - <<'PY' | tee gpurun_out/ab_many.txt
import statistics as st, collections
rows = collections.defaultdict(list)
for line in open('gpurun_out/ab_many_raw.txt'):
    w, t, r, m = line.split(); rows[w].append((float(t), float(r), float(m)))
for w, v in rows.items():
    t = [x[0] for x in v]; r = [x[1] for x in v]
    print('%-34s n %2d  tick median %6.2f min %6.2f mean %6.2f us   ray median %6.2f us   -> %6.1f M at the median' % (w, len(v), st.median(t), min(t), st.mean(t), st.median(r), 4096 / st.median(t)))
PY
