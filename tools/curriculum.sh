#!/usr/bin/env bash
# GPU box: the reference's two-stage curriculum end to end (README.md:28-40 of the reference):
# Stage-1 from scratch -> Stage-2 fine-tune from that policy -> circle test with the result.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export PYTHONPATH="$R/rl-collision-avoidance_amd"
U1="${UPDATES1:-300}"; U2="${UPDATES2:-150}"
W=/tmp/mrca_curr; rm -rf $W; mkdir -p $W "$R/gpurun_out/curriculum"; cd $W
timeout 1200 python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates $U1 --save-every $U1 2>&1 | grep "^update" > s1.log
tail -1 s1.log
cp policy/Stage1_$U1 policy/stage2.pth
timeout 1500 python -m mrca.train --stage 2 --worlds 94 --updates $U2 --save-every $U2 2>&1 | grep "^update" > s2.log
tail -1 s2.log
awk 'NR%5==1 || NR<=3' s1.log > "$R/gpurun_out/curriculum/stage1_curve.txt"
awk 'NR%5==1 || NR<=3' s2.log > "$R/gpurun_out/curriculum/stage2_curve.txt"
P=$(ls policy/stage2_*.pth | grep -v state | tail -1); echo "policy: $P"
for C in 1 200; do
  timeout 600 python -m mrca.evaluate --circles $C --policy $P --max-ticks 1000 2>/dev/null | tail -1 | tee "$R/gpurun_out/curriculum/circle_stage2policy_${C}.json"
done
