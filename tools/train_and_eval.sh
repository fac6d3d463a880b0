#!/usr/bin/env bash
# GPU box: train Stage-1 from scratch for a few minutes, keep the learning curve, then run the circle test
# (50 robots and 1000 circles = 50 000 robots) with the trained policy and with the go-to-goal stand-in.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export PYTHONPATH="$R/rl-collision-avoidance_amd"
UPD="${UPDATES:-300}"
W=/tmp/mrca_train; rm -rf $W; mkdir -p $W "$R/gpurun_out/train"; cd $W
timeout 1500 python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates $UPD --save-every $UPD 2>&1 | grep -v amdgpu.ids > train_stdout.log
grep "^update" train_stdout.log | awk 'NR%5==1 || NR<=3' > "$R/gpurun_out/train/stage1_curve.txt"; tail -3 train_stdout.log
P=$W/policy/Stage1_$UPD
ls -la $W/policy | head -5
for C in 1 1000; do
  timeout 600 python -m mrca.evaluate --circles $C --policy $P --max-ticks 900 2>/dev/null | tail -1 | tee "$R/gpurun_out/train/circle_trained_${C}.json"
done
timeout 300 python -m mrca.evaluate --circles 1000 --max-ticks 900 2>/dev/null | tail -1 | tee "$R/gpurun_out/train/circle_standin_1000.json"
tail -2 $W/log/*/ppo.log > "$R/gpurun_out/train/ppo_log_tail.txt"
