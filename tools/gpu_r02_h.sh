#!/usr/bin/env bash
# Round-2 GPU pass H: continue run D's recipe from the committed checkpoint with circles of every size in the mix and
# validate on circles of 20 / 30 / 40 / 50 robots at once (score = the minimum success rate).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/train_h; export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
S2="${S2_SECONDS:-540}"
W=/tmp/mrca_r02h; rm -rf $W; mkdir -p $W; cd $W
COMMON="--horizon 16 --batch-size 16384 --kl-target 0.01 --kl-stop 2.0 --lr 1.5e-4 --lr-max 3e-4 --max-grad-norm 1.0 --logstd-min -1.2 --save-every 100000 --log-every 25"
timeout $((S2+300)) python -m mrca.train --stage 2 --worlds 94 --mix-circles 10:8:20,20:12:20,30:16:30,40:20:20,50:25:20 --updates 1000000 --max-seconds $S2 \
    --init "$R/rl-collision-avoidance_amd/mrca/data/policy_r02_stage2_circles.pth" --epoch 1 $COMMON \
    --circle-every 250 --circle-worlds 1 --circle-ticks 1500 --circle-sizes 20:12,30:16,40:20,50:25 2>&1 \
    | grep -E "^(update|circle|per-rank|stopping|training mix|Traceback|.*Error)" > s2.log
awk 'NR<=4 || NR%8==0 || /circle/' s2.log | cut -c1-150 > "$R/gpurun_out/train_h/stage2_curve.txt"; tail -2 s2.log; grep "new best" s2.log | tail -5
cp policy/last.pth "$R/gpurun_out/train_h/last.pth"; cp policy/best_circle.pth "$R/gpurun_out/train_h/best_circle.pth" 2>/dev/null
for SPEC in "10 8" "20 12" "30 16" "40 20" "50 25"; do
  set -- $SPEC
  timeout 300 python -m mrca.evaluate --circles 20 --robots $1 --radius $2 --policy policy/best_circle.pth --max-ticks 2000 2>/dev/null | tail -1 | tee -a "$R/gpurun_out/train_h/circle_eval_best.jsonl" | cut -c1-330
done
