#!/usr/bin/env bash
# Round-2 GPU pass D: the training run with the stabilised large-batch recipe (floor on the policy's log std, KL early
# stop, gradient clipping): Stage-1 from scratch -> Stage-2 worlds mixed with circles of 10 / 20 / 30 / 50 robots;
# circle-test evaluation at every size.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/train_d; export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
S1="${S1_SECONDS:-360}"; S2="${S2_SECONDS:-660}"
W=/tmp/mrca_r02d; rm -rf $W; mkdir -p $W; cd $W
COMMON="--horizon 16 --batch-size 16384 --kl-target 0.01 --kl-stop 2.0 --lr 1.5e-4 --lr-max 3e-4 --max-grad-norm 1.0 --logstd-min -1.2 --save-every 100000 --log-every 25 --no-graph"
echo "== stage 1 ($S1 s)"
timeout $((S1+300)) python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates 1000000 --max-seconds $S1 --epoch 2 $COMMON 2>&1 \
    | grep -E "^(update|circle|per-rank|stopping|training mix|Traceback|.*Error)" > s1.log
awk 'NR<=3 || NR%6==0' s1.log | cut -c1-150 > "$R/gpurun_out/train_d/stage1_curve.txt"; tail -2 s1.log
cp policy/last.pth "$R/gpurun_out/train_d/stage1_last.pth"
echo "== stage 2 + circles ($S2 s)"
timeout $((S2+400)) python -m mrca.train --stage 2 --worlds 94 --mix-circles 10:8:40,20:12:30,30:16:20,50:25:20 --updates 1000000 --max-seconds $S2 \
    --init policy/last.pth --epoch 1 $COMMON --circle-every 1000 --circle-worlds 10 --circle-ticks 1500 2>&1 \
    | grep -E "^(update|circle|per-rank|stopping|training mix|Traceback|.*Error)" > s2.log
awk 'NR<=4 || NR%6==0 || /circle/' s2.log | cut -c1-150 > "$R/gpurun_out/train_d/stage2_curve.txt"; tail -3 s2.log; grep circle s2.log | tail -6
cp policy/last.pth "$R/gpurun_out/train_d/stage2_last.pth"; cp policy/best_circle.pth "$R/gpurun_out/train_d/best_circle.pth" 2>/dev/null
for SPEC in "10 8" "20 12" "30 16" "50 25"; do
  set -- $SPEC
  timeout 300 python -m mrca.evaluate --circles 100 --robots $1 --radius $2 --policy policy/last.pth --max-ticks 2000 2>/dev/null | tail -1 | tee -a "$R/gpurun_out/train_d/circle_eval.jsonl"
done
timeout 300 python -m mrca.evaluate --circles 1 --policy policy/last.pth --max-ticks 2000 2>/dev/null | tail -1 | tee -a "$R/gpurun_out/train_d/circle_eval.jsonl"
