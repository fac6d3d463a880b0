#!/usr/bin/env bash
# Round-2 GPU pass B: unit tests of what changed since pass A (fused policy front end, graph-captured tick, one-frame
# buffer, frame-stack shift riding behind the move kernel, big worlds), then the training run:
#   Stage-1 from scratch -> Stage-2 worlds mixed with circle worlds, circle-test validation, best checkpoint kept.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/train_b; export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
S1="${S1_SECONDS:-360}"; S2="${S2_SECONDS:-1200}"
echo "== pytest policy ops"; timeout 600 python -m pytest tests/test_gpu_policy_ops.py -m gpu -x -q > gpurun_out/pytest_policy_ops.log 2>&1; FUSED=$?; echo "rc=$FUSED"; tail -4 gpurun_out/pytest_policy_ops.log
echo "== pytest trainer"; timeout 900 python -m pytest tests/test_gpu_trainer.py -m gpu -q > gpurun_out/pytest_trainer.log 2>&1; GRAPH=$?; echo "rc=$GRAPH"; tail -6 gpurun_out/pytest_trainer.log
echo "== pytest parity + big worlds"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigworld.py -m gpu -q > gpurun_out/pytest_parity_big.log 2>&1; PAR=$?; echo "rc=$PAR"; tail -8 gpurun_out/pytest_parity_big.log
if [ "$PAR" != "0" ]; then
  # which half failed?  the training below needs the small-world path only
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_parity_only.log 2>&1; PAR=$?; echo "parity only rc=$PAR"; tail -4 gpurun_out/pytest_parity_only.log
fi
FLAGS=""
[ "$FUSED" != "0" ] && FLAGS="$FLAGS --stock-policy-path"
[ "$GRAPH" != "0" ] && FLAGS="$FLAGS --no-graph"
echo "train flags: '$FLAGS'"
if [ "$PAR" = "0" ]; then
  W=/tmp/mrca_r02b; rm -rf $W; mkdir -p $W; cd $W
  COMMON="--horizon 16 --batch-size 16384 --kl-target 0.01 --kl-stop 2.0 --lr 2e-4 --lr-max 5e-4 --max-grad-norm 1.0 --save-every 2000 --log-every 25 $FLAGS"
  echo "== stage 1 ($S1 s)"
  timeout $((S1+300)) python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates 1000000 --max-seconds $S1 --epoch 2 $COMMON 2>&1 \
      | grep -E "^(update|circle|per-rank|stopping|training mix|Traceback|.*Error)" > s1.log
  awk 'NR<=3 || NR%8==0' s1.log | cut -c1-150 > "$R/gpurun_out/train_b/stage1_curve.txt"; tail -2 s1.log
  cp policy/last.pth "$R/gpurun_out/train_b/stage1_last.pth"
  echo "== stage 2 + circle mix ($S2 s)"
  timeout $((S2+400)) python -m mrca.train --stage 2 --worlds 94 --mix-circle 40 --updates 1000000 --max-seconds $S2 --init policy/last.pth --epoch 1 $COMMON \
      --circle-every 500 --circle-worlds 10 --circle-ticks 1500 2>&1 | grep -E "^(update|circle|per-rank|stopping|training mix|Traceback|.*Error)" > s2.log
  awk 'NR<=4 || NR%8==0 || /circle/' s2.log | cut -c1-150 > "$R/gpurun_out/train_b/stage2_curve.txt"; tail -3 s2.log; grep circle s2.log | tail -8
  cp policy/last.pth "$R/gpurun_out/train_b/stage2_last.pth"; cp policy/best_circle.pth "$R/gpurun_out/train_b/best_circle.pth" 2>/dev/null
  for P in policy/best_circle.pth policy/last.pth; do
    for C in 1 200; do
      timeout 300 python -m mrca.evaluate --circles $C --policy $P --max-ticks 2000 2>/dev/null | tail -1 | tee -a "$R/gpurun_out/train_b/circle_eval.jsonl"
    done
  done
fi
