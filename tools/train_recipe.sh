#!/usr/bin/env bash
# GPU box: the reference's two-stage curriculum (README.md:28-40 of the reference) with the large-batch recipe of
# DESIGN.md 8 -- Stage-1 from scratch for S1_SECONDS, then Stage-2 worlds mixed with circles of 10-50 robots for
# S2_SECONDS -- the PPO update running through the HIP forward / backward kernels of the conv front end
# (--update-path fused, the default; UPDATE_PATH=stock for MIOpen), validation on PERTURBED circles with a held-out seed,
# then the perturbed circle test of the result on every circle size.   usage: [SEED=k] tools/train_recipe.sh <out-dir>
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="${1:-$R/gpurun_out/train}"
mkdir -p "$O"
export PYTHONPATH="$R/rl-collision-avoidance_amd"
S1="${S1_SECONDS:-300}"; S2="${S2_SECONDS:-300}"
W=/tmp/mrca_train_recipe; rm -rf $W; mkdir -p $W; cd $W
SEED="${SEED:-0}"
COMMON="--seed $SEED --horizon 16 --batch-size 16384 --kl-target 0.01 --kl-stop 2.0 --lr 1.5e-4 --lr-max 3e-4 --max-grad-norm 1.0 --logstd-min -1.2 --save-every 100000 --log-every 25 --update-path ${UPDATE_PATH:-fused}"
timeout $((S1+240)) python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates 1000000 --max-seconds $S1 --epoch 2 $COMMON 2>&1 \
    | grep -E "^(update|stopping|per-rank|Traceback|.*Error)" > s1.log
awk 'NR<=3 || NR%6==0' s1.log | cut -c1-150 > "$O/stage1_curve.txt"; tail -1 s1.log | cut -c1-150
cp policy/last.pth policy/stage2.pth
timeout $((S2+300)) python -m mrca.train --stage 2 --worlds 94 --mix-circles 10:8:20,20:12:20,30:16:30,40:20:20,50:25:20 --updates 1000000 --max-seconds $S2 \
    --epoch 1 $COMMON --circle-every 250 --circle-worlds 8 --circle-ticks 1500 --circle-sizes 20:12,30:16,40:20,50:25 --circle-perturb 0.2,0.1 2>&1 \
    | grep -E "^(update|circle|stopping|training mix|per-rank|Traceback|.*Error)" > s2.log
awk 'NR<=4 || NR%8==0 || /circle/' s2.log | cut -c1-150 > "$O/stage2_curve.txt"; tail -2 s2.log | cut -c1-150; grep "new best" s2.log | tail -3
cp policy/last.pth "$O/last.pth"; cp policy/best_circle.pth "$O/best_circle.pth" 2>/dev/null
for P in last best_circle; do
  [ -f policy/$P.pth ] || continue
  for SPEC in "10 8" "20 12" "30 16" "40 20" "50 25"; do set -- $SPEC
    timeout 300 python -m mrca.evaluate --circles 100 --robots $1 --radius $2 --policy policy/$P.pth --max-ticks 2000 --perturb 0.2,0.1 --seed 1 --fused 2>/dev/null \
      | tail -1 | tee -a "$O/circle_eval_$P.jsonl" | cut -c1-200
  done
done
