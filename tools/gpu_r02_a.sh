#!/usr/bin/env bash
# Round-2 GPU pass A: parity tests of the reworked kernels, launch-shape variants, a first env bench, the policy
# breakdown, then (only if the parity tests are green) a first run of the large-batch training recipe:
# Stage-1 from scratch -> Stage-2 from that policy with circle-test validation.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
S1="${S1_SECONDS:-330}"; S2="${S2_SECONDS:-480}"
nproc > gpurun_out/host.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== pytest parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s > gpurun_out/pytest_parity.log 2>&1; PAR=$?; echo "parity rc=$PAR"; tail -5 gpurun_out/pytest_parity.log
echo "== pytest rest"; timeout 1200 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_parity.py > gpurun_out/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -15 gpurun_out/pytest_rest.log
echo "== ablate"; timeout 600 python tools/ablate.py > gpurun_out/ablate.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/ablate.txt | head -60
echo "== bench env"; timeout 600 python bench.py --steps 1000 --warmup 100 > gpurun_out/bench_env.json 2> gpurun_out/bench.err; echo "rc=$?"; cut -c1-600 gpurun_out/bench_env.json
echo "== policy breakdown"; timeout 600 python tools/policy_breakdown.py > gpurun_out/policy_breakdown.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/policy_breakdown.txt
if [ "$PAR" = "0" ]; then
  W=/tmp/mrca_r02a; rm -rf $W; mkdir -p $W "$R/gpurun_out/train_a"; cd $W
  echo "== stage 1 ($S1 s)"
  timeout $((S1+240)) python -m mrca.train --stage 1 --worlds 128 --robots-per-world 32 --updates 100000 --max-seconds $S1 \
      --horizon 32 --epoch 2 --batch-size 16384 --kl-target 0.01 --lr 3e-4 --save-every 100000 2>&1 | grep -E "^(update|circle|per-rank|stopping)" > s1.log
  awk 'NR<=3 || NR%20==0' s1.log | cut -c1-150 > "$R/gpurun_out/train_a/stage1_curve.txt"; tail -2 s1.log
  cp policy/last.pth "$R/gpurun_out/train_a/stage1_last.pth"
  echo "== stage 2 ($S2 s)"
  timeout $((S2+300)) python -m mrca.train --stage 2 --worlds 94 --updates 100000 --max-seconds $S2 --init policy/last.pth \
      --horizon 32 --epoch 2 --batch-size 16384 --kl-target 0.01 --lr 3e-4 --save-every 100000 \
      --circle-every 100 --circle-worlds 20 --circle-ticks 1200 2>&1 | grep -E "^(update|circle|per-rank|stopping)" > s2.log
  awk 'NR<=3 || NR%20==0 || /circle/' s2.log | cut -c1-150 > "$R/gpurun_out/train_a/stage2_curve.txt"; tail -3 s2.log; grep circle s2.log | tail -5
  cp policy/last.pth "$R/gpurun_out/train_a/stage2_last.pth"; cp policy/best_circle.pth "$R/gpurun_out/train_a/best_circle.pth" 2>/dev/null
  for P in policy/last.pth policy/best_circle.pth; do
    for C in 1 200; do
      timeout 300 python -m mrca.evaluate --circles $C --policy $P --max-ticks 1500 2>/dev/null | tail -1 | tee -a "$R/gpurun_out/train_a/circle_eval.jsonl"
    done
  done
fi
