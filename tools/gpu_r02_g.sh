#!/usr/bin/env bash
# quick pass: launch-shape parity + ablation table
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; export TMPDIR=/tmp; export PYTHONPATH="$R/rl-collision-avoidance_amd"
O="$R/gpurun_out/${TAG:-g}"; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "launch_shapes or stage1_bit_exact or full_batch" > $O/pytest_shapes.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pytest_shapes.log | tail -4 | cut -c1-300
timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids > $O/ablate.txt; echo "rc=$?"; cat $O/ablate.txt
