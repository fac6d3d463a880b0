#!/usr/bin/env bash
# One parameterised GPU-box pass (replaces the per-pass one-shot scripts of rounds 1-2).
#   TAG=r03a bash tools/gpu_pass.sh smoke tests bench update_bench prof ...
# Stages (run in the order given; each bounded by its own timeout, logs under gpurun_out/$TAG/):
#   smoke            __graft_entry__.smoke()
#   tests            pytest -m gpu            (TESTS="tests/test_x.py ..." narrows it; PYTEST_ARGS adds flags)
#   bench            bench.py default line (env-only, configs[1], cpu_baseline, roofline)
#   bench_modes      bench.py rollout / train (fused and stock update) / stage2
#   update_bench     tools/update_bench.py: forward + backward kernels of the conv front end, fwd+bwd+adam per minibatch
#   ablate           tools/ablate.py (profiling build)
#   prof             rocprofv3 --kernel-trace --stats of the default bench + FETCH_SIZE / WRITE_SIZE passes -> pmc_traffic.json
#   prof_train       rocprofv3 --kernel-trace --stats of bench.py --mode train (one update)
#   prof_rollout     rocprofv3 --kernel-trace --stats of bench.py --mode rollout --no-graph
#   sq               tools/pmc_profile.sh (SQ counter sets of the env kernels)
#   pmc_policy       SQ counter sets of the policy's forward / backward kernels (tools/update_bench.py 16384)
#   bigworld         tools/bigworld_bench.py
#   sliceprobe       tools/slice_probe.py: one rank's slice of a giant world's ray cast, launch shape by launch shape + timeline
#   prof_bigworld    rocprofv3 --kernel-trace --stats of tools/bigworld_bench.py (BIGWORLD_ARGS: robot counts)
#   circle           mrca.evaluate of the committed checkpoints (POLICY=... overrides)
#   train            tools/train_recipe.sh (TRAIN_ARGS / S1_SECONDS / S2_SECONDS)
#   ldsprobe         tools/lds_conflict_probe: SQ_LDS_BANK_CONFLICT of the policy front end's LDS access patterns, one by one
#   fidelity         bench.py --fidelity (Stage's resolutions, raster collisions + raster lidar): stage1 and stage2 lines
#   boundary         tools/launch_boundary (device-side stamps: eager vs hipGraph) + its rocprofv3 kernel trace through
#                    tools/trace_gaps.py; bench.py env mode eager and --graph; rocprofv3 trace of both with the gaps
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
TAG="${TAG:-pass}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
export TMPDIR=/tmp
export PYTHONPATH="$R/rl-collision-avoidance_amd"
flt() { grep -v amdgpu.ids; }
nproc > "$O/host.txt"; rocm-smi --showproductname 2>/dev/null | head -12 >> "$O/host.txt"
for STAGE in "$@"; do
  echo "==== $STAGE"
  case "$STAGE" in
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "rc=$?"; tail -1 "$O/smoke.log" ;;
    tests)
      timeout "${TESTS_TIMEOUT:-1800}" python -m pytest ${TESTS:-tests} -m gpu -q ${PYTEST_ARGS:-} > "$O/pytest_gpu.log" 2>&1; echo "rc=$?"
      flt < "$O/pytest_gpu.log" | tail -${TESTS_TAIL:-25} | cut -c1-400 ;;
    bench)
      timeout 600 python bench.py ${BENCH_ARGS:---steps 1000 --warmup 100} > "$O/bench_env.json" 2> "$O/bench.err"; echo "rc=$?"
      cut -c1-2200 "$O/bench_env.json"; flt < "$O/bench.err" | tail -3
      timeout 600 python bench.py --steps 1000 --warmup 100 --chains 1 --no-cpu-baseline --no-extra > "$O/bench_env_one_chain.json" 2>> "$O/bench.err"; echo "one chain rc=$?"; cut -c1-260 "$O/bench_env_one_chain.json"
      timeout 600 python bench.py --steps 2000 --warmup 200 --chains 3 --no-cpu-baseline --no-extra > "$O/bench_env_three_chains.json" 2>> "$O/bench.err"; echo "three chains rc=$?"; cut -c1-260 "$O/bench_env_three_chains.json"
      timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > "$O/bench_env_20_steps.json" 2>> "$O/bench.err"; echo "20-step rc=$?"; cut -c1-260 "$O/bench_env_20_steps.json" ;;
    bench_modes)
      timeout 600 python bench.py --mode rollout --steps 400 --warmup 40 --no-cpu-baseline --no-graph > "$O/bench_rollout.json" 2>> "$O/bench.err"; echo "rollout rc=$?"; cut -c1-260 "$O/bench_rollout.json"
      timeout 900 python bench.py --mode train --steps 256 --warmup 0 --no-cpu-baseline --no-graph > "$O/bench_train.json" 2>> "$O/bench.err"; echo "train rc=$?"; cut -c1-260 "$O/bench_train.json"
      timeout 900 python bench.py --mode train --steps 256 --warmup 0 --no-cpu-baseline > "$O/bench_train_graph.json" 2>> "$O/bench.err"; echo "train (rollout tick as a hipGraph) rc=$?"; cut -c1-260 "$O/bench_train_graph.json"
      timeout 600 python bench.py --mode rollout --steps 400 --warmup 40 --no-cpu-baseline > "$O/bench_rollout_graph.json" 2>> "$O/bench.err"; echo "rollout (hipGraph) rc=$?"; cut -c1-260 "$O/bench_rollout_graph.json"
      timeout 900 python bench.py --mode train --steps 256 --warmup 0 --no-cpu-baseline --no-graph --update-path stock > "$O/bench_train_stock.json" 2>> "$O/bench.err"; echo "train stock rc=$?"; cut -c1-260 "$O/bench_train_stock.json"
      timeout 600 python bench.py --scenario stage2 --worlds 187 --steps 500 --warmup 50 --no-cpu-baseline --no-extra > "$O/bench_stage2.json" 2>> "$O/bench.err"; echo "stage2 rc=$?"; cut -c1-260 "$O/bench_stage2.json"
      flt < "$O/bench.err" | tail -4 ;;
    update_bench)
      timeout 900 python tools/update_bench.py ${UPDATE_BENCH_ARGS:-} 2>&1 | flt > "$O/update_bench.txt"; echo "rc=$?"; cut -c1-220 "$O/update_bench.txt" | head -80 ;;
    ablate)
      timeout 600 python tools/ablate.py 2>&1 | flt > "$O/ablate.txt"; echo "rc=$?"; cat "$O/ablate.txt" ;;
    prof)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o trace -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extra --schedule eager --chains 1 > "$O/prof_trace.log" 2>&1; echo "trace rc=$?"
      for C in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/prof" -o pmc_$C -- python "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extra --schedule eager --chains 1 > "$O/prof_pmc_$C.log" 2>&1; echo "pmc $C rc=$?"
      done
      cd "$R"
      f=$(find "$O/prof" -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -5 "$f" | cut -c1-200 && cp "$f" "$O/env_kernel_stats.csv"
      python tools/pmc_summary.py "$O/prof" > "$O/pmc_summary.txt" 2>&1; grep mrca "$O/pmc_summary.txt" | cut -c1-200
      python tools/pmc_summary.py "$O/prof" --traffic-json "$O/pmc_traffic.json" 4096 stage1 | cut -c1-400
      rm -rf "$O/prof" ;;
    prof_train)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_train" -o trace -- python "$R/bench.py" --mode train --steps 128 --warmup 0 --no-cpu-baseline --no-graph > "$O/prof_train.log" 2>&1; echo "rc=$?"
      cd "$R"
      f=$(find "$O/prof_train" -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-230 && cp "$f" "$O/train_kernel_stats.csv"
      rm -rf "$O/prof_train" ;;
    prof_rollout)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_rollout" -o trace -- python "$R/bench.py" --mode rollout --steps 100 --warmup 20 --no-cpu-baseline --no-graph > "$O/prof_rollout.log" 2>&1; echo "rc=$?"
      cd "$R"
      f=$(find "$O/prof_rollout" -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-230 && cp "$f" "$O/rollout_kernel_stats.csv"
      rm -rf "$O/prof_rollout" ;;
    sq)
      TAG="${TAG}_sq" timeout 900 bash tools/pmc_profile.sh > "$O/pmc_sq.log" 2>&1; cp "gpurun_out/pmc_${TAG}_sq/summary.txt" "$O/pmc_sq_summary.txt" 2>/dev/null
      grep -E "raycast|move_kernel" "$O/pmc_sq_summary.txt" | head -60 | cut -c1-200 ;;
    pmc_policy)
      # SQ counters of the policy's forward / backward kernels (counters only: no extra trace domains)
      cd /tmp; i=0
      for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU" \
                 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$O/pmc_policy" -o "set$i" -- python "$R/tools/update_bench.py" 16384 > "$O/pmc_policy_set$i.log" 2>&1; echo "set $i ($SET) rc=$?"
      done
      cd "$R"
      python tools/pmc_summary.py "$O/pmc_policy" > "$O/pmc_policy_summary.txt" 2>&1; grep -E "lidar_features|policy_tail" "$O/pmc_policy_summary.txt" | cut -c1-200 | head -40
      rm -rf "$O/pmc_policy" ;;
    bigworld)
      timeout 900 python tools/bigworld_bench.py ${BIGWORLD_ARGS:-} 2>&1 | flt | tee "$O/bigworld.jsonl" | cut -c1-300 ;;
    sliceprobe)
      timeout 600 python tools/slice_probe.py ${SLICEPROBE_ARGS:-} 2>&1 | flt | tee "$O/slice_probe.txt" | cut -c1-330
      timeout 600 python tools/slice_probe.py --sweep 2>&1 | flt | tee "$O/slice_sweep.txt" | cut -c1-200 ;;
    prof_bigworld)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_bw" -o trace -- python "$R/tools/bigworld_bench.py" ${BIGWORLD_ARGS:-50000} > "$O/prof_bigworld.log" 2>&1; echo "rc=$?"
      cd "$R"; flt < "$O/prof_bigworld.log" | tail -4 | cut -c1-300
      f=$(find "$O/prof_bw" -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-200 && cp "$f" "$O/bigworld_kernel_stats.csv"
      rm -rf "$O/prof_bw" ;;
    circle)
      for P in ${POLICY:-$R/rl-collision-avoidance_amd/mrca/data/policy_r02_stage2_circles.pth $R/rl-collision-avoidance_amd/mrca/data/policy_r02_all_circle_sizes.pth}; do
        for SPEC in "10 8" "20 12" "30 16" "40 20" "50 25"; do set -- $SPEC
          timeout 300 python -m mrca.evaluate --circles ${CIRCLES:-100} --robots $1 --radius $2 --policy "$P" --max-ticks 2000 ${EVAL_ARGS:-} 2>/dev/null | tail -1 | tee -a "$O/circle_eval.jsonl" | cut -c1-330
        done
      done ;;
    train)
      bash tools/train_recipe.sh "$O" ;;
    ldsprobe)
      [ -x tools/_build/lds_conflict_probe ] || hipcc --offload-arch=gfx950 -O3 tools/lds_conflict_probe.hip -o tools/_build/lds_conflict_probe
      timeout 120 tools/_build/lds_conflict_probe > "$O/lds_conflict_probe.txt" 2>&1; cat "$O/lds_conflict_probe.txt"
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/ldsprobe" -o p -- "$R/tools/_build/lds_conflict_probe" > "$O/ldsprobe.log" 2>&1; echo "pmc rc=$?"
      cd "$R"
      python tools/pmc_summary.py "$O/ldsprobe" > "$O/lds_conflict_probe_counters.txt" 2>&1; grep -E "stage_scan|conv[12]_" "$O/lds_conflict_probe_counters.txt" | cut -c1-160
      rm -rf "$O/ldsprobe" ;;
    fidelity)
      timeout 600 python bench.py --steps 1000 --warmup 100 --fidelity --no-extra > "$O/bench_env_fidelity.json" 2>> "$O/bench.err"; echo "fidelity rc=$?"; cut -c1-400 "$O/bench_env_fidelity.json"
      timeout 600 python bench.py --steps 500 --warmup 50 --fidelity --scenario stage2 --no-extra --no-cpu-baseline > "$O/bench_stage2_fidelity.json" 2>> "$O/bench.err"; echo "stage2 fidelity rc=$?"; cut -c1-300 "$O/bench_stage2_fidelity.json" ;;
    boundary)
      [ -x tools/_build/launch_boundary ] || { mkdir -p tools/_build; hipcc --offload-arch=gfx950 -O3 tools/launch_boundary.hip -o tools/_build/launch_boundary; }
      timeout 120 tools/_build/launch_boundary > "$O/launch_boundary.txt" 2>&1; echo "rc=$?"; cut -c1-230 "$O/launch_boundary.txt"
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/prof_lb" -o trace -- "$R/tools/_build/launch_boundary" > "$O/prof_lb.log" 2>&1; echo "trace rc=$?"
      cd "$R"
      python tools/trace_gaps.py "$O/prof_lb" stamp_kernel > "$O/launch_boundary_rocprof_gaps.txt" 2>&1; cut -c1-200 "$O/launch_boundary_rocprof_gaps.txt"
      rm -rf "$O/prof_lb"
      timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra > "$O/bench_env_eager.json" 2>> "$O/bench.err"; echo "eager rc=$?"; cut -c1-330 "$O/bench_env_eager.json"
      timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra --graph > "$O/bench_env_graph.json" 2>> "$O/bench.err"; echo "graph rc=$?"; cut -c1-330 "$O/bench_env_graph.json"
      cd /tmp
      for M in eager graph; do
        FL=""; [ $M = graph ] && FL="--graph"
        timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$M" -o trace -- python "$R/bench.py" --steps 300 --warmup 30 --no-cpu-baseline --no-extra $FL > "$O/prof_$M.log" 2>&1; echo "trace $M rc=$?"
        python "$R/tools/trace_gaps.py" "$O/prof_$M" raycast_kernel move_kernel > "$O/env_${M}_gaps.txt" 2>&1; cut -c1-200 "$O/env_${M}_gaps.txt"
        f=$(find "$O/prof_$M" -name "trace_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/env_${M}_kernel_stats.csv"
        rm -rf "$O/prof_$M"
      done
      cd "$R" ;;
    *) echo "unknown stage $STAGE" ;;
  esac
done
du -sh "$O"
