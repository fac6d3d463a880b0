#!/usr/bin/env bash
# Hardware-counter characterisation of the two env kernels: several rocprofv3 --pmc passes of a short
# env-only bench (counters only, no extra trace domains).  Output: gpurun_out/pmc_$TAG/summary.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${TAG:-r01}"
OUT="$R/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT" -o "set$i" -- \
    python "$R/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-extra --schedule eager --chains 1 ${BENCH_EXTRA:-} > "$OUT/set$i.log" 2>&1
  echo "set $i ($SET) rc=$?"
done
cd "$R"
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -f "$OUT"/*kernel_trace.csv "$OUT"/*agent_info.csv
cat "$OUT/summary.txt"
