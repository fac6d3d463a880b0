#!/usr/bin/env bash
# Build libmrca_env.so for gfx950 in-tree (cross-compiles without a GPU).
#   -ffp-contract=off + correctly rounded div/sqrt: the tick is specified as separately rounded
#   IEEE fp32 operations so a launch can be compared bit-for-bit with the oracle's fp32 mode.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
#   --profiling: the same sources with -DMRCA_PROFILING -> libmrca_env_prof.so (ablation switches compiled in,
#   mrca_set_debug_flags exported); tools/ablate.py loads it through MRCA_ENV_LIB.  The product never does.
out="${here}/../mrca/libmrca_env.so"
extra=()
if [[ "${1:-}" == "--profiling" ]]; then
    shift
    out="${here}/../mrca/libmrca_env_prof.so"
    extra=(-DMRCA_PROFILING)
fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC
    -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math
    -Wall -Wno-unused-function "${extra[@]}")
# One object per source: mrca_policy.hip is compiled with the MFMA accumulators in VGPRs (its epilogues read them lane by
# lane while the next MFMAs run; in AGPRs every read is a v_accvgpr_read_b32 first and the compiler lumps 32 of them
# behind the last MFMA of a tile pair) -- a per-file flag: the backward kernel needs AGPRs for its 242 accumulator
# registers.
pids=()
obj="$(mktemp -d)"
trap 'rm -rf "${obj}"' EXIT
for src in mrca_kernels mrca_abi mrca_policy mrca_policy_bwd mrca_policy_tail mrca_ppo_loss mrca_adam mrca_rollout_store mrca_policy_heads; do
    per=()
    [[ "${src}" == "mrca_policy" ]] && per=(-mllvm --amdgpu-mfma-vgpr-form)
    # the env kernels: the first 14 dwords of a kernel's arguments arrive in SGPRs (gfx950's kernarg preload) -- move_kernel and
    # raycast_kernel lead with the scalars and pointers their first loads need (mrca_kernels.hip)
    [[ "${src}" == "mrca_kernels" ]] && per=(-mllvm -amdgpu-kernarg-preload-count=14)
    "${HIPCC}" "${FLAGS[@]}" "${per[@]}" -c "${here}/${src}.hip" -o "${obj}/${src}.o" "$@" &
    pids+=($!)
done
# wait for EVERY compile before acting on a failure: leaving at the first one would let the EXIT trap remove the object
# directory under the compiles still running
failed=0
for pid in "${pids[@]}"; do wait "${pid}" || failed=1; done
if (( failed )); then echo "build.sh: a compile failed" >&2; exit 1; fi
"${HIPCC}" --offload-arch=gfx950 -fPIC -shared "${obj}"/*.o -o "${out}"
echo "built ${out}"
