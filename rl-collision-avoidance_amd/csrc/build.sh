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
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -Wall -Wno-unused-function "${extra[@]}" \
    "${here}/mrca_kernels.hip" "${here}/mrca_abi.hip" "${here}/mrca_policy.hip" "${here}/mrca_policy_bwd.hip" "${here}/mrca_policy_tail.hip" \
    -o "${out}" "$@"
echo "built ${out}"
