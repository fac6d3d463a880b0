#!/usr/bin/env bash
# Stage the reference's UNCHANGED Python scripts for the GPU box.
#
# The north-star sentence "ppo_stage1.py drops in unchanged" can only be tested where a checkout of the reference and an
# MI355X meet.  /root/reference does not exist on the GPU box, and reference sources must never enter this repository's
# history -- so this script packs the handful of files the unchanged-script tests execute (the three entry scripts, the
# three StageWorld modules the bridge test feeds, model/*.py) into ONE git-ignored archive, tests/_reference.tgz, which
# `gpurun` ships with the working tree like the built .so files.  tests/util.py:reference_dir() unpacks it into a
# temporary directory at test time when $MRCA_REFERENCE / /root/reference is absent.  Nothing in the product reads it.
#   MRCA_REFERENCE=/path/to/rl-collision-avoidance bash tools/stage_reference.sh
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
ref="${MRCA_REFERENCE:-/root/reference}"
out="${here}/tests/_reference.tgz"
if [[ ! -f "${ref}/ppo_stage1.py" ]]; then
    echo "stage_reference.sh: no reference checkout at ${ref} (set MRCA_REFERENCE)" >&2
    exit 1
fi
tar -C "${ref}" -czf "${out}" ppo_stage1.py ppo_stage2.py circle_test.py stage_world1.py stage_world2.py circle_world.py \
    $(cd "${ref}" && ls model/*.py)
echo "staged $(tar -tzf "${out}" | wc -l) files of ${ref} -> ${out} ($(stat -c %s "${out}") bytes, git-ignored)"
