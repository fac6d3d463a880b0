#!/usr/bin/env bash
# A/B of a kernel experiment against the committed library on ONE GPU box, alternating runs (how profiles/r04_{g,h,i,j,x,y}_ab_*
# were measured: boxes differ by a few per cent, runs on one box by a few tenths).
#
#   HERE (no GPU needed):   bash tools/ab_variant.sh build PATH/NAME.patch      # -> tools/_build/NAME/.../libmrca_env.so
#   ON THE BOX (gpurun):    bash tools/ab_variant.sh run NAME [REPS] [bench.py arguments ...]  # -> gpurun_out/ab_NAME.txt
#
# `build` copies csrc/ and include/ into tools/_build/NAME (git-ignored, travels with gpurun), applies the patch there with
# `patch -p1` and runs that copy's build.sh; the committed sources and library are not touched.  `run` alternates
# bench.py (env mode, replayed and eager) between the committed library and the variant (MRCA_ENV_LIB) REPS times and
# prints value / tick / kernel times of each run.  Adopting a variant means applying its patch to the real sources and
# re-running the full validation (tools/gpu_pass.sh smoke tests bench prof sq): a variant library is never shipped.
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R"
PKG=rl-collision-avoidance_amd
case "${1:-}" in
  build)
    patch_file="$(realpath "$2")"
    name="$(basename "$patch_file" .patch)"
    dst="tools/_build/$name"
    rm -rf "$dst"
    mkdir -p "$dst/$PKG/mrca" "$dst/include"
    cp -r "$PKG/csrc" "$dst/$PKG/csrc"
    cp include/*.h "$dst/include/"
    (cd "$dst" && patch -p1 < "$patch_file")
    bash "$dst/$PKG/csrc/build.sh"
    echo "variant library: $dst/$PKG/mrca/libmrca_env.so" ;;
  run)
    name="$2"; reps="${3:-3}"; shift; shift; [ $# -gt 0 ] && shift
    lib="$R/tools/_build/$name/$PKG/mrca/libmrca_env.so"
    [ -f "$lib" ] || { echo "no $lib: run 'build' first (here, before gpurun)" >&2; exit 2; }
    mkdir -p gpurun_out
    line() {
      python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-8s %-6s value %7.2f M  tick %6.2f us  ray %6.2f  move %6.2f' % ('$1', '$2', d['value'] / 1e6, d['ms_per_step'] * 1e3,
      r['kernel_avg_us'], r['move_kernel_avg_us']))"
    }
    for _ in $(seq "$reps"); do
      for which in base "$name"; do
        if [ "$which" = base ]; then unset MRCA_ENV_LIB; else export MRCA_ENV_LIB="$lib"; fi
        python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra "$@" 2>/dev/null | line "$which" graph
        python bench.py --steps 2000 --warmup 200 --no-graph --no-cpu-baseline --no-extra "$@" 2>/dev/null | line "$which" eager
      done
    done 2>&1 | tee "gpurun_out/ab_$name.txt" ;;
  *)
    sed -n 2,14p "$0"; exit 2 ;;
esac
