"""The A/B tooling builds a variant library from a patched COPY of the sources and leaves the committed ones alone
(tools/ab_variant.sh; no GPU needed: hipcc cross-compiles)."""
import hashlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rl-collision-avoidance_amd")


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_ab_variant_builds_a_patched_copy(tmp_path):
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    src = os.path.join(PKG, "csrc", "mrca_abi.hip")
    text = open(src).read()
    first = text.splitlines()[0]
    patch = tmp_path / "ab_selftest.patch"
    patch.write_text("--- a/rl-collision-avoidance_amd/csrc/mrca_abi.hip\n+++ b/rl-collision-avoidance_amd/csrc/mrca_abi.hip\n"
                     f"@@ -1,1 +1,2 @@\n-{first}\n+{first}\n+// (variant built by tests/test_tools.py)\n")
    before = {f: _sha(os.path.join(PKG, "csrc", f)) for f in os.listdir(os.path.join(PKG, "csrc")) if not f.startswith(".")}
    lib_before = _sha(os.path.join(PKG, "mrca", "libmrca_env.so")) if os.path.exists(os.path.join(PKG, "mrca", "libmrca_env.so")) else None
    dst = os.path.join(ROOT, "tools", "_build", "ab_selftest")
    try:
        out = subprocess.run(["bash", os.path.join(ROOT, "tools", "ab_variant.sh"), "build", str(patch)], capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        lib = os.path.join(dst, "rl-collision-avoidance_amd", "mrca", "libmrca_env.so")
        assert os.path.exists(lib) and os.path.getsize(lib) > 100_000
        assert "(variant built by tests/test_tools.py)" in open(os.path.join(dst, "rl-collision-avoidance_amd", "csrc", "mrca_abi.hip")).read()
        # the committed sources and library are untouched
        assert before == {f: _sha(os.path.join(PKG, "csrc", f)) for f in before}
        if lib_before is not None:
            assert lib_before == _sha(os.path.join(PKG, "mrca", "libmrca_env.so"))
        # `run` without the variant present refuses instead of measuring the committed library against itself
        bad = subprocess.run(["bash", os.path.join(ROOT, "tools", "ab_variant.sh"), "run", "no_such_variant"], capture_output=True,
                             text=True, timeout=60)
        assert bad.returncode == 2 and "build" in bad.stderr
    finally:
        shutil.rmtree(dst, ignore_errors=True)
