"""GPU: the recorded GEMM kernel choices (mrca/gemm_tuning.py, PyTorch TunableOp) -- the record made on an MI355X of this
image must be ACCEPTED on the box (library versions match), and a GEMM of a recorded shape must give the default kernel's
result to fp32 summation order.  (What the choices buy is measured by bench.py --mode train, profiles/r04_f_*.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_recorded_gemm_choices_are_accepted_and_change_no_result():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from mrca import gemm_tuning
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(16384, 260, device="cuda", generator=gen)          # fc2: [B, 260] x [260, 128]
    w = torch.randn(128, 260, device="cuda", generator=gen)
    gy = torch.randn(16384, 128, device="cuda", generator=gen)
    ref = (torch.nn.functional.linear(x, w), gy.t() @ x, gy @ w)       # forward, weight gradient, data gradient
    try:
        assert gemm_tuning.use_recorded_choices(), "the record was rejected: made with other library versions?"
        assert torch.cuda.tunable.is_enabled() and not torch.cuda.tunable.tuning_is_enabled()
        got = (torch.nn.functional.linear(x, w), gy.t() @ x, gy @ w)
    finally:
        torch.cuda.tunable.enable(False)
    for a, b in zip(ref, got):
        assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max())
    rows = [ln for ln in open(gemm_tuning.DEFAULT_FILE) if ln.startswith("Gemm")]
    assert len(rows) >= 10 and any("260_16384_128" in ln or "260_128_16384" in ln for ln in rows)
