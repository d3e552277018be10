"""A slice of the randomised exactness sweep (tests/stress_gpu.py) under pytest: random scenes, thresholds, match and
hypothesis counts, 1-4 ragged cameras; every bound must bound, every mixed-precision count must equal the float64 count,
every cascade must name the float64 first maximum.  The long form (20 000 configurations) is run by hand."""
import pytest

pytestmark = pytest.mark.gpu


def test_randomised_exactness_sweep_slice(capsys):
    import stress_gpu

    failures = stress_gpu.main(1500, seed=777)
    out = capsys.readouterr().out
    assert failures == 0, out[-3000:]
    assert "1500 configurations, 0 failures" in out
