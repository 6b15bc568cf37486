"""A short, fixed-seed slice of the randomised parity stress (tests/stress_cases.py; tools/stress_parity.py runs the same
generators for minutes on fresh seeds): every layer-level entry point through the HIP path against the CPU oracle, bit for
bit, over random shapes, thresholds, score distributions and forced schedules."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [20261111, 7])
def test_stress_slice(seed):
    from tests import stress_cases

    n = stress_cases.run(np.random.default_rng(seed), budget=30.0, max_cases=50)  # ~5 s; the budget is only a safety net
    assert sum(n.values()) >= 10, n
