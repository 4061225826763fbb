"""A short randomised parity sweep in the suite (tools/fuzz_parity.py runs the long one): random models / sizes / thresholds / resamplers /
covariance kinds / missing and outlying measurements / drivers (run, run after run, single steps, auxiliary filter, history, banks), the
engine against the device-order oracle bit for bit.  The seed is fixed: a failure reproduces."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_match_the_oracle(seed):
    import fuzz_parity
    bad, drivers = fuzz_parity.sweep(60, seed, verbose=False)
    assert len(drivers) >= 6, drivers
    assert not bad, "\n".join(bad)
