"""GPU parity on random lattices: every trial of tests/random_lattices.py (prices, implied vols and
flags against the CPU oracle on identical inputs, and afiv_surface against its two stages) for a
fixed range of seeds."""
import pytest

import oracle
from random_lattices import make_case, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", range(0, 240, 40))
def test_random_lattices_match_oracle(afiv, first):
    oracle.set_threads(16)
    failures = []
    for seed in range(first, first + 40):
        c = make_case(seed)
        try:
            run_case(afiv, c)
        except AssertionError as e:
            failures.append(f"seed {seed}: {c['desc']}: {e}")
    assert not failures, "\n".join(failures)
