"""The differential fuzzer (tools/fuzz_parity.py) as a bounded test: a fixed seed's first cases in both modes -- one handle against the CPU
oracle, N slabs against one handle, bit for bit -- and the recipes of what the fuzzer found in round 6 (tests/golden/fuzz_group_regressions.json:
data, i.e. the drawn parameters of those cases)."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def fuzz(pkg):
    import fuzz_parity
    return fuzz_parity


def _run(fuzz, pkg, oracle, mode, seed, n_cases, max_cells=250000):
    rng = np.random.default_rng(seed)
    ran = compared = 0
    for k in range(n_cases):
        c = fuzz.draw_case(rng, max_cells)
        if mode == "group":
            c = fuzz.draw_group(rng, c)
            bad, info = fuzz.run_group_case(pkg, pkg.engine, c)
        else:
            bad, info = fuzz.run_case(pkg, pkg.engine, oracle, c)
        ran += 1
        if info.get("error") or info.get("blown_up"):  # a reported overflow / a state that left the number range: nothing to compare
            continue
        compared += 1
        assert not bad, json.dumps({"mode": mode, "seed": seed, "case": k, "recipe": c, "mismatches": bad})
    return ran, compared


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_one_handle_against_the_oracle(pkg, oracle, fuzz, seed):
    ran, compared = _run(fuzz, pkg, oracle, "oracle", seed, 150)
    assert compared >= 120, (ran, compared)


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_slabs_against_one_handle(pkg, oracle, fuzz, seed):
    ran, compared = _run(fuzz, pkg, oracle, "group", seed, 150)
    assert compared >= 100, (ran, compared)


def test_fuzz_regressions_of_round_6(pkg, fuzz):
    recs = json.load(open(os.path.join(ROOT, "tests", "golden", "fuzz_group_regressions.json")))
    assert len(recs) >= 3
    for r in recs:
        bad, info = fuzz.run_group_case(pkg, pkg.engine, r["recipe"])
        assert not info.get("error"), info
        assert not bad, json.dumps({"recipe": r["recipe"], "mismatches": bad})
