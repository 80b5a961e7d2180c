"""Known-answer tests of the REFERENCE's own unit tests above the field layer (tests/golden/reference_kats.json, extracted from the reference checkout by
tests/golden/extract_reference_kats.py: UnivariatePoly::{evaluate, interpolate, from_evals, from_evals_and_hint}, interpolate_to_coeffs over 0 .. n - 1,
kzg::eval_univariate) against (i) the oracle's restatements and (ii) the product's host code above the ABI (jolt_host_univariate_*: what assembles every round
message of the C++ mirror; no device needed).  Small integer cases -- they pin conventions (coefficient order, the evaluation domain 0, 1, 2, ..., the hint
s(0) + s(1)), which is exactly what a restatement can get wrong silently."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))["cases"]


def m(values):
    return O.to_mont([int(v) for v in values])


IMPLS = [("oracle", O.univariate_from_evals, O.univariate_evaluate), ("product host code", ffi.host_univariate_from_evals, ffi.host_univariate_evaluate)]


@pytest.mark.parametrize("case", CASES, ids=[c["test"] for c in CASES])
@pytest.mark.parametrize("impl", IMPLS, ids=[i[0] for i in IMPLS])
def test_reference_known_answers(case, impl):
    _, from_evals, evaluate = impl
    kind = case["kind"]
    if kind == "evaluate":
        coeffs = m(case["coeffs"])
        for x, y in case["pairs"]:
            assert np.array_equal(evaluate(coeffs, m([x])[0]), m([y])[0]), case["source"]
            assert np.array_equal(O.kzg_eval_univariate(coeffs, m([x])[0]), m([y])[0]), case["source"]  # kzg.rs:51-59 is the same Horner
    elif kind == "eval_at_zero":
        assert np.array_equal(evaluate(m(case["coeffs"]), m([0])[0]), m(case["value"])[0])
        assert np.array_equal(O.kzg_eval_univariate(m(case["coeffs"]), m([0])[0]), m(case["value"])[0])
    elif kind == "interpolate_then_evaluate":
        assert [p[0] for p in case["points"]] == list(range(len(case["points"])))  # the reference's points sit on 0, 1, ...: from_evals' domain
        coeffs = from_evals(m([p[1] for p in case["points"]]))
        for x, y in case["pairs"]:
            assert np.array_equal(evaluate(coeffs, m([x])[0]), m([y])[0]), case["source"]
    elif kind == "from_evals":
        assert np.array_equal(from_evals(m(case["evals"])), m(case["coeffs"])), case["source"]
    elif kind == "interpolate_to_coeffs_prefix":
        got = from_evals(m(case["vals"]))
        k = len(case["coeffs"])
        assert np.array_equal(got[:k], m(case["coeffs"])) and not got[k:].any(), case["source"]  # the remaining coefficients are is_zero() in the reference's test
    elif kind == "from_evals_and_hint":
        # UnivariatePoly::from_evals_and_hint (univariate.rs): evals are [p(0), p(2), ...], p(1) = hint - p(0)
        p0, rest = m(case["evals"][:1])[0], m(case["evals"][1:])
        p1 = O.fr_sub(m(case["hint"]), p0.reshape(1, 4))[0]
        coeffs = from_evals(np.vstack([p0, p1, rest]))
        for x, y in case["pairs"]:
            assert np.array_equal(evaluate(coeffs, m([x])[0]), m([y])[0]), case["source"]
    else:
        raise AssertionError(kind)


def test_fixture_still_matches_the_reference_checkout():
    """where the reference checkout exists (the build container), the committed fixture is what the extraction script produces now"""
    if not os.path.isdir("/root/reference/crates"):
        pytest.skip("no reference checkout on this box")
    import subprocess
    import sys
    before = open(os.path.join(HERE, "golden", "reference_kats.json")).read()
    assert subprocess.run([sys.executable, os.path.join(HERE, "golden", "extract_reference_kats.py")], capture_output=True).returncode == 0
    assert open(os.path.join(HERE, "golden", "reference_kats.json")).read() == before
