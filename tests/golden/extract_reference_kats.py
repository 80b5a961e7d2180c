#!/usr/bin/env python3
"""Extract the known-answer tests the reference's own unit tests hold for the hot path ABOVE the field layer -- small integer cases of UnivariatePoly
(crates/jolt-poly/src/univariate.rs), interpolate_to_coeffs over the domain 0 .. n - 1 (crates/jolt-poly/src/lagrange.rs: the same map as UnivariatePoly::from_evals) and eval_univariate (crates/jolt-hyperkzg/src/kzg.rs) --
into tests/golden/reference_kats.json.  Run in a container that has the reference checkout:

    python tests/golden/extract_reference_kats.py [/root/reference]

Each case is taken from one named #[test] function: the Fr::from_u64(N) literals are read from the function's body IN ORDER and split into inputs / expected values
by the shape table below (which says how many literals form the inputs); the script fails if a function is missing or holds a different number of literals than the
shape expects, so a silent drift of the reference shows up as an extraction error rather than as a stale fixture."""
import json
import os
import re
import sys

# (file, test fn, kind, shape): kind names what tests/test_oracle_reference_kats.py does with the literals
CASES = [
    ("crates/jolt-poly/src/univariate.rs", "horner_known_polynomial", "evaluate", {"coeffs": 3, "pairs": 3}),           # coeffs, then (x, p(x)) pairs
    ("crates/jolt-poly/src/univariate.rs", "interpolate_linear", "interpolate_then_evaluate", {"points": 2, "pairs": 1}),  # (x, y) points, then (x, p(x))
    ("crates/jolt-poly/src/univariate.rs", "from_evals_quadratic", "from_evals", {"evals": 3, "coeffs": 3}),
    ("crates/jolt-poly/src/univariate.rs", "from_evals_cubic", "from_evals", {"evals": 4, "coeffs": 4}),
    ("crates/jolt-poly/src/univariate.rs", "from_evals_and_hint", "from_evals_and_hint", {"hint": 1, "evals": 2, "pairs": 3}),
    ("crates/jolt-poly/src/lagrange.rs", "interpolate_to_coeffs_constant", "interpolate_to_coeffs_prefix", {"vals": 3, "coeffs": 1}),  # only coeffs[0] is a literal; the rest is_zero()
    ("crates/jolt-poly/src/lagrange.rs", "interpolate_to_coeffs_linear", "interpolate_to_coeffs_prefix", {"vals": 2, "coeffs": 2}),
    ("crates/jolt-hyperkzg/src/kzg.rs", "eval_univariate_at_zero", "eval_at_zero", {"coeffs": 3, "value": 1}),
    ("crates/jolt-hyperkzg/src/kzg.rs", "eval_univariate_linear", "evaluate", {"coeffs": 2, "pairs": 1}),
]


def body_of(src, fn):
    m = re.search(r"fn\s+" + re.escape(fn) + r"\s*\(\s*\)\s*\{", src)
    if not m:
        raise SystemExit(f"test function {fn} not found")
    i, depth = m.end(), 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.start():i], src.count("\n", 0, m.start()) + 1, src.count("\n", 0, i) + 1


def main(root):
    out = []
    for path, fn, kind, shape in CASES:
        src = open(os.path.join(root, path)).read()
        body, first, last = body_of(src, fn)
        lits = [int(x) for x in re.findall(r"Fr::from_u64\((\d+)\)", body)]
        want = sum(2 * v if k in ("pairs", "points") else v for k, v in shape.items())
        if len(lits) != want:
            raise SystemExit(f"{path}::{fn}: {len(lits)} literals, the shape expects {want}: {lits}")
        case, pos = {"source": f"{path}:{first}-{last}", "test": fn, "kind": kind}, 0
        for k, v in shape.items():
            if k in ("pairs", "points"):
                case[k] = [[lits[pos + 2 * t], lits[pos + 2 * t + 1]] for t in range(v)]
                pos += 2 * v
            else:
                case[k] = lits[pos:pos + v]
                pos += v
        out.append(case)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    json.dump({"extracted_from": "a16z/jolt reference checkout, by tests/golden/extract_reference_kats.py", "cases": out}, open(dst, "w"), indent=1)
    print(f"{len(out)} cases -> {dst}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
