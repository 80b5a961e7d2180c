#!/usr/bin/env python3
"""Extract the reference's BN254 golden byte vectors into a JSON fixture.

Source: /root/reference/crates/jolt-field/tests/golden_bytes.rs (consts FIX_BN254_FR, FIX_BN254_FQ,
FIX_BN254_FR_CHALLENGE, FIX_BN254_FR_SCALAR_CHALLENGE, FIX_BN254_FQ_CHALLENGE, FIX_BN254_FQ_SCALAR_CHALLENGE).
Row format there: (input hex, expected canonical-LE hex).  /root/reference does not exist on the GPU box,
so the vectors are committed as tests/golden/bn254_golden_bytes.json; re-run this script to regenerate.
"""
import json
import os
import re

SRC = "/root/reference/crates/jolt-field/tests/golden_bytes.rs"
NAMES = ["FIX_BN254_FR", "FIX_BN254_FQ", "FIX_BN254_FR_CHALLENGE", "FIX_BN254_FR_SCALAR_CHALLENGE",
         "FIX_BN254_FQ_CHALLENGE", "FIX_BN254_FQ_SCALAR_CHALLENGE"]


def main():
    text = open(SRC).read()
    out = {"source": "crates/jolt-field/tests/golden_bytes.rs", "tables": {}}
    for name in NAMES:
        m = re.search(r"const " + name + r": &\[\(&str, &str\)\] = &\[(.*?)\];", text, re.S)
        assert m, name
        rows = re.findall(r'\(\s*"([0-9a-f]*)",\s*"([0-9a-f]*)",?\s*\)', m.group(1))
        assert rows, name
        out["tables"][name] = rows
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_golden_bytes.json")
    json.dump(out, open(dst, "w"), indent=1)
    print({k: len(v) for k, v in out["tables"].items()})


if __name__ == "__main__":
    main()
