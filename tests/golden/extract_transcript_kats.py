#!/usr/bin/env python3
"""Extract the known-answer vectors the reference's own tests hold for its Fiat-Shamir transcripts (crates/jolt-transcript/tests/{keccak,blake2b}_tests.rs:
`test_keccak_known_vector`, `test_blake2b_known_vector`) into tests/golden/reference_transcript_kats.json.  Run where the reference checkout is:

    python tests/golden/extract_transcript_kats.py [/root/reference]

Each vector is: Transcript::new(<label>), append_bytes(<u64>.to_be_bytes()), challenge() == Fr::from_le_bytes_mod_order(<32 bytes>).  The script reads the label, the
integer and the byte literals out of the test function's body and fails if the function is missing or has another shape."""
import json
import os
import re
import sys

CASES = [
    ("crates/jolt-transcript/tests/keccak_tests.rs", "test_keccak_known_vector", "KeccakTranscript", "keccak_sponge"),
    ("crates/jolt-transcript/tests/blake2b_tests.rs", "test_blake2b_known_vector", "Blake2bTranscript", "blake2b512_sponge"),
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
    for path, fn, ty, engine in CASES:
        src = open(os.path.join(root, path)).read()
        body, first, last = body_of(src, fn)
        label = re.search(re.escape(ty) + r"::<Fr>::new\(b\"([^\"]*)\"\)", body)
        value = re.search(r"append_bytes\(&(\d+)u64\.to_be_bytes\(\)\)", body)
        expected = re.search(r"from_le_bytes_mod_order\(&\[(.*?)\]\)", body, re.S)
        if not (label and value and expected):
            raise SystemExit(f"{path}::{fn}: shape changed")
        bytes_le = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", expected.group(1))]
        if len(bytes_le) != 32:
            raise SystemExit(f"{path}::{fn}: {len(bytes_le)} expected bytes")
        out.append({"source": f"{path}:{first}-{last}", "test": fn, "engine": engine, "label": label.group(1), "append_u64_be": int(value.group(1)),
                    "challenge_le_bytes": bytes_le})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_transcript_kats.json")
    json.dump({"extracted_from": "a16z/jolt reference checkout, by tests/golden/extract_transcript_kats.py", "cases": out}, open(dst, "w"), indent=1)
    print(f"{len(out)} cases -> {dst}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
