"""CPU-side checks of the native library: it loads, exports every symbol include/jolt_hip.h declares, refuses to run
without a gfx950 device (no CPU fallback), and its host-side field / round-message helpers agree with the oracle."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from util import rand_challenge, rand_fr

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "jolt_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jolt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ffi.lib()
    syms = declared_symbols()
    assert len(syms) > 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.jolt_abi_version() == 1


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    st = ffi.lib().jolt_ctx_create(C.c_int32(0), None, C.byref(h))
    assert st == 2 and not h  # JOLT_ERR_NO_DEVICE: the product path fails loudly instead of falling back
    with pytest.raises(ffi.JoltError):
        ffi.Context(0)


def test_host_field_ops_match_oracle():
    # field.hip.h compiled for the host: the 64-bit-limb host product (host_mul64) and the shared add/sub/inv code;
    # the kernels' 32-bit mont_rows<8> is covered by the GPU tests, mont_rows<4> below
    a, b = rand_fr(200, 1), rand_fr(200, 2)
    edge = O.to_mont([0, 1, O.R_MOD - 1, 2, O.R_MOD - 2])
    a[:5], b[:5] = edge, edge[::-1]
    want_mul, want_add, want_sub = O.fr_mul(a, b), O.fr_add(a, b), O.fr_sub(a, b)
    for i in range(200):
        assert np.array_equal(ffi.host_fr_mul(a[i], b[i]), want_mul[i])
        assert np.array_equal(ffi.host_fr_add(a[i], b[i]), want_add[i])
        assert np.array_equal(ffi.host_fr_sub(a[i], b[i]), want_sub[i])
    inv = O.fr_inv(a[5:40])
    for i in range(5, 40):
        assert np.array_equal(ffi.host_fr_inv(a[i]), inv[i - 5])
    with pytest.raises(ffi.JoltError):
        ffi.host_fr_inv(np.zeros(4, dtype=np.uint64))
    for v in (0, 1, 2**14, 2**63 + 5, 2**64 - 1):
        assert np.array_equal(ffi.host_fr_from_u64(v), O.fr_from_u64([v])[0])


def test_shifted_challenge_multiply_matches_full_multiply():
    # the 125-bit challenge fast path (4 reduction rows) must give the canonical product
    a = rand_fr(100, 3)
    for i in range(100):
        c = rand_challenge(100 + i)
        assert np.array_equal(ffi.host_fr_mul_shifted(a[i], c), O.fr_mul(a[i:i + 1], c.reshape(1, 4))[0])
    # extreme shifted operands: all-ones high limbs below the modulus, and zero
    c = np.array([0, 0, 2**64 - 1, (1 << 61) - 1], dtype=np.uint64)
    assert np.array_equal(ffi.host_fr_mul_shifted(a[0], c), O.fr_mul(a[:1], c.reshape(1, 4))[0])
    z = np.zeros(4, dtype=np.uint64)
    assert np.array_equal(ffi.host_fr_mul_shifted(a[0], z), z)
    with pytest.raises(ffi.JoltError):
        ffi.host_fr_mul_shifted(a[0], a[1])  # low limbs not zero


def test_host_round_message_assembly_matches_oracle():
    rng = random.Random(4)
    for n in (2, 3, 4, 6):
        evals = rand_fr(n, 10 + n)
        assert np.array_equal(ffi.host_univariate_from_evals(evals), O.univariate_from_evals(evals))
        x = rand_fr(1, 20 + n)[0]
        co = O.univariate_from_evals(evals)
        assert np.array_equal(ffi.host_univariate_evaluate(co, x), O.univariate_evaluate(co, x))
    args = rand_fr(5, 30)
    assert np.array_equal(ffi.host_gruen_poly_deg_3(*args), O.gruen_poly_deg_3(*args))


def test_host_g1_ops_match_oracle():
    """The host mirror's G1 helpers (Fq arithmetic through the 64-bit host product): add / equality / compressed wire
    form against the oracle, including doubling, inverse pairs and the identity."""
    g = O.g1_generator()
    pts = [O.g1_scalar_mul(g, O.to_mont([k])[0]) for k in (1, 2, 3, 12345, O.R_MOD - 1)]
    pts.append(O.g1_identity())
    for p in pts:
        for q in pts:
            got, want = ffi.host_g1_add(p, q), O.g1_add(p, q)
            assert O.g1_eq(got, want)
            assert ffi.host_g1_eq(got, want)
            assert ffi.host_g1_serialize_compressed(got) == O.g1_serialize_compressed(want)
    assert not ffi.host_g1_eq(pts[0], pts[1])


def test_deferred_reduction_accumulator_equals_plain_field_sum():
    """wide_fmadd / wide_reduce (the kernels' WideAccumulator analogue, host build): sum a_k*b_k with one REDC per block equals the
    sum of the reduced products and the oracle's restatement of the reference accumulator, including maximal operands
    (p-1)*(p-1) repeated up to the documented headroom and lengths around the flush boundary."""
    pm1 = O.to_mont([O.R_MOD - 1])[0]
    for n in (0, 1, 2, 7, 19, 20, 21, 40, 45):
        a, b = rand_fr(n, 600 + n), rand_fr(n, 700 + n)
        if n >= 2:
            a[0], b[0] = pm1, pm1
            a[n - 1], b[n - 1] = pm1, O.to_mont([1])[0]
        want = np.zeros((1, 4), dtype=np.uint64)
        if n:
            for row in O.fr_mul(a, b):
                want = O.fr_add(want, row.reshape(1, 4))
        assert np.array_equal(ffi.host_fr_wide_dot(a, b), want[0]), n
    worst = np.repeat(pm1.reshape(1, 4), 20, axis=0)  # 20 maximal products in one block: the headroom bound
    want = np.zeros((1, 4), dtype=np.uint64)
    for row in O.fr_mul(worst, worst):
        want = O.fr_add(want, row.reshape(1, 4))
    assert np.array_equal(ffi.host_fr_wide_dot(worst, worst), want[0])
    if hasattr(O, "wide_accumulate"):
        a, b = rand_fr(33, 800), rand_fr(33, 801)
        assert np.array_equal(ffi.host_fr_wide_dot(a, b), O.wide_accumulate(a, b))


def test_device_multiplication_algorithm_matches_oracle_on_host():
    """mul_limbs29 (what `mul` is on the device: nine 29-bit limbs, product scanning, b << 5 to square away the 2^261 radix) built
    for the host: Fr and Fq products against the oracle, including 0, 1, p-1, p-2 and operands with all-ones limb patterns that
    maximise every column sum."""
    rng = np.random.default_rng(29)
    for field, mod, omul in ((0, O.R_MOD, O.fr_mul), (1, O.Q_MOD, O.fq_mul)):
        ints = [0, 1, 2, mod - 1, mod - 2, (1 << 253) - 1, (1 << 254) - 1 if (1 << 254) - 1 < mod else mod - 3, 0x1FFFFFFF, (1 << 29), ((1 << 232) - 1)]
        ints += [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(150)]
        vals = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in ints], dtype=np.uint64)  # raw limbs = Montgomery residues
        for i in range(len(ints)):
            for j in (i, (i * 7 + 3) % len(ints), len(ints) - 1 - i):
                want = omul(vals[i:i + 1], vals[j:j + 1])[0]
                assert np.array_equal(ffi.host_mul_limbs29(field, vals[i], vals[j]), want), (field, i, j)


def test_rust_ffi_crate_agrees_with_the_header():
    """rust/jolt-kernels-hip/src/ffi.rs (the crate a16z/jolt would link; it cannot be compiled in this image) declares every entry
    point of include/jolt_hip.h with the same name, arity, parameter names and types, in the same order, and is up to date with its
    generator."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(root, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    header, rust = gen.parse_header(), gen.parse_rust()
    assert len(header) > 100
    assert [d[0] for d in header] == [d[0] for d in rust]
    for h, r in zip(header, rust):
        assert h[1].replace("c_void", "()") == r[1].replace("c_void", "()") or (h[1] == "c_void" and r[1] == "()"), h[0]
        assert [(p.replace("r#", ""), t) for p, t in r[2]] == h[2], h[0]
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_ffi.py"), "--check"]).returncode == 0
    # every declared symbol is also what the shared library exports (the other half is test_library_exports_every_declared_symbol)
    lib_rs = open(os.path.join(root, "rust", "jolt-kernels-hip", "src", "lib.rs")).read()
    for module in ("context", "member", "msm", "scheduler", "status", "ffi", "ops", "pcs", "backend", "rows", "streaming"):
        assert f"pub mod {module};" in lib_rs and os.path.exists(os.path.join(root, "rust", "jolt-kernels-hip", "src", module + ".rs"))


def test_hand_written_rust_calls_name_real_entry_points_with_the_right_arity():
    """The crate cannot be compiled here, so its hand-written modules are checked mechanically against the generated declarations: every `ffi::jolt_*(`
    call names a function ffi.rs declares, with as many arguments as the declaration has parameters; every other `ffi::` item (constants, types) exists."""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rust", "jolt-kernels-hip", "src")
    ffi_rs = open(os.path.join(root, "ffi.rs")).read()
    decl = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (jolt_\w+)\(([^)]*)\)", ffi_rs)}
    items = set(re.findall(r"pub (?:const|type|struct|enum) (\w+)", ffi_rs)) | set(decl)
    assert len(decl) > 200

    def split_args(text):
        args, depth, cur = [], 0, ""
        for ch in text:
            if ch in "([{<":
                depth += 1
            elif ch in ")]}>":
                depth -= 1
            if ch == "," and depth == 0:
                args.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur)
        return args

    checked = 0
    for name in sorted(os.listdir(root)):
        if not name.endswith(".rs") or name == "ffi.rs":
            continue
        src = re.sub(r"//[^\n]*", "", open(os.path.join(root, name)).read())  # comments out (doc comments mention symbols freely)
        for m in re.finditer(r"(?<!core::)(?<!std::)ffi::(\w+)", src):
            assert m.group(1) in items, (name, m.group(1))
        for m in re.finditer(r"ffi::(jolt_\w+)\s*\(", src):
            fn, i, depth = m.group(1), m.end(), 1
            start = i
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            n_args = len(split_args(src[start:i - 1]))
            n_params = len(split_args(decl[fn]))
            assert n_args == n_params, (name, fn, n_args, n_params)
            checked += 1
    assert checked > 60


def _seam_audit():
    import importlib.util
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("rust_seam_audit", os.path.join(root, "tools", "rust_seam_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rust_crate_meets_the_reference_trait_bounds():
    """The trait-bound audit of rust/jolt-kernels-hip (tools/rust_seam_audit.py; the crate cannot meet a compiler here): crate roots against Cargo.toml, every `impl Trait for
    Type` of a reference trait against the trait's required items / arities / supertraits, and every generic use (`JoltBackend::<Fr, HipHyperKzg>::optimized()`, the
    advertised `jolt_prover::dory::prove::<Fr, HipHyperKzg, ..>`) against the reference's `where` clauses through its blanket impls and cfg(feature) variants.  Runs on the
    committed fixture of the reference's surface; where the reference checkout is present the fixture must also be what a fresh extraction gives."""
    import json
    A = _seam_audit()
    fixture = json.load(open(A.FIXTURE))
    a = A.Audit(fixture)
    assert a.run() == [], a.findings
    assert len(a.crate.impls) >= 45
    names = {(i["trait"], i["type"]) for i in a.crate.impls}
    for need in (("StreamingCommitment", "HipHyperKzg"), ("ZkOpeningScheme", "HipHyperKzg"), ("ZkStreamingCommitment", "HipHyperKzg"), ("CommitmentScheme", "HipHyperKzg"),
                 ("AdditivelyHomomorphic", "HipHyperKzg"), ("CommitWitness", "HipCommitWitness")):
        assert need in names, need
    if os.path.isdir(os.path.join(A.REFERENCE, "crates")):
        live = A.Audit()
        assert live.run() == [], live.findings
        assert json.loads(json.dumps(live.surface, sort_keys=True)) == fixture, "tests/golden/reference_trait_surface.json is stale: python tools/rust_seam_audit.py --write-fixture"


def test_trait_bound_audit_catches_the_round_4_gap(tmp_path):
    """Negative control: the crate WITHOUT streaming.rs is round 4's crate -- `JoltBackend::<Fr, HipHyperKzg>::optimized()` bounded by `PCS: ModeStreamingCommitment`
    (crates/jolt-kernels/src/optimized/mod.rs:136-139) over a PCS that only implements `CommitmentScheme` -- and the audit must say so; likewise a trait impl that loses
    a required method or grows a foreign one."""
    import json
    import shutil
    A = _seam_audit()
    fixture = json.load(open(A.FIXTURE))
    crate = tmp_path / "crate"
    shutil.copytree(A.CRATE, crate)
    os.remove(crate / "src" / "streaming.rs")
    lib_rs = (crate / "src" / "lib.rs").read_text()
    (crate / "src" / "lib.rs").write_text("\n".join(l for l in lib_rs.splitlines() if "streaming" not in l))
    found = A.Audit(fixture, A.Crate(str(crate))).run()
    assert any("HipHyperKzg: ModeStreamingCommitment" in f and "backend.rs" in f for f in found), found
    assert any("HipHyperKzg: ZkOpeningScheme" in f for f in found), found
    # a required method dropped, a foreign one added, a dependency missing
    shutil.rmtree(crate)
    shutil.copytree(A.CRATE, crate)
    src = (crate / "src" / "streaming.rs").read_text()
    assert "    fn begin(_setup" in src
    (crate / "src" / "streaming.rs").write_text(src.replace("    fn begin(_setup", "    fn begin_renamed(_setup", 1))
    cargo = (crate / "Cargo.toml").read_text()
    (crate / "Cargo.toml").write_text(cargo.replace("thiserror = { workspace = true }\n", ""))
    found = A.Audit(fixture, A.Crate(str(crate))).run()
    assert any("required method `begin` is missing" in f for f in found), found
    assert any("`begin_renamed` is not an item of the trait" in f for f in found), found
    assert any("thiserror" in f for f in found), found


@pytest.mark.parametrize("c,log_n", [(23, 16), (20, 16), (16, 14)])
def test_capacity_regions_hold_the_digits_of_uniform_scalars(c, log_n):
    """The capacity sort of the fixed-base MSM (msm_fixed.hip section 2d) gives every segment of 256 buckets a region sized from the digit MODEL of uniform scalars instead
    of counting the digits first.  Here the digits of 2^log_n uniform scalars are recoded by an independent Python restatement of the recoding (checked against the
    library's jolt_host_fx_digits on a sample), counted per segment, and every count must fit its region; regions where no digit can land keep the minimum; and at the
    benchmarked size (n = 2^26, c = 23) the regions together exceed the expected number of entries by less than 6 %."""
    from jolt_amd import ffi
    import oracle_lib as O
    r = O.R_MOD
    rng = random.Random(1000 * c + log_n)
    n, W, B = 1 << log_n, (253 + c - 1) // c, 1 << (c - 1)
    counts = {}
    for i in range(n):
        s = rng.randrange(r)
        t = s if s <= (r - 1) // 2 else r - s
        carry, digits = 0, []
        for w in range(W - 1):
            raw = ((t >> (w * c)) & ((1 << c) - 1)) + carry
            if raw > B:
                digits.append((1 << c) - raw)
                carry = 1
            else:
                digits.append(raw)
                carry = 0
        digits.append((t >> ((W - 1) * c)) + carry)
        if i < 40:  # the restatement above IS the library's recoding (magnitudes)
            got, _ = ffi.host_fx_digits(O.to_mont([s])[0], c)
            assert [abs(d) for d in got] == digits, (s, c)
        for d in digits:
            if d:
                counts[d >> 8] = counts.get(d >> 8, 0) + 1
    top_max = (((r - 1) // 2) >> ((W - 1) * c)) + 1
    n_segments = (max(B, top_max) >> 8) + 1
    caps = {seg: ffi.host_fx_segment_capacity(n, c, seg) for seg in set(counts) | {0, (B >> 8), (B >> 8) + 1, (top_max >> 8), n_segments - 1}}
    for seg, cnt in counts.items():
        assert seg < n_segments and cnt <= caps[seg], (seg, cnt, caps[seg])
    assert all(v % 4 == 0 and v >= 64 for v in caps.values())
    beyond = (max(B, top_max) >> 8) + 5
    assert ffi.host_fx_segment_capacity(n, c, beyond) == 64  # no digit reaches it: the floor only
    if c == 23:  # the benchmarked MSM: 2^26 terms -- 738 M entries expected, the regions' total within 6 % of it
        N = 1 << 26
        total = sum(ffi.host_fx_segment_capacity(N, 23, seg) for seg in (0, 1)) // 2  # a typical low segment
        low, high = total, ffi.host_fx_segment_capacity(N, 23, (B >> 8) + 2)
        n_low, n_high = B >> 8, (top_max >> 8) - (B >> 8)
        regions = n_low * low + n_high * high
        expected = N * 10 * (1 - 2.0 ** -23) + N  # ten signed windows (a zero digit is not stored) + the top window
        assert expected < regions < 1.06 * expected, (regions, expected)


def test_host_on_curve_check_accepts_group_elements_and_refuses_everything_else():
    """g1_is_on_curve (g1.hip.h) through jolt_host_g1_is_on_curve: what jolt_host_hyperkzg_open_with_levels applies to the level commitments a caller supplies.  Oracle
    points in their many projective representations (scalar multiples, sums, doublings, P + (-P) = the identity) pass; a flipped bit in any coordinate, a coordinate
    that is not reduced mod q and the affine 'infinity' (0, 0, 1) do not."""
    import oracle_lib as O
    from jolt_amd import ffi
    g = O.g1_generator()
    pts = [g, O.g1_identity(), O.g1_double(g), O.g1_neg(g)]
    rng = np.random.default_rng(5)
    for k in range(12):
        s = O.to_mont([int(rng.integers(1, 2**62)) * (1 + k)])[0]
        p = O.g1_scalar_mul(g, s)
        pts += [p, O.g1_add(p, pts[k % len(pts)]), O.g1_double(p)]
    pts.append(O.g1_add(pts[5], O.g1_neg(pts[5])))
    for p in pts:
        assert ffi.host_g1_is_on_curve(np.asarray(p, dtype=np.uint64)), p
    for p in pts[4:16]:
        q = np.asarray(p, dtype=np.uint64).copy()
        if O.g1_is_identity(q):
            continue
        for limb in (0, 5, 9):  # one bit in X, Y, Z
            bad = q.copy()
            bad[limb] ^= np.uint64(1 << 7)
            assert not ffi.host_g1_is_on_curve(bad), (limb, p)
        unreduced = q.copy()
        unreduced[0:4] = np.uint64(2**64 - 1)  # X >= q
        assert not ffi.host_g1_is_on_curve(unreduced)
    one_q = np.asarray(g, dtype=np.uint64)[0:4]
    assert not ffi.host_g1_is_on_curve(np.concatenate([np.zeros(8, dtype=np.uint64), one_q]))  # (0, 0, 1): 0 != 3


def test_integration_doc_lists_the_sources_the_build_compiles():
    """INTEGRATION.md section 4 names exactly jolt_amd/build.py:SOURCES (round-1 review: the list had gone stale)"""
    import re
    from jolt_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"jolt_amd/csrc/\{([a-z_0-9,]+)\}\.hip", doc)
    assert m, "source list not found in INTEGRATION.md"
    assert m.group(1).split(",") == [os.path.basename(s)[:-4] for s in build.SOURCES]


def test_limb_form_field_arithmetic_matches_oracle_on_host():
    """fq_limb.hip.h (2^261 Montgomery radix, lazy reduction, dedicated squaring, two-product reduction, lazily reduced differences) built
    for the host: every operation against the oracle's Fq arithmetic on corner values (0, 1, p - 1, all-ones limb patterns) and random
    operands -- the differences with the subtrahend LARGER than the minuend included, which is what the added multiples of p are for."""
    rng = np.random.default_rng(61)
    mod = O.Q_MOD
    ints = [0, 1, 2, mod - 1, mod - 2, (1 << 253) - 1, mod - 3, 0x1FFFFFFF, 1 << 29, (1 << 232) - 1] + [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(60)]
    vals = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in ints], dtype=np.uint64)
    n = len(ints)
    mul = lambda x, y: O.fq_mul(x.reshape(1, 4), y.reshape(1, 4))[0]
    sub = lambda x, y: O.fq_sub(x.reshape(1, 4), y.reshape(1, 4))[0]
    add = lambda x, y: O.fq_add(x.reshape(1, 4), y.reshape(1, 4))[0]
    for i in range(n):
        a, b, c, d = vals[i], vals[(7 * i + 3) % n], vals[n - 1 - i], vals[(5 * i + 11) % n]
        assert np.array_equal(ffi.host_fq_limb_op(0, a, b), mul(a, b)), i
        assert np.array_equal(ffi.host_fq_limb_op(1, a), mul(a, a)), i
        assert np.array_equal(ffi.host_fq_limb_op(2, a, b, c, d), add(mul(a, b), mul(c, d))), i
        assert np.array_equal(ffi.host_fq_limb_op(3, a, b, c), mul(sub(a, b), c)), i
        assert np.array_equal(ffi.host_fq_limb_op(4, a, b, c, d), mul(sub(sub(a, b), add(c, c)), d)), i


def test_limb_form_bucket_sum_matches_oracle_on_host():
    """g1xl_add_mixed (XYZZ accumulator in limb form) over lists of affine points: random points, the same point twice and three times in a
    row (the doubling branch, then an ordinary addition onto 2P), P followed by -P (identity, then a fresh start), points at infinity in
    the list, negated entries -- equal as group elements to the oracle's sum."""
    rng = np.random.default_rng(62)
    g = O.g1_generator()
    jac = [O.g1_scalar_mul(g, np.array([int(rng.integers(1, 2**62)), 0, 0, 0], dtype=np.uint64)) for _ in range(12)]
    aff = O.g1_to_affine(np.stack(jac))
    inf = np.zeros(8, dtype=np.uint64)

    def check(order, negate=None):
        pts = np.stack([aff[k] if k >= 0 else inf for k in order]) if order else np.zeros((0, 8), dtype=np.uint64)
        neg = np.zeros(len(order), dtype=np.uint8) if negate is None else np.array(negate, dtype=np.uint8)
        want = O.g1_identity()
        for k, s in zip(order, neg):
            if k >= 0:
                want = O.g1_add(want, O.g1_neg(jac[k]) if s else jac[k])
        assert O.g1_eq(ffi.host_g1_sum_limb_form(pts, neg), want), (order, negate)

    check([])
    check([-1, -1])
    check([3])
    check(list(range(12)))
    check([0, 0])                      # P + P
    check([0, 0, 0, 1])                # 2P + P, then another point
    check([2, 2], [0, 1])              # P + (-P) = identity
    check([2, 2, 5, 5, 5], [0, 1, 0, 0, 0])  # identity, then a fresh start with a doubling
    check([4, -1, 4, 7, -1, 7, 7], [1, 0, 1, 0, 0, 0, 1])
    check([int(k) for k in rng.integers(0, 12, size=200)], [int(b) for b in rng.integers(0, 2, size=200)])


@pytest.mark.parametrize("c", [10, 13, 19, 23, 26])
def test_fixed_base_digit_recoding_reconstructs_the_scalar(c):
    """k_fx_digits' recoding on the host: sum_w digit_w 2^(c w) = s mod r (scalars above r / 2 come out negated), every signed digit is
    within +-2^(c-1), every magnitude addresses a bucket below the table set's bucket count, and the worst cases (r - 1, (r - 1) / 2 and
    its neighbours, all-ones windows that carry through every window) are among the samples."""
    R = O.R_MOD
    rng = np.random.default_rng(70 + c)
    samples = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R - 1) // 2 + 1, (R + 1) // 2 + 1, (1 << 253) - 1, (1 << 252), (1 << (c - 1)), (1 << (c - 1)) + 1, (1 << c) - 1,
               sum(((1 << c) - 1) << (c * w) for w in range(253 // c)) % R, sum((1 << (c - 1)) << (c * w) for w in range(253 // c)) % R]
    samples += [int.from_bytes(rng.bytes(32), "little") % R for _ in range(300)]
    for s in samples:
        digits, buckets = ffi.host_fx_digits(O.to_mont([s])[0], c)
        assert len(digits) == -(-253 // c)
        assert sum(d << (c * w) for w, d in enumerate(digits)) % R == s, hex(s)
        assert all(abs(d) <= 1 << (c - 1) for d in digits[:-1]) and all(abs(d) <= buckets for d in digits), hex(s)
        assert buckets >= 1 << (c - 1)
    if c == 23:
        assert buckets == 6342813  # ((r - 1) / 2 >> 230) + 1: DESIGN.md section 3.5c


def test_small_scalar_accumulator_built_for_the_host():
    """small_scalar.hip.h (field x machine-integer products summed unreduced in 13 limbs, one REDC) on the host: equal to the plain sum of
    reduced products for i128 scalars incl. the corners, to the oracle's restatement of FrSmallScalarAccumulator where that one's five limbs
    suffice, and still exact for 5000 full-range 128-bit terms (headroom 2^34 terms)"""
    rng = np.random.default_rng(90)
    R = O.R_MOD
    for n in (0, 1, 4, 300, 5000):
        values = rand_fr(max(n, 1), 91 + n)[:n]
        scalars = [int(rng.integers(-2**62, 2**62)) * int(rng.integers(0, 2**63)) for _ in range(n)]
        scalars[: min(n, 4)] = [0, -(2**127), 2**127 - 1, -1][: min(n, 4)]
        v_int = O.from_mont(values) if n else []
        want = sum(v * s for v, s in zip(v_int, scalars)) % R
        got = ffi.host_small_scalar_dot(values if n else np.zeros((0, 4), dtype=np.uint64), scalars)
        assert O.from_mont(got.reshape(1, 4))[0] == want, n
    values = rand_fr(200, 95)
    small = rng.integers(-2**48, 2**48, size=200, dtype=np.int64)
    assert np.array_equal(ffi.host_small_scalar_dot(values, [int(x) for x in small]), O.small_scalar_accumulate(values, small))


def test_block_cyclic_ownership_partitions_every_prefix():
    """Block-cyclic term assignment of the sharded PCS legs (term i belongs to rank (i / block) % world): the owned-term count of
    every prefix matches a direct count, the ranks' counts add up to the prefix, and the rank's terms of a prefix are a PREFIX of
    its compact index order (what lets one set of per-rank window tables serve every level of an opening)."""
    from jolt_amd import ffi
    for world, block in [(1, 4), (2, 8), (4, 4), (8, 2), (3, 5)]:
        total = block * world * 3
        owner = (np.arange(total) // block) % world
        for n in list(range(0, total + 1, 3)) + [total]:
            counts = [ffi.host_owned_terms(n, block, g, world) for g in range(world)]
            assert counts == [int((owner[:n] == g).sum()) for g in range(world)], (world, block, n)
            assert sum(counts) == n
        with pytest.raises(ffi.JoltError):
            ffi.host_owned_terms(5, 0, 0, world)
        with pytest.raises(ffi.JoltError):
            ffi.host_owned_terms(5, block, world, world)


def test_subtree_ownership_hooks_agree_with_the_specification():
    """jolt_host_subtree_term_index / _owned_terms (term_map.hip.h, the maps the device kernels use) against tests/subtree_model.py."""
    import subtree_model as M
    from jolt_amd import ffi
    for world in (1, 2, 4, 8):
        gamma = world.bit_length() - 1
        for g in range(world):
            for c in list(range(0, 300, 7)) + [4096, 4097, (1 << 26) - 1, 1 << 26]:
                assert ffi.host_subtree_term_index(c, g, world) == M.insert(c, g, gamma)
            for n in list(range(0, 70)) + [255, 256, 257, (1 << 20) - 1, 1 << 20, (1 << 29) - 1, 1 << 29]:
                assert ffi.host_subtree_owned_terms(n, g, world) == M.owned(n, g, gamma), (n, g, world)
    with pytest.raises(ffi.JoltError):
        ffi.host_subtree_owned_terms(8, 0, 3)  # the world size must be a power of two
    with pytest.raises(ffi.JoltError):
        ffi.host_subtree_term_index(1, 4, 4)


def test_host_booleanity_address_rounds_match_the_oracle():
    """jolt_host_booleanity_address_{round,bind} (the K-domain loop of stage 6a the stage driver runs on the host) against oracle/onehot.c's restatement"""
    import oracle_lib as O
    from jolt_amd import ffi
    from util import rand_challenge, rand_fr
    n_polys, log_k = 5, 4
    masses = rand_fr(n_polys * (1 << log_k), 71).reshape(n_polys, 1 << log_k, 4)
    gamma, ref = rand_fr(1, 72)[0], rand_fr(log_k, 73)
    dev, orc = ffi.HostBooleanityAddress(masses, gamma, ref), O.BooleanityAddress(masses, gamma, ref)
    assert np.array_equal(dev.weights, orc.weights) and np.array_equal(dev.eq, orc.eq)
    for rnd in range(log_k):
        assert np.array_equal(dev.round(), orc.round()), rnd
        r = rand_challenge(80 + rnd, shifted=(rnd % 2 == 0))
        dev.bind(r)
        orc.bind(r)
    assert np.array_equal(dev.intermediate(), orc.intermediate())


def test_host_transcript_is_the_oracles_byte_for_byte():
    """jolt_host_transcript_* (the deterministic test transcript above the ABI, re-implemented in host_mirror.hip from the text of oracle/mock_transcript.h): field
    elements, raw bytes of every length around the absorb block size (what jolt_host_hyperkzg_open appends for compressed points), both challenge shapes, interleaved --
    the same challenges as the oracle's transcript from the same label"""
    import oracle_lib as O
    from jolt_amd import ffi
    from util import rand_fr
    rng = np.random.default_rng(5)
    for label in (0, 7, 2**63 + 11):
        mine, theirs = ffi.HostTranscript(label), O.MockTranscript(label)
        vals = rand_fr(9, 400 + label % 97)
        for step in range(40):
            kind = step % 4
            if kind == 0:
                k = int(rng.integers(1, 4))
                mine.append(vals[:k])
                for v in vals[:k]:
                    theirs.append_fr(v)
            elif kind == 1:
                data = bytes(rng.integers(0, 256, size=int(rng.integers(1, 100)), dtype=np.uint8))
                mine.append_bytes(data)
                theirs.append_bytes(data)
            elif kind == 2:
                assert np.array_equal(mine.challenge(), theirs.challenge()), (label, step)
            else:
                assert np.array_equal(mine.challenge(full_width=True), theirs.challenge_scalar()), (label, step)
        mine.close()
