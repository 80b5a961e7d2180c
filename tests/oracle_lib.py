"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Field elements travel as numpy uint64 arrays of shape (..., 4): the 4 little-endian Montgomery limbs of
jolt_field::Fr (reference crates/jolt-field/src/bn254/mod.rs:37-43).  G1 points are (..., 12) uint64
(Jacobian x,y,z of Montgomery Fq limbs = ark_bn254::G1Projective layout).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
MONT_R = 1 << 256


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fr_array(n):
    return np.zeros((n, 4), dtype=np.uint64)


# ---------------------------------------------------------------- Python <-> limb conversions (big-int side)
def int_to_limbs(v, n=4):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs_to_int(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def to_mont(values, mod=R_MOD):
    """canonical python ints -> Montgomery limb array (computed with Python big ints, not the oracle)."""
    out = np.zeros((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        out[i] = int_to_limbs((v % mod) * MONT_R % mod)
    return out


def from_mont(arr, mod=R_MOD):
    rinv = pow(MONT_R, -1, mod)
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [limbs_to_int(row) * rinv % mod for row in arr]


# ---------------------------------------------------------------- vector field ops
def _binop(name, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    o = np.zeros_like(a)
    getattr(lib(), name)(_p(a), _p(b), _p(o), C.c_size_t(a.shape[0]))
    return o


def fr_add(a, b): return _binop("orc_fr_add_vec", a, b)
def fr_sub(a, b): return _binop("orc_fr_sub_vec", a, b)
def fr_mul(a, b): return _binop("orc_fr_mul_vec", a, b)
def fq_add(a, b): return _binop("orc_fq_add_vec", a, b)
def fq_sub(a, b): return _binop("orc_fq_sub_vec", a, b)
def fq_mul(a, b): return _binop("orc_fq_mul_vec", a, b)


def _unop(name, a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    o = np.zeros_like(a)
    getattr(lib(), name)(_p(a), _p(o), C.c_size_t(a.shape[0]))
    return o


def fr_neg(a): return _unop("orc_fr_neg_vec", a)
def fr_inv(a): return _unop("orc_fr_inv_vec", a)
def fq_inv(a): return _unop("orc_fq_inv_vec", a)
def fr_to_canonical(a): return _unop("orc_fr_to_canonical_vec", a)
def fr_from_canonical(a): return _unop("orc_fr_from_canonical_vec", a)
def fq_to_canonical(a): return _unop("orc_fq_to_canonical_vec", a)
def fq_from_canonical(a): return _unop("orc_fq_from_canonical_vec", a)


def fr_from_u64(vals):
    v = np.ascontiguousarray(vals, dtype=np.uint64)
    o = fr_array(v.shape[0])
    lib().orc_fr_from_u64_vec(_p(v), _p(o), C.c_size_t(v.shape[0]))
    return o


def fr_from_i64(vals):
    v = np.ascontiguousarray(vals, dtype=np.int64)
    o = fr_array(v.shape[0])
    lib().orc_fr_from_i64_vec(_p(v), _p(o), C.c_size_t(v.shape[0]))
    return o


def fr_from_u128(v):
    o = fr_array(1)
    lib().orc_fr_from_u128(C.c_uint64(v & (2**64 - 1)), C.c_uint64(v >> 64), _p(o))
    return o[0]


def fr_from_i128(v):
    m = abs(v)
    o = fr_array(1)
    lib().orc_fr_from_i128(C.c_uint64(m & (2**64 - 1)), C.c_uint64(m >> 64), C.c_int(1 if v < 0 else 0), _p(o))
    return o[0]


def fr_mul_u64(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = fr_array(1)
    lib().orc_fr_mul_u64(_p(a), C.c_uint64(b), _p(o))
    return o[0]


def fr_mul_u128(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = fr_array(1)
    lib().orc_fr_mul_u128(_p(a), C.c_uint64(b & (2**64 - 1)), C.c_uint64(b >> 64), _p(o))
    return o[0]


def fr_mul_pow_2(a, k):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    o = fr_array(1)
    lib().orc_fr_mul_pow_2(_p(a), C.c_uint(k), _p(o))
    return o[0]


def _from_bytes(name, b):
    buf = (C.c_uint8 * len(b)).from_buffer_copy(bytes(b)) if len(b) else (C.c_uint8 * 1)()
    o = fr_array(1)
    getattr(lib(), name)(buf, C.c_size_t(len(b)), _p(o))
    return o[0]


def fr_from_bytes_le_reduced(b): return _from_bytes("orc_fr_from_bytes_le_reduced", b)
def fr_from_challenge_bytes(b): return _from_bytes("orc_fr_from_challenge_bytes", b)
def fr_from_scalar_challenge_bytes(b): return _from_bytes("orc_fr_from_scalar_challenge_bytes", b)


def fr_to_bytes_le(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = (C.c_uint8 * 32)()
    lib().orc_fr_to_bytes_le(_p(a), out)
    return bytes(out)


def fr_from_montgomery_reduce(limbs):
    l = np.ascontiguousarray(limbs, dtype=np.uint64)
    o = fr_array(1)
    lib().orc_fr_from_montgomery_reduce(_p(l), C.c_size_t(l.shape[0]), _p(o))
    return o[0]


def wide_accumulate(a, b, adds=None):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    adds = fr_array(0) if adds is None else np.ascontiguousarray(adds, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1)
    lib().orc_wide_accumulate(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(adds), C.c_size_t(adds.shape[0]), _p(o))
    return o[0]


# ---------------------------------------------------------------- poly
def bind_high_to_low(table, r):
    t = np.array(table, dtype=np.uint64, copy=True).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    lib().orc_bind_high_to_low(_p(t), C.c_size_t(t.shape[0]), _p(r))
    return t[: t.shape[0] // 2].copy()


def bind_low_to_high(table, r):
    t = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    o = fr_array(t.shape[0] // 2)
    lib().orc_bind_low_to_high(_p(t), C.c_size_t(t.shape[0]), _p(r), _p(o))
    return o


def bind_to_field_u64(table, r):
    t = np.ascontiguousarray(table, dtype=np.uint64)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    o = fr_array(t.shape[0] // 2)
    lib().orc_bind_to_field_u64(_p(t), C.c_size_t(t.shape[0]), _p(r), _p(o))
    return o


def eq_evals(r, scale=None):
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    n = r.shape[0]
    o = fr_array(1 << n)
    s = None if scale is None else np.ascontiguousarray(scale, dtype=np.uint64)
    lib().orc_eq_evals(_p(r), C.c_size_t(n), _p(s) if s is not None else None, _p(o))
    return o


def eq_evaluations(r):
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1 << r.shape[0])
    lib().orc_eq_evaluations(_p(r), C.c_size_t(r.shape[0]), _p(o))
    return o


def eq_evals_aligned_block(r, start, block):
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(block)
    lib().orc_eq_evals_aligned_block(_p(r), C.c_size_t(r.shape[0]), C.c_size_t(start), C.c_size_t(block), _p(o))
    return o


def eq_mle(x, y):
    x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    y = np.ascontiguousarray(y, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1)
    lib().orc_eq_mle(_p(x), _p(y), C.c_size_t(x.shape[0]), _p(o))
    return o[0]


def poly_evaluate(evals, point):
    e = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1)
    lib().orc_poly_evaluate(_p(e), C.c_size_t(p.shape[0]), _p(p), _p(o))
    return o[0]


def lt_evals(r):
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1 << r.shape[0])
    lib().orc_lt_evals(_p(r), C.c_size_t(r.shape[0]), _p(o))
    return o


def eq_plus_one_evals(r, scale=None):
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    n = r.shape[0]
    e, e1 = fr_array(1 << n), fr_array(1 << n)
    s = None if scale is None else np.ascontiguousarray(scale, dtype=np.uint64)
    lib().orc_eq_plus_one_evals(_p(r), C.c_size_t(n), _p(s) if s is not None else None, _p(e), _p(e1))
    return e, e1


def univariate_from_evals(evals):
    e = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(e.shape[0])
    lib().orc_univariate_from_evals(_p(e), C.c_size_t(e.shape[0]), _p(o))
    return o


def univariate_evaluate(coeffs, x):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    o = fr_array(1)
    lib().orc_univariate_evaluate(_p(c), C.c_size_t(c.shape[0]), _p(x), _p(o))
    return o[0]


def split_eq_current_dims(n, bound):
    a, b = C.c_size_t(), C.c_size_t()
    lib().orc_split_eq_current_dims(C.c_size_t(n), C.c_size_t(bound), C.byref(a), C.byref(b))
    return a.value, b.value


def gruen_poly_deg_3(scalar, point_i, q0, qinf, claim):
    args = [np.ascontiguousarray(x, dtype=np.uint64) for x in (scalar, point_i, q0, qinf, claim)]
    o = fr_array(4)
    rc = lib().orc_gruen_poly_deg_3(*[_p(a) for a in args], _p(o))
    assert rc == 0
    return o


# ---------------------------------------------------------------- sumcheck members
ORDER_LOW_TO_HIGH, ORDER_HIGH_TO_LOW = 0, 1


class Member:
    """Oracle twin of a jolt_sumcheck::ProveRounds member."""

    def __init__(self, handle, degree, n_tables, gruen=False):
        self.h, self.degree, self.n_tables, self.gruen = handle, degree, n_tables, gruen

    @classmethod
    def expr(cls, tables, terms, degree, order=ORDER_LOW_TO_HIGH, skip_one=False):
        """terms = [(coeff_limbs, [table indices...]), ...]"""
        L = lib()
        L.orc_member_create_expr.restype = C.c_void_p
        tabs = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
        ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
        offs, facs = [0], []
        for _, f in terms:
            facs.extend(f)
            offs.append(len(facs))
        offs = np.array(offs, dtype=np.uint32)
        facs_a = np.array(facs if facs else [0], dtype=np.uint32)
        coeffs = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c, _ in terms])).reshape(-1, 4)
        h = L.orc_member_create_expr(ptrs, C.c_uint32(len(tabs)), C.c_size_t(tabs[0].shape[0]), C.c_uint32(len(terms)),
                                     _p(offs), _p(facs_a), _p(coeffs), C.c_uint32(degree), C.c_int(order),
                                     C.c_int(1 if skip_one else 0))
        return cls(C.c_void_p(h), degree, len(tabs))

    @classmethod
    def gruen_product(cls, a, b, w, scale=None):
        L = lib()
        L.orc_member_create_gruen_product.restype = C.c_void_p
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4)
        s = None if scale is None else np.ascontiguousarray(scale, dtype=np.uint64)
        h = L.orc_member_create_gruen_product(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(w), _p(s) if s is not None else None)
        return cls(C.c_void_p(h), 3, 2, gruen=True)

    def num_rounds(self):
        lib().orc_member_num_rounds.restype = C.c_size_t
        return lib().orc_member_num_rounds(self.h)

    def prove_round(self, bind, previous_claim):
        o = fr_array(self.degree + 1)
        b = None if bind is None else np.ascontiguousarray(bind, dtype=np.uint64)
        pc = np.ascontiguousarray(previous_claim, dtype=np.uint64)
        rc = lib().orc_member_prove_round(self.h, _p(b) if b is not None else None, _p(pc), _p(o))
        if rc != 0:
            raise RuntimeError(f"oracle prove_round failed rc={rc}")
        return o

    def round_sums(self, bind, shard_scale=None):
        o = fr_array(2 if self.gruen else self.degree + 1)
        b = None if bind is None else np.ascontiguousarray(bind, dtype=np.uint64)
        sc = None if shard_scale is None else np.ascontiguousarray(shard_scale, dtype=np.uint64)
        rc = lib().orc_member_round_sums(self.h, _p(b) if b is not None else None, _p(sc) if sc is not None else None, _p(o))
        assert rc == 0
        return o

    def finish_rounds(self, bind):
        b = np.ascontiguousarray(bind, dtype=np.uint64)
        lib().orc_member_finish_rounds(self.h, _p(b))

    def final_values(self):
        o = fr_array(self.n_tables + (1 if self.gruen else 0))
        rc = lib().orc_member_final_values(self.h, _p(o))
        assert rc == 0
        return o

    def current_tables(self):
        """The partially bound evaluations of every table: (n_tables, current_len, 4)."""
        lib().orc_member_current_len.restype = C.c_size_t
        n = lib().orc_member_current_len(self.h)
        o = fr_array(self.n_tables * n).reshape(self.n_tables, n, 4)
        for t in range(self.n_tables):
            rc = lib().orc_member_copy_table(self.h, C.c_uint32(t), _p(o[t]), C.c_size_t(n))
            assert rc == 0
        return o

    def input_claim(self):
        o = fr_array(1)
        lib().orc_member_input_claim(self.h, _p(o))
        return o[0]

    def close(self):
        if self.h:
            lib().orc_member_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove_batch(members, input_claims, coefficients, offsets, max_num_vars, max_degree, label=0, challenge_mode=0):
    n = len(members)
    hs = (C.c_void_p * n)(*[m.h for m in members])
    ic = np.ascontiguousarray(np.stack(input_claims), dtype=np.uint64).reshape(-1, 4)
    co = np.ascontiguousarray(np.stack(coefficients), dtype=np.uint64).reshape(-1, 4)
    offs = (C.c_size_t * n)(*offsets)
    polys = fr_array(max_num_vars * (max_degree + 1))
    chal = fr_array(max_num_vars)
    mclaims = fr_array(n)
    final = fr_array(1)
    rc = lib().orc_prove_batch(hs, C.c_size_t(n), _p(ic), _p(co), offs, C.c_size_t(max_num_vars), C.c_size_t(max_degree),
                               C.c_uint64(label), C.c_int(challenge_mode), _p(polys), _p(chal), _p(mclaims), _p(final))
    if rc != 0:
        raise RuntimeError(f"oracle prove_batch failed rc={rc}")
    return dict(polys=polys.reshape(max_num_vars, max_degree + 1, 4), challenges=chal, member_claims=mclaims, final_claim=final[0])


def triple_product_round_evals(a, b, c):
    a, b, c = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (a, b, c)]
    o = fr_array(3)
    lib().orc_triple_product_round_evals(_p(a), _p(b), _p(c), C.c_size_t(a.shape[0]), _p(o))
    return o


TRANSCRIPT_MOCK, TRANSCRIPT_BLAKE2B, TRANSCRIPT_KECCAK, TRANSCRIPT_BLAKE2B_SPONGE = 0, 1 << 62, 2 << 62, 3 << 62  # the two top bits of a transcript label select the engine (oracle/mock_transcript.h)


class MockTranscript:
    """The oracle's transcript (oracle/mock_transcript.h): the engine is selected by the label's two top bits -- 0 the deterministic stand-in, TRANSCRIPT_BLAKE2B the
    reference's LegacyBlake2bTranscript, TRANSCRIPT_KECCAK its KeccakTranscript -- or explicitly with a byte label (`kind` 1 / 2, the reference's `Transcript::new(b"...")`)."""

    def __init__(self, label=0, kind=None):
        L = lib()
        L.orc_mt_sizeof.restype = C.c_size_t
        self.s = (C.c_uint8 * L.orc_mt_sizeof())()
        if isinstance(label, (bytes, bytearray)):
            buf = (C.c_uint8 * max(1, len(label))).from_buffer_copy(bytes(label) if len(label) else b"\0")
            if L.orc_mt_init_bytes(self.s, C.c_uint64(kind), buf, C.c_size_t(len(label))) != 0:
                raise ValueError("transcript label longer than 32 bytes or unknown kind")
        else:
            L.orc_mt_init(self.s, C.c_uint64(label))

    def append_bytes(self, b):
        buf = (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0")
        lib().orc_mt_append_bytes(self.s, buf, C.c_size_t(len(b)))

    def append_fr(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        lib().orc_mt_append_fr(self.s, _p(a))

    def append_label(self, label):
        lib().orc_mt_append_label(self.s, C.c_char_p(label))

    def append_label_with_count(self, label, count):
        lib().orc_mt_append_label_with_count(self.s, C.c_char_p(label), C.c_uint64(count))

    def append_u64_word(self, v):
        lib().orc_mt_append_u64_word(self.s, C.c_uint64(v))

    def append_round_poly(self, coeffs, label=b"sumcheck_poly"):
        c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        lib().orc_mt_append_round_poly(self.s, C.c_char_p(label), _p(c), C.c_size_t(c.shape[0]))

    def state(self):
        out = (C.c_uint8 * 32)()
        lib().orc_mt_state(self.s, out)
        return bytes(out)

    def challenge(self):
        o = fr_array(1)
        lib().orc_mt_challenge(self.s, _p(o))
        return o[0]

    def challenge_scalar(self):
        o = fr_array(1)
        lib().orc_mt_challenge_scalar(self.s, _p(o))
        return o[0]


def blake2b_digest(data, outlen=32):
    out = (C.c_uint8 * outlen)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) if len(data) else b"\0")
    lib().orc_blake2b_digest(buf, C.c_size_t(len(data)), C.c_size_t(outlen), out)
    return bytes(out)


# ---------------------------------------------------------------- G1 / MSM / HyperKZG
def g1_array(n):
    return np.zeros((n, 12), dtype=np.uint64)


def g1_generator():
    o = g1_array(1)
    lib().orc_g1_generator(_p(o))
    return o[0]


def g1_identity():
    o = g1_array(1)
    lib().orc_g1_identity(_p(o))
    return o[0]


def _g1(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def g1_add(p, q):
    o = g1_array(1)
    lib().orc_g1_add(_p(_g1(p)), _p(_g1(q)), _p(o))
    return o[0]


def g1_double(p):
    o = g1_array(1)
    lib().orc_g1_double(_p(_g1(p)), _p(o))
    return o[0]


def g1_neg(p):
    o = g1_array(1)
    lib().orc_g1_neg(_p(_g1(p)), _p(o))
    return o[0]


def g1_scalar_mul(p, s):
    o = g1_array(1)
    lib().orc_g1_scalar_mul(_p(_g1(p)), _p(np.ascontiguousarray(s, dtype=np.uint64)), _p(o))
    return o[0]


def g1_eq(p, q):
    return bool(lib().orc_g1_eq(_p(_g1(p)), _p(_g1(q))))


def g1_is_identity(p):
    return bool(lib().orc_g1_is_identity(_p(_g1(p))))


def g1_on_curve(p):
    return bool(lib().orc_g1_on_curve(_p(_g1(p))))


def g1_to_affine(ps):
    ps = _g1(ps).reshape(-1, 12)
    o = np.zeros((ps.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_to_affine_vec(_p(ps), _p(o), C.c_size_t(ps.shape[0]))
    return o


def g1_serialize_compressed(p):
    out = (C.c_uint8 * 32)()
    lib().orc_g1_serialize_compressed(_p(_g1(p)), out)
    return bytes(out)


def g1_msm_naive(bases, scalars):
    b = _g1(bases).reshape(-1, 12)
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    o = g1_array(1)
    lib().orc_g1_msm_naive(_p(b), _p(s), C.c_size_t(b.shape[0]), _p(o))
    return o[0]


def g1_msm_pippenger(bases, scalars):
    b = _g1(bases).reshape(-1, 12)
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert b.shape[0] == s.shape[0], "msm: bases/scalars length mismatch"
    o = g1_array(1)
    lib().orc_g1_msm_pippenger(_p(b), _p(s), C.c_size_t(b.shape[0]), _p(o))
    return o[0]


def dory_commit_rows(bases, values, kind, row_width):
    """streaming.rs feed_u64 / feed_i128 per row: values = uint64 / int64 array, or (n,2) uint64 for i128."""
    b = _g1(bases).reshape(-1, 12)
    v = np.ascontiguousarray(values)
    count = v.shape[0]
    rows = count // row_width
    o = g1_array(max(rows, 1))
    lib().orc_dory_commit_rows(_p(b), v.ctypes.data_as(C.c_void_p), C.c_int({"u64": 0, "i64": 1, "i128": 2}[kind]), C.c_size_t(count), C.c_size_t(row_width), _p(o))
    return o[:rows]


def dory_onehot_chunk(bases, idx, k):
    b = _g1(bases).reshape(-1, 12)
    i = np.ascontiguousarray(idx, dtype=np.uint8)
    o = g1_array(k)
    lib().orc_dory_onehot_chunk(_p(b), i.ctypes.data_as(C.c_void_p), C.c_size_t(k), C.c_size_t(i.shape[0]), _p(o))
    return o


def srs_setup_from_secret(beta, count):
    o = g1_array(count)
    lib().orc_srs_setup_from_secret(_p(np.ascontiguousarray(beta, dtype=np.uint64)), C.c_size_t(count), _p(o))
    return o


def kzg_commit(coeffs, srs):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    s = _g1(srs).reshape(-1, 12)
    o = g1_array(1)
    rc = lib().orc_kzg_commit(_p(c), C.c_size_t(c.shape[0]), _p(s), C.c_size_t(s.shape[0]), _p(o))
    if rc != 0:
        raise ValueError("SrsTooSmall")
    return o[0]


def kzg_witness_polynomial(f, u):
    f = np.ascontiguousarray(f, dtype=np.uint64).reshape(-1, 4)
    h = fr_array(max(f.shape[0] - 1, 0))
    lib().orc_kzg_witness_polynomial(_p(f), C.c_size_t(f.shape[0]), _p(np.ascontiguousarray(u, dtype=np.uint64)), _p(h))
    return h


def kzg_eval_univariate(coeffs, u):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1)
    lib().orc_kzg_eval_univariate(_p(c), C.c_size_t(c.shape[0]), _p(np.ascontiguousarray(u, dtype=np.uint64)), _p(o))
    return o[0]


def hyperkzg_fold_polynomials(evals, point):
    e = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    ell = p.shape[0]
    out = fr_array(2 * e.shape[0])
    lib().orc_hyperkzg_fold_polynomials(_p(e), C.c_size_t(ell), _p(p), _p(out))
    polys, off, ln = [], 0, e.shape[0]
    for _ in range(ell):
        polys.append(out[off:off + ln].copy())
        off += ln
        ln //= 2
    return polys


def hyperkzg_open(srs, evals, point, label=0):
    s = _g1(srs).reshape(-1, 12)
    e = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    ell = p.shape[0]
    com, w, v, ch = g1_array(max(ell - 1, 1)), g1_array(3), fr_array(3 * ell), fr_array(3)
    rc = lib().orc_hyperkzg_open(_p(s), C.c_size_t(s.shape[0]), _p(e), C.c_size_t(ell), _p(p), C.c_uint64(label),
                                 _p(com), _p(w), _p(v), _p(ch))
    if rc != 0:
        raise ValueError(f"hyperkzg open failed rc={rc}")
    return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)


# ---------------------------------------------------------------- cpu baseline drivers (bench.py only)
def baseline_threads():
    return lib().orc_baseline_threads()


def baseline_set_threads(n):
    lib().orc_baseline_set_threads(C.c_int(n))


def baseline_bind_low_to_high(table, r, out):
    lib().orc_baseline_bind_low_to_high(_p(table), C.c_size_t(table.shape[0]), _p(np.ascontiguousarray(r, dtype=np.uint64)), _p(out))


def baseline_round_evals(tables, terms, degree):
    tabs = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
    ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
    offs, facs = [0], []
    for _, f in terms:
        facs.extend(f)
        offs.append(len(facs))
    offs = np.array(offs, dtype=np.uint32)
    facs_a = np.array(facs if facs else [0], dtype=np.uint32)
    coeffs = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64) for c, _ in terms])).reshape(-1, 4)
    o = fr_array(degree)
    lib().orc_baseline_round_evals(ptrs, C.c_uint32(len(tabs)), C.c_size_t(tabs[0].shape[0]), C.c_uint32(len(terms)),
                                   _p(offs), _p(facs_a), _p(coeffs), C.c_uint32(degree), _p(o))
    return o


def baseline_member_sumcheck(tables, groups, degree, challenges):
    """groups in resolved LC form: [[(const|None, [(coeff, table_idx), ...]), ...], ...]"""
    tabs = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
    ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
    goff, foff, consts, has_c, ltab, lcoef, lone = [0], [0], [], [], [], [], []
    one = to_mont([1])[0]
    zero = np.zeros(4, dtype=np.uint64)
    for g in groups:
        for const, entries in g:
            consts.append(zero if const is None else np.asarray(const, dtype=np.uint64))
            has_c.append(0 if const is None else 1)
            for c, ti in entries:
                c = np.asarray(c, dtype=np.uint64)
                lcoef.append(c)
                lone.append(1 if np.array_equal(c, one) else 0)
                ltab.append(ti)
            foff.append(len(ltab))
        goff.append(len(consts))
    goff, foff = np.array(goff, dtype=np.uint32), np.array(foff, dtype=np.uint32)
    consts_a = np.ascontiguousarray(np.stack(consts))
    has_a = np.array(has_c, dtype=np.uint32)
    ltab_a, lone_a = np.array(ltab, dtype=np.uint32), np.array(lone, dtype=np.uint32)
    lcoef_a = np.ascontiguousarray(np.stack(lcoef))
    ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
    out = fr_array(8)
    n_rounds = 0
    while (1 << n_rounds) < tabs[0].shape[0]:
        n_rounds += 1
    lib().orc_baseline_member_sumcheck(ptrs, C.c_uint32(len(tabs)), C.c_size_t(tabs[0].shape[0]), C.c_uint32(len(groups)), _p(goff),
                                       _p(foff), _p(consts_a), _p(has_a), _p(ltab_a), _p(lcoef_a), _p(lone_a), C.c_uint32(degree),
                                       _p(ch), C.c_size_t(n_rounds), _p(out))
    return out[:degree]


def baseline_prepare_bases(bases):
    """affine copies of a base array, cached inside the oracle by address: keep `bases` alive while the parallel MSM is in use"""
    b = _g1(bases).reshape(-1, 12)
    lib().orc_baseline_prepare_bases(_p(b), C.c_size_t(b.shape[0]))
    return b


def baseline_set_max_window(c):
    lib().orc_baseline_set_max_window(C.c_uint(c))


def baseline_use_parallel_msm(on):
    lib().orc_baseline_use_parallel_msm(C.c_int(1 if on else 0))


def baseline_msm(bases, scalars):
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    o = g1_array(1)
    lib().orc_baseline_msm(_p(bases), _p(s), C.c_size_t(s.shape[0]), _p(o))
    return o[0]


def baseline_msm_many(bases, scalar_arrays):
    """several MSMs over prefixes of the prepared bases as one pool of (msm, window, chunk) tasks"""
    ss = [np.ascontiguousarray(s, dtype=np.uint64).reshape(-1, 4) for s in scalar_arrays]
    ptrs = (C.c_void_p * len(ss))(*[s.ctypes.data for s in ss])
    lens = (C.c_size_t * len(ss))(*[s.shape[0] for s in ss])
    o = g1_array(len(ss))
    lib().orc_baseline_msm_many(_p(bases), ptrs, lens, C.c_size_t(len(ss)), _p(o))
    return o


def baseline_grid_onehot_sums(bases, idx):
    """commitments of one-hot columns on the K x T grid: idx (n_polys, cycles) hot addresses (0xFF = cold), cycle-major placement"""
    i = np.ascontiguousarray(idx, dtype=np.uint8)
    n_polys, cycles = i.shape
    o = g1_array(n_polys)
    lib().orc_baseline_grid_onehot_sums(_p(bases), i.ctypes.data_as(C.c_void_p), C.c_size_t(n_polys), C.c_size_t(cycles), _p(o))
    return o


def baseline_grid_joint(idx, k_grid, scalars, dense, dense_scalars):
    i = np.ascontiguousarray(idx, dtype=np.uint8)
    n_polys, cycles = i.shape
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    dn = [np.ascontiguousarray(d, dtype=np.uint64).reshape(-1, 4) for d in dense]
    ptrs = (C.c_void_p * max(len(dn), 1))(*[d.ctypes.data for d in dn])
    dsc = np.ascontiguousarray(dense_scalars, dtype=np.uint64).reshape(-1, 4) if len(dn) else fr_array(1)
    out = fr_array(k_grid * cycles)
    lib().orc_baseline_grid_joint(i.ctypes.data_as(C.c_void_p), C.c_uint32(n_polys), C.c_size_t(cycles), C.c_uint32(k_grid), _p(sc), ptrs,
                                  C.c_uint32(len(dn)), _p(dsc), _p(out))
    return out


# ---- one-hot selector columns (oracle/onehot.c) ---------------------------------------------------------------------
def onehot_values(table, width, k_entries, col, cycles):
    table = np.ascontiguousarray(table, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.uint8)
    out = fr_array(cycles // width)
    lib().orc_onehot_values(_p(table), C.c_size_t(width), C.c_size_t(k_entries), col.ctypes.data_as(C.c_void_p), C.c_size_t(cycles), _p(out))
    return out


def onehot_double_branches(table, challenge):
    table = np.ascontiguousarray(table, dtype=np.uint64)
    out = fr_array(2 * table.shape[0])
    lib().orc_onehot_double_branches(_p(table), C.c_size_t(table.shape[0]), _p(np.ascontiguousarray(challenge, dtype=np.uint64)), _p(out))
    return out


def onehot_pushforward(col, k_entries, w):
    col = np.ascontiguousarray(col, dtype=np.uint8)
    w = np.ascontiguousarray(w, dtype=np.uint64)
    out = fr_array(k_entries)
    lib().orc_onehot_pushforward(col.ctypes.data_as(C.c_void_p), C.c_size_t(col.shape[0]), C.c_size_t(k_entries), _p(w), _p(out))
    return out


class HammingWeight:
    """HammingWeightKernel (crates/jolt-kernels/src/optimized/hamming_weight_claim_reduction.rs:150-300) over pushforward masses (n_polys, K, 4)"""

    def __init__(self, masses, gamma, r_address, virtualization_points):
        self.g = np.ascontiguousarray(masses, dtype=np.uint64).copy()
        self.n_polys, self.k = self.g.shape[0], self.g.shape[1]
        self.len = self.k
        log_k = self.k.bit_length() - 1
        vp = np.ascontiguousarray(virtualization_points, dtype=np.uint64).reshape(self.n_polys, log_k, 4)
        one = to_mont([1])
        eq_bool = eq_evals(r_address) if log_k else one
        eq_virt = np.ascontiguousarray(np.stack([eq_evals(vp[i]) if log_k else one for i in range(self.n_polys)]))
        self.w = fr_array(self.n_polys * self.k).reshape(self.n_polys, self.k, 4)
        lib().orc_hamming_weights(_p(np.ascontiguousarray(gamma, dtype=np.uint64)), _p(np.ascontiguousarray(eq_bool)), _p(eq_virt), C.c_size_t(self.n_polys), C.c_size_t(self.k), _p(self.w))

    def round(self):
        o = fr_array(3)
        lib().orc_pair_tables_round(_p(self.g), _p(self.w), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(o))
        return o

    def bind(self, r):
        lib().orc_pair_tables_bind(_p(self.g), _p(self.w), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(np.ascontiguousarray(r, dtype=np.uint64)))
        self.len //= 2

    def output_claims(self):
        assert self.len == 1
        return self.g[:, 0].copy()


def fold_cycles(keys, k_entries, w):
    """RamAccessColumns::fold_cycles (optimized/ram_trace.rs:150-162): out[k] = sum of w over the cycles with key k; keys >= k_entries are cold"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    w = np.ascontiguousarray(w, dtype=np.uint64)
    out = fr_array(k_entries)
    lib().orc_fold_cycles(_p(keys), C.c_size_t(keys.shape[0]), C.c_size_t(k_entries), _p(w), _p(out))
    return out


def stage_pushforwards(points, pcs, k_entries):
    """stage_pushforwards (optimized/bytecode_read_raf.rs:152-237): points (n_stages, log_t, 4) big-endian; -> (n_stages, k_entries, 4)"""
    points = np.ascontiguousarray(points, dtype=np.uint64)
    pcs = np.ascontiguousarray(pcs, dtype=np.uint64)
    n_stages, log_t = points.shape[0], points.shape[1]
    assert pcs.shape[0] == 1 << log_t
    out = fr_array(n_stages * k_entries)
    rc = lib().orc_stage_pushforwards(_p(points), C.c_size_t(n_stages), C.c_size_t(log_t), _p(pcs), C.c_size_t(k_entries), _p(out))
    if rc:
        raise ValueError("bytecode index outside the padded bytecode domain" if rc == -2 else "out of memory")
    return out.reshape(n_stages, k_entries, 4)


def last_value(keys, post, k_entries, init):
    """the word an address holds after its last access (ram_val_final); init where it is never accessed"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    post = np.ascontiguousarray(post, dtype=np.uint64)
    init = np.ascontiguousarray(init, dtype=np.uint64)
    out = fr_array(k_entries)
    lib().orc_last_value(_p(keys), _p(post), C.c_size_t(keys.shape[0]), C.c_size_t(k_entries), _p(init), _p(out))
    return out


class BooleanityAddress:
    """OptimizedBooleanityAddressKernel (crates/jolt-kernels/src/optimized/booleanity.rs:283-427) over pushforward masses (n_polys, K, 4)"""

    def __init__(self, masses, gamma, reference_address):
        self.linear = np.ascontiguousarray(masses, dtype=np.uint64).copy()
        self.squared = self.linear.copy()
        self.n_polys, self.k = self.linear.shape[0], self.linear.shape[1]
        self.len = self.k
        g2 = fr_mul(np.asarray(gamma).reshape(1, 4), np.asarray(gamma).reshape(1, 4))[0]
        w, cur = [], to_mont([1])[0]
        for _ in range(self.n_polys):
            w.append(cur)
            cur = fr_mul(cur.reshape(1, 4), g2.reshape(1, 4))[0]
        self.weights = np.ascontiguousarray(np.stack(w))
        self.eq = eq_evals(reference_address).copy()

    def round(self):
        o = fr_array(4)
        lib().orc_booleanity_address_round(_p(self.linear), _p(self.squared), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(self.weights), _p(self.eq), _p(o))
        return o

    def bind(self, r):
        lib().orc_booleanity_address_bind(_p(self.linear), _p(self.squared), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(self.eq),
                                          _p(np.ascontiguousarray(r, dtype=np.uint64)))
        self.len //= 2

    def intermediate(self):
        assert self.len == 1
        acc = np.zeros(4, dtype=np.uint64)
        for i in range(self.n_polys):
            acc = fr_add(acc.reshape(1, 4), fr_mul(self.weights[i].reshape(1, 4), fr_sub(self.squared[i, 0].reshape(1, 4), self.linear[i, 0].reshape(1, 4))))[0]
        return fr_mul(self.eq[0].reshape(1, 4), acc.reshape(1, 4))[0]


# ---- sparse read-write matrix (oracle/rw_matrix.c) --------------------------------------------------------------------
RW_NO_ACCESS = 0xFFFFFFFFFFFFFFFF


def split_eq_bind_scalar(scalar, point_i, challenge):
    o = fr_array(1)
    lib().orc_split_eq_bind_scalar(*[_p(np.ascontiguousarray(x, dtype=np.uint64)) for x in (scalar, point_i, challenge)], _p(o))
    return o[0]


class SplitEqState:
    """Host half of GruenSplitEqPolynomial::new(w, LowToHigh) (crates/jolt-poly/src/split_eq.rs:187-363): current scalar, current
    E_out / E_in tables, the coordinate the next round binds."""

    def __init__(self, w):
        self.w = np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4)
        self.n, self.bound = self.w.shape[0], 0
        self.scalar = to_mont([1])[0]

    def tables(self):
        out_bits, in_bits = split_eq_current_dims(self.n, self.bound)
        out_len = min(self.n // 2, self.n - 1 if self.n else 0)
        return eq_evals(self.w[:out_bits]), eq_evals(self.w[out_len:out_len + in_bits]), in_bits

    def point(self):
        return self.w[self.n - self.bound - 1]

    def bind(self, r):
        self.scalar = split_eq_bind_scalar(self.scalar, self.point(), r)
        self.bound += 1


class RwMatrix:
    """CycleMajorMatrix / AddressMajorMatrix of the RAM read/write-checking kernel (crates/jolt-kernels/src/optimized/rw_matrix.rs)"""

    def __init__(self, addresses, pre, post):
        a, p, q = (np.ascontiguousarray(x, dtype=np.uint64) for x in (addresses, pre, post))
        lib().orc_rw_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_rw_create(_p(a), _p(p), _p(q), C.c_size_t(a.shape[0])))

    def __len__(self):
        lib().orc_rw_len.restype = C.c_size_t
        return lib().orc_rw_len(self.h)

    def export(self):
        n = len(self)
        rows, cols = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        val, ra, prev, nxt = fr_array(n), fr_array(n), fr_array(n), fr_array(n)
        lib().orc_rw_export(self.h, _p(rows), _p(cols), _p(val), _p(ra), _p(prev), _p(nxt))
        return dict(rows=rows, cols=cols, val=val, ra=ra, prev=prev, next=nxt)

    def cycle_round(self, e_out, e_in, in_bits, inc, gamma):
        o = fr_array(2)
        lib().orc_rw_cycle_round(self.h, _p(np.ascontiguousarray(e_out)), _p(np.ascontiguousarray(e_in)), C.c_size_t(in_bits), _p(np.ascontiguousarray(inc)),
                                 _p(np.ascontiguousarray(gamma)), _p(o))
        return o

    def cycle_bind(self, r):
        lib().orc_rw_cycle_bind(self.h, _p(np.ascontiguousarray(r, dtype=np.uint64)))

    def into_address_major(self):
        assert lib().orc_rw_into_address_major(self.h) == 0

    def address_round(self, val_init, inc, eq, gamma):
        o = fr_array(2)
        lib().orc_rw_address_round(self.h, _p(np.ascontiguousarray(val_init)), _p(np.ascontiguousarray(inc)), _p(np.ascontiguousarray(eq)),
                                   _p(np.ascontiguousarray(gamma)), _p(o))
        return o

    def address_bind(self, r, val_init):
        """binds the matrix and `val_init` (returned, half as long)"""
        v = np.ascontiguousarray(val_init, dtype=np.uint64).copy()
        lib().orc_rw_address_bind(self.h, _p(np.ascontiguousarray(r, dtype=np.uint64)), _p(v), C.c_size_t(v.shape[0]))
        return v[: v.shape[0] // 2].copy()

    def final_values(self, val_init):
        ra, val = fr_array(1), fr_array(1)
        lib().orc_rw_final_values(self.h, _p(np.ascontiguousarray(val_init)), _p(ra), _p(val))
        return ra[0], val[0]

    def close(self):
        if self.h:
            lib().orc_rw_destroy(self.h)
            self.h = None


class RegMatrix:
    """The sparse cycle-major matrix of the registers read/write-checking kernel (oracle/registers_rw.c;
    crates/jolt-kernels/src/optimized/registers_read_write/sparse.rs).  Columns: register indices as uint8 (0xFF = none), values uint64."""

    def __init__(self, rs1, rs1_val, rs2, rs2_val, rd, rd_pre, rd_post, gamma):
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        cols = [u8(rs1), u64(rs1_val), u8(rs2), u64(rs2_val), u8(rd), u64(rd_pre), u64(rd_post)]
        lib().orc_regrw_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_regrw_create(*[c.ctypes.data_as(C.c_void_p) for c in cols], C.c_size_t(cols[0].shape[0]), _p(np.ascontiguousarray(gamma, dtype=np.uint64))))

    def __len__(self):
        lib().orc_regrw_len.restype = C.c_size_t
        return lib().orc_regrw_len(self.h)

    def export(self):
        n = len(self)
        rows, cols, prev, nxt = (np.zeros(max(n, 1), dtype=np.uint64) for _ in range(4))
        val, ra, wa = fr_array(max(n, 1)), fr_array(max(n, 1)), fr_array(max(n, 1))
        lib().orc_regrw_export(self.h, _p(rows), _p(cols), _p(val), _p(ra), _p(wa), _p(prev), _p(nxt))
        return dict(rows=rows[:n], cols=cols[:n], val=val[:n], ra=ra[:n], wa=wa[:n], prev=prev[:n], next=nxt[:n])

    def cycle_round(self, e_out, e_in, inc):
        o = fr_array(2)
        e_in = np.ascontiguousarray(e_in, dtype=np.uint64).reshape(-1, 4)
        lib().orc_regrw_cycle_round(self.h, _p(np.ascontiguousarray(e_out)), _p(e_in), C.c_size_t(e_in.shape[0]), _p(np.ascontiguousarray(inc)), _p(o))
        return o

    def cycle_bind(self, r):
        lib().orc_regrw_cycle_bind(self.h, _p(np.ascontiguousarray(r, dtype=np.uint64)))

    def into_dense(self, k):
        ra, wa, val = fr_array(k), fr_array(k), fr_array(k)
        assert lib().orc_regrw_into_dense(self.h, C.c_size_t(k), _p(ra), _p(wa), _p(val)) == 0
        return ra, wa, val

    def close(self):
        if self.h:
            lib().orc_regrw_destroy(self.h)
            self.h = None


def regrw_address_round(ra, wa, val, inc_scalar, eq_scalar):
    o = fr_array(4)
    lib().orc_regrw_address_round(_p(np.ascontiguousarray(ra)), _p(np.ascontiguousarray(wa)), _p(np.ascontiguousarray(val)), C.c_size_t(ra.shape[0]),
                                  _p(np.ascontiguousarray(inc_scalar, dtype=np.uint64)), _p(np.ascontiguousarray(eq_scalar, dtype=np.uint64)), _p(o))
    return o


def regrw_operand_claim(idx, eq_address, eq_cycle):
    i = np.ascontiguousarray(idx, dtype=np.uint8)
    o = fr_array(1)
    lib().orc_regrw_operand_claim(i.ctypes.data_as(C.c_void_p), C.c_size_t(i.shape[0]), _p(np.ascontiguousarray(eq_address)), _p(np.ascontiguousarray(eq_cycle)), _p(o))
    return o[0]


# ---- Spartan outer T-scale sums (oracle/r1cs.c) ---------------------------------------------------------------------------------
def _ptrs(tables):
    tabs = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
    return tabs, (C.c_void_p * max(len(tabs), 1))(*[t.ctypes.data for t in tabs])


def r1cs_row_values(inputs, rows):
    """rows: [[(column, coeff), ...], ...] with column 0 = the constant -> (n_rows, cycles, 4)"""
    tabs, ptrs = _ptrs(inputs)
    cycles = tabs[0].shape[0]
    offs, cols, coefs = [0], [], []
    for row in rows:
        for c, a in row:
            cols.append(c)
            coefs.append(np.asarray(a, dtype=np.uint64))
        offs.append(len(cols))
    out = fr_array(len(rows) * cycles)
    lib().orc_r1cs_row_values(ptrs, C.c_size_t(cycles), C.c_uint32(len(rows)), _p(np.array(offs, dtype=np.uint32)), _p(np.array(cols if cols else [0], dtype=np.uint32)),
                              _p(np.ascontiguousarray(np.stack(coefs)) if coefs else fr_array(1)), _p(out))
    return out.reshape(len(rows), cycles, 4)


def r1cs_uniskip_sums_rows(az_rows, bz_rows, eq, row_weights):
    n_rows, cycles = az_rows.shape[0], az_rows.shape[1]
    w = np.ascontiguousarray(row_weights, dtype=np.uint64).reshape(-1, 2, n_rows, 4)
    out = fr_array(w.shape[0])
    lib().orc_r1cs_uniskip_sums_rows(_p(np.ascontiguousarray(az_rows)), _p(np.ascontiguousarray(bz_rows)), C.c_uint32(n_rows), C.c_size_t(cycles),
                                     _p(np.ascontiguousarray(eq)), _p(w), C.c_uint32(w.shape[0]), _p(out))
    return out


def r1cs_uniskip_sums(inputs, eq, a_weights, b_weights):
    tabs, ptrs = _ptrs(inputs)
    wa = np.ascontiguousarray(a_weights, dtype=np.uint64).reshape(-1, 2, 1 + len(tabs), 4)
    wb = np.ascontiguousarray(b_weights, dtype=np.uint64).reshape(-1, 2, 1 + len(tabs), 4)
    out = fr_array(wa.shape[0])
    lib().orc_r1cs_uniskip_sums(ptrs, C.c_uint32(len(tabs)), C.c_size_t(tabs[0].shape[0]), _p(np.ascontiguousarray(eq)), _p(wa), _p(wb), C.c_uint32(wa.shape[0]), _p(out))
    return out


def r1cs_materialize(inputs, a_weights, b_weights):
    tabs, ptrs = _ptrs(inputs)
    cycles = tabs[0].shape[0]
    wa = np.ascontiguousarray(a_weights, dtype=np.uint64).reshape(2, 1 + len(tabs), 4)
    wb = np.ascontiguousarray(b_weights, dtype=np.uint64).reshape(2, 1 + len(tabs), 4)
    az, bz = fr_array(2 * cycles), fr_array(2 * cycles)
    lib().orc_r1cs_materialize(ptrs, C.c_uint32(len(tabs)), C.c_size_t(cycles), _p(wa), _p(wb), _p(az), _p(bz))
    return az, bz


def small_scalar_accumulate(values, scalars):
    """FrSmallScalarAccumulator: sum_k values[k] * scalars[k] (int64 scalars), one deferred reduction"""
    v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
    sc = np.ascontiguousarray(scalars, dtype=np.int64)
    o = fr_array(1)
    lib().orc_small_scalar_accumulate(_p(v), sc.ctypes.data_as(C.c_void_p), C.c_size_t(v.shape[0]), _p(o))
    return o[0]


# ---- Spartan product virtualization (oracle/r1cs.c) -------------------------------------------------------------------------------
def spartan_product_extension_coefficients():
    out = np.zeros((5, 3), dtype=np.int64)
    lib().orc_spartan_product_extension_coefficients(out.ctypes.data_as(C.c_void_p))
    return out


def _product_args(rows):
    """rows: dict of numpy columns left_input u64, lookup_output u64, jump u8, right_input (n, 2) u64 (i128 two's complement), branch u8, next_is_noop u8"""
    cols = [np.ascontiguousarray(rows["left_input"], dtype=np.uint64), np.ascontiguousarray(rows["lookup_output"], dtype=np.uint64),
            np.ascontiguousarray(rows["jump"], dtype=np.uint8), np.ascontiguousarray(rows["right_input"], dtype=np.uint64).reshape(-1, 2),
            np.ascontiguousarray(rows["branch"], dtype=np.uint8), np.ascontiguousarray(rows["next_is_noop"], dtype=np.uint8)]
    return cols, [c.ctypes.data_as(C.c_void_p) for c in cols], C.c_size_t(cols[0].shape[0])


def spartan_product_t1(rows, eq):
    cols, ptrs, n = _product_args(rows)
    e = np.ascontiguousarray(eq, dtype=np.uint64).reshape(-1, 4)
    out = fr_array(5)
    lib().orc_spartan_product_t1(*ptrs, n, _p(e), _p(out))
    return out


def spartan_product_tables(rows, weights):
    cols, ptrs, n = _product_args(rows)
    w = np.ascontiguousarray(weights, dtype=np.uint64).reshape(3, 4)
    left, right = fr_array(cols[0].shape[0]), fr_array(cols[0].shape[0])
    lib().orc_spartan_product_tables(*ptrs, n, _p(w), _p(left), _p(right))
    return left, right


# ---- instruction read+RAF checking scans (oracle/read_raf.c) ------------------------------------------------------------------------
NUM_SUFFIX_KINDS = 48


def suffix_mle(kind, bits, length):
    lib().orc_suffix_mle.restype = C.c_uint64
    b = bits % (1 << length) if length < 128 else bits
    return int(lib().orc_suffix_mle(C.c_uint32(kind), C.c_uint64(b & (2**64 - 1)), C.c_uint64(b >> 64), C.c_uint32(length)))


def suffix_is_01_valued(kind):
    return bool(lib().orc_suffix_is_01_valued(C.c_uint32(kind)))


def _rr_args(lookup_index, table_index, raf_flag):
    idx = np.ascontiguousarray(lookup_index, dtype=np.uint64).reshape(-1, 2)
    tab = np.ascontiguousarray(table_index, dtype=np.uint8)
    raf = np.ascontiguousarray(raf_flag, dtype=np.uint8)
    return idx, tab, raf


def read_raf_phase_scan(lookup_index, table_index, raf_flag, n_tables, u, suffix_len, address_bits, suffix_lists, canonical=False):
    idx, tab, raf = _rr_args(lookup_index, table_index, raf_flag)
    offs = np.zeros(n_tables + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(l) for l in suffix_lists])
    kinds = np.array([k for l in suffix_lists for k in l] + [0], dtype=np.uint8)
    uu = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, 4)
    raf_out, suf = fr_array(6 * 256), fr_array(max(int(offs[-1]) * 256, 1))
    lib().orc_read_raf_phase_scan(idx.ctypes.data_as(C.c_void_p), tab.ctypes.data_as(C.c_void_p), raf.ctypes.data_as(C.c_void_p), C.c_size_t(idx.shape[0]), C.c_uint32(n_tables),
                                  _p(uu), C.c_uint32(suffix_len), C.c_uint32(address_bits), C.c_int(1 if canonical else 0), offs.ctypes.data_as(C.c_void_p),
                                  kinds.ctypes.data_as(C.c_void_p), _p(raf_out), _p(suf))
    return raf_out.reshape(6, 256, 4), suf[: int(offs[-1]) * 256].reshape(-1, 256, 4)


def read_raf_condense(lookup_index, u, v_table, shift):
    idx = np.ascontiguousarray(lookup_index, dtype=np.uint64).reshape(-1, 2)
    out = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, 4).copy()
    v = np.ascontiguousarray(v_table, dtype=np.uint64).reshape(256, 4)
    lib().orc_read_raf_condense(idx.ctypes.data_as(C.c_void_p), C.c_size_t(idx.shape[0]), _p(v), C.c_uint32(shift), _p(out))
    return out


def read_raf_cycle_tables(lookup_index, table_index, raf_flag, table_values, raf_interleaved, raf_identity, v_tables, address_bits, ra_count):
    idx, tab, raf = _rr_args(lookup_index, table_index, raf_flag)
    tv = np.ascontiguousarray(table_values, dtype=np.uint64).reshape(-1, 4)
    vt = np.ascontiguousarray(v_tables, dtype=np.uint64).reshape(-1, 256, 4)
    T = idx.shape[0]
    combined, ra = fr_array(T), fr_array(ra_count * T)
    lib().orc_read_raf_cycle_tables(idx.ctypes.data_as(C.c_void_p), tab.ctypes.data_as(C.c_void_p), raf.ctypes.data_as(C.c_void_p), C.c_size_t(T), _p(tv),
                                    _p(np.ascontiguousarray(raf_interleaved, dtype=np.uint64).reshape(4)), _p(np.ascontiguousarray(raf_identity, dtype=np.uint64).reshape(4)),
                                    _p(vt), C.c_uint32(vt.shape[0]), C.c_uint32(address_bits), C.c_uint32(ra_count), _p(combined), _p(ra))
    return combined, ra.reshape(ra_count, T, 4)


# ---- lookup tables as full multilinear extensions and the address rounds from the definition (oracle/lookup_tables.c) ---------------------------
TABLE_KINDS = ["RangeCheck", "RangeCheckAligned", "And", "Andn", "Or", "Xor", "Equal", "SignedGreaterThanEqual", "UnsignedGreaterThanEqual", "NotEqual", "SignedLessThan",
               "UnsignedLessThan", "SignMask", "UpperWord", "UnsignedLessThanEqual", "ValidUnsignedRemainder", "ValidDiv0", "HalfwordAlignment", "WordAlignment", "LowerHalfWord",
               "SignExtendWord", "Pow2", "Pow2W", "ShiftRightBitmask", "VirtualRev8W", "VirtualSRL", "VirtualSRA", "VirtualROTR", "VirtualROTRW", "VirtualChangeDivisor",
               "VirtualChangeDivisorW", "MulUNoOverflow", "VirtualXORROT32", "VirtualXORROT24", "VirtualXORROT16", "VirtualXORROT63", "VirtualXORROTW16", "VirtualXORROTW12",
               "VirtualXORROTW8", "VirtualXORROTW7", "WindowMaskW", "PextSigned"]  # enum LookupTableKind, tables/mod.rs:121-166


def table_count():
    lib().orc_table_count.restype = C.c_uint32
    return int(lib().orc_table_count())


def table_materialize_entry(kind, index):
    lib().orc_table_materialize_entry.restype = C.c_uint64
    return int(lib().orc_table_materialize_entry(C.c_uint32(kind), C.c_uint64(index & (2**64 - 1)), C.c_uint64(index >> 64)))


def table_evaluate_mle(kind, r):
    """r: (128, 4) Montgomery limbs, r[0] the variable of index bit 127"""
    rr = np.ascontiguousarray(r, dtype=np.uint64).reshape(128, 4)
    out = fr_array(1)
    lib().orc_table_evaluate_mle(C.c_uint32(kind), _p(rr), _p(out))
    return out[0]


def read_raf_input_claim(lookup_index, table_index, raf_flag, u, gamma, canonical=False):
    idx, tab, raf = _rr_args(lookup_index, table_index, raf_flag)
    uu = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, 4)
    out = fr_array(1)
    lib().orc_read_raf_input_claim(_p(idx), _p(tab), _p(raf), C.c_size_t(idx.shape[0]), _p(uu), _p(np.ascontiguousarray(gamma, dtype=np.uint64).reshape(4)),
                                   C.c_int(1 if canonical else 0), _p(out))
    return out[0]


def read_raf_address_rounds(lookup_index, table_index, raf_flag, u, gamma, challenges, canonical=False):
    """-> (evals (128, 3, 4): s_i(0), s_i(1), s_i(2); table_values (42, 4); operands (4, 4): left, right, identity, upper_all_ones at r_address)"""
    idx, tab, raf = _rr_args(lookup_index, table_index, raf_flag)
    uu = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, 4)
    ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(128, 4)
    evals, tv, ops = fr_array(128 * 3), fr_array(len(TABLE_KINDS)), fr_array(4)
    lib().orc_read_raf_address_rounds(_p(idx), _p(tab), _p(raf), C.c_size_t(idx.shape[0]), _p(uu), _p(np.ascontiguousarray(gamma, dtype=np.uint64).reshape(4)),
                                      C.c_int(1 if canonical else 0), _p(ch), _p(evals), _p(tv), _p(ops))
    return evals.reshape(128, 3, 4), tv, ops


class ReadRafAddressDirect:
    """the address rounds from the definition, one at a time (orc_read_raf_address_round / _bind): for a prover whose challenges come from a transcript"""

    def __init__(self, lookup_index, table_index, raf_flag, u, gamma, canonical=False):
        self.idx, self.tab, self.raf = _rr_args(lookup_index, table_index, raf_flag)
        self.weight = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, 4).copy()
        self.gamma = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(4).copy()
        self.canonical = 1 if canonical else 0
        self.challenges = fr_array(128)
        self.i = 0

    def restart(self, i, weight, challenges=None):
        """jump to round i: `weight` = eq(r_reduction, j) * eq(r_{<i}, k_j[<i]) per row (at a phase boundary that is the oracle's condensed u of the phase,
        read_raf_condense), `challenges` = the i challenges drawn so far"""
        self.weight = np.ascontiguousarray(weight, dtype=np.uint64).reshape(-1, 4).copy()
        assert self.weight.shape[0] == self.idx.shape[0]
        self.challenges[:] = 0
        if i:
            self.challenges[:i] = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)[:i]
        self.i = i

    def round(self):
        out = fr_array(3)
        lib().orc_read_raf_address_round(_p(self.idx), _p(self.tab), _p(self.raf), C.c_size_t(self.idx.shape[0]), _p(self.weight), _p(self.gamma), C.c_int(self.canonical),
                                         _p(self.challenges), C.c_uint32(self.i), _p(out))
        return out

    def bind(self, r):
        rr = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        lib().orc_read_raf_address_bind(_p(self.idx), C.c_size_t(self.idx.shape[0]), _p(self.weight), C.c_uint32(self.i), _p(rr))
        self.challenges[self.i] = rr
        self.i += 1

    def values(self):
        assert self.i == 128
        tv, ops = fr_array(len(TABLE_KINDS)), fr_array(4)
        lib().orc_read_raf_address_values(_p(self.challenges), _p(tv), _p(ops))
        return tv, ops


def read_raf_address_values(challenges):
    """Val_t(r_address) of the 42 tables by evaluate_mle, and the operand polynomials (left, right, identity, upper_all_ones) at r_address"""
    ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(128, 4)
    tv, ops = fr_array(len(TABLE_KINDS)), fr_array(4)
    lib().orc_read_raf_address_values(_p(ch), _p(tv), _p(ops))
    return tv, ops
