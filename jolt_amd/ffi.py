"""ctypes binding of libjolt_hip.so (include/jolt_hip.h) -- the same C ABI a Rust FFI crate would bind.

Field elements are numpy uint64 arrays of shape (..., 4) (Montgomery limbs of jolt_field::Fr); G1 points are
(..., 12) uint64 (Jacobian, ark_bn254::G1Projective layout).  There is no CPU fallback: if the shared library is
missing or no gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjolt_hip.so")

STATUS = {0: "ok", 1: "invalid argument", 2: "no device", 3: "oom", 4: "hip error", 5: "size mismatch", 6: "unsupported",
          7: "not fully bound", 8: "round check failed", 9: "srs too small", 10: "empty point", 11: "not invertible"}
ORDER_LOW_TO_HIGH, ORDER_HIGH_TO_LOW = 0, 1
MEMBER_FLAG_SKIP_ONE = 1
MEMBER_FLAG_BORROW_TABLES = 2


class JoltError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status} ({STATUS.get(status, '?')}) {detail}")


class MemberDesc(C.Structure):
    _fields_ = [("n_tables", C.c_uint32), ("n_terms", C.c_uint32), ("degree", C.c_uint32), ("order", C.c_int32),
                ("term_offsets", C.c_void_p), ("factors", C.c_void_p), ("coeffs", C.c_void_p)]


class MemberLcDesc(C.Structure):
    _fields_ = [("n_tables", C.c_uint32), ("n_groups", C.c_uint32), ("n_factors", C.c_uint32), ("n_lc", C.c_uint32),
                ("degree", C.c_uint32), ("order", C.c_int32), ("flags", C.c_uint32),
                ("group_factor_offsets", C.c_void_p), ("factor_lc_offsets", C.c_void_p), ("factor_consts", C.c_void_p),
                ("lc_tables", C.c_void_p), ("lc_coeffs", C.c_void_p)]


_lib = None


def lib():
    """Load the native library (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m jolt_amd.build` (there is no CPU fallback)")
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (capi.hip sets the same default when the library is loaded; here for processes whose HIP runtime starts before it)
        _lib = C.CDLL(LIB_PATH)
        _lib.jolt_status_string.restype = C.c_char_p
        _lib.jolt_last_error.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fr(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def fr_array(n):
    return np.zeros((n, 4), dtype=np.uint64)


def g1_array(n):
    return np.zeros((n, 12), dtype=np.uint64)


def _ck(status, where, ctx=None):
    if status != 0:
        detail = ""
        if ctx is not None and ctx.h:
            detail = lib().jolt_last_error(ctx.h).decode()
        raise JoltError(status, where, detail)


class Context:
    def __init__(self, device_id=0, stream=None):
        self.h = C.c_void_p()
        self.device_id = device_id
        st = lib().jolt_ctx_create(C.c_int32(device_id), C.c_void_p(stream) if stream else None, C.byref(self.h))
        if st != 0:
            self.h = C.c_void_p()
            raise JoltError(st, "jolt_ctx_create")

    def close(self):
        if self.h:
            lib().jolt_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def bind_thread(self):
        """select this context's device on the calling thread (a worker thread starts on device 0)"""
        _ck(lib().jolt_ctx_bind_thread(self.h), "jolt_ctx_bind_thread", self)

    def synchronize_foreground(self):
        """every stream but the background one (the opening hint's class sums)"""
        _ck(lib().jolt_ctx_synchronize_foreground(self.h), "jolt_ctx_synchronize_foreground", self)

    def synchronize(self):
        _ck(lib().jolt_ctx_synchronize(self.h), "jolt_ctx_synchronize", self)

    def timer_begin(self):
        _ck(lib().jolt_timer_begin(self.h), "jolt_timer_begin", self)

    def timer_end(self):
        ms = C.c_float()
        _ck(lib().jolt_timer_end(self.h, C.byref(ms)), "jolt_timer_end", self)
        return ms.value

    # ---- tables
    def upload(self, arr):
        a = fr(arr).reshape(-1, 4)
        h = C.c_void_p()
        _ck(lib().jolt_table_upload(self.h, _p(a), C.c_size_t(a.shape[0]), C.byref(h)), "jolt_table_upload", self)
        return Table(self, h)

    def from_device(self, ptr, length):
        h = C.c_void_p()
        _ck(lib().jolt_table_from_device(self.h, C.c_void_p(ptr), C.c_size_t(length), C.byref(h)), "jolt_table_from_device", self)
        return Table(self, h)

    def alloc(self, length):
        h = C.c_void_p()
        _ck(lib().jolt_table_alloc(self.h, C.c_size_t(length), C.byref(h)), "jolt_table_alloc", self)
        return Table(self, h)

    def from_u64(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.uint64)
        h = C.c_void_p()
        _ck(lib().jolt_table_from_u64(self.h, _p(v), C.c_size_t(v.shape[0]), C.byref(h)), "jolt_table_from_u64", self)
        return Table(self, h)

    def from_i64(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.int64)
        h = C.c_void_p()
        _ck(lib().jolt_table_from_i64(self.h, _p(v), C.c_size_t(v.shape[0]), C.byref(h)), "jolt_table_from_i64", self)
        return Table(self, h)

    def bind(self, tables, r, order=ORDER_LOW_TO_HIGH):
        hs = (C.c_void_p * len(tables))(*[t.h for t in tables])
        _ck(lib().jolt_bind(self.h, hs, C.c_size_t(len(tables)), _p(fr(r)), C.c_int32(order)), "jolt_bind", self)

    def eq_evals(self, r, scale=None):
        r = fr(r).reshape(-1, 4)
        h = C.c_void_p()
        _ck(lib().jolt_eq_evals(self.h, _p(r), C.c_size_t(r.shape[0]), _p(fr(scale)) if scale is not None else None, C.byref(h)),
            "jolt_eq_evals", self)
        return Table(self, h)

    def eq_evals_aligned_block(self, r, start, block):
        r = fr(r).reshape(-1, 4)
        h = C.c_void_p()
        _ck(lib().jolt_eq_evals_aligned_block(self.h, _p(r), C.c_size_t(r.shape[0]), C.c_size_t(start), C.c_size_t(block), C.byref(h)),
            "jolt_eq_evals_aligned_block", self)
        return Table(self, h)

    def lt_evals(self, r):
        r = fr(r).reshape(-1, 4)
        h = C.c_void_p()
        _ck(lib().jolt_lt_evals(self.h, _p(r), C.c_size_t(r.shape[0]), C.byref(h)), "jolt_lt_evals", self)
        return Table(self, h)

    def eq_plus_one_evals(self, r, scale=None):
        r = fr(r).reshape(-1, 4)
        h1, h2 = C.c_void_p(), C.c_void_p()
        _ck(lib().jolt_eq_plus_one_evals(self.h, _p(r), C.c_size_t(r.shape[0]), _p(fr(scale)) if scale is not None else None,
                                         C.byref(h1), C.byref(h2)), "jolt_eq_plus_one_evals", self)
        return Table(self, h1), Table(self, h2)

    def evaluate(self, table, point):
        p = fr(point).reshape(-1, 4)
        o = fr_array(1)
        _ck(lib().jolt_table_evaluate(self.h, table.h, _p(p), C.c_size_t(p.shape[0]), _p(o)), "jolt_table_evaluate", self)
        return o[0]

    def table_sum(self, table):
        o = fr_array(1)
        _ck(lib().jolt_table_sum(self.h, table.h, _p(o)), "jolt_table_sum", self)
        return o[0]

    # ---- members
    def member_expr(self, tables, terms, degree, order=ORDER_LOW_TO_HIGH):
        """terms = [(coeff_limbs, [table indices]), ...]; takes ownership of the tables."""
        offs, facs = [0], []
        for _, f in terms:
            facs.extend(f)
            offs.append(len(facs))
        offs = np.array(offs, dtype=np.uint32)
        facs_a = np.array(facs if facs else [0], dtype=np.uint32)
        coeffs = np.ascontiguousarray(np.stack([fr(c) for c, _ in terms])).reshape(-1, 4)
        d = MemberDesc(len(tables), len(terms), degree, order, offs.ctypes.data, facs_a.ctypes.data, coeffs.ctypes.data)
        hs = (C.c_void_p * len(tables))(*[t.h for t in tables])
        h = C.c_void_p()
        _ck(lib().jolt_member_create_expr(self.h, hs, C.byref(d), C.byref(h)), "jolt_member_create_expr", self)
        for t in tables:
            t.h = None  # ownership moved into the member
        return Member(self, h, degree, len(tables), False, False)

    def member_lc(self, tables, groups, degree, order=ORDER_LOW_TO_HIGH, skip_one=False, borrow=False, eq_point=None, eq_scale=None,
                  shard_scale=None):
        """groups = [[factor, ...], ...]; factor = (const_limbs_or_None, [(coeff_limbs, table_idx), ...]).
        borrow=True: the member only reads `tables` (caller keeps ownership, tables must outlive the member).
        eq_point: the summand is eq(eq_point, j) * (these groups) with the eq weight factored out (jolt_member_create_split_eq_lc);
        `degree` is then the inner degree and prove_round returns q(0), q(2), .., q(degree)."""
        goff, foff, consts, ltab, lcoef = [0], [0], [], [], []
        zero = np.zeros(4, dtype=np.uint64)
        for g in groups:
            for const, entries in g:
                consts.append(zero if const is None else fr(const))
                for c, ti in entries:
                    lcoef.append(fr(c))
                    ltab.append(ti)
                foff.append(len(ltab))
            goff.append(len(consts))
        goff = np.array(goff, dtype=np.uint32)
        foff = np.array(foff, dtype=np.uint32)
        consts_a = np.ascontiguousarray(np.stack(consts)) if consts else fr_array(1)
        ltab_a = np.array(ltab if ltab else [0], dtype=np.uint32)
        lcoef_a = np.ascontiguousarray(np.stack(lcoef)) if lcoef else fr_array(1)
        flags = (MEMBER_FLAG_SKIP_ONE if skip_one else 0) | (MEMBER_FLAG_BORROW_TABLES if borrow else 0)
        d = MemberLcDesc(len(tables), len(groups), len(consts), len(ltab), degree, order, flags,
                         goff.ctypes.data, foff.ctypes.data, consts_a.ctypes.data, ltab_a.ctypes.data, lcoef_a.ctypes.data)
        small = any(isinstance(t, Ints) for t in tables)  # compact-scalar slots: resident u64 columns read unpromoted in round 0 (jolt_member_create_lc_small)
        hs = (C.c_void_p * len(tables))(*[None if isinstance(t, Ints) else t.h for t in tables])
        h = C.c_void_p()
        if small:
            assert borrow and order == ORDER_LOW_TO_HIGH, "integer-backed slots: borrowed tables, LowToHigh"
            ih = (C.c_void_p * len(tables))(*[t.h if isinstance(t, Ints) else None for t in tables])
            w = fr(eq_point).reshape(-1, 4) if eq_point is not None else None
            _ck(lib().jolt_member_create_lc_small(self.h, hs, ih, C.byref(d), _p(w) if w is not None else None, C.c_size_t(0 if w is None else w.shape[0]),
                                                  _p(fr(eq_scale)) if eq_scale is not None else None, _p(fr(shard_scale)) if shard_scale is not None else None,
                                                  C.byref(h)), "jolt_member_create_lc_small", self)
            if eq_point is not None:
                m = Member(self, h, degree + 1, len(tables), True, True)
                m.n_evals = degree
                m.uniform = True
                m.eq_weighted = True
            else:
                m = Member(self, h, degree, len(tables), False, skip_one)
        elif eq_point is not None:
            w = fr(eq_point).reshape(-1, 4)
            _ck(lib().jolt_member_create_split_eq_lc(self.h, hs, C.byref(d), _p(w), C.c_size_t(w.shape[0]), _p(fr(eq_scale)) if eq_scale is not None else None,
                                                     _p(fr(shard_scale)) if shard_scale is not None else None, C.byref(h)), "jolt_member_create_split_eq_lc", self)
            m = Member(self, h, degree + 1, len(tables), True, True)
            m.n_evals = degree
            m.uniform = True  # same host-side shape as the uniform member: dq sums, gruen_poly_from_q
            m.eq_weighted = True
        else:
            _ck(lib().jolt_member_create_lc(self.h, hs, C.byref(d), C.byref(h)), "jolt_member_create_lc", self)
            m = Member(self, h, degree, len(tables), False, skip_one)
        if borrow:
            m._keepalive = list(tables)
        else:
            for t in tables:
                t.h = None
        return m

    def member_split_eq_product(self, a, b, w, scale=None, borrow=False):
        w = fr(w).reshape(-1, 4)
        h = C.c_void_p()
        fn = lib().jolt_member_create_split_eq_product_borrowed if borrow else lib().jolt_member_create_split_eq_product
        _ck(fn(self.h, a.h, b.h, _p(w), C.c_size_t(w.shape[0]), _p(fr(scale)) if scale is not None else None, C.byref(h)),
            "jolt_member_create_split_eq_product", self)
        m = Member(self, h, 3, 2, True, False)
        if borrow:
            m._keepalive = [a, b]
        else:
            a.h = None
            b.h = None
        return m

    def member_split_eq_product_sharded(self, a, b, w_local, scale=None, shard_scale=None):
        """one rank's shard of the split-eq product member (jolt_member_create_split_eq_product_sharded): a, b hold the rank's block of rows (borrowed), w_local the
        low log2(rows) coordinates of the point, shard_scale = eq(w_hi, rank)"""
        w = fr(w_local).reshape(-1, 4)
        h = C.c_void_p()
        _ck(lib().jolt_member_create_split_eq_product_sharded(self.h, a.h, b.h, _p(w), C.c_size_t(w.shape[0]), _p(fr(scale)) if scale is not None else None,
                                                              _p(fr(shard_scale)) if shard_scale is not None else None, C.byref(h)),
            "jolt_member_create_split_eq_product_sharded", self)
        m = Member(self, h, 3, 2, True, False)
        m._keepalive = [a, b]
        return m

    def member_split_eq_uniform(self, tables, V, F, coeffs, w, scale=None, shard_scale=None, borrow=False):
        """eq(w,.) * sum_v coeffs[v] * prod_{i<F} tables[v*F+i]; prove_round returns q(0), q(2), .., q(F)."""
        w = fr(w).reshape(-1, 4)
        co = np.ascontiguousarray(np.stack([fr(c) for c in coeffs])).reshape(-1, 4)
        hs = (C.c_void_p * len(tables))(*[t.h for t in tables])
        h = C.c_void_p()
        _ck(lib().jolt_member_create_split_eq_uniform(self.h, hs, C.c_uint32(V), C.c_uint32(F), _p(co), _p(w), C.c_size_t(w.shape[0]),
                                                      _p(fr(scale)) if scale is not None else None,
                                                      _p(fr(shard_scale)) if shard_scale is not None else None,
                                                      C.c_uint32(MEMBER_FLAG_BORROW_TABLES if borrow else 0), C.byref(h)),
            "jolt_member_create_split_eq_uniform", self)
        m = Member(self, h, F + 1, len(tables), True, False)
        m.n_evals = F
        m.uniform = True
        if borrow:
            m._keepalive = list(tables)
        else:
            for t in tables:
                t.h = None
        return m

    def round_group_prove(self, members, binds):
        hs = (C.c_void_p * len(members))(*[m.h for m in members])
        bstore = [None if b is None else fr(b) for b in binds]
        bp = (C.c_void_p * len(members))(*[None if b is None else b.ctypes.data for b in bstore])
        total = sum(m.n_evals for m in members)
        out = fr_array(total)
        _ck(lib().jolt_round_group_prove(self.h, hs, C.c_size_t(len(members)), bp, _p(out), C.c_size_t(total)), "jolt_round_group_prove", self)
        res, off = [], 0
        for m in members:
            res.append(out[off:off + m.n_evals].copy())
            off += m.n_evals
        return res

    def prove_batch(self, members, input_claims, coefficients, offsets, max_num_vars, max_degree, label=0, challenge_mode=0,
                    use_round_group=True):
        n = len(members)
        hs = (C.c_void_p * n)(*[m.h for m in members])
        ic = np.ascontiguousarray(np.stack(input_claims), dtype=np.uint64).reshape(-1, 4)
        co = np.ascontiguousarray(np.stack(coefficients), dtype=np.uint64).reshape(-1, 4)
        offs = (C.c_size_t * n)(*offsets)
        polys, chal = fr_array(max_num_vars * (max_degree + 1)), fr_array(max_num_vars)
        mclaims, final = fr_array(n), fr_array(1)
        _ck(lib().jolt_host_prove_batch(self.h, hs, C.c_size_t(n), _p(ic), _p(co), offs, C.c_size_t(max_num_vars), C.c_size_t(max_degree),
                                        C.c_uint64(label), C.c_int32(challenge_mode), C.c_int32(1 if use_round_group else 0),
                                        _p(polys), _p(chal), _p(mclaims), _p(final)), "jolt_host_prove_batch", self)
        return dict(polys=polys.reshape(max_num_vars, max_degree + 1, 4), challenges=chal, member_claims=mclaims, final_claim=final[0])

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Table:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        n = C.c_size_t()
        _ck(lib().jolt_table_len(self.h, C.byref(n)), "jolt_table_len", self.ctx)
        return n.value

    def download(self, offset=0, length=None):
        n = len(self) - offset if length is None else length
        out = fr_array(n)
        _ck(lib().jolt_table_download(self.ctx.h, self.h, C.c_size_t(offset), C.c_size_t(n), _p(out)), "jolt_table_download", self.ctx)
        return out

    def clone(self):
        h = C.c_void_p()
        _ck(lib().jolt_table_clone(self.ctx.h, self.h, C.byref(h)), "jolt_table_clone", self.ctx)
        return Table(self.ctx, h)

    def device_ptr(self):
        p = C.c_void_p()
        _ck(lib().jolt_table_device_ptr(self.h, C.byref(p)), "jolt_table_device_ptr", self.ctx)
        return p.value

    def free(self):
        if self.h and self.ctx.h:
            lib().jolt_table_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Member:
    """Handle of a device-resident sumcheck member (jolt_sumcheck::ProveRounds, device half)."""

    def __init__(self, ctx, handle, degree, n_tables, split_eq, skip_one):
        self.ctx, self.h, self.degree, self.n_tables, self.split_eq, self.skip_one = ctx, handle, degree, n_tables, split_eq, skip_one
        self.n_evals = 2 if split_eq else (degree if skip_one else degree + 1)
        self.uniform = False

    def set_scale(self, scale):
        _ck(lib().jolt_member_set_scale(self.h, _p(fr(scale))), "jolt_member_set_scale", self.ctx)

    def num_rounds(self):
        n = C.c_size_t()
        _ck(lib().jolt_member_num_rounds(self.h, C.byref(n)), "jolt_member_num_rounds", self.ctx)
        return n.value

    def prove_round(self, bind=None, want_aux=False):
        out, aux = fr_array(self.n_evals), fr_array(3)
        _ck(lib().jolt_member_prove_round(self.h, _p(fr(bind)) if bind is not None else None, _p(out), C.c_size_t(self.n_evals), _p(aux)),
            "jolt_member_prove_round", self.ctx)
        return (out, aux) if want_aux else out

    def finish(self, bind):
        _ck(lib().jolt_member_finish(self.h, _p(fr(bind))), "jolt_member_finish", self.ctx)

    def final_values(self):
        k = self.n_tables + (1 if self.split_eq else 0)
        out = fr_array(k)
        _ck(lib().jolt_member_final_values(self.h, _p(out), C.c_size_t(k)), "jolt_member_final_values", self.ctx)
        return out

    def input_claim(self):
        out = fr_array(1)
        _ck(lib().jolt_member_input_claim(self.h, _p(out)), "jolt_member_input_claim", self.ctx)
        return out[0]

    def reset(self):
        _ck(lib().jolt_member_reset(self.h), "jolt_member_reset", self.ctx)

    def destroy(self):
        if self.h and self.ctx.h:
            lib().jolt_member_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ---- host-side helpers (no GPU needed) ------------------------------------------------------------------------------
def host_fr_binop(name, a, b):
    o = fr_array(1)
    _ck(getattr(lib(), name)(_p(fr(a)), _p(fr(b)), _p(o)), name)
    return o[0]


def host_fr_mul(a, b): return host_fr_binop("jolt_host_fr_mul", a, b)


def host_mul_limbs29(field, a, b):
    """the device multiplication algorithm (29-bit limbs) built for the host; field 0 = Fr, 1 = Fq"""
    o = fr_array(1)
    _ck(lib().jolt_host_mul_limbs29(C.c_int32(field), _p(fr(a)), _p(fr(b)), _p(o)), "jolt_host_mul_limbs29")
    return o[0]


def host_fr_wide_dot(a, b):
    """sum_k a[k]*b[k] through the deferred-reduction accumulator (one REDC per block of products)"""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    o = fr_array(1)
    _ck(lib().jolt_host_fr_wide_dot(_p(a), _p(b), C.c_size_t(a.shape[0]), _p(o)), "jolt_host_fr_wide_dot")
    return o[0]

def host_fr_add(a, b): return host_fr_binop("jolt_host_fr_add", a, b)
def host_fr_sub(a, b): return host_fr_binop("jolt_host_fr_sub", a, b)
def host_fr_mul_shifted(a, c): return host_fr_binop("jolt_host_fr_mul_shifted", a, c)


def host_fr_inv(a):
    o = fr_array(1)
    _ck(lib().jolt_host_fr_inv(_p(fr(a)), _p(o)), "jolt_host_fr_inv")
    return o[0]


def host_fr_from_u64(v):
    o = fr_array(1)
    _ck(lib().jolt_host_fr_from_u64(C.c_uint64(v), _p(o)), "jolt_host_fr_from_u64")
    return o[0]


def host_eq_evals(r, scale=None):
    p = fr(r).reshape(-1, 4)
    out = fr_array(1 << p.shape[0])
    st = lib().jolt_host_eq_evals(_p(p), C.c_size_t(p.shape[0]), _p(fr(scale)) if scale is not None else None, _p(out))
    if st != 0:
        raise JoltError(st, "jolt_host_eq_evals")
    return out


def host_univariate_from_evals(evals):
    e = fr(evals).reshape(-1, 4)
    o = fr_array(e.shape[0])
    _ck(lib().jolt_host_univariate_from_evals(_p(e), C.c_size_t(e.shape[0]), _p(o)), "jolt_host_univariate_from_evals")
    return o


def host_univariate_evaluate(coeffs, x):
    c = fr(coeffs).reshape(-1, 4)
    o = fr_array(1)
    _ck(lib().jolt_host_univariate_evaluate(_p(c), C.c_size_t(c.shape[0]), _p(fr(x)), _p(o)), "jolt_host_univariate_evaluate")
    return o[0]


def host_gruen_poly_from_q(scalar, point_i, q_evals, claim):
    q = fr(q_evals).reshape(-1, 4)
    o = fr_array(q.shape[0] + 2)
    _ck(lib().jolt_host_gruen_poly_from_q(_p(fr(scalar)), _p(fr(point_i)), _p(q), C.c_size_t(q.shape[0]), _p(fr(claim)), _p(o)),
        "jolt_host_gruen_poly_from_q")
    return o


class HostBooleanityAddress:
    """The K-domain state of the booleanity address phase on the host (jolt_host_booleanity_address_*): masses (n_polys, K, 4) from the pushforward"""

    def __init__(self, masses, gamma, reference_address):
        self.linear = np.ascontiguousarray(masses, dtype=np.uint64).copy()
        self.squared = self.linear.copy()
        self.n_polys, self.k = self.linear.shape[0], self.linear.shape[1]
        self.len = self.k
        g2 = host_fr_mul(gamma, gamma)
        w, cur = [], host_fr_from_u64(1)
        for _ in range(self.n_polys):
            w.append(cur)
            cur = host_fr_mul(cur, g2)
        self.weights = np.ascontiguousarray(np.stack(w))
        self.eq = np.ascontiguousarray(host_eq_evals(reference_address)).copy()

    def round(self):
        o = fr_array(4)
        _ck(lib().jolt_host_booleanity_address_round(_p(self.linear), _p(self.squared), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(self.weights), _p(self.eq),
                                                     _p(o)), "jolt_host_booleanity_address_round")
        return o

    def bind(self, r):
        _ck(lib().jolt_host_booleanity_address_bind(_p(self.linear), _p(self.squared), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(self.eq), _p(fr(r))),
            "jolt_host_booleanity_address_bind")
        self.len //= 2

    def intermediate(self):
        assert self.len == 1
        acc = np.zeros(4, dtype=np.uint64)
        for i in range(self.n_polys):
            acc = host_fr_add(acc, host_fr_mul(self.weights[i], host_fr_sub(self.squared[i, 0], self.linear[i, 0])))
        return host_fr_mul(self.eq[0], acc)


class HostHammingWeight:
    """HammingWeightKernel's K_chunk-domain state on the host (jolt_host_hamming_weights / jolt_host_pair_tables_*): masses (n_polys, K, 4) from the pushforward"""

    def __init__(self, masses, gamma, r_address, virtualization_points):
        self.g = np.ascontiguousarray(masses, dtype=np.uint64).copy()
        self.n_polys, self.k = self.g.shape[0], self.g.shape[1]
        self.len = self.k
        log_k = self.k.bit_length() - 1
        ra = np.ascontiguousarray(fr(r_address), dtype=np.uint64).reshape(-1, 4)
        vp = np.ascontiguousarray(virtualization_points, dtype=np.uint64).reshape(-1, 4)
        assert ra.shape[0] == log_k and vp.shape[0] == self.n_polys * log_k
        self.w = fr_array(self.n_polys * self.k).reshape(self.n_polys, self.k, 4)
        _ck(lib().jolt_host_hamming_weights(_p(fr(gamma)), _p(ra), _p(vp), C.c_size_t(self.n_polys), C.c_size_t(log_k), _p(self.w)), "jolt_host_hamming_weights")

    def round(self):
        o = fr_array(3)
        _ck(lib().jolt_host_pair_tables_round(_p(self.g), _p(self.w), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(o)), "jolt_host_pair_tables_round")
        return o

    def bind(self, r):
        _ck(lib().jolt_host_pair_tables_bind(_p(self.g), _p(self.w), C.c_size_t(self.n_polys), C.c_size_t(self.k), C.c_size_t(self.len), _p(fr(r))), "jolt_host_pair_tables_bind")
        self.len //= 2

    def output_claims(self):
        assert self.len == 1
        return self.g[:, 0].copy()


TRANSCRIPT_TEST, TRANSCRIPT_BLAKE2B, TRANSCRIPT_KECCAK, TRANSCRIPT_BLAKE2B_SPONGE = 0, 1 << 62, 2 << 62, 3 << 62  # OR into any `label` / `transcript_label`: the engine behind it (include/jolt_hip.h)


class HostTranscript:
    """jolt_host_transcript_*: the library's transcripts for members driven round by round from here.  `label`: an integer whose two top bits select the engine
    (TRANSCRIPT_TEST: the deterministic test transcript; TRANSCRIPT_BLAKE2B: the reference's LegacyBlake2bTranscript, the one its benchmark profile proves with;
    TRANSCRIPT_KECCAK: its KeccakTranscript), or a byte label with `kind` 1 / 2 -- the reference's `Transcript::new(b"...")`."""

    def __init__(self, label, kind=None):
        self.h = C.c_void_p()
        if isinstance(label, (bytes, bytearray)):
            buf = (C.c_uint8 * max(1, len(label))).from_buffer_copy(bytes(label) if len(label) else b"\0")
            _ck(lib().jolt_host_transcript_create_labelled(C.c_int32(kind), buf, C.c_size_t(len(label)), C.byref(self.h)), "jolt_host_transcript_create_labelled")
        else:
            _ck(lib().jolt_host_transcript_create(C.c_uint64(label), C.byref(self.h)), "jolt_host_transcript_create")

    def append(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        _ck(lib().jolt_host_transcript_append_fr(self.h, _p(v), C.c_size_t(v.shape[0])), "jolt_host_transcript_append_fr")

    def append_bytes(self, data):
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) if len(data) else b"\0")
        _ck(lib().jolt_host_transcript_append_bytes(self.h, buf, C.c_size_t(len(data))), "jolt_host_transcript_append_bytes")

    def append_label(self, label, count=None):
        _ck(lib().jolt_host_transcript_append_label(self.h, C.c_char_p(label), C.c_int32(0 if count is None else 1), C.c_uint64(count or 0)), "jolt_host_transcript_append_label")

    def append_u64_word(self, value):
        _ck(lib().jolt_host_transcript_append_u64_word(self.h, C.c_uint64(value)), "jolt_host_transcript_append_u64_word")

    def append_round_poly(self, coefficients, label=b"sumcheck_poly"):
        v = np.ascontiguousarray(coefficients, dtype=np.uint64).reshape(-1, 4)
        _ck(lib().jolt_host_transcript_append_round_poly(self.h, C.c_char_p(label), _p(v), C.c_size_t(v.shape[0])), "jolt_host_transcript_append_round_poly")

    def state(self):
        out = (C.c_uint8 * 32)()
        _ck(lib().jolt_host_transcript_state(self.h, out), "jolt_host_transcript_state")
        return bytes(out)

    def challenge(self, full_width=False):
        o = fr_array(1)
        _ck(lib().jolt_host_transcript_challenge(self.h, C.c_int32(1 if full_width else 0), _p(o)), "jolt_host_transcript_challenge")
        return o[0]

    def close(self):
        if self.h:
            lib().jolt_host_transcript_destroy(self.h)
            self.h = None


def host_blake2b(data, outlen=32):
    out = (C.c_uint8 * outlen)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) if len(data) else b"\0")
    _ck(lib().jolt_host_blake2b(buf, C.c_size_t(len(data)), C.c_size_t(outlen), out), "jolt_host_blake2b")
    return bytes(out)


def host_keccak_f1600(state):
    buf = (C.c_uint8 * 200).from_buffer_copy(bytes(state))
    _ck(lib().jolt_host_keccak_f1600(buf), "jolt_host_keccak_f1600")
    return bytes(buf)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_gruen_poly_deg_3(scalar, point_i, q0, qinf, claim):
    o = fr_array(4)
    _ck(lib().jolt_host_gruen_poly_deg_3(_p(fr(scalar)), _p(fr(point_i)), _p(fr(q0)), _p(fr(qinf)), _p(fr(claim)), _p(o)),
        "jolt_host_gruen_poly_deg_3")
    return o


# ---- G1 / MSM / HyperKZG ---------------------------------------------------------------------------------------------
class Srs:
    """Device-resident affine bases (HyperKZGProverSetup.g1_powers)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __len__(self):
        n = C.c_size_t()
        _ck(lib().jolt_srs_len(self.h, C.byref(n)), "jolt_srs_len", self.ctx)
        return n.value

    def download(self, offset=0, n=None):
        n = len(self) - offset if n is None else n
        out = g1_array(n)
        _ck(lib().jolt_srs_download(self.ctx.h, self.h, C.c_size_t(offset), C.c_size_t(n), _p(out)), "jolt_srs_download", self.ctx)
        return out

    def free(self):
        if self.h and self.ctx.h:
            lib().jolt_srs_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _srs_upload(self, bases):
    b = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 12)
    h = C.c_void_p()
    _ck(lib().jolt_srs_upload_g1(self.h, _p(b), C.c_size_t(b.shape[0]), C.byref(h)), "jolt_srs_upload_g1", self)
    return Srs(self, h)


def _srs_setup_from_secret(self, beta, count, g1):
    h = C.c_void_p()
    _ck(lib().jolt_srs_setup_from_secret(self.h, _p(fr(beta)), C.c_size_t(count), _p(np.ascontiguousarray(g1, dtype=np.uint64)), C.byref(h)),
        "jolt_srs_setup_from_secret", self)
    return Srs(self, h)


def host_owned_terms(n, block, rank, world):
    """Terms of the prefix [0, n) that `rank` owns under the block-cyclic assignment (host only)."""
    out = C.c_size_t()
    _ck(lib().jolt_host_owned_terms(C.c_size_t(n), C.c_size_t(block), C.c_int32(rank), C.c_int32(world), C.byref(out)), "jolt_host_owned_terms")
    return out.value


def host_subtree_owned_terms(n, rank, world):
    """Terms of the prefix [0, n) that `rank` owns under the subtree assignment (host only)."""
    out = C.c_size_t()
    _ck(lib().jolt_host_subtree_owned_terms(C.c_size_t(n), C.c_int32(rank), C.c_int32(world), C.byref(out)), "jolt_host_subtree_owned_terms")
    return out.value


def host_subtree_term_index(slot, rank, world):
    """Index of the rank's compact slot under the subtree assignment (host only)."""
    out = C.c_size_t()
    _ck(lib().jolt_host_subtree_term_index(C.c_size_t(slot), C.c_int32(rank), C.c_int32(world), C.byref(out)), "jolt_host_subtree_term_index")
    return out.value


def _srs_setup_from_secret_subtree(self, beta, count_global, g1, rank, world):
    """One rank's compact share of the powers under the subtree assignment."""
    h = C.c_void_p()
    _ck(lib().jolt_srs_setup_from_secret_subtree(self.h, _p(fr(beta)), C.c_size_t(count_global), _p(np.ascontiguousarray(g1, dtype=np.uint64)), C.c_int32(rank),
                                                 C.c_int32(world), C.byref(h)), "jolt_srs_setup_from_secret_subtree", self)
    return Srs(self, h)


def _msm_subtree(self, srs, table, n, rank, world):
    """The rank's share of sum_{i<n} table[i] * SRS[i] under the subtree assignment (srs = the rank's compact SRS)."""
    out = g1_array(1)
    _ck(lib().jolt_msm_g1_table_subtree(self.h, srs.h, table.h, C.c_size_t(n), C.c_int32(rank), C.c_int32(world), _p(out)), "jolt_msm_g1_table_subtree", self)
    return out[0]


def _hyperkzg_open_subtree(self, srs, evals_compact, point, label, rank, world, gather_fn, gather_user):
    """jolt_host_hyperkzg_open_subtree: the opening with the polynomial sharded over the ranks (evals_compact: the rank's compact array)."""
    p = fr(point).reshape(-1, 4)
    ell = p.shape[0]
    com, w, v, ch = g1_array(max(ell - 1, 1)), g1_array(3), fr_array(3 * ell), fr_array(3)
    _ck(lib().jolt_host_hyperkzg_open_subtree(self.h, srs.h, evals_compact.h, _p(p), C.c_size_t(ell), C.c_uint64(label), C.c_int32(rank), C.c_int32(world),
                                              gather_fn, gather_user, _p(com), _p(w), _p(v), _p(ch)), "jolt_host_hyperkzg_open_subtree", self)
    return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)


def _srs_setup_from_secret_blocks(self, beta, count_global, g1, block, rank, world):
    """One rank's compact share of the powers under the block-cyclic term assignment (term i belongs to rank (i / block) % world)."""
    h = C.c_void_p()
    _ck(lib().jolt_srs_setup_from_secret_blocks(self.h, _p(fr(beta)), C.c_size_t(count_global), _p(np.ascontiguousarray(g1, dtype=np.uint64)), C.c_size_t(block),
                                                C.c_int32(rank), C.c_int32(world), C.byref(h)), "jolt_srs_setup_from_secret_blocks", self)
    return Srs(self, h)


def _msm_blocks(self, srs, table, n, block, rank, world):
    """The rank's share of sum_{i<n} table[i] * SRS[i] under the block-cyclic assignment (srs = the rank's compact SRS)."""
    out = g1_array(1)
    _ck(lib().jolt_msm_g1_table_blocks(self.h, srs.h, table.h, C.c_size_t(n), C.c_size_t(block), C.c_int32(rank), C.c_int32(world), _p(out)),
        "jolt_msm_g1_table_blocks", self)
    return out[0]


class PendingMsms:
    """jolt_msm_g1_tables_begin ... _finish: up to three prefix MSMs in flight on the side lanes while the caller works on the main stream"""

    def __init__(self, ctx, srs, tables, ns):
        hs = (C.c_void_p * len(tables))(*[t.h for t in tables])
        nn = (C.c_size_t * len(tables))(*[int(x) for x in ns])
        h = C.c_void_p()
        _ck(lib().jolt_msm_g1_tables_begin(ctx.h, srs.h, hs, nn, C.c_size_t(len(tables)), C.byref(h)), "jolt_msm_g1_tables_begin", ctx)
        self.ctx, self.h, self.count = ctx, h, len(tables)

    def finish(self):
        out = g1_array(self.count)
        h, self.h = self.h, None
        _ck(lib().jolt_msm_g1_tables_finish(self.ctx.h, h, _p(out)), "jolt_msm_g1_tables_finish", self.ctx)
        return out


def _msm_tables_begin(self, srs, tables, ns):
    return PendingMsms(self, srs, tables, ns)


def _msm(self, srs, scalars, n=None, full_width=False):
    """JoltGroup::msm(bases = srs[..n], scalars); scalars = numpy (n,4) host array or a device Table.  full_width: the caller knows the scalars to be uniform field
    elements (jolt_msm_g1_table_full_width)."""
    out = g1_array(1)
    if isinstance(scalars, Table):
        n = len(scalars) if n is None else n
        fn = "jolt_msm_g1_table_full_width" if full_width else "jolt_msm_g1_table"
        _ck(getattr(lib(), fn)(self.h, srs.h, scalars.h, C.c_size_t(n), _p(out)), fn, self)
    else:
        s = fr(scalars).reshape(-1, 4)
        n = s.shape[0] if n is None else n
        _ck(lib().jolt_msm_g1(self.h, srs.h, _p(s), C.c_size_t(n), _p(out)), "jolt_msm_g1", self)
    return out[0]


def _msm_profile_buckets(self, enable=True):
    """HIP events around the bucket-sum kernel of every later fixed-base MSM (jolt_msm_profile_buckets)"""
    _ck(lib().jolt_msm_profile_buckets(self.h, C.c_int32(1 if enable else 0)), "jolt_msm_profile_buckets", self)


def _msm_profile_buckets_last(self):
    """(milliseconds, mixed additions) of the last profiled bucket-sum launch"""
    ms, adds = C.c_float(), C.c_uint64()
    _ck(lib().jolt_msm_profile_buckets_last(self.h, C.byref(ms), C.byref(adds)), "jolt_msm_profile_buckets_last", self)
    return float(ms.value), int(adds.value)


def _measure_mad_peak(self, target_ms=50.0):
    """jolt_ctx_measure_mad_peak: (v_mad_u64_u32 lane-operations per second chip-wide, kernel milliseconds timed, launches)"""
    rate, ms, n = C.c_double(), C.c_float(), C.c_uint32()
    _ck(lib().jolt_ctx_measure_mad_peak(self.h, C.c_float(target_ms), C.byref(rate), C.byref(ms), C.byref(n)), "jolt_ctx_measure_mad_peak", self)
    return float(rate.value), float(ms.value), int(n.value)


def _msm_window(self, srs, base_offset, values, kind=None, acc=None):
    """One window of a streamed commitment (jolt_msm_g1_window: StreamingCommitment::feed / feed_u64 / feed_i128 for a KZG-type scheme): acc + sum_i values[i] *
    srs[base_offset + i].  values: (n, 4) uint64 field elements (kind "fr"), uint64 / int64 arrays, or i128 as an (n, 2) uint64 array (kind "i128")."""
    v = np.ascontiguousarray(values)
    if kind is None:
        kind = "fr" if (v.ndim == 2 and v.shape[1] == 4) else ({np.dtype(np.uint64): "u64", np.dtype(np.int64): "i64"}[v.dtype] if v.ndim == 1 else "i128")
    code = {"u64": 0, "i64": 1, "i128": 2, "fr": 3}[kind]
    out = g1_array(1)
    a = None if acc is None else np.ascontiguousarray(acc, dtype=np.uint64).reshape(-1)
    _ck(lib().jolt_msm_g1_window(self.h, srs.h, C.c_size_t(base_offset), C.c_int32(code), v.ctypes.data_as(C.c_void_p), C.c_size_t(v.shape[0]),
                                 None if a is None else _p(a), _p(out)), "jolt_msm_g1_window", self)
    return out[0]


def _hyperkzg_fold(self, evals, point):
    p = fr(point).reshape(-1, 4)
    ell = p.shape[0]
    hs = (C.c_void_p * max(ell, 1))()
    _ck(lib().jolt_hyperkzg_fold(self.h, evals.h, _p(p), C.c_size_t(ell), hs), "jolt_hyperkzg_fold", self)
    return [Table(self, C.c_void_p(hs[i])) for i in range(ell)]


def _hyperkzg_eval3(self, levels, u):
    hs = (C.c_void_p * len(levels))(*[t.h for t in levels])
    uu = fr(u).reshape(3, 4)
    out = fr_array(3 * len(levels))
    _ck(lib().jolt_hyperkzg_eval3(self.h, hs, C.c_size_t(len(levels)), _p(uu), _p(out)), "jolt_hyperkzg_eval3", self)
    return out.reshape(3, len(levels), 4)


def _hyperkzg_rlc(self, levels, q):
    hs = (C.c_void_p * len(levels))(*[t.h for t in levels])
    h = C.c_void_p()
    _ck(lib().jolt_hyperkzg_rlc(self.h, hs, C.c_size_t(len(levels)), _p(fr(q)), C.byref(h)), "jolt_hyperkzg_rlc", self)
    return Table(self, h)


def _hyperkzg_witness_poly(self, f, u):
    h = C.c_void_p()
    _ck(lib().jolt_hyperkzg_witness_poly(self.h, f.h, _p(fr(u)), C.byref(h)), "jolt_hyperkzg_witness_poly", self)
    return Table(self, h)


def _hyperkzg_commit(self, srs, evals):
    out = g1_array(1)
    _ck(lib().jolt_host_hyperkzg_commit(self.h, srs.h, evals.h, _p(out)), "jolt_host_hyperkzg_commit", self)
    return out[0]


def _hyperkzg_open(self, srs, evals, point, label=0, known_levels=None):
    """known_levels: (n, 12) commitments of the first n folded polynomials, computed by the caller (jolt_host_hyperkzg_open_with_levels skips their MSMs)"""
    p = fr(point).reshape(-1, 4)
    ell = p.shape[0]
    com, w, v, ch = g1_array(max(ell - 1, 1)), g1_array(3), fr_array(3 * max(ell, 1)), fr_array(3)
    if known_levels is not None and len(known_levels):
        kl = np.ascontiguousarray(known_levels, dtype=np.uint64).reshape(-1, 12)
        _ck(lib().jolt_host_hyperkzg_open_with_levels(self.h, srs.h, evals.h, _p(p), C.c_size_t(ell), C.c_uint64(label), None, None, _p(kl), C.c_size_t(kl.shape[0]),
                                                      _p(com), _p(w), _p(v), _p(ch)), "jolt_host_hyperkzg_open_with_levels", self)
        return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)
    _ck(lib().jolt_host_hyperkzg_open(self.h, srs.h, evals.h, _p(p), C.c_size_t(ell), C.c_uint64(label), _p(com), _p(w), _p(v), _p(ch)),
        "jolt_host_hyperkzg_open", self)
    return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)


class GridHint:
    """The commitment grid's opening hint (jolt_grid_hint_begin): the class sums of the one-hot columns for the first `levels` level commitments of an opening, in
    flight on the context's background stream (or its main stream) from the moment this returns"""

    def __init__(self, ctx, srs, sources, levels, background=True):
        hs = (C.c_void_p * len(sources))(*[s_.h for s_ in sources])
        h = C.c_void_p()
        _ck(lib().jolt_grid_hint_begin(ctx.h, srs.h, hs, C.c_size_t(len(sources)), C.c_uint32(levels), C.c_int32(1 if background else 0), C.byref(h)), "jolt_grid_hint_begin", ctx)
        self.ctx, self.h, self.levels, self.n_cols, self._keep = ctx, h, levels, sum(s_.n_polys for s_ in sources), list(sources)

    def wait(self):
        _ck(lib().jolt_grid_hint_wait(self.ctx.h, self.h), "jolt_grid_hint_wait", self.ctx)

    def download(self, level):
        out = g1_array(self.n_cols << level)
        _ck(lib().jolt_grid_hint_download(self.ctx.h, self.h, C.c_uint32(level), _p(out)), "jolt_grid_hint_download", self.ctx)
        return out.reshape(1 << level, self.n_cols, 12)

    def free(self):
        if self.h:
            lib().jolt_grid_hint_free(self.ctx.h, self.h)
            self.h = None
        self._keep = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _hyperkzg_open_grid(self, srs, evals, point, hint, levels, onehot_scalars, dense, dense_scalars, label=0):
    """jolt_host_hyperkzg_open_grid: the opening of the grid's joint polynomial with its first `levels` level commitments by linearity from the commit-time hint"""
    p = fr(point).reshape(-1, 4)
    ell = p.shape[0]
    com, w, v, ch = g1_array(max(ell - 1, 1)), g1_array(3), fr_array(3 * max(ell, 1)), fr_array(3)
    osc = np.ascontiguousarray(onehot_scalars, dtype=np.uint64).reshape(-1, 4)
    dsc = np.ascontiguousarray(dense_scalars, dtype=np.uint64).reshape(-1, 4) if len(dense) else None
    hs = (C.c_void_p * max(len(dense), 1))(*[t.h for t in dense])
    _ck(lib().jolt_host_hyperkzg_open_grid(self.h, srs.h, evals.h, _p(p), C.c_size_t(ell), C.c_uint64(label), None, None, hint.h, C.c_uint32(levels), _p(osc), hs,
                                           C.c_size_t(len(dense)), _p(dsc) if dsc is not None else None, _p(com), _p(w), _p(v), _p(ch)), "jolt_host_hyperkzg_open_grid", self)
    return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)


def _srs_precompute_windows(self, srs, window_bits=0, min_terms=0):
    """Fixed-base tables 2^(c*w) * srs[i] (one bucket set for all windows of an MSM): ceil(255/c) copies of the bases in HBM."""
    _ck(lib().jolt_srs_precompute_windows(self.h, srs.h, C.c_uint32(window_bits), C.c_size_t(min_terms)), "jolt_srs_precompute_windows", self)


Context.srs_precompute_windows = _srs_precompute_windows
Context.srs_upload = _srs_upload
Context.srs_setup_from_secret = _srs_setup_from_secret
Context.msm = _msm
Context.msm_window = _msm_window
Context.msm_profile_buckets = _msm_profile_buckets
Context.msm_profile_buckets_last = _msm_profile_buckets_last
Context.measure_mad_peak = _measure_mad_peak
Context.srs_setup_from_secret_blocks = _srs_setup_from_secret_blocks
Context.msm_blocks = _msm_blocks
Context.srs_setup_from_secret_subtree = _srs_setup_from_secret_subtree
Context.msm_subtree = _msm_subtree
Context.hyperkzg_open_subtree = _hyperkzg_open_subtree
Context.hyperkzg_fold = _hyperkzg_fold
Context.msm_tables_begin = _msm_tables_begin
Context.hyperkzg_eval3 = _hyperkzg_eval3
Context.hyperkzg_rlc = _hyperkzg_rlc
Context.hyperkzg_witness_poly = _hyperkzg_witness_poly
Context.hyperkzg_commit = _hyperkzg_commit
Context.hyperkzg_open = _hyperkzg_open
Context.hyperkzg_open_grid = _hyperkzg_open_grid
Context.grid_hint = lambda self, srs, sources, levels, background=True: GridHint(self, srs, sources, levels, background)

OPEN_TRANSCRIPT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p)


def _hyperkzg_open_with_transcript(self, srs, evals, point, absorb_points, absorb_values, challenge):
    """jolt_host_hyperkzg_open_with_transcript: the opening under the CALLER's transcript.  absorb_points((n, 12) Jacobian limbs) / absorb_values((n, 4)) absorb what
    the prover sends at a step, challenge() -> (4,) draws the challenge after it (three steps: level commitments -> r, evaluations -> q, witness commitments -> d_0)."""
    p = fr(point).reshape(-1, 4)
    ell = p.shape[0]
    com, w, v, ch = g1_array(max(ell - 1, 1)), g1_array(3), fr_array(3 * max(ell, 1)), fr_array(3)
    err = []

    def cb(user, phase, points, n_points, values, n_values, out):
        try:
            if n_points:
                absorb_points(np.ctypeslib.as_array(C.cast(points, C.POINTER(C.c_uint64)), shape=(n_points, 12)).copy())
            if n_values:
                absorb_values(np.ctypeslib.as_array(C.cast(values, C.POINTER(C.c_uint64)), shape=(n_values, 4)).copy())
            c = np.ascontiguousarray(challenge(), dtype=np.uint64).reshape(4)
            C.memmove(out, c.ctypes.data, 32)
            return 0
        except Exception as e:  # never unwind through the C frame
            err.append(e)
            return 4

    fn = OPEN_TRANSCRIPT_FN(cb)
    st = lib().jolt_host_hyperkzg_open_with_transcript(self.h, srs.h, evals.h, _p(p), C.c_size_t(ell), fn, None, _p(com), _p(w), _p(v), _p(ch))
    if err:
        raise err[0]
    _ck(st, "jolt_host_hyperkzg_open_with_transcript", self)
    return dict(com=com[: ell - 1], w=w, v=v.reshape(3, ell, 4), challenges=ch)


Context.hyperkzg_open_with_transcript = _hyperkzg_open_with_transcript


def host_g1_add(p, q):
    o = g1_array(1)
    _ck(lib().jolt_host_g1_add(_p(np.ascontiguousarray(p, dtype=np.uint64)), _p(np.ascontiguousarray(q, dtype=np.uint64)), _p(o)), "jolt_host_g1_add")
    return o[0]


def host_g1_is_on_curve(p):
    e = C.c_int32()
    _ck(lib().jolt_host_g1_is_on_curve(_p(np.ascontiguousarray(p, dtype=np.uint64)), C.byref(e)), "jolt_host_g1_is_on_curve")
    return bool(e.value)


def host_g1_eq(p, q):
    e = C.c_int32()
    _ck(lib().jolt_host_g1_eq(_p(np.ascontiguousarray(p, dtype=np.uint64)), _p(np.ascontiguousarray(q, dtype=np.uint64)), C.byref(e)), "jolt_host_g1_eq")
    return bool(e.value)


def host_g1_serialize_compressed(p):
    out = (C.c_uint8 * 32)()
    _ck(lib().jolt_host_g1_serialize_compressed(_p(np.ascontiguousarray(p, dtype=np.uint64)), out), "jolt_host_g1_serialize_compressed")
    return bytes(out)


class PinnedBuffer:
    """Page-locked host memory (jolt_host_pinned_alloc) as a numpy uint8 array of the given shape: the staging block a tracer fills and Rows uploads from"""

    def __init__(self, ctx, shape):
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = int(np.prod(shape))
        p = C.c_void_p()
        _ck(lib().jolt_host_pinned_alloc(ctx.h, C.c_size_t(n), C.byref(p)), "jolt_host_pinned_alloc", ctx)
        self.ctx, self.p = ctx, p
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).reshape(shape)

    def free(self):
        if self.p:
            self.array = None
            _ck(lib().jolt_host_pinned_free(self.ctx.h, self.p), "jolt_host_pinned_free", self.ctx)
            self.p = None


class Rows:
    """Packed typed witness rows (numpy structured or 2-D uint8 array) resident on the device; columns are expanded there."""

    def __init__(self, ctx, rows):
        raw = np.ascontiguousarray(rows)
        self.n_rows, self.row_bytes = raw.shape[0], raw.dtype.itemsize if raw.ndim == 1 else raw.shape[1]
        h = C.c_void_p()
        _ck(lib().jolt_rows_upload(ctx.h, raw.ctypes.data_as(C.c_void_p), C.c_size_t(self.n_rows), C.c_size_t(self.row_bytes), C.byref(h)), "jolt_rows_upload", ctx)
        self.ctx, self.h = ctx, h

    @classmethod
    def begin(cls, ctx, pinned_rows):
        """jolt_rows_upload_begin: the copy starts now on the context's copy stream and runs beside whatever the context does next; `pinned_rows` is a PinnedBuffer's
        array (page-locked) and must stay unchanged until wait()"""
        raw = pinned_rows
        assert raw.flags["C_CONTIGUOUS"] and raw.ndim == 2
        self = cls.__new__(cls)
        self.n_rows, self.row_bytes = raw.shape[0], raw.shape[1]
        h = C.c_void_p()
        _ck(lib().jolt_rows_upload_begin(ctx.h, raw.ctypes.data_as(C.c_void_p), C.c_size_t(self.n_rows), C.c_size_t(self.row_bytes), C.byref(h)), "jolt_rows_upload_begin", ctx)
        self.ctx, self.h = ctx, h
        return self

    def wait(self):
        _ck(lib().jolt_rows_upload_wait(self.ctx.h, self.h), "jolt_rows_upload_wait", self.ctx)

    def table(self, offset, width, signed=False):
        h = C.c_void_p()
        _ck(lib().jolt_table_from_rows(self.ctx.h, self.h, C.c_size_t(offset), C.c_uint32(width), C.c_int32(1 if signed else 0), C.byref(h)), "jolt_table_from_rows",
            self.ctx)
        return Table(self.ctx, h)

    def ints(self, offset, width, signed=False):
        """the field as a resident integer column (jolt_ints_from_rows): compact scalars, no promotion"""
        h = C.c_void_p()
        _ck(lib().jolt_ints_from_rows(self.ctx.h, self.h, C.c_size_t(offset), C.c_uint32(width), C.c_int32(1 if signed else 0), C.byref(h)), "jolt_ints_from_rows", self.ctx)
        v = Ints.__new__(Ints)
        v.ctx, v.kind, v.count, v.h = self.ctx, "i64" if signed else "u64", self.n_rows, h
        return v

    def ints_many(self, fields):
        """[(offset, width, signed)] -> one Ints per field, all extracted in ONE pass over the rows (jolt_ints_from_rows_many)"""
        n = len(fields)
        offs = (C.c_size_t * n)(*[f[0] for f in fields])
        widths = (C.c_uint32 * n)(*[f[1] for f in fields])
        signed = (C.c_int32 * n)(*[1 if f[2] else 0 for f in fields])
        hs = (C.c_void_p * n)()
        _ck(lib().jolt_ints_from_rows_many(self.ctx.h, self.h, offs, widths, signed, C.c_size_t(n), hs), "jolt_ints_from_rows_many", self.ctx)
        out = []
        for k, f in enumerate(fields):
            v = Ints.__new__(Ints)
            v.ctx, v.kind, v.count, v.h = self.ctx, "i64" if f[2] else "u64", self.n_rows, C.c_void_p(hs[k])
            out.append(v)
        return out

    def onehot(self, offset, width, shifts, log_k, valid_offset=None):
        sh = (C.c_uint32 * len(shifts))(*shifts)
        h = C.c_void_p()
        vo = C.c_size_t(valid_offset if valid_offset is not None else 2**64 - 1)
        _ck(lib().jolt_onehot_from_rows(self.ctx.h, self.h, C.c_size_t(offset), C.c_uint32(width), sh, C.c_size_t(len(shifts)), C.c_uint32(log_k), vo, C.byref(h)),
            "jolt_onehot_from_rows", self.ctx)
        src = OneHot.__new__(OneHot)
        src.ctx, src.n_polys, src.cycles, src.k, src.h, src.wide = self.ctx, len(shifts), self.n_rows, 1 << log_k, h, log_k > 7
        return src

    def free(self):
        if self.h:
            lib().jolt_rows_free(self.ctx.h, self.h)
            self.h = None


class SplitLt:
    """LT(., r) + constant from split tables, bound low-to-high (optimized/support.rs:640-760)."""

    def __init__(self, ctx, r_cycle, constant=None):
        r = fr(r_cycle).reshape(-1, 4)
        h = C.c_void_p()
        _ck(lib().jolt_split_lt_create(ctx.h, _p(r) if r.shape[0] else None, C.c_size_t(r.shape[0]), _p(fr(constant)) if constant is not None else None,
                                       C.byref(h)), "jolt_split_lt_create", ctx)
        self.ctx, self.h = ctx, h

    def __len__(self):
        n = C.c_size_t()
        _ck(lib().jolt_split_lt_len(self.h, C.byref(n)), "jolt_split_lt_len", self.ctx)
        return n.value

    def bind(self, r):
        _ck(lib().jolt_split_lt_bind(self.ctx.h, self.h, _p(fr(r))), "jolt_split_lt_bind", self.ctx)

    def to_dense(self):
        h = C.c_void_p()
        _ck(lib().jolt_split_lt_to_dense(self.ctx.h, self.h, C.byref(h)), "jolt_split_lt_to_dense", self.ctx)
        return Table(self.ctx, h)

    def final_value(self):
        o = fr_array(1)
        _ck(lib().jolt_split_lt_final_value(self.ctx.h, self.h, _p(o)), "jolt_split_lt_final_value", self.ctx)
        return o[0]

    def free(self):
        if self.h:
            lib().jolt_split_lt_free(self.ctx.h, self.h)
            self.h = None


class OneHot:
    """Hot indices of N one-hot selector columns (uint8 [n_polys, cycles], 0xFF = cold cycle) resident on the device."""

    def __init__(self, ctx, indices, k):
        """indices: uint8 (0xFF = cold, k <= 255) or uint16 (0xFFFF = cold: the K = 256 chunks of long traces)"""
        wide = np.asarray(indices).dtype == np.uint16
        idx = np.ascontiguousarray(indices, dtype=np.uint16 if wide else np.uint8)
        assert idx.ndim == 2
        self.ctx, self.n_polys, self.cycles, self.k, self.wide = ctx, idx.shape[0], idx.shape[1], k, wide
        h = C.c_void_p()
        fn = lib().jolt_onehot_upload16 if wide else lib().jolt_onehot_upload
        _ck(fn(ctx.h, idx.ctypes.data_as(C.c_void_p), C.c_size_t(idx.shape[0]), C.c_size_t(idx.shape[1]), C.c_uint32(k), C.byref(h)), "jolt_onehot_upload", ctx)
        self.h = h

    def download(self):
        wide = self.wide
        out = np.empty((self.n_polys, self.cycles), dtype=np.uint16 if wide else np.uint8)
        fn = lib().jolt_onehot_download16 if wide else lib().jolt_onehot_download
        _ck(fn(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p)), "jolt_onehot_download", self.ctx)
        return out

    def materialize(self, poly, scale_table):
        h = C.c_void_p()
        _ck(lib().jolt_onehot_materialize(self.ctx.h, self.h, C.c_size_t(poly), scale_table.h, C.byref(h)), "jolt_onehot_materialize", self.ctx)
        return Table(self.ctx, h)

    def pushforward(self, weights):
        h = C.c_void_p()
        _ck(lib().jolt_onehot_pushforward(self.ctx.h, self.h, weights.h, C.byref(h)), "jolt_onehot_pushforward", self.ctx)
        return Table(self.ctx, h)

    def free(self):
        if self.h:
            lib().jolt_onehot_free(self.ctx.h, self.h)
            self.h = None


class Ints:
    """Small machine integers resident on the device for the Dory tier-1 row commitments: uint64 / int64 numpy arrays, or
    i128 as an (n, 2) uint64 array (low, high; two's complement) / a list of Python ints with kind="i128"."""
    KINDS = {"u64": 0, "i64": 1, "i128": 2}

    def __init__(self, ctx, values, kind=None):
        if kind == "i128" and not isinstance(values, np.ndarray):
            values = np.array([[int(v) & (2**64 - 1), (int(v) >> 64) & (2**64 - 1)] for v in values], dtype=np.uint64).reshape(-1, 2)
        v = np.ascontiguousarray(values)
        if kind is None:
            kind = {np.dtype(np.uint64): "u64", np.dtype(np.int64): "i64"}[v.dtype] if v.ndim == 1 else "i128"
        self.ctx, self.kind = ctx, kind
        self.count = v.shape[0]
        h = C.c_void_p()
        _ck(lib().jolt_ints_upload(ctx.h, v.ctypes.data_as(C.c_void_p), C.c_int32(self.KINDS[kind]), C.c_size_t(self.count), C.byref(h)), "jolt_ints_upload", ctx)
        self.h = h

    def free(self):
        if self.h:
            lib().jolt_ints_free(self.ctx.h, self.h)
            self.h = None


def _dory_commit_rows(self, srs, values, row_width):
    """DoryScheme::feed_u64 / feed_i128 over every row_width window of `values` (an Ints): one G1 point per row."""
    rows = values.count // row_width if row_width else 0
    out = g1_array(max(rows, 1))
    _ck(lib().jolt_dory_commit_rows(self.h, srs.h, values.h, C.c_size_t(row_width), _p(out)), "jolt_dory_commit_rows", self)
    return out[:rows]


def _dory_commit_onehot(self, srs, source, poly, chunk_width):
    """DoryScheme::process_one_hot_chunks_with for hot-index column `poly`: (chunks, k) G1 points."""
    chunks = source.cycles // chunk_width if chunk_width else 0
    out = g1_array(max(chunks * source.k, 1))
    _ck(lib().jolt_dory_commit_onehot(self.h, srs.h, source.h, C.c_size_t(poly), C.c_size_t(chunk_width), _p(out)), "jolt_dory_commit_onehot", self)
    return out[:chunks * source.k].reshape(chunks, source.k, -1)


def _member_lazy_ra_uniform(self, source, scale_tables, V, F, coeffs, w, scale=None, shard_scale=None):
    """eq(w,.) * sum_v coeffs[v] * prod_{i<F} ra_{vF+i}, ra_p(j) = scale_tables[p][index(p,j)], lazily bound (LazyFoldedRa)."""
    w = fr(w).reshape(-1, 4)
    co = np.ascontiguousarray(np.stack([fr(c) for c in coeffs])).reshape(-1, 4)
    st = np.ascontiguousarray(scale_tables, dtype=np.uint64).reshape(-1, 4)
    h = C.c_void_p()
    if shard_scale is not None:
        _ck(lib().jolt_member_create_lazy_ra_uniform_sharded(self.h, source.h, _p(st), C.c_uint32(V), C.c_uint32(F), _p(co), _p(w), C.c_size_t(w.shape[0]),
                                                             _p(fr(scale)) if scale is not None else None, _p(fr(shard_scale)), C.byref(h)),
            "jolt_member_create_lazy_ra_uniform_sharded", self)
    else:
        _ck(lib().jolt_member_create_lazy_ra_uniform(self.h, source.h, _p(st), C.c_uint32(V), C.c_uint32(F), _p(co), _p(w), C.c_size_t(w.shape[0]),
                                                     _p(fr(scale)) if scale is not None else None, C.byref(h)), "jolt_member_create_lazy_ra_uniform", self)
    m = Member(self, h, F + 1, V * F, True, False)
    m.n_evals = F
    m.uniform = True
    m._keepalive = [source]
    return m


def _member_lazy_booleanity(self, source, scale_tables, rho, w, scale=None):
    """eq(w,.) * sum_i (H_i^2 - rho_i H_i), H_i(j) = scale_tables[i][index(i,j)] (gamma-pre-scaled), lazily bound; 2 sums per round."""
    w = fr(w).reshape(-1, 4)
    st = np.ascontiguousarray(scale_tables, dtype=np.uint64).reshape(-1, 4)
    rh = np.ascontiguousarray(np.stack([fr(c) for c in rho])).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_member_create_lazy_booleanity(self.h, source.h, _p(st), _p(rh), _p(w), C.c_size_t(w.shape[0]),
                                                 _p(fr(scale)) if scale is not None else None, C.byref(h)), "jolt_member_create_lazy_booleanity", self)
    m = Member(self, h, 3, rh.shape[0], True, False)
    m.n_evals = 2
    m._keepalive = [source]
    return m


Context.member_lazy_booleanity = _member_lazy_booleanity
Context.member_lazy_ra_uniform = _member_lazy_ra_uniform
Context.onehot = lambda self, indices, k: OneHot(self, indices, k)
Context.ints = lambda self, values, kind=None: Ints(self, values, kind)
Context.dory_commit_rows = _dory_commit_rows
Context.dory_commit_onehot = _dory_commit_onehot


def _table_op2(name):
    def f(self, a, b):
        h = C.c_void_p()
        _ck(getattr(lib(), name)(self.h, a.h, b.h, C.byref(h)), name, self)
        return Table(self, h)
    return f


def _tile(self, base, copies):
    h = C.c_void_p()
    _ck(lib().jolt_tile(self.h, base.h, C.c_size_t(copies), C.byref(h)), "jolt_tile", self)
    return Table(self, h)


def _replicate_stream_lsb(self, base):
    h = C.c_void_p()
    _ck(lib().jolt_replicate_stream_lsb(self.h, base.h, C.byref(h)), "jolt_replicate_stream_lsb", self)
    return Table(self, h)


def _rlc(self, tables, scalars):
    hs = (C.c_void_p * len(tables))(*[t.h for t in tables])
    sc = fr(scalars).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_rlc(self.h, hs, C.c_size_t(len(tables)), _p(sc), C.byref(h)), "jolt_rlc", self)
    return Table(self, h)


Context.address_fold = _table_op2("jolt_address_fold")
Context.cycle_fold = _table_op2("jolt_cycle_fold")
Context.tile = _tile
Context.replicate_stream_lsb = _replicate_stream_lsb
Context.rlc = _rlc


# ---- device-resident integer promotion, commitment-grid pieces (pcs.hip), memory pool
def _table_from_ints(self, values, offset=0, length=None):
    """Ring::from_u64 / from_i64 / from_i128 of entries [offset, offset+length) of an Ints (already in HBM) as a field table."""
    length = values.count - offset if length is None else length
    h = C.c_void_p()
    _ck(lib().jolt_table_from_ints(self.h, values.h, C.c_size_t(offset), C.c_size_t(length), C.byref(h)), "jolt_table_from_ints", self)
    return Table(self, h)


def _grid_commit_onehot(self, srs, source):
    """KZG commitments of the one-hot columns of `source` over the K x T commitment grid (index k*T + j): (n_polys, 12)."""
    out = g1_array(source.n_polys)
    _ck(lib().jolt_grid_commit_onehot(self.h, srs.h, source.h, _p(out)), "jolt_grid_commit_onehot", self)
    return out


def _grid_commit_onehot_classes(self, srs, source, shift):
    """(2^shift, n_polys, 12): per residue class c of the cycle mod 2^shift the sums of the bases at (hot * T + j) >> shift (jolt_grid_commit_onehot_classes)"""
    out = g1_array(source.n_polys << shift)
    _ck(lib().jolt_grid_commit_onehot_classes(self.h, srs.h, source.h, C.c_uint32(shift), _p(out)), "jolt_grid_commit_onehot_classes", self)
    return out.reshape(1 << shift, source.n_polys, 12)


def _grid_joint_polynomial(self, sources, onehot_scalars, dense, dense_scalars, log_k):
    """Joint polynomial of the stage-8 batch opening over the 2^log_k x T grid (HomomorphicBatch::prove_batch's RLC)."""
    hs = (C.c_void_p * max(len(sources), 1))(*[s.h for s in sources])
    ds = (C.c_void_p * max(len(dense), 1))(*[t.h for t in dense])
    osc = fr(np.stack([fr(c) for c in onehot_scalars])).reshape(-1, 4) if len(onehot_scalars) else None
    dsc = fr(np.stack([fr(c) for c in dense_scalars])).reshape(-1, 4) if len(dense_scalars) else None
    h = C.c_void_p()
    _ck(lib().jolt_grid_joint_polynomial(self.h, hs if sources else None, C.c_size_t(len(sources)), _p(osc), ds if dense else None, C.c_size_t(len(dense)),
                                         _p(dsc), C.c_uint32(log_k), C.byref(h)), "jolt_grid_joint_polynomial", self)
    return Table(self, h)


def _grid_joint_polynomial_subtree(self, sources, onehot_scalars, dense, dense_scalars, log_k, rank, world):
    """The rank's compact array (subtree assignment) of the same joint polynomial: 2^log_k * T / world coefficients."""
    hs = (C.c_void_p * max(len(sources), 1))(*[s.h for s in sources])
    ds = (C.c_void_p * max(len(dense), 1))(*[t.h for t in dense])
    osc = fr(np.stack([fr(c) for c in onehot_scalars])).reshape(-1, 4) if len(onehot_scalars) else None
    dsc = fr(np.stack([fr(c) for c in dense_scalars])).reshape(-1, 4) if len(dense_scalars) else None
    h = C.c_void_p()
    _ck(lib().jolt_grid_joint_polynomial_subtree(self.h, hs if sources else None, C.c_size_t(len(sources)), _p(osc), ds if dense else None, C.c_size_t(len(dense)),
                                                 _p(dsc), C.c_uint32(log_k), C.c_int32(rank), C.c_int32(world), C.byref(h)), "jolt_grid_joint_polynomial_subtree", self)
    return Table(self, h)


def _trim(self):
    _ck(lib().jolt_ctx_trim(self.h), "jolt_ctx_trim", self)


def _memory_stats(self):
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _ck(lib().jolt_ctx_memory_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), "jolt_ctx_memory_stats", self)
    out = {"live_bytes": a.value, "cached_bytes": b.value, "peak_bytes": c.value}
    la, ba = C.c_size_t(), C.c_size_t()
    _ck(lib().jolt_ctx_workspace_stats(self.h, C.byref(la), C.byref(ba)), "jolt_ctx_workspace_stats", self)
    out["msm_lanes_bytes"], out["msm_batch_bytes"] = la.value, ba.value  # outside the pool: grow-only MSM workspaces
    return out


Context.table_from_ints = _table_from_ints
Context.grid_commit_onehot = _grid_commit_onehot
Context.grid_commit_onehot_classes = _grid_commit_onehot_classes
Context.grid_joint_polynomial = _grid_joint_polynomial
Context.grid_joint_polynomial_subtree = _grid_joint_polynomial_subtree
Context.trim = _trim
Context.memory_stats = _memory_stats


class RwMatrix:
    """Device twin of the optimized RAM read/write-checking kernel's sparse matrix (jolt_rw_matrix_*): same interface as the oracle's
    RwMatrix wrapper plus the fused prove_round."""

    def __init__(self, ctx, addresses, pre, post, inc, val_init, tau_low, gamma):
        """addresses / pre / post: host uint64 arrays (uploaded here), or three u64 Ints already resident in HBM"""
        tau = fr(tau_low).reshape(-1, 4)
        self.ctx = ctx
        h = C.c_void_p()
        if isinstance(addresses, Ints):
            _ck(lib().jolt_rw_matrix_create_resident(ctx.h, addresses.h, pre.h, post.h, inc.h, val_init.h, _p(tau), _p(fr(gamma)), C.byref(h)),
                "jolt_rw_matrix_create_resident", ctx)
        else:
            a, p, q = (np.ascontiguousarray(x, dtype=np.uint64) for x in (addresses, pre, post))
            _ck(lib().jolt_rw_matrix_create(ctx.h, _p(a), _p(p), _p(q), C.c_size_t(a.shape[0]), inc.h, val_init.h, _p(tau), _p(fr(gamma)), C.byref(h)),
                "jolt_rw_matrix_create", ctx)
        self.h = h

    def __len__(self):
        n = C.c_size_t()
        _ck(lib().jolt_rw_matrix_len(self.h, C.byref(n)), "jolt_rw_matrix_len", self.ctx)
        return n.value

    def prove_round(self, bind=None):
        evals, aux = fr_array(2), fr_array(3)
        _ck(lib().jolt_rw_matrix_prove_round(self.h, _p(fr(bind)) if bind is not None else None, _p(evals), _p(aux)), "jolt_rw_matrix_prove_round", self.ctx)
        return evals, aux

    def finish(self, bind):
        _ck(lib().jolt_rw_matrix_finish(self.h, _p(fr(bind))), "jolt_rw_matrix_finish", self.ctx)

    def final_values(self):
        out = fr_array(4)
        _ck(lib().jolt_rw_matrix_final_values(self.h, _p(out)), "jolt_rw_matrix_final_values", self.ctx)
        return out

    def download(self):
        n = len(self)
        rows, cols = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint64)
        val, ra, prev, nxt = fr_array(max(n, 1)), fr_array(max(n, 1)), fr_array(max(n, 1)), fr_array(max(n, 1))
        _ck(lib().jolt_rw_matrix_download(self.h, _p(rows), _p(cols), _p(val), _p(ra), _p(prev), _p(nxt)), "jolt_rw_matrix_download", self.ctx)
        return dict(rows=rows[:n], cols=cols[:n], val=val[:n], ra=ra[:n], prev=prev[:n], next=nxt[:n])

    def free(self):
        if self.h:
            lib().jolt_rw_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


Context.rw_matrix = lambda self, *a, **k: RwMatrix(self, *a, **k)


# ---- the sharded form of both sparse matrices (jolt_rw_matrix_hold_row / _bind / _export_row / _create_merged): shared by RwMatrix and RegistersRw handles
def rw_hold_row(m):
    _ck(lib().jolt_rw_matrix_hold_row(m.h), "jolt_rw_matrix_hold_row", m.ctx)


def rw_bind(m, bind):
    _ck(lib().jolt_rw_matrix_bind(m.h, _p(fr(bind))), "jolt_rw_matrix_bind", m.ctx)


def rw_export_row(m, registers):
    """the single row a held local matrix is left with: dict(cols, prev, next (u64), val, ra[, wa] (n, 4), inc, scalar)"""
    n = len(m)
    cap = max(n, 1)
    cols, prev, nxt = (np.zeros(cap, dtype=np.uint64) for _ in range(3))
    val, ra, wa = fr_array(cap), fr_array(cap), fr_array(cap)
    inc, scalar = fr_array(1), fr_array(1)
    got = C.c_size_t()
    _ck(lib().jolt_rw_matrix_export_row(m.h, C.c_size_t(cap), _p(cols), _p(prev), _p(nxt), _p(val), _p(ra), _p(wa) if registers else None, _p(inc), _p(scalar), C.byref(got)),
        "jolt_rw_matrix_export_row", m.ctx)
    n = got.value
    out = dict(cols=cols[:n], prev=prev[:n], next=nxt[:n], val=val[:n], ra=ra[:n], inc=inc[0], scalar=scalar[0])
    if registers:
        out["wa"] = wa[:n]
    return out


class MergedRw:
    """jolt_rw_matrix_create_merged: the matrix of the remaining log G cycle variables built from the ranks' exported rows (rank order); continues with the
    ordinary rounds.  registers: the registers flavour (four address-round points, five final values)."""

    def __init__(self, ctx, registers, log_rows, log_k, rows, cols, prev, nxt, val, ra, wa, inc, val_init, w, scalar, gamma):
        self.ctx, self.registers = ctx, bool(registers)
        u = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        rows, cols, prev, nxt = u(rows), u(cols), u(prev), u(nxt)
        n = rows.shape[0]
        val, ra = u(val).reshape(-1, 4), u(ra).reshape(-1, 4)
        wa = u(wa).reshape(-1, 4) if registers else None
        inc = u(inc).reshape(-1, 4)
        assert inc.shape[0] == 1 << log_rows
        h = C.c_void_p()
        _ck(lib().jolt_rw_matrix_create_merged(ctx.h, C.c_int32(1 if registers else 0), C.c_size_t(log_rows), C.c_size_t(log_k), C.c_size_t(n), _p(rows), _p(cols), _p(prev), _p(nxt),
                                               _p(val), _p(ra), _p(wa) if registers else None, _p(inc), val_init.h if val_init is not None else None,
                                               _p(fr(w).reshape(-1, 4)), _p(fr(scalar)), _p(fr(gamma)), C.byref(h)), "jolt_rw_matrix_create_merged", ctx)
        self.h = h

    def __len__(self):
        n = C.c_size_t()
        _ck(lib().jolt_rw_matrix_len(self.h, C.byref(n)), "jolt_rw_matrix_len", self.ctx)
        return n.value

    def prove_round(self, bind=None):
        evals, aux = fr_array(4 if self.registers else 2), fr_array(3)
        fn = lib().jolt_registers_rw_prove_round if self.registers else lib().jolt_rw_matrix_prove_round
        _ck(fn(self.h, _p(fr(bind)) if bind is not None else None, _p(evals), _p(aux)), "jolt_rw_matrix_prove_round (merged)", self.ctx)
        return evals, aux

    def finish(self, bind):
        _ck(lib().jolt_rw_matrix_finish(self.h, _p(fr(bind))), "jolt_rw_matrix_finish", self.ctx)

    def final_values(self):
        out = fr_array(5 if self.registers else 4)
        fn = lib().jolt_registers_rw_final_values if self.registers else lib().jolt_rw_matrix_final_values
        _ck(fn(self.h, _p(out)), "jolt_rw_matrix_final_values (merged)", self.ctx)
        return out

    def free(self):
        if self.h:
            lib().jolt_rw_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class KeyIndex:
    """Rows of a resident key column (u64 Ints: bytecode PCs, RAM word addresses; a key >= k is a cold cycle) sorted by key once, for pushforwards
    of cycle weights onto the k-entry address domain (jolt_key_index_*)."""

    def __init__(self, ctx, keys, k):
        self.ctx, self.k = ctx, int(k)
        h = C.c_void_p()
        _ck(lib().jolt_key_index_create(ctx.h, keys.h, C.c_uint64(self.k), C.byref(h)), "jolt_key_index_create", ctx)
        self.h = h

    def size(self):
        cycles, k, items = C.c_size_t(), C.c_uint64(), C.c_uint32()
        _ck(lib().jolt_key_index_size(self.h, C.byref(cycles), C.byref(k), C.byref(items)), "jolt_key_index_size", self.ctx)
        return cycles.value, k.value, items.value

    def pushforward(self, weights):
        """weights: list of <= 8 Tables over the cycles -> list of Tables of k entries, out[s][a] = sum of weights[s] over the cycles with key a"""
        n = len(weights)
        hs = (C.c_void_p * n)(*[w.h for w in weights])
        out = (C.c_void_p * n)()
        _ck(lib().jolt_key_index_pushforward(self.ctx.h, self.h, hs, C.c_size_t(n), out), "jolt_key_index_pushforward", self.ctx)
        return [Table(self.ctx, C.c_void_p(out[i])) for i in range(n)]

    def last_value(self, values, init):
        """values: u64 Ints over the cycles; init: Table of k entries -> Table: the value at the latest cycle of every key, init where a key never occurs"""
        h = C.c_void_p()
        _ck(lib().jolt_key_index_last_value(self.ctx.h, self.h, values.h, init.h, C.byref(h)), "jolt_key_index_last_value", self.ctx)
        return Table(self.ctx, h)

    def free(self):
        if self.h:
            lib().jolt_key_index_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


Context.key_index = lambda self, keys, k: KeyIndex(self, keys, k)


class RegistersRw:
    """Device twin of the optimized registers read/write-checking kernel (jolt_registers_rw_*): `regs` a OneHot with the columns rs1, rs2, rd,
    the value columns u64 Ints, `inc` the RdInc table."""

    def __init__(self, ctx, regs, rs1_val, rs2_val, rd_pre, rd_post, inc, r_cycle, gamma):
        self.ctx, self.regs = ctx, regs
        h = C.c_void_p()
        _ck(lib().jolt_registers_rw_create(ctx.h, regs.h, rs1_val.h, rs2_val.h, rd_pre.h, rd_post.h, inc.h, _p(fr(r_cycle).reshape(-1, 4)), _p(fr(gamma)), C.byref(h)),
            "jolt_registers_rw_create", ctx)
        self.h = h

    def __len__(self):
        n = C.c_size_t()
        _ck(lib().jolt_rw_matrix_len(self.h, C.byref(n)), "jolt_rw_matrix_len", self.ctx)
        return n.value

    def prove_round(self, bind=None):
        evals, aux = fr_array(4), fr_array(3)
        _ck(lib().jolt_registers_rw_prove_round(self.h, _p(fr(bind)) if bind is not None else None, _p(evals), _p(aux)), "jolt_registers_rw_prove_round", self.ctx)
        return evals, aux

    def finish(self, bind):
        _ck(lib().jolt_rw_matrix_finish(self.h, _p(fr(bind))), "jolt_rw_matrix_finish", self.ctx)

    def final_values(self):
        out = fr_array(5)
        _ck(lib().jolt_registers_rw_final_values(self.h, _p(out)), "jolt_registers_rw_final_values", self.ctx)
        return out

    def download(self):
        n = len(self)
        rows, cols, prev, nxt = (np.zeros(max(n, 1), dtype=np.uint64) for _ in range(4))
        val, ra, wa = fr_array(max(n, 1)), fr_array(max(n, 1)), fr_array(max(n, 1))
        _ck(lib().jolt_registers_rw_download(self.h, _p(rows), _p(cols), _p(val), _p(ra), _p(wa), _p(prev), _p(nxt)), "jolt_registers_rw_download", self.ctx)
        return dict(rows=rows[:n], cols=cols[:n], val=val[:n], ra=ra[:n], wa=wa[:n], prev=prev[:n], next=nxt[:n])

    def free(self):
        if self.h:
            lib().jolt_rw_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


Context.registers_rw = lambda self, *a, **k: RegistersRw(self, *a, **k)


def _table_dot(self, a, b, deferred=False):
    out = fr_array(1)
    _ck(lib().jolt_table_dot(self.h, a.h, b.h, C.c_int32(1 if deferred else 0), _p(out)), "jolt_table_dot", self)
    return out[0]


Context.table_dot = _table_dot


# ---- Spartan outer T-scale sums (r1cs.hip)
def _handles(tables):
    return (C.c_void_p * max(len(tables), 1))(*[t.h for t in tables])


def _r1cs_uniskip_sums(self, inputs, eq, a_weights, b_weights):
    wa = fr(a_weights).reshape(-1, 2, 1 + len(inputs), 4)
    wb = fr(b_weights).reshape(-1, 2, 1 + len(inputs), 4)
    out = fr_array(wa.shape[0])
    _ck(lib().jolt_r1cs_uniskip_sums(self.h, _handles(inputs), C.c_size_t(len(inputs)), eq.h, _p(wa), _p(wb), C.c_size_t(wa.shape[0]), _p(out)), "jolt_r1cs_uniskip_sums", self)
    return out


def _r1cs_materialize(self, inputs, a_weights, b_weights):
    wa, wb = fr(a_weights).reshape(2, 1 + len(inputs), 4), fr(b_weights).reshape(2, 1 + len(inputs), 4)
    az, bz = C.c_void_p(), C.c_void_p()
    _ck(lib().jolt_r1cs_materialize(self.h, _handles(inputs), C.c_size_t(len(inputs)), _p(wa), _p(wb), C.byref(az), C.byref(bz)), "jolt_r1cs_materialize", self)
    return Table(self, az), Table(self, bz)


def _tables_evaluate(self, tables, point):
    p = fr(point).reshape(-1, 4)
    out = fr_array(len(tables))
    _ck(lib().jolt_tables_evaluate(self.h, _handles(tables), C.c_size_t(len(tables)), _p(p), C.c_size_t(p.shape[0]), _p(out)), "jolt_tables_evaluate", self)
    return out


def _r1cs_uniskip_sums_small(self, inputs, eq, a_weights, b_weights, streams=2):
    """inputs: Ints columns; a_weights / b_weights: int64 [node][stream][1 + n] (integer Lagrange-extension column weights);
    streams = 2: Spartan outer, streams = 1: product virtualization (no stream variable)"""
    wa = np.ascontiguousarray(a_weights, dtype=np.int64).reshape(-1, streams, 1 + len(inputs))
    wb = np.ascontiguousarray(b_weights, dtype=np.int64).reshape(-1, streams, 1 + len(inputs))
    out = fr_array(wa.shape[0])
    _ck(lib().jolt_r1cs_uniskip_sums_small(self.h, _handles(inputs), C.c_size_t(len(inputs)), eq.h, C.c_uint32(streams), wa.ctypes.data_as(C.c_void_p),
                                           wb.ctypes.data_as(C.c_void_p), C.c_size_t(wa.shape[0]), _p(out)), "jolt_r1cs_uniskip_sums_small", self)
    return out


def _r1cs_materialize_small(self, inputs, a_weights, b_weights, streams=2):
    wa, wb = fr(a_weights).reshape(streams, 1 + len(inputs), 4), fr(b_weights).reshape(streams, 1 + len(inputs), 4)
    az, bz = C.c_void_p(), C.c_void_p()
    _ck(lib().jolt_r1cs_materialize_small(self.h, _handles(inputs), C.c_size_t(len(inputs)), C.c_uint32(streams), _p(wa), _p(wb), C.byref(az), C.byref(bz)),
        "jolt_r1cs_materialize_small", self)
    return Table(self, az), Table(self, bz)


def _ints_evaluate(self, columns, point):
    p = fr(point).reshape(-1, 4)
    out = fr_array(len(columns))
    _ck(lib().jolt_ints_evaluate(self.h, _handles(columns), C.c_size_t(len(columns)), _p(p), C.c_size_t(p.shape[0]), _p(out)), "jolt_ints_evaluate", self)
    return out


Context.r1cs_uniskip_sums_small = _r1cs_uniskip_sums_small
Context.r1cs_materialize_small = _r1cs_materialize_small
Context.ints_evaluate = _ints_evaluate
Context.r1cs_uniskip_sums = _r1cs_uniskip_sums
Context.r1cs_materialize = _r1cs_materialize
Context.tables_evaluate = _tables_evaluate


def _rows_window_table(self, offset, width, signed=False, lookahead=0, cycles=None, padding_value=0, none_value=0):
    """RandomAccessRows::window: field of row j (or j + 1) for every cycle of the padded domain, padding / None rows as given."""
    cycles = self.n_rows if cycles is None else cycles
    h = C.c_void_p()
    _ck(lib().jolt_table_from_rows_window(self.ctx.h, self.h, C.c_size_t(offset), C.c_uint32(width), C.c_int32(1 if signed else 0), C.c_int32(lookahead),
                                          C.c_size_t(cycles), C.c_int64(padding_value), C.c_int64(none_value), C.byref(h)), "jolt_table_from_rows_window", self.ctx)
    return Table(self.ctx, h)


def _rows_onehot_sentinel(self, offset, width, shifts, log_k, cycles=None):
    """hot-index columns from a `value + 1, 0 = none` packed address field (InstructionCycleRow)"""
    cycles = self.n_rows if cycles is None else cycles
    sh = (C.c_uint32 * len(shifts))(*shifts)
    h = C.c_void_p()
    _ck(lib().jolt_onehot_from_rows_sentinel(self.ctx.h, self.h, C.c_size_t(offset), C.c_uint32(width), sh, C.c_size_t(len(shifts)), C.c_uint32(log_k),
                                             C.c_size_t(cycles), C.byref(h)), "jolt_onehot_from_rows_sentinel", self.ctx)
    src = OneHot.__new__(OneHot)
    src.ctx, src.n_polys, src.cycles, src.k, src.h, src.wide = self.ctx, len(shifts), cycles, 1 << log_k, h, log_k > 7
    return src


Rows.window_table = _rows_window_table
Rows.onehot_sentinel = _rows_onehot_sentinel


# ---- instruction read+RAF checking scans (read_raf.hip) ---------------------------------------------------------------------------
class ReadRaf:
    """lookup_index: (T, 2) uint64 (lo, hi); table_index: uint8 (0xFF = none); raf_flag: uint8"""

    def __init__(self, ctx, lookup_index, table_index, raf_flag, n_tables):
        idx = np.ascontiguousarray(lookup_index, dtype=np.uint64).reshape(-1, 2)
        tab = np.ascontiguousarray(table_index, dtype=np.uint8)
        raf = np.ascontiguousarray(raf_flag, dtype=np.uint8)
        assert idx.shape[0] == tab.shape[0] == raf.shape[0]
        self.ctx, self.cycles, self.n_tables = ctx, idx.shape[0], n_tables
        h = C.c_void_p()
        _ck(lib().jolt_read_raf_create(ctx.h, idx.ctypes.data_as(C.c_void_p), tab.ctypes.data_as(C.c_void_p), raf.ctypes.data_as(C.c_void_p), C.c_size_t(self.cycles),
                                       C.c_uint32(n_tables), C.byref(h)), "jolt_read_raf_create", ctx)
        self.h = h

    def phase_scan(self, u, suffix_len, address_bits, suffix_lists, canonical=False):
        """suffix_lists: per table the list of suffix kind ids -> (raf (6, 256, 4), suffix sums (total, 256, 4))"""
        offs = np.zeros(self.n_tables + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(l) for l in suffix_lists])
        kinds = np.array([k for l in suffix_lists for k in l], dtype=np.uint8)
        raf = fr_array(6 * 256)
        suf = fr_array(max(int(offs[-1]) * 256, 1))
        _ck(lib().jolt_read_raf_phase_scan(self.ctx.h, self.h, u.h, C.c_uint32(suffix_len), C.c_uint32(address_bits), C.c_int32(1 if canonical else 0),
                                           offs.ctypes.data_as(C.c_void_p), kinds.ctypes.data_as(C.c_void_p) if kinds.size else None, _p(raf), _p(suf)),
            "jolt_read_raf_phase_scan", self.ctx)
        return raf.reshape(6, 256, 4), suf[: int(offs[-1]) * 256].reshape(-1, 256, 4)

    def condense(self, u, v_table, shift):
        v = fr(v_table).reshape(256, 4)
        _ck(lib().jolt_read_raf_condense(self.ctx.h, self.h, u.h, _p(v), C.c_uint32(shift)), "jolt_read_raf_condense", self.ctx)

    def cycle_tables(self, table_values, raf_interleaved, raf_identity, v_tables, address_bits, ra_count):
        tv = fr(table_values).reshape(self.n_tables, 4)
        vt = fr(v_tables).reshape(-1, 256, 4)
        combined = C.c_void_p()
        ra = (C.c_void_p * ra_count)()
        _ck(lib().jolt_read_raf_cycle_tables(self.ctx.h, self.h, _p(tv), _p(fr(raf_interleaved).reshape(4)), _p(fr(raf_identity).reshape(4)), _p(vt), C.c_uint32(vt.shape[0]),
                                             C.c_uint32(address_bits), C.c_uint32(ra_count), C.byref(combined), ra), "jolt_read_raf_cycle_tables", self.ctx)
        return Table(self.ctx, combined), [Table(self.ctx, C.c_void_p(p)) for p in ra]

    def free(self):
        if self.h:
            lib().jolt_read_raf_destroy(self.ctx.h, self.h)
            self.h = None


Context.read_raf = lambda self, lookup_index, table_index, raf_flag, n_tables: ReadRaf(self, lookup_index, table_index, raf_flag, n_tables)


def host_fq_limb_op(op, a, b=None, c=None, d=None):
    """fq_limb.hip.h on the host: op 0 a*b, 1 a^2, 2 a*b + c*d, 3 (a - b)*c, 4 (a - b - 2c)*d over canonical Fq (standard Montgomery limbs)"""
    o = fr_array(1)
    arg = lambda x: _p(fr(x)) if x is not None else None
    _ck(lib().jolt_host_fq_limb_op(C.c_int32(op), arg(a), arg(b), arg(c), arg(d), _p(o)), "jolt_host_fq_limb_op")
    return o[0]


def host_g1_sum_limb_form(points, negate=None):
    """sum of affine points ((n, 8) uint64: x, y in standard Montgomery form; (0, 0) = infinity) through the limb-form XYZZ accumulator"""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    neg = np.ascontiguousarray(negate if negate is not None else np.zeros(pts.shape[0]), dtype=np.uint8)
    out = g1_array(1)
    _ck(lib().jolt_host_g1_sum_limb_form(pts.ctypes.data_as(C.c_void_p), neg.ctypes.data_as(C.c_void_p), C.c_size_t(pts.shape[0]), _p(out)), "jolt_host_g1_sum_limb_form")
    return out[0]


def host_fx_digits(scalar, window_bits):
    """(signed digits as Python ints, bucket count) of the fixed-base MSM's recoding of one Montgomery-form scalar"""
    keys = np.zeros(64, dtype=np.uint32)
    nw, nb = C.c_uint32(), C.c_uint32()
    _ck(lib().jolt_host_fx_digits(_p(fr(scalar)), C.c_uint32(window_bits), keys.ctypes.data_as(C.c_void_p), C.byref(nw), C.byref(nb)), "jolt_host_fx_digits")
    return [(-1 if int(k) >> 31 else 1) * (int(k) & 0x7FFFFFFF) for k in keys[: nw.value]], nb.value


def host_fx_segment_capacity(n, window_bits, segment):
    """the region (in entries) the capacity sort of an n-term fixed-base MSM gives segment `segment` of 256 buckets (jolt_host_fx_segment_capacity)"""
    cap = C.c_uint32()
    _ck(lib().jolt_host_fx_segment_capacity(C.c_uint64(n), C.c_uint32(window_bits), C.c_uint32(segment), C.byref(cap)), "jolt_host_fx_segment_capacity")
    return int(cap.value)


def host_suffix_mle(kind, bits, length):
    out = C.c_uint64()
    _ck(lib().jolt_host_suffix_mle(C.c_uint32(kind), C.c_uint64(bits & (2**64 - 1)), C.c_uint64((bits >> 64) & (2**64 - 1)), C.c_uint32(length), C.byref(out)), "jolt_host_suffix_mle")
    return out.value


def host_small_scalar_dot(values, scalars):
    """sum_k values[k] * scalars[k] (Python ints within i128) through the device's small-scalar accumulator, on the host"""
    v = fr(values).reshape(-1, 4)
    sc = np.array([[int(x) & (2**64 - 1), (int(x) >> 64) & (2**64 - 1)] for x in scalars], dtype=np.uint64).reshape(-1, 2)
    o = fr_array(1)
    _ck(lib().jolt_host_small_scalar_dot(_p(v), sc.ctypes.data_as(C.c_void_p), C.c_size_t(v.shape[0]), _p(o)), "jolt_host_small_scalar_dot")
    return o[0]


# ---- instruction read+RAF: the address rounds on the host (read_raf_address.hip, lookup_tables.hpp) ---------------------------------------------------------
NUM_LOOKUP_TABLES, NUM_LOOKUP_PREFIXES = 42, 49


def lookup_table_suffixes(kind):
    out, n = np.zeros(8, dtype=np.uint8), C.c_uint32()
    _ck(lib().jolt_lookup_table_suffixes(C.c_uint32(kind), _p(out), C.byref(n)), "jolt_lookup_table_suffixes")
    return [int(k) for k in out[: n.value]]


def lookup_table_prefixes(kind):
    out, n = np.zeros(8, dtype=np.uint8), C.c_uint32()
    _ck(lib().jolt_lookup_table_prefixes(C.c_uint32(kind), _p(out), C.byref(n)), "jolt_lookup_table_prefixes")
    return [int(k) for k in out[: n.value]]


def lookup_suffix_lists():
    """LookupTableKind::suffixes() of the 42 tables: the suffix_lists argument of ReadRaf.phase_scan for real tables"""
    offs = np.zeros(NUM_LOOKUP_TABLES + 1, dtype=np.uint32)
    _ck(lib().jolt_lookup_suffix_layout(_p(offs), None), "jolt_lookup_suffix_layout")
    kinds = np.zeros(int(offs[-1]), dtype=np.uint8)
    _ck(lib().jolt_lookup_suffix_layout(_p(offs), _p(kinds)), "jolt_lookup_suffix_layout")
    return [[int(k) for k in kinds[offs[t]: offs[t + 1]]] for t in range(NUM_LOOKUP_TABLES)]


def host_lookup_prefix_default_checkpoints():
    out = fr_array(NUM_LOOKUP_PREFIXES)
    _ck(lib().jolt_host_lookup_prefix_default_checkpoints(_p(out)), "jolt_host_lookup_prefix_default_checkpoints")
    return out


def host_lookup_prefix_evaluate(prefix, checkpoints, b, b_len, suffix_len):
    out = fr_array(1)
    _ck(lib().jolt_host_lookup_prefix_evaluate(C.c_uint32(prefix), _p(fr(checkpoints).reshape(NUM_LOOKUP_PREFIXES, 4)), C.c_uint32(b), C.c_uint32(b_len), C.c_uint32(suffix_len),
                                               _p(out)), "jolt_host_lookup_prefix_evaluate")
    return out[0]


def host_lookup_prefix_table(prefix, checkpoints, b_len, suffix_len):
    out = fr_array(1 << b_len)
    _ck(lib().jolt_host_lookup_prefix_table(C.c_uint32(prefix), _p(fr(checkpoints).reshape(NUM_LOOKUP_PREFIXES, 4)), C.c_uint32(b_len), C.c_uint32(suffix_len), _p(out)),
        "jolt_host_lookup_prefix_table")
    return out


def host_lookup_table_combine(kind, prefixes, suffixes):
    out = fr_array(1)
    _ck(lib().jolt_host_lookup_table_combine(C.c_uint32(kind), _p(fr(prefixes).reshape(NUM_LOOKUP_PREFIXES, 4)), _p(fr(suffixes).reshape(-1, 4)), _p(out)),
        "jolt_host_lookup_table_combine")
    return out[0]


class HostReadRafAddress:
    """The 256-entry state of the 16 address phases (jolt_host_read_raf_address_*): prefix polynomials, suffix accumulators, RAF decompositions, checkpoints"""

    def __init__(self, gamma, table_present, canonical=False):
        present = np.zeros(NUM_LOOKUP_TABLES, dtype=np.uint8)
        present[: len(table_present)] = np.asarray(table_present, dtype=np.uint8)
        h = C.c_void_p()
        _ck(lib().jolt_host_read_raf_address_create(_p(fr(gamma).reshape(4)), _p(present), C.c_int32(1 if canonical else 0), C.byref(h)), "jolt_host_read_raf_address_create")
        self.h = h

    def init_phase(self, phase, raf_sums, suffix_sums):
        _ck(lib().jolt_host_read_raf_address_init_phase(self.h, C.c_uint32(phase), _p(fr(raf_sums).reshape(6 * 256, 4)), _p(fr(suffix_sums).reshape(-1, 4))),
            "jolt_host_read_raf_address_init_phase")

    def message(self, previous_claim=None):
        """s(0), s(1), s(2); without a running claim s(1) is summed from the tables (round 0: s(0) + s(1) = the input claim)"""
        o = fr_array(3)
        _ck(lib().jolt_host_read_raf_address_message(self.h, _p(fr(previous_claim).reshape(4)) if previous_claim is not None else None, _p(o)), "jolt_host_read_raf_address_message")
        return o

    def bind(self, r):
        done = C.c_int32()
        _ck(lib().jolt_host_read_raf_address_bind(self.h, _p(fr(r).reshape(4)), C.byref(done)), "jolt_host_read_raf_address_bind")
        return bool(done.value)

    def prove_phase(self, claim, transcript):
        """the 8 rounds of the open phase against a HostTranscript: -> (claim after the phase, coefficients (8, 3, 4), challenges (8, 4))"""
        c = fr(claim).reshape(4).copy()
        coeffs, challenges = fr_array(24), fr_array(8)
        _ck(lib().jolt_host_read_raf_address_prove_phase(self.h, _p(c), None, None, transcript.h, _p(coeffs), _p(challenges)), "jolt_host_read_raf_address_prove_phase")
        return c, coeffs.reshape(8, 3, 4), challenges

    def v_table(self, phase):
        o = fr_array(256)
        _ck(lib().jolt_host_read_raf_address_v_table(self.h, C.c_uint32(phase), _p(o)), "jolt_host_read_raf_address_v_table")
        return o

    def finish(self):
        tv, a, b = fr_array(NUM_LOOKUP_TABLES), fr_array(1), fr_array(1)
        _ck(lib().jolt_host_read_raf_address_finish(self.h, _p(tv), _p(a), _p(b)), "jolt_host_read_raf_address_finish")
        return tv, a[0], b[0]

    def close(self):
        if self.h:
            lib().jolt_host_read_raf_address_destroy(self.h)
            self.h = None


def host_small_round_pair(n_tables, groups, is_int, int_pairs, fr_pairs, n_evals, skip_one):
    """jolt_host_small_round_pair: the per-pair evaluation of the integer round kernel (small_round.hip.h) compiled for the host.  groups as in Context.member_lc;
    int_pairs (n_tables, 2) uint64, fr_pairs (n_tables, 2, 4) Montgomery limbs -> (n_evals, 4)"""
    goff, foff, consts, ltab, lcoef = [0], [0], [], [], []
    zero = np.zeros(4, dtype=np.uint64)
    for g in groups:
        for const, entries in g:
            consts.append(zero if const is None else fr(const))
            for c, ti in entries:
                lcoef.append(fr(c))
                ltab.append(ti)
            foff.append(len(ltab))
        goff.append(len(consts))
    goff, foff = np.array(goff, dtype=np.uint32), np.array(foff, dtype=np.uint32)
    consts_a = np.ascontiguousarray(np.stack(consts)) if consts else fr_array(1)
    ltab_a = np.array(ltab if ltab else [0], dtype=np.uint32)
    lcoef_a = np.ascontiguousarray(np.stack(lcoef)) if lcoef else fr_array(1)
    d = MemberLcDesc(n_tables, len(groups), len(consts), len(ltab), n_evals, ORDER_LOW_TO_HIGH, 0, goff.ctypes.data, foff.ctypes.data, consts_a.ctypes.data, ltab_a.ctypes.data,
                     lcoef_a.ctypes.data)
    mask = np.ascontiguousarray(is_int, dtype=np.uint8)
    ip = np.ascontiguousarray(int_pairs, dtype=np.uint64).reshape(n_tables, 2)
    fp = np.ascontiguousarray(fr_pairs, dtype=np.uint64).reshape(n_tables, 2, 4)
    out = fr_array(n_evals)
    _ck(lib().jolt_host_small_round_pair(C.byref(d), _p(mask), _p(ip), _p(fp), C.c_uint32(n_evals), C.c_int32(1 if skip_one else 0), _p(out)), "jolt_host_small_round_pair")
    return out


# ---- stage operators as ProveRounds objects (stage_ops.hip): one per backend slot ---------------------------------------------------------------------------
class StageOp:
    """jolt_stage_op: prove_round / finish_rounds / output_claims of one stage operator.  `keep`: Python objects the operator borrows (they must outlive it)."""

    def __init__(self, ctx, handle, keep=()):
        self.ctx, self.h, self._keep = ctx, handle, list(keep)

    @property
    def rounds(self):
        n = C.c_size_t()
        _ck(lib().jolt_stage_op_num_rounds(self.h, C.byref(n)), "jolt_stage_op_num_rounds", self.ctx)
        return n.value

    @property
    def degree(self):
        n = C.c_size_t()
        _ck(lib().jolt_stage_op_degree(self.h, C.byref(n)), "jolt_stage_op_degree", self.ctx)
        return n.value

    def input_claim(self):
        o = fr_array(1)
        _ck(lib().jolt_stage_op_input_claim(self.h, _p(o)), "jolt_stage_op_input_claim", self.ctx)
        return o[0]

    def prove_round(self, bind, rnd, previous_claim):
        """ProveRounds::prove_round: the round message as coefficients (n, 4)"""
        cap = self.degree + 1
        out, n = fr_array(cap), C.c_size_t()
        _ck(lib().jolt_stage_op_prove_round(self.h, _p(fr(bind)) if bind is not None else None, C.c_size_t(rnd), _p(fr(previous_claim)), _p(out), C.c_size_t(cap), C.byref(n)),
            "jolt_stage_op_prove_round", self.ctx)
        return out[: n.value].copy()

    def finish_rounds(self, bind):
        _ck(lib().jolt_stage_op_finish_rounds(self.h, _p(fr(bind))), "jolt_stage_op_finish_rounds", self.ctx)

    def output_claims(self):
        n = C.c_size_t()
        out = fr_array(256)
        _ck(lib().jolt_stage_op_output_claims(self.h, _p(out), C.c_size_t(256), C.byref(n)), "jolt_stage_op_output_claims", self.ctx)
        return out[: n.value].copy()

    def kept(self, key):
        n = C.c_size_t()
        _ck(lib().jolt_stage_op_kept(self.h, key.encode(), None, C.c_size_t(0), C.byref(n)), "jolt_stage_op_kept", self.ctx)
        out = fr_array(max(n.value, 1))
        _ck(lib().jolt_stage_op_kept(self.h, key.encode(), _p(out), C.c_size_t(n.value), C.byref(n)), "jolt_stage_op_kept", self.ctx)
        return out[: n.value]

    def window(self, first, n):
        h = C.c_void_p()
        _ck(lib().jolt_stage_op_window(self.h, C.c_size_t(first), C.c_size_t(n), C.byref(h)), "jolt_stage_op_window", self.ctx)
        return StageOp(self.ctx, h, keep=[self])

    def prove_alone(self, transcript, claim):
        """jolt_host_stage_op_prove_alone: -> dict(polys = per round the message's coefficients, challenges, final_claim)"""
        rounds, stride = self.rounds, self.degree + 1
        c = fr(claim).reshape(4).copy()
        coeffs, counts, chal = fr_array(max(rounds * stride, 1)), np.zeros(max(rounds, 1), dtype=np.uint32), fr_array(max(rounds, 1))
        _ck(lib().jolt_host_stage_op_prove_alone(self.h, transcript.h, _p(c), _p(coeffs), C.c_size_t(stride), counts.ctypes.data_as(C.c_void_p), _p(chal)),
            "jolt_host_stage_op_prove_alone", self.ctx)
        coeffs = coeffs.reshape(-1, stride, 4)
        return dict(polys=[coeffs[r, : counts[r]].copy() for r in range(rounds)], challenges=chal[:rounds].copy() if rounds else np.zeros((0, 4), dtype=np.uint64), final_claim=c)

    def destroy(self):
        if self.h:
            lib().jolt_stage_op_destroy(self.h)
            self.h = None
        self._keep = []

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def stage_host_expr(tables, terms, degree):
    """jolt_stage_host_expr_create: the dense member over HOST tables as a stage operator (no device, no context): terms = [(coeff_limbs, [table indices]), ...]"""
    tabs = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
    offs, facs = [0], []
    for _, f in terms:
        facs.extend(f)
        offs.append(len(facs))
    offs = np.array(offs, dtype=np.uint32)
    facs_a = np.array(facs if facs else [0], dtype=np.uint32)
    coeffs = np.ascontiguousarray(np.stack([fr(c) for c, _ in terms])).reshape(-1, 4)
    d = MemberDesc(len(tabs), len(terms), degree, ORDER_LOW_TO_HIGH, offs.ctypes.data, facs_a.ctypes.data, coeffs.ctypes.data)
    ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
    h = C.c_void_p()
    _ck(lib().jolt_stage_host_expr_create(ptrs, C.c_size_t(tabs[0].shape[0]), C.byref(d), C.byref(h)), "jolt_stage_host_expr_create")
    return StageOp(None, h)


def prove_batch_ops(ops, input_claims, coefficients, offsets, max_num_vars, max_degree, label=0, challenge_mode=0):
    """jolt_host_prove_batch_ops without a context (host-only operators)"""
    return _prove_batch_ops(None, ops, input_claims, coefficients, offsets, max_num_vars, max_degree, label, challenge_mode)


def _prove_batch_ops(self, ops, input_claims, coefficients, offsets, max_num_vars, max_degree, label=0, challenge_mode=0):
    """prove_batch (prover.rs:193-362) over stage operators (jolt_host_prove_batch_ops): the same outputs as Context.prove_batch"""
    n = len(ops)
    hs = (C.c_void_p * n)(*[o.h for o in ops])
    ic = np.ascontiguousarray(np.stack(input_claims), dtype=np.uint64).reshape(-1, 4)
    co = np.ascontiguousarray(np.stack(coefficients), dtype=np.uint64).reshape(-1, 4)
    offs = (C.c_size_t * n)(*offsets)
    polys, chal = fr_array(max(max_num_vars * (max_degree + 1), 1)), fr_array(max(max_num_vars, 1))
    mclaims, final = fr_array(n), fr_array(1)
    _ck(lib().jolt_host_prove_batch_ops(self.h if self is not None else None, hs, C.c_size_t(n), _p(ic), _p(co), offs, C.c_size_t(max_num_vars), C.c_size_t(max_degree), C.c_uint64(label),
                                        C.c_int32(challenge_mode), _p(polys), _p(chal), _p(mclaims), _p(final)), "jolt_host_prove_batch_ops", self)
    return dict(polys=polys[: max_num_vars * (max_degree + 1)].reshape(max_num_vars, max_degree + 1, 4), challenges=chal[:max_num_vars], member_claims=mclaims, final_claim=final[0])


def _stage_spartan_uniskip_sums(self, cols, tau, a_weights, b_weights, streams):
    wa = np.ascontiguousarray(a_weights, dtype=np.int64).reshape(-1, streams, 1 + len(cols))
    wb = np.ascontiguousarray(b_weights, dtype=np.int64).reshape(-1, streams, 1 + len(cols))
    t = fr(tau).reshape(-1, 4)
    out = fr_array(wa.shape[0])
    _ck(lib().jolt_stage_spartan_uniskip_sums(self.h, _handles(cols), C.c_size_t(len(cols)), C.c_uint32(streams), _p(t), C.c_size_t(t.shape[0]), wa.ctypes.data_as(C.c_void_p),
                                              wb.ctypes.data_as(C.c_void_p), C.c_size_t(wa.shape[0]), _p(out)), "jolt_stage_spartan_uniskip_sums", self)
    return out


def _stage_spartan_remainder(self, cols, fa, fb, tau, scale, streams):
    wa, wb = fr(fa).reshape(streams, 1 + len(cols), 4), fr(fb).reshape(streams, 1 + len(cols), 4)
    t = fr(tau).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_stage_spartan_remainder_create(self.h, _handles(cols), C.c_size_t(len(cols)), C.c_uint32(streams), _p(wa), _p(wb), _p(t), C.c_size_t(t.shape[0]),
                                                  _p(fr(scale)) if scale is not None else None, C.byref(h)), "jolt_stage_spartan_remainder_create", self)
    return StageOp(self, h, keep=cols)


def _stage_ram_read_write(self, addresses, pre, post, inc, val_init, tau_low, gamma):
    h = C.c_void_p()
    _ck(lib().jolt_stage_ram_read_write_create(self.h, addresses.h, pre.h, post.h, inc.h, val_init.h, _p(fr(tau_low).reshape(-1, 4)), _p(fr(gamma)), C.byref(h)),
        "jolt_stage_ram_read_write_create", self)
    return StageOp(self, h, keep=[addresses, pre, post, inc, val_init])


def _stage_registers_read_write(self, regs, rs1_val, rs2_val, rd_pre, rd_post, inc, r_cycle, gamma):
    h = C.c_void_p()
    _ck(lib().jolt_stage_registers_read_write_create(self.h, regs.h, rs1_val.h, rs2_val.h, rd_pre.h, rd_post.h, inc.h, _p(fr(r_cycle).reshape(-1, 4)), _p(fr(gamma)), C.byref(h)),
        "jolt_stage_registers_read_write_create", self)
    return StageOp(self, h, keep=[regs, rs1_val, rs2_val, rd_pre, rd_post, inc])


def _stage_booleanity_address(self, cols, reference_cycle, reference_address, gamma):
    rc = fr(reference_cycle).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_stage_booleanity_address_create(self.h, cols.h, _p(rc), C.c_size_t(rc.shape[0]), _p(fr(reference_address).reshape(-1, 4)), _p(fr(gamma)), C.byref(h)),
        "jolt_stage_booleanity_address_create", self)
    return StageOp(self, h, keep=[cols])


def _stage_booleanity_cycle(self, cols, r_address, reference_address, reference_cycle, gamma):
    rc = fr(reference_cycle).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_stage_booleanity_cycle_create(self.h, cols.h, _p(fr(r_address).reshape(-1, 4)), _p(fr(reference_address).reshape(-1, 4)), _p(rc), C.c_size_t(rc.shape[0]),
                                                 _p(fr(gamma)), C.byref(h)), "jolt_stage_booleanity_cycle_create", self)
    return StageOp(self, h, keep=[cols])


def _stage_hamming_weight(self, cols, r_cycle, r_address, virtualization_points, gamma):
    rc = fr(r_cycle).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_stage_hamming_weight_create(self.h, cols.h, _p(rc), C.c_size_t(rc.shape[0]), _p(fr(r_address).reshape(-1, 4)),
                                               _p(np.ascontiguousarray(virtualization_points, dtype=np.uint64).reshape(-1, 4)), _p(fr(gamma)), C.byref(h)),
        "jolt_stage_hamming_weight_create", self)
    return StageOp(self, h, keep=[cols])


def _stage_instruction_read_raf(self, rows, claim_columns, r_reduction, gamma, table_present, ra_count):
    rr = fr(r_reduction).reshape(-1, 4)
    present = np.zeros(NUM_LOOKUP_TABLES, dtype=np.uint8)
    present[: len(table_present)] = np.asarray(table_present, dtype=np.uint8)
    h = C.c_void_p()
    _ck(lib().jolt_stage_instruction_read_raf_create(self.h, rows.h, claim_columns.h, _p(rr), C.c_size_t(rr.shape[0]), _p(fr(gamma)), _p(present), C.c_uint32(ra_count), C.byref(h)),
        "jolt_stage_instruction_read_raf_create", self)
    return StageOp(self, h, keep=[rows, claim_columns])


def _stage_bytecode_read_raf_address(self, pc_index, stage_points, stage_values, gamma, first_pc, entry_index):
    sp = np.ascontiguousarray(stage_points, dtype=np.uint64).reshape(5, -1, 4)
    sv = np.ascontiguousarray(stage_values, dtype=np.uint64).reshape(5, -1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_stage_bytecode_read_raf_address_create(self.h, pc_index.h, _p(sp), C.c_size_t(sp.shape[1]), _p(sv), _p(fr(gamma)), C.c_uint64(int(first_pc)),
                                                          C.c_uint64(int(entry_index)), C.byref(h)), "jolt_stage_bytecode_read_raf_address_create", self)
    return StageOp(self, h, keep=[pc_index])


def _stage_bytecode_read_raf_cycle(self, address_op, pc_chunks, chunk_bits):
    h = C.c_void_p()
    _ck(lib().jolt_stage_bytecode_read_raf_cycle_create(self.h, address_op.h, pc_chunks.h, C.c_uint32(chunk_bits), C.byref(h)), "jolt_stage_bytecode_read_raf_cycle_create", self)
    return StageOp(self, h, keep=[pc_chunks])


def _stage_ram_raf_evaluation(self, ram_index, tau_low, lowest_address):
    t = fr(tau_low).reshape(-1, 4)
    h = C.c_void_p()
    _ck(lib().jolt_stage_ram_raf_evaluation_create(self.h, ram_index.h, _p(t), C.c_size_t(t.shape[0]), C.c_uint64(int(lowest_address)), C.byref(h)),
        "jolt_stage_ram_raf_evaluation_create", self)
    return StageOp(self, h, keep=[ram_index])


def _stage_ram_output_check(self, ram_index, post_values, val_init, val_io, io_lo, io_len, r_address):
    vi, vo = np.ascontiguousarray(val_init, dtype=np.uint64), np.ascontiguousarray(val_io, dtype=np.uint64)
    h = C.c_void_p()
    _ck(lib().jolt_stage_ram_output_check_create(self.h, ram_index.h, post_values.h, _p(vi), _p(vo), C.c_uint64(int(io_lo)), C.c_uint64(int(io_len)), _p(fr(r_address).reshape(-1, 4)),
                                                 C.byref(h)), "jolt_stage_ram_output_check_create", self)
    return StageOp(self, h, keep=[ram_index, post_values])


Context.prove_batch_ops = _prove_batch_ops
Context.stage_spartan_uniskip_sums = _stage_spartan_uniskip_sums
Context.stage_spartan_remainder = _stage_spartan_remainder
Context.stage_ram_read_write = _stage_ram_read_write
Context.stage_registers_read_write = _stage_registers_read_write
Context.stage_booleanity_address = _stage_booleanity_address
Context.stage_booleanity_cycle = _stage_booleanity_cycle
Context.stage_hamming_weight = _stage_hamming_weight
Context.stage_instruction_read_raf = _stage_instruction_read_raf
Context.stage_bytecode_read_raf_address = _stage_bytecode_read_raf_address
Context.stage_bytecode_read_raf_cycle = _stage_bytecode_read_raf_cycle
Context.stage_ram_raf_evaluation = _stage_ram_raf_evaluation
Context.stage_ram_output_check = _stage_ram_output_check
