#!/usr/bin/env python3
"""Generate rust/jolt-kernels-hip/src/ffi.rs from include/jolt_hip.h (and parse either side for the agreement test).

The image has no Rust toolchain, so the FFI crate cannot be compiled here; what CAN be guaranteed mechanically is that the Rust
declarations name every entry point of the C header with the same arity and the same types.  tests/test_abi_cpu.py re-parses both
files with the functions below and compares them declaration by declaration.

    python tools/gen_rust_ffi.py            # rewrite ffi.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if ffi.rs is stale
"""
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HEADER = os.path.join(ROOT, "include", "jolt_hip.h")
FFI_RS = os.path.join(ROOT, "rust", "jolt-kernels-hip", "src", "ffi.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "uint16_t": "u16", "size_t": "usize", "float": "f32", "double": "f64",
           "void": "c_void", "char": "c_char"}
def opaque_handles(path=None):
    """every `typedef struct X X;` of the header, in declaration order: the handle types ffi.rs must define (a hand-kept list went stale in round 3)"""
    return re.findall(r"typedef\s+struct\s+(jolt_\w+)\s+\1\s*;", open(path or HEADER).read())


FNPTR = {"jolt_local_round_fn", "jolt_gather_fn", "jolt_round_transcript_fn", "jolt_open_transcript_fn"}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def c_type_to_rust(ctype):
    """'const jolt_fr_t *const *' -> '*const *const jolt_fr_t' (pointer levels read right to left)."""
    t = ctype.strip()
    toks = re.findall(r"\*|\w+", t)
    base, i = None, 0
    base_const = False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        elif toks[i] != "struct":
            base = toks[i]
        i += 1
    rust = SCALARS.get(base, base)
    const_here = base_const
    while i < len(toks):
        assert toks[i] == "*", ctype
        rust = ("*const " if const_here else "*mut ") + rust
        i += 1
        const_here = False
        if i < len(toks) and toks[i] == "const":
            const_here = True
            i += 1
    if base in FNPTR and not rust.startswith("*"):
        return base
    return rust


def parse_header(path=HEADER):
    """-> list of (name, ret_rust, [(param_name, rust_type)]) in declaration order"""
    src = strip_comments(open(path).read())
    src = re.sub(r"#.*", "", src)
    src = re.sub(r'extern\s+"C"\s*\{', "", src)
    decls = []
    for m in re.finditer(r"(?:^|;|\})\s*((?:const\s+)?\w+\s*\**)\s*(jolt_\w+)\s*\(([^;{}]*?)\)\s*(?=;)", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        stmt_start = max(src.rfind(";", 0, m.start(2)), src.rfind("}", 0, m.start(2))) + 1
        if "typedef" in src[stmt_start:m.start(2)]:
            continue
        plist = []
        params = " ".join(params.split())
        if params and params != "void":
            for k, p in enumerate(params.split(",")):
                p = p.strip()
                arr = re.match(r"(.*?)(\w+)\s*\[\s*\d*\s*\]$", p)  # `uint8_t out[32]` / `const jolt_fr_t u[3]` decay to pointers
                if arr:
                    ctype, pname = arr.group(1).strip() + " *", arr.group(2)
                else:
                    mm = re.match(r"(.*?)(\w+)$", p)
                    ctype, pname = mm.group(1).strip(), mm.group(2)
                    if not ctype:  # unnamed parameter
                        ctype, pname = p, f"arg{k}"
                plist.append((pname, c_type_to_rust(ctype)))
        decls.append((name, c_type_to_rust(ret), plist))
    return decls


def parse_rust(path=FFI_RS):
    src = re.sub(r"//.*", "", open(path).read())
    decls = []
    for m in re.finditer(r"pub fn (jolt_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", src, flags=re.S):
        name, params, ret = m.group(1), " ".join(m.group(2).split()), (m.group(3) or "()").strip()
        plist = []
        if params:
            for p in re.split(r",\s*(?=(?:r#)?\w+\s*:)", params.rstrip(", ")):
                pname, ptype = p.split(":", 1)
                plist.append((pname.strip(), ptype.strip()))
        decls.append((name, ret, plist))
    return decls


RUST_KEYWORDS = {"type", "ref", "box", "in", "fn", "loop", "match", "move", "self", "where", "use", "mod", "impl", "struct", "enum"}


def render(decls):
    out = []
    out.append("//! Raw FFI declarations of `libjolt_hip.so` -- GENERATED from `include/jolt_hip.h` by `tools/gen_rust_ffi.py`; do not edit.")
    out.append("//! One declaration per entry point of the C header, same order, same arity, same types (checked by tests/test_abi_cpu.py).")
    out.append("//! `jolt_fr_t` is bit-identical to `jolt_field::Fr` (4 x u64 Montgomery limbs, crates/jolt-field/src/bn254/mod.rs:33-43) and")
    out.append("//! `jolt_g1_t` to `jolt_crypto::Bn254G1` (ark_bn254::G1Projective, crates/jolt-crypto/src/ec/bn254/mod.rs:17-24).")
    out.append("#![allow(non_camel_case_types, clippy::too_many_arguments, clippy::missing_safety_doc)]")
    out.append("use core::ffi::{c_char, c_void};")
    out.append("")
    out.append("pub const JOLT_HIP_ABI_VERSION: i32 = 1;")
    out.append("")
    out.append("#[repr(C)]\n#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]\npub struct jolt_fr_t {\n    pub l: [u64; 4],\n}")
    out.append("#[repr(C)]\n#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]\npub struct jolt_fq_t {\n    pub l: [u64; 4],\n}")
    out.append("#[repr(C)]\n#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]\npub struct jolt_g1_t {\n    pub x: jolt_fq_t,\n    pub y: jolt_fq_t,\n    pub z: jolt_fq_t,\n}")
    for o in opaque_handles():
        out.append(f"#[repr(C)]\npub struct {o} {{\n    _private: [u8; 0],\n}}")
    out.append("")
    out.append("/// Status codes (`enum` of the header); see `crate::status` for the mapping onto the reference's error types.")
    for k, name in enumerate(["JOLT_OK", "JOLT_ERR_INVALID_ARG", "JOLT_ERR_NO_DEVICE", "JOLT_ERR_OOM", "JOLT_ERR_HIP", "JOLT_ERR_SIZE_MISMATCH", "JOLT_ERR_UNSUPPORTED",
                              "JOLT_ERR_NOT_FULLY_BOUND", "JOLT_ERR_ROUND_CHECK", "JOLT_ERR_SRS_TOO_SMALL", "JOLT_ERR_EMPTY_POINT", "JOLT_ERR_NOT_INVERTIBLE"]):
        out.append(f"pub const {name}: i32 = {k};")
    out.append("pub const JOLT_ORDER_LOW_TO_HIGH: i32 = 0;\npub const JOLT_ORDER_HIGH_TO_LOW: i32 = 1;")
    out.append("pub const JOLT_MEMBER_FLAG_SKIP_ONE: u32 = 1;\npub const JOLT_MEMBER_FLAG_BORROW_TABLES: u32 = 2;")
    out.append("pub const JOLT_INT_U64: i32 = 0;\npub const JOLT_INT_I64: i32 = 1;\npub const JOLT_INT_I128: i32 = 2;\npub const JOLT_SCALAR_FR: i32 = 3;")
    out.append("pub const JOLT_MAX_MEMBER_TABLES: usize = 40;\npub const JOLT_MAX_MEMBER_TERMS: usize = 16;\npub const JOLT_MAX_MEMBER_FACTORS: usize = 64;\npub const JOLT_MAX_DEGREE: usize = 7;")
    out.append("")
    out.append("#[repr(C)]\npub struct jolt_member_desc {\n    pub n_tables: u32,\n    pub n_terms: u32,\n    pub degree: u32,\n    pub order: i32,\n"
               "    pub term_offsets: *const u32,\n    pub factors: *const u32,\n    pub coeffs: *const jolt_fr_t,\n}")
    out.append("#[repr(C)]\npub struct jolt_member_lc_desc {\n    pub n_tables: u32,\n    pub n_groups: u32,\n    pub n_factors: u32,\n    pub n_lc: u32,\n    pub degree: u32,\n"
               "    pub order: i32,\n    pub flags: u32,\n    pub group_factor_offsets: *const u32,\n    pub factor_lc_offsets: *const u32,\n    pub factor_consts: *const jolt_fr_t,\n"
               "    pub lc_tables: *const u32,\n    pub lc_coeffs: *const jolt_fr_t,\n}")
    out.append("pub type jolt_local_round_fn = Option<\n    unsafe extern \"C\" fn(user: *mut c_void, active: *const usize, n_active: usize, binds: *const *const jolt_fr_t, evals_out: *mut jolt_fr_t, evals_count: usize) -> i32,\n>;")
    out.append("pub type jolt_gather_fn = Option<unsafe extern \"C\" fn(user: *mut c_void, local: *const jolt_fr_t, count: usize, gathered: *mut jolt_fr_t) -> i32>;")
    out.append("pub type jolt_round_transcript_fn = Option<unsafe extern \"C\" fn(user: *mut c_void, compressed_coeffs: *const jolt_fr_t, n_coeffs: usize, challenge_out: *mut jolt_fr_t) -> i32>;")
    out.append("pub type jolt_open_transcript_fn = Option<\n    unsafe extern \"C\" fn(user: *mut c_void, phase: i32, points: *const jolt_g1_t, n_points: usize, values: *const jolt_fr_t, n_values: usize, challenge_out: *mut jolt_fr_t) -> i32,\n>;")
    out.append("")
    out.append('#[link(name = "jolt_hip")]\nextern "C" {')
    for name, ret, params in decls:
        ps = ", ".join(f"{('r#' + p) if p in RUST_KEYWORDS else p}: {t}" for p, t in params)
        out.append(f"    pub fn {name}({ps})" + (f" -> {ret};" if ret != "c_void" else ";"))
    out.append("}")
    return "\n".join(out) + "\n"


def main():
    text = render(parse_header())
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(FFI_RS) and open(FFI_RS).read() == text else 1)
    os.makedirs(os.path.dirname(FFI_RS), exist_ok=True)
    open(FFI_RS, "w").write(text)
    print(f"{FFI_RS}: {len(parse_header())} entry points")


if __name__ == "__main__":
    main()
