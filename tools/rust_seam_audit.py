#!/usr/bin/env python3
"""Static audit of rust/jolt-kernels-hip against the reference's trait surface -- the crate cannot meet a compiler in this image (no cargo / rustc), so the
checks a type-checker would make at the SEAM are made here, mechanically:

  1. crates: every path root the sources name (`use x::..`, `x::item`) is a dependency in Cargo.toml, the crate itself, or std / core;
  2. trait impls: for every `impl Trait for Type` of a REFERENCE trait, the required items (methods and associated types without a default) are all there,
     nothing is defined that the trait does not declare, every method has the trait's number of parameters, the trait's reference supertraits are implemented
     for the same type, and associated types the trait bounds by derivable std traits (Clone, Debug, ...) derive them when they are in-tree types;
  4. backend slots: every `backend.<slot> = ..` of `mi355x()` is a field of the reference's `JoltBackend` (crates/jolt-kernels/src/backend.rs:126-171) and the assigned
     type implements that slot's trait FOR THAT SLOT'S RELATION (`PrepareKernel<Fr, R>` / `UniskipKernel<Fr, R>` / `ResolveLeaves<R>` behind `with_relation`); >= 25 slots;
  5. imports: every `use jolt_x::a::b::Item` resolves to a pub item (or pub re-export) of module a::b of the reference crate (round 6: two modules had imported relation
     types from `jolt_verifier::stages::relations`, which does not export them);
  3. generic uses: for every `Name::<A, B>::func(` over a reference type (and the uses INTEGRATION.md advertises, ADVERTISED below), the `where` clauses of the
     reference's `impl<..> Name<..>` / `fn func<..>` are parsed, the concrete arguments substituted, and every bound on an in-tree type is checked against the
     in-tree impls -- through the reference's blanket impls (`impl<P: StreamingCommitment> ModeStreamingCommitment for P`), under every cfg(feature) variant the
     reference declares, including associated-type equalities (`CommitmentScheme<Field = F>`).

The reference side is read from /root/reference when present and frozen into tests/golden/reference_trait_surface.json (`--write-fixture`), which is what the test
suite uses where the reference is absent.  Round 4's `JoltBackend::<Fr, HipHyperKzg>::optimized()` without `StreamingCommitment for HipHyperKzg` is finding class 3.

    python tools/rust_seam_audit.py                  # audit, findings on stdout, exit 1 if any
    python tools/rust_seam_audit.py --write-fixture  # re-extract the reference surface the crate touches
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "jolt-kernels-hip")
REFERENCE = "/root/reference"
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_trait_surface.json")
STD_ROOTS = {"std", "core", "alloc", "crate", "self", "super"}
PRIMITIVES = {"u8", "u16", "u32", "u64", "u128", "usize", "i8", "i16", "i32", "i64", "i128", "isize", "f32", "f64", "bool", "char", "str"}
DERIVABLE = {"Clone", "Copy", "Debug", "Default", "PartialEq", "Eq", "Hash", "PartialOrd", "Ord"}
AUTO = {"Send", "Sync", "Sized", "Unpin", "'static"}
# generic uses the documentation advertises but the crate does not spell: the transparent prover over the device PCS
ADVERTISED = [
    {"crate": "jolt_prover", "item": "prove", "args": {"F": "Fr", "PCS": "HipHyperKzg", "VC": "Pedersen<Bn254G1>"}, "projections": {"VC::Output": "Bn254G1"}, "file": "dory/prover.rs",
     "why": "jolt_prover::dory::prove::<Fr, HipHyperKzg, Pedersen<Bn254G1>, _, _> (crates/jolt-prover/src/dory/prover.rs:112-131)"},
]


# ---- lexing helpers ---------------------------------------------------------------------------------------------------------------------------------------
def strip(src):
    """comments, string and char literals blanked (same length, newlines kept): braces and commas that remain are syntax"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif src.startswith("/*", i):
            depth, j = 1, i + 2
            while j < n and depth:
                if src.startswith("/*", j):
                    depth, j = depth + 1, j + 2
                elif src.startswith("*/", j):
                    depth, j = depth - 1, j + 2
                else:
                    j += 1
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j]))
            i = j
        elif c == '"' or (c == "r" and re.match(r'r#*"', src[i:])):
            if c == "r":
                m = re.match(r'r(#*)"', src[i:])
                close = '"' + m.group(1)
                j = src.find(close, i + len(m.group(0)))
                j = n if j < 0 else j + len(close)
            else:
                j = i + 1
                while j < n and src[j] != '"':
                    j += 2 if src[j] == "\\" else 1
                j += 1
            out.append('"' + "".join(ch if ch == "\n" else " " for ch in src[i + 1:j - 1]) + '"')
            i = j
        elif c == "'" and re.match(r"'(?:\\[^']+|[^'\\])'", src[i:]):
            m = re.match(r"'(?:\\[^']+|[^'\\])'", src[i:])
            out.append(" " * len(m.group(0)))
            i += len(m.group(0))
        else:
            out.append(c)
            i += 1
    return "".join(out)


def match_close(src, i, open_ch="{", close_ch="}"):
    """index just past the bracket that closes the one at src[i]"""
    depth = 0
    while i < len(src):
        if src[i] == open_ch:
            depth += 1
        elif src[i] == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    return len(src)


def split_top(text, sep=","):
    """split on `sep` outside (), [], {}, <> -- `->` and `=>` are not brackets"""
    parts, depth, cur, i = [], 0, "", 0
    while i < len(text):
        ch = text[i]
        if text.startswith("->", i) or text.startswith("=>", i):
            cur += text[i:i + 2]
            i += 2
            continue
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def last_segment(path):
    """`a::b::Trait<X = Y>` -> (`Trait`, `X = Y`)"""
    path = path.strip()
    m = re.match(r"^(?:for\s*<[^>]*>\s*)?\??\s*((?:\w+::)*)(\w+)\s*(?:<(.*)>)?$", path, re.S)
    if not m:
        return path, ""
    return m.group(2), (m.group(3) or "")


def parse_bounds(text):
    """`A + b::C<X = Y> + 'static` -> [(name, generic args text)]"""
    return [last_segment(b) for b in split_top(text, "+") if b.strip()]


def parse_generics(text):
    """`<'a, F: JoltField, PCS>` (without the angle brackets) -> ordered [(name, [bounds])], lifetimes dropped"""
    out = []
    for g in split_top(text):
        g = g.strip()
        if not g or g.startswith("'"):
            continue
        g = re.sub(r"^const\s+", "", g)
        name, _, bounds = g.partition(":")
        out.append((name.strip(), parse_bounds(bounds.split("=")[0]) if bounds else []))
    return out


def parse_where(text):
    """`where A: X + Y, B::C: Z` -> {subject: [bounds]}"""
    out = {}
    for clause in split_top(text):
        if ":" not in clause:
            continue
        subject, bounds = re.split(r"(?<!:):(?!:)", clause, maxsplit=1)
        out.setdefault(subject.strip(), []).extend(parse_bounds(bounds))
    return out


def header_end(src, i):
    """from i (inside an item header) to its `{` or `;` at bracket depth 0 -> (index, char)"""
    depth = 0
    while i < len(src):
        ch = src[i]
        if src.startswith("->", i):
            i += 2
            continue
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        elif ch in "{;" and depth <= 0:
            return i, ch
        i += 1
    return len(src), ";"


def cfg_before(src, start):
    """the #[cfg(..)] attributes directly above the item that starts at `start` (attributes and whitespace only in between)"""
    head = src[:start]
    cfgs = []
    while True:
        m = re.search(r"#\[([^\]]*)\]\s*$", head)
        if not m:
            break
        if m.group(1).strip().startswith("cfg("):
            cfgs.append(re.sub(r"\s+", "", m.group(1)))
        head = head[:m.start()]
    return sorted(cfgs)


def items_of(body):
    """top-level items of a trait / impl body (text between its braces): [(kind, name, text, has_body)] for fn / type / const"""
    out, i, n = [], 0, len(body)
    while i < n:
        m = re.compile(r"\b(fn|type|const)\s+(\w+)").search(body, i)
        if not m:
            break
        # skip matches that sit inside a nested block of a previous item (cannot happen: we jump past bodies below)
        end, ch = header_end(body, m.end())
        text = body[m.start():end]
        if ch == "{":
            close = match_close(body, end)
            out.append((m.group(1), m.group(2), text, True))
            i = close
        else:
            out.append((m.group(1), m.group(2), text, "=" in text.split("where")[0] if m.group(1) != "fn" else False))
            i = end + 1
    return out


def fn_arity(text):
    m = re.search(r"\bfn\s+\w+\s*(<)?", text)
    i = m.end()
    if m.group(1):
        i = match_close(text, m.end() - 1, "<", ">")
    i = text.index("(", i)
    j = match_close(text, i, "(", ")")
    return len(split_top(text[i + 1:j - 1]))


# ---- the reference side ------------------------------------------------------------------------------------------------------------------------------------
def crate_dir(crate):
    return os.path.join(REFERENCE, "crates", crate.replace("_", "-"), "src")


def crate_sources(crate):
    d = crate_dir(crate)
    for base, _, files in os.walk(d):
        for f in sorted(files):
            if f.endswith(".rs"):
                yield os.path.relpath(os.path.join(base, f), REFERENCE), strip(open(os.path.join(base, f)).read())


def extract_trait(crate, name):
    """every definition of `trait name` in the crate (cfg variants): supertraits, required / provided items, associated-type bounds"""
    found = []
    for rel, src in crate_sources(crate):
        for m in re.finditer(r"\bpub(?:\([^)]*\))?\s+(?:unsafe\s+)?trait\s+%s\b" % re.escape(name), src):
            end, ch = header_end(src, m.end())
            header = src[m.end():end]
            generics = ""
            h = header.strip()
            if h.startswith("<"):
                g_end = match_close(h, 0, "<", ">")
                generics, h = h[1:g_end - 1], h[g_end:]
            supers, _, where = h.partition("where")
            supers = supers.strip()
            sup = parse_bounds(supers[1:]) if supers.startswith(":") else []
            wh = parse_where(where)
            sup += wh.get("Self", [])
            body = src[end + 1:match_close(src, end) - 1] if ch == "{" else ""
            methods, types = {}, {}
            for kind, item, text, has_default in items_of(body):
                if kind == "fn":
                    methods[item] = {"required": not has_default, "arity": fn_arity(text)}
                elif kind == "type":
                    bounds = text.split("=")[0].partition(":")[2]
                    types[item] = {"required": not has_default, "bounds": [b for b, _ in parse_bounds(bounds)] if bounds.strip() else []}
            found.append({"file": rel, "line": src.count("\n", 0, m.start()) + 1, "cfg": cfg_before(src, m.start()), "generics": [g for g, _ in parse_generics(generics)],
                          "supertraits": [{"name": s, "args": a} for s, a in sup], "methods": methods, "types": types})
    return found


def extract_blanket_impls(crate, name):
    """`impl<P: A + B> name for P {}` in the crate -> [{cfg, bounds}]"""
    out = []
    for rel, src in crate_sources(crate):
        for m in re.finditer(r"\bimpl\s*<([^{;]*?)>\s*(?:\w+::)*%s\b(?:<[^{;]*?>)?\s+for\s+(\w+)\s*(where[^{]*)?\{" % re.escape(name), src):
            gens = parse_generics(m.group(1))
            names = [g for g, _ in gens]
            if m.group(2) not in names:
                continue
            bounds = dict(gens).get(m.group(2), []) + parse_where(m.group(3)[5:] if m.group(3) else "").get(m.group(2), [])
            out.append({"file": rel, "cfg": cfg_before(src, m.start()), "bounds": [{"name": b, "args": a} for b, a in bounds]})
    return out


def extract_generic_item(crate, owner, func):
    """the reference's `impl<..> owner<..> { fn func }` blocks (owner a type) or free `fn func<..>` (owner None): generics, type arguments, where clauses, cfg"""
    out = []
    for rel, src in crate_sources(crate):
        if owner:
            for m in re.finditer(r"\bimpl\s*(<)", src):
                g_end = match_close(src, m.end() - 1, "<", ">")
                rest_end, ch = header_end(src, g_end)
                header = src[g_end:rest_end]
                hm = re.match(r"\s*(?:\w+::)*%s\s*<(.*?)>\s*(where.*)?$" % re.escape(owner), header, re.S)
                if not hm or ch != "{" or re.search(r"\bfor\b", header.split("where")[0]):
                    continue
                body = src[rest_end + 1:match_close(src, rest_end) - 1]
                for kind, item, text, _ in items_of(body):
                    if kind == "fn" and item == func:
                        fn_where = parse_where(text.partition("where")[2]) if "where" in text else {}
                        where = parse_where(hm.group(2)[5:]) if hm.group(2) else {}
                        for k, v in fn_where.items():
                            where.setdefault(k, []).extend(v)
                        gens = parse_generics(src[m.end():g_end - 1])
                        fpos = body.find(text)
                        out.append({"file": rel, "line": src.count("\n", 0, m.start()) + 1, "cfg": sorted(set(cfg_before(src, m.start()) + cfg_before(body, fpos))),
                                    "generics": [{"name": g, "bounds": [{"name": b, "args": a} for b, a in bs]} for g, bs in gens],
                                    "type_args": split_top(hm.group(1)), "where": {k: [{"name": b, "args": a} for b, a in v] for k, v in where.items()}})
        else:
            for m in re.finditer(r"\bpub\s+fn\s+%s\s*<" % re.escape(func), src):
                g_end = match_close(src, m.end() - 1, "<", ">")
                end, _ = header_end(src, g_end)
                text = src[m.start():end]
                where = parse_where(text.partition("where")[2]) if "where" in text else {}
                gens = parse_generics(src[m.end():g_end - 1])
                out.append({"file": rel, "line": src.count("\n", 0, m.start()) + 1, "cfg": cfg_before(src, m.start()),
                            "generics": [{"name": g, "bounds": [{"name": b, "args": a} for b, a in bs]} for g, bs in gens],
                            "type_args": [g for g, _ in gens], "where": {k: [{"name": b, "args": a} for b, a in v] for k, v in where.items()}})
    return out


# ---- the crate side ----------------------------------------------------------------------------------------------------------------------------------------
def flatten_use(tree, prefix=""):
    """`a::{b, c::d as e}` -> {local name: full path}"""
    tree = tree.strip()
    m = re.match(r"^((?:\w+::)*)\{(.*)\}$", tree, re.S)
    if m:
        out = {}
        for part in split_top(m.group(2)):
            out.update(flatten_use(part, prefix + m.group(1)))
        return out
    m = re.match(r"^((?:\w+::)*)(\w+|\*)(?:\s+as\s+(\w+))?$", tree)
    if not m:
        return {}
    name = m.group(3) or m.group(2)
    if name == "self":
        name = m.group(1).rstrip(":").split("::")[-1]
        return {name: (prefix + m.group(1)).rstrip(":")}
    return {name: prefix + m.group(1) + m.group(2)}


class Crate:
    def __init__(self, root=CRATE):
        self.root = root
        self.files = {}
        src_dir = os.path.join(root, "src")
        for f in sorted(os.listdir(src_dir)):
            if f.endswith(".rs"):
                self.files[f] = strip(open(os.path.join(src_dir, f)).read())
        cargo = open(os.path.join(root, "Cargo.toml")).read()
        deps = re.search(r"\[dependencies\](.*?)(?:\n\[|\Z)", cargo, re.S).group(1)
        self.deps = {m.group(1).replace("-", "_") for m in re.finditer(r"^([\w-]+)\s*=", deps, re.M)}
        self.modules = {f[:-3] for f in self.files if f not in ("lib.rs",)}
        self.uses = {f: self._uses(src) for f, src in self.files.items()}
        self.impls = [i for f, src in self.files.items() for i in self._impls(f, src)]
        self.derives = self._derives()

    @staticmethod
    def _uses(src):
        out = {}
        for m in re.finditer(r"\buse\s+([^;]+);", src):
            out.update(flatten_use(re.sub(r"\s+", " ", m.group(1))))
        return out

    def _impls(self, fname, src):
        for m in re.finditer(r"\b(unsafe\s+)?impl\b", src):
            i = m.end()
            generics = ""
            rest = src[i:].lstrip()
            i = len(src) - len(rest)
            if rest.startswith("<"):
                g_end = match_close(src, i, "<", ">")
                generics, i = src[i + 1:g_end - 1], g_end
            end, ch = header_end(src, i)
            if ch != "{":
                continue
            header = src[i:end]
            head, _, where = header.partition(" where ")
            fm = re.match(r"^\s*(.*?)\s+for\s+(.*?)\s*$", head, re.S)
            if not fm:
                continue  # inherent impl
            trait, targs = last_segment(fm.group(1))
            ty, ty_args = last_segment(fm.group(2))
            body = src[end + 1:match_close(src, end) - 1]
            methods, types = {}, {}
            for kind, item, text, _ in items_of(body):
                if kind == "fn":
                    methods[item] = fn_arity(text)
                elif kind == "type":
                    types[item] = re.sub(r"\s+", " ", text.partition("=")[2]).strip()
            yield {"file": fname, "line": src.count("\n", 0, m.start()) + 1, "trait": trait, "trait_path": fm.group(1).strip(), "trait_args": targs, "type": ty, "type_args": ty_args,
                   "generics": generics, "methods": methods, "types": types}

    def _derives(self):
        out = {}
        for src in self.files.values():
            for m in re.finditer(r"#\[derive\(([^)]*)\)\]\s*(?:#\[[^\]]*\]\s*)*(?:pub(?:\([^)]*\))?\s+)?(?:struct|enum)\s+(\w+)", src):
                out.setdefault(m.group(2), set()).update(last_segment(d)[0] for d in m.group(1).split(","))
        return out

    def defines(self, ty):
        return any(re.search(r"\b(?:struct|enum)\s+%s\b" % re.escape(ty), src) for src in self.files.values())

    def crate_of(self, fname, name, path=None):
        """the dependency crate a trait / type name used in `fname` comes from (through the file's `use`s or an explicit path)"""
        if path and "::" in path:
            root = path.split("::")[0].lstrip("<")
            if root in self.deps:
                return root
            if root in self.uses[fname] and "::" in self.uses[fname][root]:
                return self.uses[fname][root].split("::")[0]
        full = self.uses[fname].get(name)
        if full:
            root = full.split("::")[0]
            return root if root in self.deps else None
        return None


# ---- the audit ---------------------------------------------------------------------------------------------------------------------------------------------
class Audit:
    def __init__(self, surface=None, crate=None):
        self.crate = crate or Crate()
        self.live = surface is None
        self.surface = surface if surface is not None else {"traits": {}, "blanket": {}, "generic": {}}
        self.findings = []

    # reference lookups, memoised into self.surface (which is what --write-fixture saves)
    def trait(self, crate, name):
        key = f"{crate}::{name}"
        if key not in self.surface["traits"]:
            if not self.live:
                return None
            self.surface["traits"][key] = extract_trait(crate, name)
            self.surface["blanket"][key] = extract_blanket_impls(crate, name)
        return self.surface["traits"][key]

    def generic_item(self, crate, owner, func):
        key = f"{crate}::{owner or ''}::{func}"
        if key not in self.surface["generic"]:
            if not self.live:
                return None
            self.surface["generic"][key] = extract_generic_item(crate, owner, func)
        return self.surface["generic"][key]

    def find_trait_anywhere(self, name, hint_crates):
        for c in hint_crates:
            t = self.trait(c, name)
            if t:
                return c, t
        return None, None

    def note(self, text):
        if text not in self.findings:
            self.findings.append(text)

    # 1 ------------------------------------------------------------------------------------------------------------------------------------------------
    def check_crate_roots(self):
        c = self.crate
        for fname, src in c.files.items():
            local = set(c.uses[fname]) | c.modules | STD_ROOTS | PRIMITIVES
            for m in re.finditer(r"\buse\s+(\w+)::", src):
                if m.group(1) not in c.deps | STD_ROOTS | c.modules:
                    self.note(f"{fname}: `use {m.group(1)}::..` but `{m.group(1)}` is not a dependency in Cargo.toml")
            for m in re.finditer(r"(?<![\w:>$])([a-z][a-z0-9_]*)::(?=[A-Za-z_])", src):
                root = m.group(1)
                if root in local or root in c.deps or root in ("clippy", "rustfmt"):  # tool lints inside attributes
                    continue
                # a module of a dependency brought in by `use dep::module;` is in `local`; anything else lower-case and unknown is a missing crate
                self.note(f"{fname}:{src.count(chr(10), 0, m.start()) + 1}: path root `{root}::` is neither a dependency in Cargo.toml, an imported name nor std")

    # 2 ------------------------------------------------------------------------------------------------------------------------------------------------
    def implements(self, ty, trait, cfg_features=frozenset(), seen=()):
        """does the in-tree type `ty` implement the reference trait `trait` (by name) under the given enabled features?  -> (bool, how)"""
        c = self.crate
        if trait in AUTO:
            return True, "auto"
        for i in c.impls:
            if i["type"] == ty and i["trait"] == trait:
                return True, f"{i['file']}:{i['line']}"
        if trait in DERIVABLE and trait in c.derives.get(ty, ()):
            return True, "derive"
        if (ty, trait) in seen:
            return False, "cycle"
        for key, blankets in self.surface["blanket"].items():
            if key.split("::")[-1] != trait:
                continue
            for b in blankets:
                if not cfg_holds(b["cfg"], cfg_features):
                    continue
                ok = all(self.implements(ty, bb["name"], cfg_features, seen + ((ty, trait),))[0] for bb in b["bounds"])
                if ok:
                    return True, f"blanket impl {b['file']}"
        return False, "no impl"

    def check_trait_impls(self):
        c = self.crate
        for i in c.impls:
            crate = c.crate_of(i["file"], i["trait"], i["trait_path"])
            if crate is None or not crate.startswith("jolt_"):
                continue
            variants = self.trait(crate, i["trait"])
            where = f"{i['file']}:{i['line']} impl {i['trait']} for {i['type']}"
            if not variants:
                self.note(f"{where}: no `pub trait {i['trait']}` in the reference crate {crate}")
                continue
            for v in variants:
                for name, meth in v["methods"].items():
                    if meth["required"] and name not in i["methods"]:
                        self.note(f"{where}: required method `{name}` is missing ({v['file']}:{v['line']})")
                for name, arity in i["methods"].items():
                    if name not in v["methods"]:
                        self.note(f"{where}: method `{name}` is not an item of the trait ({v['file']}:{v['line']})")
                    elif v["methods"][name]["arity"] != arity:
                        self.note(f"{where}: method `{name}` takes {arity} parameter(s), the trait declares {v['methods'][name]['arity']} ({v['file']}:{v['line']})")
                for name, t in v["types"].items():
                    if t["required"] and name not in i["types"]:
                        self.note(f"{where}: associated type `{name}` is missing ({v['file']}:{v['line']})")
                    assigned = last_segment(i["types"].get(name, ""))[0]
                    if assigned and c.defines(assigned):
                        for b in t["bounds"]:
                            if b in DERIVABLE and not self.implements(assigned, b)[0]:
                                self.note(f"{where}: `type {name} = {assigned}` must be `{b}` ({v['file']}:{v['line']}) and {assigned} neither derives nor implements it")
                for name in i["types"]:
                    if name not in v["types"]:
                        self.note(f"{where}: associated type `{name}` is not an item of the trait ({v['file']}:{v['line']})")
                if c.defines(i["type"]):
                    for s in v["supertraits"]:
                        feats = features_of(v["cfg"])
                        if s["name"] in AUTO or s["name"] == "MaybeAllocative":
                            continue
                        sup_crate, _ = self.find_trait_anywhere(s["name"], [crate] + sorted(d for d in c.deps if d.startswith("jolt_")))
                        if sup_crate is None and s["name"] not in DERIVABLE:
                            continue  # a std trait we do not model (Sized, Fn..)
                        if not self.implements(i["type"], s["name"], feats)[0]:
                            self.note(f"{where}: supertrait `{s['name']}` ({v['file']}:{v['line']}) is not implemented for {i['type']}")

    # 3 ------------------------------------------------------------------------------------------------------------------------------------------------
    def assoc_type(self, ty, trait, name):
        for i in self.crate.impls:
            if i["type"] == ty and i["trait"] == trait and name in i["types"]:
                return i["types"][name]
        return None

    def check_bound(self, where, ty, bound, subst, feats, projections):
        name, args = bound["name"], bound["args"]
        if name in AUTO or name.startswith("'"):
            return
        c = self.crate
        if not c.defines(ty):
            return  # a reference / std type: the reference's own impls are the compiler's business there
        ok, how = self.implements(ty, name, feats)
        if not ok:
            self.note(f"{where}: `{ty}: {name}` is required{' with features ' + '+'.join(sorted(feats)) if feats else ''} and {ty} does not implement it ({how})")
            return
        for eq in split_top(args):
            m = re.match(r"^(\w+)\s*=\s*(.+)$", eq.strip(), re.S)
            if not m:
                continue
            want = re.sub(r"\s+", "", m.group(2))
            want = projections.get(want, subst.get(want, want))
            got = self.assoc_type(ty, name, m.group(1))
            if got is not None and re.sub(r"\s+", "", got).replace("Self::", "") != want and last_segment(got)[0] != last_segment(want)[0]:
                self.note(f"{where}: `{ty}: {name}<{m.group(1)} = {want}>` is required, the impl has `type {m.group(1)} = {got}`")

    def check_generic_use(self, where, crate, owner, func, concrete, projections=None, only_file=None):
        variants = [v for v in (self.generic_item(crate, owner, func) or []) if only_file is None or v["file"].endswith(only_file)]
        if not variants:
            if owner and self.is_alias_or_trait_fn(crate, owner, func):
                return  # `Owner` is a type alias / `func` a trait's associated function there: no inherent-impl bounds to check
            self.note(f"{where}: no generic `{(owner + '::') if owner else ''}{func}` found in the reference crate {crate}")
            return
        for v in variants:
            feats = features_of(v["cfg"])
            if isinstance(concrete, dict):
                subst = dict(concrete)
            else:
                subst = {p.strip(): a for p, a in zip(v["type_args"], concrete)}
            for g in v["generics"]:
                ty = last_segment(subst.get(g["name"], ""))[0]
                for b in g["bounds"]:
                    if ty:
                        self.check_bound(f"{where} -> {v['file']}:{v['line']}", ty, b, subst, feats, projections or {})
            for subject, bounds in v["where"].items():
                ty = last_segment(subst.get(subject, ""))[0]
                for b in bounds:
                    if ty:
                        self.check_bound(f"{where} -> {v['file']}:{v['line']}", ty, b, subst, feats, projections or {})

    def is_alias_or_trait_fn(self, crate, owner, func):
        key = f"{crate}::alias::{owner}::{func}"
        if key not in self.surface["generic"]:
            if not self.live:
                return False
            found = False
            for _, src in crate_sources(crate):
                if re.search(r"\bpub\s+type\s+%s\b" % re.escape(owner), src):  # an alias: `func` is an associated function of a trait its target implements
                    found = True
                    break
            self.surface["generic"][key] = found
        return self.surface["generic"][key]

    def check_generic_uses(self):
        c = self.crate
        for fname, src in c.files.items():
            for m in re.finditer(r"\b((?:\w+::)*)(\w+)::<([^;(){}]*?)>::(\w+)\s*\(", src):
                owner, func = m.group(2), m.group(4)
                crate = c.crate_of(fname, owner, m.group(1) + owner)
                if crate is None or not crate.startswith("jolt_"):
                    continue
                line = src.count("\n", 0, m.start()) + 1
                self.check_generic_use(f"{fname}:{line} {owner}::<{m.group(3)}>::{func}()", crate, owner, func, split_top(m.group(3)))
        for adv in ADVERTISED:
            self.check_generic_use(adv["why"], adv["crate"], None, adv["item"], adv["args"], adv.get("projections"), adv.get("file"))

    # ---- 4. the backend's slots: every `backend.<slot> = ..` of mi355x() against the registry (crates/jolt-kernels/src/backend.rs:126-171) ---------------------------------
    def registry(self):
        """slot name -> (trait, relation or None), from the reference's `pub struct JoltBackend` (live) or the fixture"""
        if not self.live:
            return self.surface.get("slots", {})
        src = strip(open(os.path.join(crate_dir("jolt_kernels"), "backend.rs")).read())
        m = re.search(r"pub struct JoltBackend<[^>]*>\s*(?:where[^{]*)?\{", src)
        body = src[m.end():match_close(src, m.end() - 1) - 1]
        slots = {}
        for f in split_top(body):
            fm = re.match(r"\s*pub\s+(\w+)\s*:\s*Box<\s*dyn\s+(\w+)<(.*)>\s*>\s*$", re.sub(r"\s+", " ", f), re.S)
            if not fm:
                continue
            args = split_top(fm.group(3))
            relation = None
            if fm.group(2) in ("PrepareKernel", "UniskipKernel") and len(args) >= 2:
                relation = re.sub(r"<.*", "", args[1].strip())
            slots[fm.group(1)] = [fm.group(2), relation]
        self.surface["slots"] = slots
        # the slots `optimized()` overwrites (optimized/mod.rs:136-196): what a backend "as complete as the optimized tier" serves
        opt = strip(open(os.path.join(crate_dir("jolt_kernels"), "optimized", "mod.rs")).read())
        self.surface["optimized_slots"] = sorted(set(re.findall(r"backend\.(\w+)\s*=", opt)))
        return slots

    def check_backend_slots(self):
        slots = self.registry()
        src = self.crate.files.get("backend.rs", "")
        m = re.search(r"pub fn mi355x\b[^{]*\{", src)
        if not m or not slots:
            self.findings.append("backend.rs: pub fn mi355x not found (or no registry to check it against)")
            return
        body = src[m.end():match_close(src, m.end() - 1) - 1]
        impls = {}
        for i in self.crate.impls:
            impls.setdefault(i["type"], []).append(i)
        served = {}
        for am in re.finditer(r"backend\.(\w+)\s*=\s*([^;]+);", body):
            slot, expr = am.group(1), re.sub(r"\s+", " ", am.group(2))
            if slot not in slots:
                self.findings.append(f"mi355x(): `backend.{slot}` is not a slot of JoltBackend")
                continue
            trait, relation = slots[slot]
            wm = re.search(r"with_relation\(\s*\w+\s*,\s*\w+\s*,\s*(?:\w+::)*(\w+)\s*\)", expr)
            if wm:  # the generic device member over a leaf resolver: the resolver must be `ResolveLeaves<that slot's relation>`
                leaves = wm.group(1)
                rels = [re.sub(r"<.*", "", a.strip()) for i in impls.get(leaves, []) if i["trait"] == "ResolveLeaves" for a in split_top(i["trait_args"] or "")[:1]]
                if trait != "PrepareKernel":
                    self.findings.append(f"mi355x(): slot `{slot}` is a {trait}, not a PrepareKernel: with_relation does not fit")
                elif relation not in rels:
                    self.findings.append(f"mi355x(): slot `{slot}` proves {relation}, but {leaves} resolves the leaves of {rels or 'nothing'}")
                served[slot] = f"HipPrepare<{relation}, {leaves}>"
                continue
            tm = re.search(r"Box::new\(\s*((?:\w+::)*\w+)", expr)
            if tm:  # the type is the last CamelCase segment of the path (`stage::HipX::new(ctx)`, `HipUniskip::<R>::new(..)`, `stage::HipY { .. }`)
                camel = [seg for seg in tm.group(1).split("::") if seg[:1].isupper()]
                tm = re.match(r"(\w+)", camel[-1]) if camel else None
            if not tm:
                self.findings.append(f"mi355x(): cannot read the type assigned to `backend.{slot}`: {expr[:80]}")
                continue
            ty = tm.group(1)
            fits = []
            for i in impls.get(ty, []):
                if i["trait"] != trait:
                    continue
                args = [re.sub(r"<.*", "", a.strip()) for a in split_top(i["trait_args"] or "")]
                if relation is None or relation in args or any(a in ("R", "$relation") for a in args):
                    fits.append(i)
            if not fits:
                self.findings.append(f"mi355x(): slot `{slot}` needs `impl {trait}<Fr{', ' + relation if relation else ''}..> for {ty}`; none in the crate")
            served[slot] = ty
        self.slots_served = served
        # placeholders that only live inside a mem::replace do not count
        real = {k: v for k, v in served.items()}
        if len(real) < 25:
            self.findings.append(f"mi355x() overwrites {len(real)} of {len(slots)} slots; the round-5 review asks for >= 25")

    # ---- 5. imports: every `use <reference crate>::path::Item` names a pub item of that crate at that path ---------------------------------------------------------------
    def check_imports(self):
        if not self.live:
            return
        cache = {}

        def module_sources(crate, mods):
            """sources of module `mods` of `crate`: [(path, text)] of <mods>.rs / <mods>/mod.rs (the crate root: lib.rs); None when the module does not exist"""
            key = (crate, tuple(mods))
            if key not in cache:
                base = crate_dir(crate)
                cands = [os.path.join(base, "lib.rs")] if not mods else [os.path.join(base, *mods) + ".rs", os.path.join(base, *mods, "mod.rs")]
                found = [(c, strip(open(c).read())) for c in cands if os.path.isfile(c)]
                if not found and mods:  # an inline `pub mod name { .. }` of the parent
                    parent = module_sources(crate, mods[:-1])
                    if parent:
                        for path, text in parent:
                            im = re.search(r"\bmod\s+%s\s*\{" % re.escape(mods[-1]), text)
                            if im:
                                found.append((path, text[im.end():match_close(text, im.end() - 1) - 1]))
                cache[key] = found or None
            return cache[key]

        def declares(text, name):
            if re.search(r"\bpub(?:\([^)]*\))?\s+(?:unsafe\s+)?(?:struct|enum|trait|fn|type|const|static|mod|union)\s+%s\b" % re.escape(name), text):
                return True
            if re.search(r"macro_rules!\s+%s\b" % re.escape(name), text):
                return True
            for um in re.finditer(r"\bpub(?:\([^)]*\))?\s+use\s+([^;]+);", text):
                flat = flatten_use(re.sub(r"\s+", " ", um.group(1)))
                if name in flat or "*" in flat:
                    return True
            return False

        for fname, uses in self.crate.uses.items():
            for local, full in uses.items():
                parts = full.split("::")
                crate = parts[0]
                if crate not in self.crate.deps or not crate.startswith("jolt_") or len(parts) < 2 or parts[-1] == "*":
                    continue
                mods, item = parts[1:-1], parts[-1]
                if "__private" in mods:
                    continue  # doc-hidden re-exports a derive macro uses
                srcs = module_sources(crate, mods)
                if srcs is None:
                    self.findings.append(f"{fname}: `use {full}`: {crate} has no module `{'::'.join(mods)}`")
                    continue
                if not any(declares(text, item) for _, text in srcs):
                    # a name that is itself a module file
                    if module_sources(crate, mods + [item]):
                        continue
                    self.findings.append(f"{fname}: `use {full}`: `{item}` is not a pub item of {crate}::{'::'.join(mods) or '(crate root)'}")

    def run(self):
        self.check_crate_roots()
        # trait lookups for blanket impls need the traits of the bounds too: prime the surface with every reference trait the crate names
        self.check_trait_impls()
        for extra in ("ModeStreamingCommitment",):
            self.trait("jolt_kernels", extra)
        self.check_generic_uses()
        self.check_backend_slots()
        self.check_imports()
        return self.findings


def features_of(cfgs):
    """cfg attribute list -> the set of features that must be ON for the item to exist (`not(feature = ..)` items exist with the feature off: empty set)"""
    out = set()
    for c in cfgs:
        if "not(" in c:
            continue
        out.update(re.findall(r'feature="(\w+)"', c))
    return frozenset(out)


def cfg_holds(cfgs, features):
    for c in cfgs:
        for f in re.findall(r'feature="(\w+)"', c):
            negated = bool(re.search(r'not\(feature="%s"\)' % f, c))
            if negated == (f in features):
                return False
    return True


def main():
    live = os.path.isdir(os.path.join(REFERENCE, "crates"))
    if "--write-fixture" in sys.argv:
        if not live:
            sys.exit("the reference checkout is not here")
        a = Audit()
        a.run()
        os.makedirs(os.path.dirname(FIXTURE), exist_ok=True)
        with open(FIXTURE, "w") as f:
            json.dump(a.surface, f, indent=1, sort_keys=True)
        print(f"{FIXTURE}: {len(a.surface['traits'])} traits, {len(a.surface['generic'])} generic items; {len(a.findings)} finding(s)")
        return
    a = Audit() if live else Audit(json.load(open(FIXTURE)))
    findings = a.run()
    for f in findings:
        print("FINDING", f)
    served = getattr(a, "slots_served", {})
    print(f"{len(findings)} finding(s); {len(a.crate.impls)} trait impls, mi355x() serves {len(served)} of {len(a.surface.get('slots', {}))} JoltBackend slots, reference surface {'live' if live else 'from the fixture'}")
    if "--slots" in sys.argv:
        for slot, (trait, relation) in sorted(a.surface.get("slots", {}).items()):
            print(f"  {slot:36s} {trait:26s} {relation or '':34s} {served.get(slot, '-- optimized() --')}")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
