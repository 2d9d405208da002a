"""The Rust binding against the C header, mechanically (CPU suite; no Rust toolchain is needed or present).

`bindings/rust/lightdock-hip/src/lib.rs` and the `src/hip.rs` snippet of INTEGRATION.md declare, in Rust, the part of
`include/lightdock_hip.h` that a lightdock-rust `impl Score` (src/scoring.rs:11-19; the factory that would select it is
src/bin/lightdock-rust.rs:276-316) links against.  Neither can be compiled in this image, so a drift between the header and
those declarations -- a reordered field of `ld_molecule`, a `usize` where the header says `uint32_t`, a missing argument --
would ship unnoticed.  This test parses all three texts and compares, per `extern "C"` function: the name exists in the
header, the argument count, every argument's and the return value's pointer depth / constness / pointee width; per
`#[repr(C)]` struct: field names, order and types.  It also proves that it can fail: the same comparison on deliberately
damaged copies of the Rust text must report the damage.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lightdock_hip.h")
LIB_RS = os.path.join(ROOT, "bindings", "rust", "lightdock-hip", "src", "lib.rs")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")

# struct ld_x of the header <-> #[repr(C)] struct of the binding
STRUCTS = {"ld_molecule": "LdMolecule", "ld_scorer_desc": "LdScorerDesc"}
OPAQUE = {"ld_scorer", "ld_gso", "ld_model"}          # handles: *mut c_void / *const c_void on the Rust side

C_SCALARS = {"size_t": "usize", "int": "c_int", "uint8_t": "u8", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64",
             "double": "f64", "float": "f32", "char": "c_char", "void": "c_void"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def canon_c_type(decl):
    """'const double *coordinates' / 'const double translation[3]' / 'ld_scorer *s' -> (canonical Rust-like type, name)."""
    decl = " ".join(decl.replace("*", " * ").split())
    array = False
    m = re.search(r"\[\s*\d*\s*\]\s*$", decl)
    if m:
        array = True
        decl = decl[:m.start()].strip()
    tokens = decl.split(" ")
    name = None
    if tokens and re.fullmatch(r"[A-Za-z_]\w*", tokens[-1]) and tokens[-1] not in C_SCALARS and tokens[-1] not in OPAQUE \
            and tokens[-1] != "const" and not tokens[-1].startswith("ld_"):
        name = tokens.pop()
    # tokens: [const] base [* [const]]...
    base_const = False
    i = 0
    if tokens[i] == "const":
        base_const = True
        i += 1
    if tokens[i] == "struct":
        i += 1
    base = tokens[i]
    i += 1
    if i < len(tokens) and tokens[i] == "const":      # 'char const'
        base_const = True
        i += 1
    levels = []                                       # constness of what each * points to, innermost first
    pointee_const = base_const
    while i < len(tokens):
        assert tokens[i] == "*", decl
        levels.append(pointee_const)
        i += 1
        pointee_const = False
        if i < len(tokens) and tokens[i] == "const":
            pointee_const = True
            i += 1
    if array:
        levels.append(base_const if not levels else False)
    if base in OPAQUE:
        rust = "c_void"
    elif base in STRUCTS:
        rust = STRUCTS[base]
    elif base in C_SCALARS:
        rust = C_SCALARS[base]
    elif base.startswith("ld_"):
        rust = base                                   # a struct / enum the binding does not mirror: stays under its C name
    else:
        raise AssertionError("unknown C type %r in %r" % (base, decl))
    for const in levels:
        rust = ("*const " if const else "*mut ") + rust
    if rust == "c_void" and not levels:
        rust = "()"
    return rust, name


def parse_header(text):
    text = strip_c_comments(text)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                t, name = canon_c_type(decl)
                fields.append((name, t))
        structs[m.group(1)] = fields
    funcs = {}
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ld_\w+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        ret_t, _ = canon_c_type(ret + " _ret") if not ret.endswith("*") else canon_c_type(ret)
        arg_list = []
        if args and args != "void":
            for a in args.split(","):
                arg_list.append(canon_c_type(a.strip()))
        funcs[name] = (ret_t, arg_list)
    return structs, funcs


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", " ", text)


def canon_rust_type(t):
    t = " ".join(t.replace("*", " *").split()).replace("* const", "*const").replace("* mut", "*mut")
    t = re.sub(r"\bstd::os::raw::", "", t)
    return t.strip()


def parse_rust(text):
    text = strip_rust_comments(text)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(","):
            decl = decl.strip()
            if not decl:
                continue
            name, t = decl.split(":", 1)
            fields.append((name.replace("pub", "").strip(), canon_rust_type(t)))
        structs[m.group(1)] = fields
    funcs = {}
    for block in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", block.group(1), flags=re.S):
            name, args, ret = m.group(1), m.group(2).strip(), (m.group(4) or "()").strip()
            arg_list = []
            if args:
                for a in args.split(","):
                    a = a.strip()
                    if a:
                        an, at = a.split(":", 1)
                        arg_list.append((canon_rust_type(at), an.strip()))
            funcs[name] = (canon_rust_type(ret), arg_list)
    return structs, funcs


def integration_snippet():
    text = open(INTEGRATION).read()
    blocks = re.findall(r"```rust\n(.*?)```", text, flags=re.S)
    assert blocks, "INTEGRATION.md holds no rust block"
    return "\n".join(blocks)


def compare(header_text, rust_text, where, need_funcs=()):
    """Every mismatch between the Rust declarations in `rust_text` and the header, as a list of strings."""
    hs, hf = parse_header(header_text)
    rs, rf = parse_rust(rust_text)
    problems = []
    for c_name, r_name in STRUCTS.items():
        if r_name not in rs:
            problems.append("%s: #[repr(C)] struct %s (== %s) is missing" % (where, r_name, c_name))
            continue
        want, got = hs[c_name], rs[r_name]
        if [n for n, _ in want] != [n for n, _ in got]:
            problems.append("%s: %s fields %s, header's %s has %s" % (where, r_name, [n for n, _ in got], c_name, [n for n, _ in want]))
            continue
        for (n, tw), (_, tg) in zip(want, got):
            if tw != tg:
                problems.append("%s: %s.%s is %s, the header says %s" % (where, r_name, n, tg, tw))
    if not rf:
        problems.append("%s: no extern \"C\" function found" % where)
    for name in need_funcs:
        if name not in rf:
            problems.append("%s: extern fn %s is not declared" % (where, name))
    for name, (ret, args) in rf.items():
        if name not in hf:
            problems.append("%s: extern fn %s is not in the header" % (where, name))
            continue
        hret, hargs = hf[name]
        if ret != hret:
            problems.append("%s: %s returns %s, the header says %s" % (where, name, ret, hret))
        if len(args) != len(hargs):
            problems.append("%s: %s takes %d arguments, the header says %d" % (where, name, len(args), len(hargs)))
            continue
        for k, ((tg, _), (tw, hn)) in enumerate(zip(args, hargs)):
            if tg != tw:
                problems.append("%s: %s argument %d (%s) is %s, the header says %s" % (where, name, k, hn, tg, tw))
    return problems


SCORE_PATH = ("ld_init", "ld_last_error", "ld_scorer_create", "ld_scorer_destroy", "ld_scorer_energy", "ld_scorer_energy_batch")


def test_header_parses_to_what_it_declares():
    """The parser itself, on facts one can read off the header: ld_molecule's 13 fields in order, Score::energy's mirror."""
    hs, hf = parse_header(open(HEADER).read())
    assert [n for n, _ in hs["ld_molecule"]] == ["n_atoms", "coordinates", "dfire_types", "ele_charges", "vdw_charges", "vdw_radii",
                                                "n_membrane", "membrane", "n_restraint_groups", "restraint_offsets",
                                                "restraint_atoms", "num_anm", "nmodes"]
    assert dict(hs["ld_molecule"])["dfire_types"] == "*const u32" and dict(hs["ld_molecule"])["n_atoms"] == "usize"
    assert [t for _, t in hs["ld_scorer_desc"]] == ["c_int", "c_int", "LdMolecule", "LdMolecule", "*const f64"]
    ret, args = hf["ld_scorer_energy"]                 # == Score::energy, src/scoring.rs:11-19
    assert ret == "c_int" and [t for t, _ in args] == ["*mut c_void", "*const f64", "*const f64", "*const f64", "*const f64", "*mut f64"]
    assert hf["ld_scorer_create"] == ("*mut c_void", [("*const LdScorerDesc", "desc")])
    assert hf["ld_gso_save_many"][1][3][0] == "*const *const c_char"
    assert hf["ld_last_error"] == ("*const c_char", [])
    assert hf["ld_scorer_pose_len"][1][0][0] == "*const c_void"


def test_binding_crate_matches_the_header():
    problems = compare(open(HEADER).read(), open(LIB_RS).read(), "bindings/rust/lightdock-hip/src/lib.rs",
                       SCORE_PATH + ("ld_gso_create", "ld_gso_destroy", "ld_gso_run", "ld_gso_save"))
    assert not problems, "\n".join(problems)


def test_integration_snippet_matches_the_header():
    problems = compare(open(HEADER).read(), integration_snippet(), "INTEGRATION.md", SCORE_PATH)
    assert not problems, "\n".join(problems)


@pytest.mark.parametrize("source", ["crate", "doc"])
def test_the_check_fails_on_a_damaged_binding(source):
    """Reorder two fields of LdMolecule; narrow an integer; drop an argument; rename a function: each must be reported."""
    header = open(HEADER).read()
    rust = open(LIB_RS).read() if source == "crate" else integration_snippet()
    assert not compare(header, rust, source)

    def damaged(pattern, repl, count=1):
        out, n = re.subn(pattern, repl, rust, count=count, flags=re.S)
        assert n == count, pattern
        return out

    # membrane and n_restraint_groups swapped
    swapped = damaged(r"((?:pub )?membrane: \*const u32,[^\n]*\n)(\s*(?:pub )?n_restraint_groups: usize,[^\n]*\n)", r"\2\1")
    assert any("fields" in p for p in compare(header, swapped, source))
    narrowed = damaged(r"n_atoms: usize", "n_atoms: u32")
    assert any("n_atoms is u32" in p for p in compare(header, narrowed, source))
    mutable = damaged(r"coordinates: \*const f64", "coordinates: *mut f64")
    assert any("coordinates is *mut f64" in p for p in compare(header, mutable, source))
    dropped = damaged(r"(fn ld_scorer_energy_batch\([^)]*?), stride: usize", r"\1")
    assert any("ld_scorer_energy_batch takes 4 arguments" in p for p in compare(header, dropped, source))
    renamed = damaged(r"fn ld_scorer_destroy\(", "fn ld_scorer_free(")
    got = compare(header, renamed, source, SCORE_PATH)
    assert any("ld_scorer_free is not in the header" in p for p in got) and any("ld_scorer_destroy is not declared" in p for p in got)
    wrong_ret = damaged(r"fn ld_scorer_create\(([^)]*)\) -> \*mut c_void", r"fn ld_scorer_create(\1) -> *const c_void")
    assert any("ld_scorer_create returns" in p for p in compare(header, wrong_ret, source))
