"""The Rust extern block (rust/prism-gpu-sys/src/lib.rs) against include/prism_gpu.h: same
functions with the same number of arguments, same #[repr(C)] structs with the same fields in
the same order, same constants.  (No Rust toolchain in the build image: the crate is never
compiled here, so this is what keeps it from drifting.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _split_args(s: str):
    """Top-level comma split (function pointer members carry their own parentheses)."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_functions():
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "prism_gpu.h")).read())
    fns = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(prism_gpu_[a-z_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        fns[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return fns


def c_structs():
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "prism_gpu.h")).read())
    out = {}
    for m in re.finditer(r"typedef struct (prism_gpu_[a-z_]+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in _split_top_level_semicolons(m.group(2)):
            fp = re.search(r"\(\s*\*\s*([a-z_0-9]+)\s*\)", decl)  # function pointer member
            if fp:
                fields.append(fp.group(1))
                continue
            names = re.sub(r"\[[^\]]*\]", "", decl).split(",")
            first = names[0].split()[-1].lstrip("*")
            fields.append(first)
            fields += [n.strip().lstrip("*") for n in names[1:]]
        out[m.group(1)] = fields
    return out


def _split_top_level_semicolons(body: str):
    out, depth, cur = [], 0, ""
    for ch in body:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == ";" and depth == 0:
            if cur.strip():
                out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    return out


def c_constants():
    text = _strip_c_comments(open(os.path.join(ROOT, "include", "prism_gpu.h")).read())
    consts = {}
    for m in re.finditer(r"#define\s+(PRISM_[A-Z0-9_]+)\s+(-?\d+)u?\b", text):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"\b(PRISM_[A-Z0-9_]+)\s*=\s*(-?\d+)", text):
        consts[m.group(1)] = int(m.group(2))
    return consts


def rust_source():
    text = open(os.path.join(ROOT, "rust", "prism-gpu-sys", "src", "lib.rs")).read()
    return re.sub(r"//.*", "", text)


def rust_functions():
    text = rust_source()
    block = text[text.index('extern "C" {'):]
    fns = {}
    for m in re.finditer(r"pub fn (prism_gpu_[a-z_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        fns[m.group(1)] = 0 if not args else len(_split_args(args))
    return fns


def rust_structs():
    text = rust_source()
    out = {}
    for m in re.finditer(r"pub struct (prism_gpu_[a-z_]+)\s*\{\n(.*?)\n\}", text, flags=re.S):
        fields = re.findall(r"pub ([a-z_0-9]+)\s*:", m.group(2))
        if fields:
            out[m.group(1)] = fields
    return out


def rust_constants():
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (PRISM_[A-Z0-9_]+)\s*:\s*[a-z0-9_]+\s*=\s*(-?\d+)\s*;", rust_source())}


def test_extern_block_matches_the_header():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 35
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, n in c.items():
        assert r[name] == n, f"{name}: {n} arguments in the header, {r[name]} in the extern block"


def test_repr_c_structs_match_the_header():
    c, r = c_structs(), rust_structs()
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, fields in c.items():
        assert r[name] == fields, f"{name}: header {fields}, Rust {r[name]}"


def test_constants_match_the_header():
    c, r = c_constants(), rust_constants()
    missing = sorted(set(c) - set(r) - {"PRISM_GPU_H"})
    # column types are not needed by the binding (the schema travels as Arrow)
    missing = [m for m in missing if not m.startswith(("PRISM_COL_", "PRISM_SLOT_"))]
    assert not missing, missing
    for name, v in r.items():
        assert c.get(name) == v, f"{name}: Rust {v}, header {c.get(name)}"


def test_exec_crate_only_calls_declared_functions():
    text = open(os.path.join(ROOT, "rust", "prism-gpu-exec", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(prism_gpu_[a-z_]+)\s*\(", text))
    assert used and used <= set(rust_functions()), used - set(rust_functions())
