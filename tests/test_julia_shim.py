"""Static checks of julia/*.jl (no Julia runtime exists in the build image): every `ccall` names a symbol the
header declares, passes as many argument types and arguments as the C prototype has parameters, with pointer / integer /
floating-point kinds that match; block keywords and brackets balance.  Not a substitute for running the shim — it catches
the drift between `include/tfhe_mi355x.h` and the binding that would otherwise only show up on a machine with Julia."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tfhe_mi355x.h")
JULIA = [os.path.join(ROOT, "julia", f) for f in ("TFHEMI355X.jl", "mint_fixtures.jl")]


def c_prototypes():
    """name -> list of parameter kinds ('ptr' | 'int' | 'float') parsed from the header."""
    src = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int32_t|int64_t|void|const\s+char\s*\*)\s*(tfhe_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",")]
        kinds = []
        for p in params:
            if p in ("void", ""):
                continue
            kinds.append("ptr" if "*" in p else "float" if re.search(r"\b(double|float)\b", p) else "int")
        protos[m.group(1)] = kinds
    return protos


def strip_julia(src):
    """Comments and string literals replaced by blanks (lengths kept)."""
    out, i, n = [], 0, len(src)
    while i < n:
        ch = src[i]
        if ch == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif ch == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        else:
            out.append(ch); i += 1
    return "".join(out)


def split_top(s):
    """Split at top-level commas."""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip()); cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def ccalls(src):
    """(symbol, [argument types], number of arguments) for every ccall((:sym, LIB), ...)."""
    res = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\)", src):
        i, depth = m.start() + len("ccall"), 0
        j = i
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        args = split_top(src[i + 1:j])
        types = split_top(args[2].strip()[1:-1]) if args[2].strip().startswith("(") else None
        res.append((m.group(1), types, len(args) - 3))
    return res


def julia_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
        return "ptr"
    if t in ("Float64", "Cdouble", "Float32"):
        return "float"
    return "int"


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    assert len(protos) >= 30 and "tfhe_gates_batch" in protos and protos["tfhe_keygen_cloud_key"].count("float") == 2
    seen = set()
    for path in JULIA:
        for sym, types, nargs in ccalls(strip_julia(open(path).read())):
            assert sym in protos, f"{os.path.basename(path)}: ccall of {sym}, which include/tfhe_mi355x.h does not declare"
            assert types is not None, f"{sym}: argument types are not a tuple literal"
            want = protos[sym]
            assert len(types) == len(want) == nargs, f"{sym}: {len(types)} types, {nargs} arguments, {len(want)} C parameters"
            got = [julia_kind(t) for t in types]
            assert got == want, f"{sym}: Julia argument kinds {got} vs C {want}"
            seen.add(sym)
    # the shim binds the whole single-key and multi-key path
    for need in ("tfhe_ctx_create", "tfhe_ctx_create_multi", "tfhe_load_bootstrap_key_c128", "tfhe_load_keyswitch_key", "tfhe_gates_batch",
                 "tfhe_mk_load_bootstrap_key_c128", "tfhe_mk_load_keyswitch_key", "tfhe_mk_gate_nand_batch", "tfhe_keygen_cloud_key",
                 "tfhe_ctx_destroy", "tfhe_last_error"):
        assert need in seen, need


def test_julia_blocks_and_brackets_balance():
    openers = r"\b(function|struct|if|for|while|do|begin|let|quote|module|try|macro)\b"
    for path in JULIA:
        src = strip_julia(open(path).read())
        for a, b in ("()", "[]", "{}"):
            assert src.count(a) == src.count(b), f"{os.path.basename(path)}: unbalanced {a}{b}"
        # the contents of [...] open no blocks: `end` in an index expression (a[end]) and the `for` / `if` of a comprehension
        # need no terminator (innermost brackets first, until nothing changes); one-line `f(x) = ...` definitions open nothing
        code = src
        while True:
            code2 = re.sub(r"\[[^\[\]\n]*\]", " ", code)
            if code2 == code:
                break
            code = code2
        n_open = len(re.findall(openers, code))
        n_end = len(re.findall(r"\bend\b", code))
        assert n_open == n_end, f"{os.path.basename(path)}: {n_open} block openers, {n_end} `end`"
