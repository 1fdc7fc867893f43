"""CPU test of the reference-side half of the boundary: integration/julia/*.jl (the `ccall` shims a maintainer adds to
AbstractBayesOpt.jl) against include/abo_hip.h.

No Julia toolchain exists in the build image, so the shims cannot be executed here; what CAN be checked mechanically is what
breaks silently when the C-ABI moves: every `@ccall` / `@abocall LIBABO.<name>(arg::Type, …)::Int32` is parsed and compared with
the header's prototype of <name> — existence, arity, and the C type of every argument (Int32 ↔ int32_t, Int64 ↔ int64_t,
UInt64 ↔ uint64_t, Float64 ↔ double, Csize_t ↔ size_t, Ptr{Float64} ↔ double*, Ptr{Cvoid} ↔ an opaque handle, …) — the mirrored
structs are compared field by field, and ABO_ABI with ABO_ABI_VERSION.  Deleting or retyping one argument of any call fails this."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "abo_hip.h")
SHIMS = [os.path.join(ROOT, "integration", "julia", f) for f in ("HipStandardGP.jl", "HipGradientGP.jl")]

OPAQUE = {"abo_gp", "abo_cand", "abo_mgpu", "abo_mcand", "void"}
SCALARS = {"Int32": "int32_t", "Int64": "int64_t", "UInt64": "uint64_t", "Float64": "double", "Csize_t": "size_t"}
POINTEES = {"Float64": "double", "Int64": "int64_t", "Int32": "int32_t", "UInt8": "char", "AboParams": "abo_params",
            "AboRefineOpts": "abo_refine_opts", "AboAcqTerm": "abo_acq_term", "AboTimings": "abo_timings",
            "AboQeiStats": "abo_qei_stats"}


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_prototypes():
    """name → list of (base type, pointer depth) per parameter"""
    text = _strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"\bint32_t\s+(abo_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                depth = a.count("*")
                toks = [t for t in re.sub(r"[*]", " ", a).split() if t not in ("const", "struct")]
                # the last token is the parameter name unless the declaration is unnamed
                base = toks[0] if len(toks) >= 1 else ""
                assert 1 <= len(toks) <= 2, (name, a)
                params.append((base, depth))
        protos[name] = params
    return protos


def c_struct(name):
    """[(field, c type)] of `typedef struct <name> { … } <name>;`"""
    text = _strip_c_comments(open(HEADER).read())
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S)
    assert m, f"struct {name} not found in the header"
    out = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        ty, rest = decl.split(" ", 1)
        for f in rest.split(","):
            f = f.strip()
            arr = re.match(r"(\w+)\[(\d+)\]$", f)
            out.append((arr.group(1), f"{ty}[{arr.group(2)}]") if arr else (f, ty))
    return out


def _strip_jl_comments(text):
    return "\n".join(line.split("#", 1)[0] if '"' not in line.split("#", 1)[0] else line for line in text.splitlines())


def _split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def julia_calls(path):
    """[(name, [julia type per argument], return type, line)] of every `LIBABO.<name>(…)::T` in the file"""
    text = _strip_jl_comments(open(path).read())
    calls = []
    for m in re.finditer(r"LIBABO\.(\w+)\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        args = text[m.end():i - 1]
        ret = re.match(r"\s*::\s*(\w+)", text[i:])
        assert ret, f"{path}: call of {m.group(1)} without a return type"
        types = []
        for a in _split_top(" ".join(args.split())):
            # the type is what follows the LAST top-level `::`
            depth, cut = 0, -1
            for j, ch in enumerate(a):
                depth += {"(": 1, "[": 1, "{": 1, ")": -1, "]": -1, "}": -1}.get(ch, 0)
                if depth == 0 and a[j:j + 2] == "::":
                    cut = j
            assert cut > 0, f"{path}: argument `{a}` of {m.group(1)} carries no ::Type"
            types.append(a[cut + 2:].strip())
        calls.append((m.group(1), types, ret.group(1), text.count("\n", 0, m.start()) + 1))
    return calls


def julia_struct(path, name):
    text = _strip_jl_comments(open(path).read())
    m = re.search(r"\bstruct\s+%s\b(.*?)\bend\b" % name, text, flags=re.S)
    assert m, f"struct {name} not found in {path}"
    return [tuple(x.strip() for x in f.split("::")) for f in re.split(r"[;\n]", m.group(1)) if "::" in f]


def jl_matches_c(jl, c):
    base, depth = c
    if jl in SCALARS:
        return depth == 0 and SCALARS[jl] == base
    m = re.match(r"Ptr\{(.+)\}$", jl)
    if not m:
        return False
    inner = m.group(1)
    if inner == "Cvoid":
        return depth == 1 and base in OPAQUE
    if inner == "Ptr{Cvoid}":
        return depth == 2 and base in OPAQUE
    return depth == 1 and POINTEES.get(inner) == base


def test_every_ccall_matches_its_prototype():
    protos = c_prototypes()
    assert len(protos) > 60, "the header parser lost the prototypes"
    seen, total = set(), 0
    for path in SHIMS:
        calls = julia_calls(path)
        assert calls, path
        # every mention of the library in code is a parsed call (nothing escapes the parser)
        code = _strip_jl_comments(open(path).read())
        assert len(re.findall(r"LIBABO\.\w+\(", code)) == len(calls)
        for name, types, ret, line in calls:
            where = f"{os.path.basename(path)}:{line} {name}"
            assert name in protos, f"{where}: not declared in include/abo_hip.h"
            assert ret == "Int32", f"{where}: returns {ret}, every entry point returns int32_t"
            want = protos[name]
            assert len(types) == len(want), f"{where}: {len(types)} arguments, the header declares {len(want)}"
            for k, (jl, c) in enumerate(zip(types, want)):
                assert jl_matches_c(jl, c), f"{where}: argument {k + 1} is {jl}, the header says {c[0]}{'*' * c[1]}"
            seen.add(name)
            total += 1
    assert total >= 40
    # the entry points the driver's call sites need are all bound
    for need in ("abo_create", "abo_fit", "abo_retain", "abo_destroy", "abo_predict", "abo_acq", "abo_nlml", "abo_nlml_grad",
                 "abo_optimize_acquisition", "abo_optimize_acquisition_terms", "abo_mgpu_optimize_acquisition_terms", "abo_acq_lhs",
                 "abo_acq_terms", "abo_mgpu_create", "abo_mgpu_fit", "abo_mgpu_acq", "abo_mgpu_acq_lhs", "abo_append", "abo_mgpu_append",
                 "abo_create_grad", "abo_predict_grad", "abo_predict_grad_cov", "abo_append_grad", "abo_abi_version", "abo_last_error"):
        assert need in seen, f"{need} is not called by the shims"


@pytest.mark.parametrize("jl_name,c_name", [("AboParams", "abo_params"), ("AboRefineOpts", "abo_refine_opts"),
                                            ("AboAcqTerm", "abo_acq_term"), ("AboQeiStats", "abo_qei_stats")])
def test_mirrored_structs_have_the_header_layout(jl_name, c_name):
    jl = julia_struct(SHIMS[0], jl_name)
    c = c_struct(c_name)
    assert [f for f, _ in jl] == [f for f, _ in c], (jl, c)
    for (f, jt), (_, ct) in zip(jl, c):
        assert SCALARS.get(jt) == ct, f"{jl_name}.{f}: {jt} against {ct}"


def test_abi_constant_and_version_guard():
    hdr = open(HEADER).read()
    abi = int(re.search(r"#define ABO_ABI_VERSION (\d+)", hdr).group(1))
    jl = open(SHIMS[0]).read()
    assert int(re.search(r"const ABO_ABI = Int32\((\d+)\)", jl).group(1)) == abi
    for path in SHIMS:
        code = _strip_jl_comments(open(path).read())
        # Julia 1.11 (reference Project.toml:37) has no `gc_safe`: it may appear only inside the version-guarded macro
        uses = [m.start() for m in re.finditer(r"gc_safe", code)]
        if path == SHIMS[0]:
            guard = re.search(r"@static if VERSION >= v\"1\.12[^\n]*\n\s*macro abocall\(ex\) esc\(:\(@ccall gc_safe=true \$ex\)\) end\s*\n"
                              r"else\s*\n\s*macro abocall\(ex\) esc\(:\(@ccall \$ex\)\) end\s*\nend", code)
            assert guard, "the @abocall macro must be defined under @static if VERSION >= v\"1.12…\""
            assert len(uses) == 1 and guard.start() < uses[0] < guard.end()
        else:
            assert not uses
    # no model is refitted to reach the grid stage (round 3's _group_of)
    assert "_group_of" not in open(SHIMS[0]).read()


def test_the_parser_notices_a_dropped_argument(tmp_path):
    """the guard guards: a copy of the shim with one argument deleted from one call is caught"""
    src = open(SHIMS[0]).read()
    broken = src.replace("LIBABO.abo_fit(hd.ptr::Ptr{Cvoid}, X::Ptr{Float64}, N::Int64,", "LIBABO.abo_fit(hd.ptr::Ptr{Cvoid}, X::Ptr{Float64},", 1)
    assert broken != src
    p = tmp_path / "broken.jl"
    p.write_text(broken)
    protos = c_prototypes()
    bad = [(n, len(t), len(protos[n])) for n, t, _, _ in julia_calls(str(p)) if len(t) != len(protos[n])]
    assert bad == [("abo_fit", 6, 7)]
    retyped = src.replace("LIBABO.abo_abi_version()::Int32", "LIBABO.abo_abi_version()::Int64", 1)
    p.write_text(retyped)
    assert any(r != "Int32" for _, _, r, _ in julia_calls(str(p)))


def test_private_helpers_called_by_the_shims_are_defined_in_them():
    """A call of a helper that does not exist (`_kind(acqf)` for `_acq_args(acqf)`) only shows at run time in Julia — and the
    shims never run here.  Every `_name(` the two files call must be defined in one of them (`function _name(` or `_name(…) =`)."""
    code = "\n".join(re.sub(r"#[^\n]*", "", open(p).read()) for p in SHIMS)
    called = set(re.findall(r"(?<![\w.!])(_[a-z][a-z0-9_]*!?)\(", code))
    defined = set(re.findall(r"^\s*function\s+(_[a-z][a-z0-9_]*!?)\s*[({]", code, flags=re.M))
    defined |= set(re.findall(r"^\s*(_[a-z][a-z0-9_]*!?)\([^\n]*\)\s*(?:where[^\n=]*)?=", code, flags=re.M))
    # helpers the shims take from the package they are included into (src/surrogates/*.jl)
    from_reference = {"_get_minimum", "_update_model_parameters"}
    missing = sorted(called - defined - from_reference)
    assert not missing, f"helpers called but not defined in integration/julia/*.jl: {missing}"
