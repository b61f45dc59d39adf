"""The Go side (integration/**/*.go, cgo) has never met a compiler in this image (no Go toolchain).  What can be held without
one: every `C.dp_*` / `C.dph_*` call names a function one of the two published headers declares and passes as many arguments
as the declaration has parameters; every `C.dp_*` type or constant a Go file mentions exists in the headers; every Go method
INTEGRATION.md names for the pipeline boundary exists in gpuhost/chost.go.  Header drift then breaks a CPU test instead of a
maintainer's build."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(txt):
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"//[^\n]*", "", txt)


def _split_top(s):
    """splits on commas outside parentheses / brackets / braces"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _declared():
    """name -> parameter count, plus the set of type / enumerator names, from both headers"""
    funcs, names = {}, set()
    for h in ("downpore_hip.h", "downpore_host.h"):
        txt = _strip_c(open(os.path.join(ROOT, "include", h)).read())
        for m in re.finditer(r"\b(dph?_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
            params = m.group(2).strip()
            n = 0 if params in ("", "void") else len(_split_top(params))
            funcs[m.group(1)] = n
        names |= set(re.findall(r"\b(?:struct|enum)\s+(dp_[a-z_0-9]+)", txt))
        names |= set(re.findall(r"\}\s*(dp_[a-z_0-9]+)\s*;", txt))
        names |= set(re.findall(r"\btypedef\s+[^;{]*?\b(dp_[a-z_0-9]+)\s*;", txt))
        names |= set(re.findall(r"\b(DPH?_[A-Z_0-9]+)\b", txt))
    return funcs, names


def _go_files():
    fs = sorted(glob.glob(os.path.join(ROOT, "integration", "**", "*.go"), recursive=True))
    assert len(fs) >= 7
    return fs


def _calls(src):
    """(name, argument count, line) of every C.dp*_( ... ) call"""
    out = []
    for m in re.finditer(r"\bC\.(dph?_[a-z_0-9]+)\s*\(", src):
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += src[i] in "([{"
            depth -= src[i] in ")]}"
            i += 1
        args = src[m.end():i - 1].strip()
        out.append((m.group(1), 0 if not args else len(_split_top(args)), src.count("\n", 0, m.start()) + 1))
    return out


def test_every_cgo_call_matches_a_declaration():
    funcs, names = _declared()
    assert len(funcs) > 120
    seen = set()
    for f in _go_files():
        src = re.sub(r"//[^\n]*", "", open(f).read())
        body = src.split('import "C"', 1)[1] if 'import "C"' in src else src
        for name, n, line in _calls(body):
            where = "%s:%d" % (os.path.relpath(f, ROOT), line)
            if name in names and name not in funcs:
                continue  # a conversion to a C type: C.dp_seq_meta(x)
            assert name in funcs, "%s calls C.%s, which no header declares" % (where, name)
            assert n == funcs[name], "%s passes %d arguments to %s (declared with %d)" % (where, n, name, funcs[name])
            seen.add(name)
    assert len(seen) > 60, len(seen)


def test_every_c_type_and_constant_exists():
    funcs, names = _declared()
    for f in _go_files():
        src = re.sub(r"//[^\n]*", "", open(f).read())
        for m in re.finditer(r"\bC\.((?:struct_)?)(dph?_[a-z_0-9]+|DPH?_[A-Z_0-9]+)\b(?!\s*\()", src):
            nm = m.group(2)
            assert nm in names or nm in funcs, "%s mentions C.%s, which no header declares" % (os.path.relpath(f, ROOT), nm)


def test_integration_md_names_only_go_methods_that_exist():
    """INTEGRATION.md section 0 names the Go methods of the pipeline boundary in backticks; each must be defined in gpuhost."""
    chost = open(os.path.join(ROOT, "integration", "gpuhost", "chost.go")).read()
    defined = set(re.findall(r"^func (?:\([a-z]+ \*?[A-Za-z]+\) )?([A-Z][A-Za-z0-9]*)\(", chost, flags=re.M))
    for want in ("ReadsFromFile", "OpenOverlap", "Init", "Step", "RoundPAF", "CommUniqueID", "InitComms", "StepSharded", "InitComm",
                 "SetRanks", "Superstep", "RunRoundParallel", "Done", "KeepText", "RunMap"):
        assert want in defined, want
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    para = md[md.index("Multi-GPU with one process per GPU"):]
    para = para[:para.index("\n\n")]
    for nm in re.findall(r"`(?:gpuhost\.|Overlap\.)?([A-Z][A-Za-z]+)`", para):
        assert nm in defined, "INTEGRATION.md names %s, gpuhost/chost.go does not define it" % nm


def test_go_commands_use_only_methods_gpuhost_defines():
    chost = open(os.path.join(ROOT, "integration", "gpuhost", "chost.go")).read()
    defined = set(re.findall(r"^func (?:\([a-z]+ \*?[A-Za-z]+\) )?([A-Z][A-Za-z0-9]*)\(", chost, flags=re.M))
    defined |= set(re.findall(r"^type ([A-Z][A-Za-z0-9]*) ", chost, flags=re.M))
    for f in glob.glob(os.path.join(ROOT, "integration", "commands", "*.go")):
        src = re.sub(r"//[^\n]*", "", open(f).read())
        for nm in re.findall(r"\bgpuhost\.([A-Z][A-Za-z0-9]*)", src):
            assert nm in defined, "%s uses gpuhost.%s" % (os.path.basename(f), nm)
        for recv in ("ov", "reads"):
            for nm in re.findall(r"\b%s\.([A-Z][A-Za-z0-9]*)\(" % recv, src):
                assert nm in defined, "%s calls %s.%s" % (os.path.basename(f), recv, nm)
