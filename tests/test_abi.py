"""not-gpu: the C-ABI library loads and exports every symbol include/ganrev.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ganrev.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    import ganrev._lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in ganrev.h but not exported: {missing}"
    # and the ctypes binding covers the same set
    assert sorted(L.EXPORTED_SYMBOLS) == syms, sorted(set(syms) ^ set(L.EXPORTED_SYMBOLS))


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a gfx950 device the product path must raise, never compute on the CPU."""
    import pytest
    import torch
    import ganrev._lib as L
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(L.GanrevError):
        L.Context(0)
    from ganrev import models, synth
    R = models.create_R((1, 8, 8), 4)
    with pytest.raises(L.GanrevError):
        R.forward(synth.uniform((2, 1, 8, 8), 1))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan-reverser_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".lua")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "ganrev_oracle" not in src and "from oracle" not in src and "import oracle" not in src, os.path.join(dp, f)


# ---------------------------------------------------------------------------------------------------------------------
# lua/hipnn.lua declares its share of the ABI by hand in an ffi.cdef block.  There is no Lua runtime here or on the GPU box, so
# a text-level comparison with include/ganrev.h is the only guard against the two drifting apart:
# every prototype of the cdef must exist in the header with the same return type and the same parameter TYPES in the same order
# (parameter names are free, and may be absent, on either side).
def _strip_comments(txt):
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


_C_TYPE_WORDS = {"const", "unsigned", "signed", "struct", "int", "float", "double", "char", "void", "long", "short",
                 "int32_t", "int64_t", "uint8_t", "uint64_t", "uint32_t", "size_t", "gr_ctx", "gr_net", "gr_layer_desc", "gr_hyper", "gr_exchange_fn"}


def _param_type(p):
    """'const float* in_host' -> 'const float*'; 'gr_net*' -> 'gr_net*'; 'int' -> 'int' (the identifier, if any, is dropped)"""
    p = p.strip()
    stars = p.count("*")
    words = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", p)
    if len(words) > 1 and words[-1] not in _C_TYPE_WORDS:
        words = words[:-1]                         # trailing parameter name
    assert all(w in _C_TYPE_WORDS for w in words), f"unknown type word in parameter {p!r}"
    return " ".join(words) + "*" * stars


def prototypes(txt):
    """{name: (return type, [parameter types])} of every `type gr_xxx(params);` in a piece of C."""
    txt = _strip_comments(txt)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(gr_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ret = " ".join(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", ret)) + "*" * ret.count("*")
        plist = [] if params.strip() in ("", "void") else [_param_type(p) for p in params.split(",")]
        assert name not in out, f"{name} declared twice"
        out[name] = (ret, plist)
    return out


def test_lua_ffi_cdef_matches_the_header():
    lua = open(os.path.join(ROOT, "gan-reverser_amd", "lua", "hipnn.lua")).read()
    m = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S)
    assert m, "no ffi.cdef block in hipnn.lua"
    cdef = m.group(1)
    hdr = open(os.path.join(ROOT, "include", "ganrev.h")).read()
    lp, hp = prototypes(cdef), prototypes(hdr)
    assert len(lp) >= 40 and len(hp) >= 60, (len(lp), len(hp))
    for name, sig in sorted(lp.items()):
        assert name in hp, f"hipnn.lua declares {name}, which include/ganrev.h does not"
        assert sig == hp[name], f"{name}: hipnn.lua {sig} != ganrev.h {hp[name]}"
    # every C function the Lua code calls through the library handle is declared in its cdef
    called = set(re.findall(r"\bC\.(gr_[a-z0-9_]+)", lua))
    assert called and called <= set(lp), f"called but not declared in the cdef: {sorted(called - set(lp))}"
    # and the structs the cdef restates have the header's layout
    norm = lambda s: re.sub(r"\s+", " ", _strip_comments(s)).strip()
    for struct in ("gr_layer_desc", "gr_hyper"):
        a = re.search(r"typedef struct \{([^}]*)\}\s*" + struct, norm(cdef)).group(1)
        b = re.search(r"typedef struct \{([^}]*)\}\s*" + struct, norm(hdr)).group(1)
        fields = lambda body: [(_param_type(d.split(",")[0]).replace("*", ""), len(d.split(","))) for d in body.split(";") if d.strip()]
        flat = lambda body: [t for t, n in fields(body) for _ in range(n)]
        assert flat(a) == flat(b), f"{struct}: {flat(a)} != {flat(b)}"


def test_every_runtime_knob_is_documented_and_the_shipping_library_reads_few_environment_variables():
    """the shipping library keeps <= 10 documented runtime knobs.  Every gr_set_tuning key of the non-ablation build must be
    described in include/ganrev.h's comment on gr_set_tuning, every environment variable gr_init reads must be named there and in README.md, and the kernels'
    sources may call getenv at most 10 times (everything else is GR_KNOB: a constant outside the ablation build)."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "gan-reverser_amd", "csrc")
    net = open(os.path.join(csrc, "net.hip")).read()
    body = net[net.index('extern "C" int gr_set_tuning('):]
    body = body[:body.index("\n}\n")]
    shipping = re.sub(r"#ifdef GR_ABLATE.*?#endif", "", body, flags=re.S)
    keys = re.findall(r'!strcmp\(key, "(\w+)"\)', shipping)
    assert 1 <= len(keys) <= 10, keys
    header = open(os.path.join(root, "include", "ganrev.h")).read()
    doc = header[header.index("/* Runtime knobs."):header.index("int gr_set_tuning(")]
    readme = open(os.path.join(root, "README.md")).read()
    for k in keys:
        assert f'"{k}"' in doc, f"gr_set_tuning key {k} is not documented in include/ganrev.h"
        assert f"`{k}`" in readme, f"gr_set_tuning key {k} is not listed in README.md"
    n_getenv, envs = 0, set()
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            src = open(os.path.join(csrc, f)).read()
            n_getenv += len(re.findall(r"\bgetenv\(", src))
            envs |= set(re.findall(r'getenv\("(GR_\w+)"\)', src))
    assert n_getenv <= 10, n_getenv
    for e in envs:
        assert e in doc and e in readme, f"environment variable {e} is read by the library but not documented"
