"""The Rust side of the boundary cannot be compiled in this image (no rustc / cargo): bindings/rust/vaporetto-hip is source a
maintainer builds.  What keeps it from rotting is this test: it parses the crate's `extern "C"` block and the C header
INDEPENDENTLY (neither through bindings/rust/gen_ffi.py) and compares, for every exported function, the arity, every argument's
pointer depth / constness / pointee width, and the return type; the #[repr(C)] struct field by field; and checks that the safe
layer only calls functions the block declares."""
import os
import re

from vaporetto_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust", "vaporetto-hip")

C_BASE = {"int": ("i", 32), "unsigned": ("u", 32), "size_t": ("u", "ptr"), "uint64_t": ("u", 64), "uint32_t": ("u", 32), "int32_t": ("i", 32),
          "uint8_t": ("u", 8), "float": ("f", 32), "char": ("char", 8), "void": ("void", 0), "vpt_status": ("i", 32),
          "vpt_predictor": ("opaque", "vpt_predictor"), "vpt_batch": ("opaque", "vpt_batch"), "vpt_model_info": ("struct", "vpt_model_info")}
RUST_BASE = {"c_int": ("i", 32), "c_uint": ("u", 32), "usize": ("u", "ptr"), "u64": ("u", 64), "u32": ("u", 32), "i32": ("i", 32), "u8": ("u", 8),
             "f32": ("f", 32), "c_char": ("char", 8), "c_void": ("void", 0),
             "vpt_predictor": ("opaque", "vpt_predictor"), "vpt_batch": ("opaque", "vpt_batch"), "vpt_model_info": ("struct", "vpt_model_info")}


def c_type(decl: str):
    """-> (base, [constness of each pointer level, outermost first])"""
    decl = decl.strip()
    levels = []
    # peel pointer levels from the right: `const vpt_predictor *const *` -> outer pointer to const pointer to const struct
    while True:
        m = re.match(r"^(.*)\*\s*(const)?\s*$", decl)
        if not m:
            break
        decl = m.group(1).strip()
        levels.append(m.group(2) == "const")      # constness of THIS pointer object (irrelevant for an argument), kept for depth
    toks = decl.split()
    base_const = "const" in toks
    base = C_BASE[" ".join(t for t in toks if t != "const")]
    # what a pointer level points AT is const when the thing to its left is const
    n = len(levels)
    pointee_const = []
    for k in range(n):            # k = 0: outermost pointer; it points at level k+1's pointer object, or at the base
        inner = levels[k + 1] if k + 1 < n else base_const
        pointee_const.append(inner)
    return base, pointee_const


def rust_type(t: str):
    t = t.strip()
    consts = []
    while t.startswith("*"):
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        consts.append(m.group(1) == "const")
        t = m.group(2).strip()
    return RUST_BASE[t], consts


def header_functions():
    text = open(os.path.join(ROOT, "include", "vaporetto_hip.h"), encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(vpt_status|void|const char \*)\s*(vpt_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text):
        params = " ".join(m.group(3).split())
        args = []
        if params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.match(r"^(.*?)\b\w+\s*\[\d+\]$", p)
                args.append(c_type(arr.group(1) + " *") if arr else c_type(re.match(r"^(.*?)\b\w+$", p).group(1)))
        ret = {"vpt_status": (("i", 32), []), "void": None, "const char *": (("char", 8), [True])}[m.group(1).strip()]
        out[m.group(2)] = (args, ret)
    return out, text


def rust_functions():
    text = open(os.path.join(CRATE, "src", "ffi.rs"), encoding="utf-8").read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (vpt_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = [rust_type(a.split(":", 1)[1]) for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (args, rust_type(m.group(3)) if m.group(3) else None)
    return out, text


def test_every_exported_function_has_the_same_shape_in_rust():
    c, _ = header_functions()
    r, _ = rust_functions()
    assert set(c) == set(r) == set(_lib.SIGNATURES), set(c) ^ set(r)
    for name in sorted(c):
        (cargs, cret), (rargs, rret) = c[name], r[name]
        assert len(cargs) == len(rargs), name
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            assert ca == ra, (name, i, ca, ra)
        assert cret == rret, (name, cret, rret)


def test_the_info_struct_matches_field_by_field():
    _, htext = header_functions()
    body = re.search(r"typedef struct vpt_model_info \{(.*?)\} vpt_model_info;", htext, flags=re.S).group(1)
    c_fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if decl:
            ty, names = decl.split(" ", 1)
            c_fields += [(n.strip(), C_BASE[ty]) for n in names.split(",")]
    _, rtext = rust_functions()
    rbody = re.search(r"pub struct vpt_model_info \{(.*?)\}", rtext, flags=re.S).group(1)
    r_fields = [(m.group(1), RUST_BASE[m.group(2)]) for m in re.finditer(r"pub (\w+): (\w+),", rbody)]
    assert c_fields == r_fields
    assert "#[repr(C)]" in rtext.split("pub struct vpt_model_info")[0].splitlines()[-3:][0] or "#[repr(C)]" in rtext
    # and the ctypes mirror has the same fields in the same order
    assert [n for n, _ in c_fields] == [n for n, _ in _lib.ModelInfo._fields_]


def test_status_codes_and_flags_match_the_header():
    _, htext = header_functions()
    raw_h = open(os.path.join(ROOT, "include", "vaporetto_hip.h"), encoding="utf-8").read()
    _, rtext = rust_functions()
    for name in ("VPT_OK", "VPT_INVALID_MODEL", "VPT_INVALID_ARGUMENT", "VPT_RUNTIME_ERROR"):
        hv = re.search(r"\b%s\s*=\s*(\d+)" % name, raw_h) or re.search(r"#define\s+%s\s+(\d+)" % name, raw_h)
        rv = re.search(r"pub const %s: c_int = (\d+);" % name, rtext)
        assert hv and rv and hv.group(1) == rv.group(1), name
    lib_rs = open(os.path.join(CRATE, "src", "lib.rs"), encoding="utf-8").read()
    assert re.search(r"\bVPT_FLAG_KYTEA_FULLWIDTH\s*=\s*1\b", raw_h) and "FLAG_KYTEA_FULLWIDTH: u32 = 1;" in lib_rs
    assert re.search(r"\bVPT_FLAG_SPLIT_LINEBREAKS\s*=\s*1u?\s*<<\s*7", raw_h) and "FLAG_SPLIT_LINEBREAKS: u32 = 1 << 7;" in lib_rs
    assert re.search(r"#define\s+VPT_FLAG_WSCONST\(\w+\)\s+\(1u\s*<<\s*\(\w+\)\)", raw_h) and "1 << char_type" in lib_rs


def test_the_safe_layer_only_calls_declared_functions_and_the_crate_is_whole():
    r, _ = rust_functions()
    lib_rs = open(os.path.join(CRATE, "src", "lib.rs"), encoding="utf-8").read()
    used = set(re.findall(r"ffi::(vpt_[a-z_0-9]+)\s*\(", lib_rs))
    assert used and used <= set(r), used - set(r)
    for f in ("Cargo.toml", "build.rs", os.path.join("src", "lib.rs"), os.path.join("src", "ffi.rs")):
        assert os.path.getsize(os.path.join(CRATE, f)) > 0
    assert 'links = "vaporetto_hip"' in open(os.path.join(CRATE, "Cargo.toml")).read()
    assert "rustc-link-lib=dylib=vaporetto_hip" in open(os.path.join(CRATE, "build.rs")).read()


def test_safe_wrappers_check_the_sizes_they_hand_to_c():
    """`predict_batch_sharded` is a safe `pub fn` over caller-sized slices -- it must validate them before the FFI call
    (an empty `boff` underflowed `len() - 1`; short `scores` / `labels` let the C side write past them).  No rustc here: the checks are
    pinned as source."""
    src = open(os.path.join(ROOT, "bindings", "rust", "vaporetto-hip", "src", "lib.rs")).read()
    body = src[src.index("pub fn predict_batch_sharded"):]
    body = body[:body.index("\n}\n")]
    call = body.index("ffi::vpt_predict_batch_sharded")
    for check in ("boff.is_empty()", "ooff.len() != boff.len()", "(utf8.len() as u64) < boff[n]", "(scores.len() as u64) < ooff[n]", "(labels.len() as u64) < ooff[n]",
                  "preds.is_empty()"):
        assert check in body and body.index(check) < call, check
    assert "boff.len() - 1" not in body[call:]
