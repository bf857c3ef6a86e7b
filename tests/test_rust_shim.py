"""CPU: the Rust shim crate (rust/ark-ec-vrfs-hip, source only: no Rust toolchain in this image) against the C header.

The only validation possible without rustc: parse `extern "C" { ... }` of src/ffi.rs and the prototypes of
include/vrfhip.h with two INDEPENDENT small parsers (not the generator's) and assert, item by item, the same symbol
set, arity, and per-argument kind (pointer depth, constness, pointee / integer width), the same `#[repr(C)]` layout of
the suite descriptor, the same enum values and flag constants; that ffi.rs is what tools/gen_rust_ffi.py produces
from the current header; that every symbol is exported by libvrfhip.so; that the safe wrapper only calls functions
that exist; and that the crate keeps the reference's manifest surface (/root/reference Cargo.toml:11-17, restated
here because /root/reference does not travel)."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "ark-ec-vrfs-hip")
HEADER = open(os.path.join(ROOT, "include", "vrfhip.h")).read()
FFI = open(os.path.join(CRATE, "src", "ffi.rs")).read()
LIB_RS = open(os.path.join(CRATE, "src", "lib.rs")).read()

C_WIDTH = {"uint8_t": ("u", 8), "uint32_t": ("u", 32), "int32_t": ("i", 32), "uint64_t": ("u", 64), "size_t": ("u", "ptr"),
           "double": ("f", 64), "char": ("c", 8), "void": ("v", 0), "vrfhip_suite": ("i", 32), "vrfhip_curve": ("i", 32)}
R_WIDTH = {"u8": ("u", 8), "u32": ("u", 32), "i32": ("i", 32), "u64": ("u", 64), "usize": ("u", "ptr"), "f64": ("f", 64),
           "c_char": ("c", 8), "c_void": ("v", 0)}
STRUCTS = ("vrfhip_ctx", "vrfhip_keyset", "vrfhip_suite_desc")


def c_kind(decl):
    """('const uint8_t* x' | 'uint8_t seed[32]' | 'size_t') -> (base kind, [constness of each pointee, outermost last])"""
    decl = " ".join(decl.replace("*", " * ").split())
    decl = re.sub(r"\b\w+\s*\[\s*\d*\s*\]$", "* arr", decl)            # array parameter == pointer
    toks = decl.split(" ")
    base_toks, i = [], 0
    while i < len(toks) and toks[i] != "*":
        base_toks.append(toks[i]); i += 1
    names = [t for t in base_toks if t != "const"]
    base = names[0]
    kind = ("s", base) if base in STRUCTS else C_WIDTH[base]
    consts = ["const" in base_toks]
    while i < len(toks):
        if toks[i] == "*":
            j = i + 1
            q = False
            while j < len(toks) and toks[j] != "*":
                q = q or toks[j] == "const"
                j += 1
            consts.append(q)
            i = j
        else:
            i += 1
    depth = len(consts) - 1
    return kind, consts[:depth]             # constness of what each pointer level points to


def r_kind(t):
    t = t.strip()
    consts = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)", t)
        consts.append(m.group(1) == "const")
        t = m.group(2)
    kind = ("s", t) if t in STRUCTS else R_WIDTH[t]
    return kind, list(reversed(consts))     # innermost pointee first, like c_kind


def c_prototypes():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    text = "\n".join(ln for ln in text.split("\n") if not ln.strip().startswith("#"))
    text = re.sub(r"typedef\s+(enum|struct)\b.*?;\s*\n", "", text, flags=re.S) if False else text
    out = {}
    for m in re.finditer(r"\b(vrfhip_\w+)\s*\(([^()]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        head = text[:m.start()].rstrip()
        ret_m = re.search(r"([A-Za-z_][\w \*]*)$", head.split(";")[-1].split("}")[-1].split("{")[-1])
        ret = ret_m.group(1).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = ([c_kind(re.sub(r"\b\w+$", "", p).strip() if not p.rstrip().endswith("]") else p) for p in params],
                     None if ret == "void" else c_kind(ret))
    return out


def rust_prototypes():
    block = re.search(r'extern "C" \{(.*?)\n\}', FFI, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        params = [] if not args else [a.split(":", 1)[1] for a in args.split(", ")]
        out[name] = ([r_kind(p) for p in params], None if ret is None else r_kind(ret))
    return out


def test_extern_block_matches_header_item_by_item():
    c, r = c_prototypes(), rust_prototypes()
    assert set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    assert len(c) >= 50
    for name in sorted(c):
        cp, cr = c[name]
        rp, rr = r[name]
        assert len(cp) == len(rp), (name, "arity", len(cp), len(rp))
        for k, (a, b) in enumerate(zip(cp, rp)):
            assert a == b, (name, "argument %d" % k, a, b)
        assert cr == rr, (name, "return", cr, rr)


def test_descriptor_layout_enums_and_constants_match():
    body = re.search(r"typedef struct vrfhip_suite_desc\s*\{(.*?)\}", re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S), flags=re.S).group(1)
    c_fields = []
    for line in body.split(";"):
        line = " ".join(line.split())
        if line:
            m = re.match(r"(\w+) (\w+)(?:\[(\d+)\])?$", line)
            c_fields.append((m.group(2), C_WIDTH[m.group(1)], int(m.group(3) or 0)))
    rb = re.search(r"pub struct vrfhip_suite_desc \{(.*?)\}", FFI, flags=re.S).group(1)
    r_fields = []
    for m in re.finditer(r"pub (\w+): (?:\[(\w+); (\d+)\]|(\w+)),", rb):
        r_fields.append((m.group(1), R_WIDTH[m.group(2) or m.group(4)], int(m.group(3) or 0)))
    assert c_fields == r_fields and len(c_fields) == 10
    assert "#[repr(C)]" in FFI.split("pub struct vrfhip_suite_desc")[0][-60:]
    # same layout as the ctypes mirror, whose size is checked against the library in test_abi
    from ark_ec_vrfs_amd import _lib
    assert ctypes.sizeof(_lib.SuiteDescStruct) == sum((w if n == 0 else n * w) // 8 for _, (_, w), n in c_fields)
    # enums and #defines
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    for m in re.finditer(r"(VRFHIP_(?:SUITE|CURVE|ST|ERR|SUCCESS)\w*)\s*=\s*(-?\d+)", text):
        assert re.search(r"pub const %s: i32 = %s;" % (m.group(1), m.group(2)), FFI), m.group(1)
    for m in re.finditer(r"#define\s+(VRFHIP_(?:FLAG|POINT|SCALAR|HASH)\w*)\s+(\d+)u?", text):
        assert re.search(r"pub const %s: \w+ = %s;" % (m.group(1), m.group(2)), FFI), m.group(1)


def test_ffi_rs_is_generated_from_the_current_header():
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]).returncode
    assert rc == 0, "rust/ark-ec-vrfs-hip/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"


def test_every_rust_symbol_is_exported_by_the_library():
    from ark_ec_vrfs_amd import _lib
    lib = _lib.load()
    for name in rust_prototypes():
        assert hasattr(lib, name), name
    assert set(rust_prototypes()) == set(_lib.SYMBOLS)


def test_safe_wrapper_calls_only_existing_functions_and_keeps_the_reference_surface():
    r = rust_prototypes()
    called = set(re.findall(r"ffi::(vrfhip_\w+)\s*\(", LIB_RS))
    assert called and called <= set(r), called - set(r)
    for name in called:                                   # arity of every call site
        for m in re.finditer(r"ffi::%s\s*\(" % name, LIB_RS):
            depth, i, args, cur = 1, m.end(), [], ""
            while depth:
                ch = LIB_RS[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if depth == 1 and ch == ",":
                    args.append(cur); cur = ""
                elif depth:
                    cur += ch
                i += 1
            if cur.strip():
                args.append(cur)
            assert len(args) == len(r[name][0]), (name, len(args), len(r[name][0]))
    # the reference's re-export list (src/lib.rs:13-17), name for name
    surface = ["codec", "ietf", "pedersen", "reexports", "ring", "ring_suite_types", "suite_types", "suites", "utils",
               "AffinePoint", "BaseField", "CurveConfig", "Error", "HashOutput", "Input", "Output", "Public", "ScalarField",
               "Secret", "Suite"]
    use = re.search(r"pub use ark_vrf::\{(.*?)\};", LIB_RS, flags=re.S).group(1)
    assert sorted(x.strip() for x in use.split(",") if x.strip()) == sorted(surface)
    for item in ("struct GpuBatch", "fn ietf_prove", "fn ietf_verify", "fn pedersen_prove", "fn pedersen_verify", "impl<S: GpuSuite> Drop"):
        assert item in LIB_RS, item
    # manifest: the reference's dependency and feature switches (Cargo.toml:11-17)
    toml = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert re.search(r'ark-vrf\s*=\s*\{\s*version\s*=\s*"0\.1\.0",\s*default-features\s*=\s*false\s*\}', toml)
    assert 'default = ["std", "full"]' in toml and 'std = ["ark-vrf/std"]' in toml and 'full = ["ark-vrf/full"]' in toml
    assert 'links = "vrfhip"' in toml and os.path.exists(os.path.join(CRATE, "build.rs"))
