"""bindings/rust/hip.rs cannot be compiled in this image (no rustc): lint its `extern "C"` block
against include/oddio_hip.h instead -- every declared function must exist in the header with the same
number of arguments and compatible C types, so the shim cannot drift from the ABI silently."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_TO_C = {
    "c_int": {"int"},
    "u32": {"uint32_t"},
    "f32": {"float"},
    "f64": {"double"},
    "usize": {"size_t"},
    "*const f32": {"const float*", "const float[3]", "const float[4]"},
    "*mut f32": {"float*"},
    "*mut u32": {"uint32_t*"},
    "*mut c_int": {"int*"},
    "*const c_char": {"const char*"},
    "*const c_void": {"const void*"},
    "*mut c_void": {"void*"},
    "*mut RawFrames": {"oddio_hip_frames*"},
    "*mut *mut RawFrames": {"oddio_hip_frames**"},
    "*mut RawScene": {"oddio_hip_scene*"},
    "*mut *mut RawScene": {"oddio_hip_scene**"},
    "*mut RawMixer": {"oddio_hip_mixer*"},
    "*mut *mut RawMixer": {"oddio_hip_mixer**"},
    "*const RawFilter": {"const oddio_hip_filter*"},
    "*const *mut RawFrames": {"oddio_hip_frames* const*"},
    "*const f64": {"const double*"},
    "*const c_int": {"const int*"},
    "*const u32": {"const uint32_t*"},
}


def c_declarations():
    text = open(os.path.join(ROOT, "include", "oddio_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for ret, name, args in re.findall(r"\b(int|const char\*)\s+(oddio_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        params = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            m = re.match(r"(.*?)(\w+)(\[\d+\])?$", a)      # type, name, optional array suffix
            ty = re.sub(r"\s+", " ", m.group(1)).strip().replace(" *", "*") + (m.group(3) or "")
            params.append(ty)
        decls[name] = (ret, params)
    return decls


def rust_declarations():
    text = open(os.path.join(ROOT, "bindings", "rust", "hip.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    decls = {}
    for name, args, ret in re.findall(r"fn (oddio_hip_[a-z0-9_]+)\((.*?)\)\s*->\s*([^;]+);", block, flags=re.S):
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        decls[name] = (ret.strip(), params)
    return decls


def test_every_extern_fn_matches_the_header():
    c, rs = c_declarations(), rust_declarations()
    assert len(rs) >= 30
    for name, (ret, params) in rs.items():
        assert name in c, f"{name} is not declared in include/oddio_hip.h"
        c_ret, c_params = c[name]
        assert c_ret in RUST_TO_C[ret], (name, ret, c_ret)
        assert len(params) == len(c_params), f"{name}: {len(params)} Rust args vs {len(c_params)} in the header"
        for i, (r, cc) in enumerate(zip(params, c_params)):
            assert r in RUST_TO_C, f"{name} arg {i}: unmapped Rust type {r!r}"
            assert cc in RUST_TO_C[r], f"{name} arg {i}: Rust {r!r} vs C {cc!r}"


def test_every_extern_fn_is_used_by_the_shim():
    text = open(os.path.join(ROOT, "bindings", "rust", "hip.rs")).read()
    body = text.split('extern "C" {', 1)[1].split("\n}", 1)[1]
    for name in rust_declarations():
        assert re.search(rf"\b{name}\(", body), f"{name} is declared but never called"
