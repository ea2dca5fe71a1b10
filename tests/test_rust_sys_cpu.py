"""The Rust `-sys` crate (rust/cheetah-hip-sys, source only: no cargo / rustc in this pipeline) against the C ABI.

Parses include/cheetah_hip.h and rust/cheetah-hip-sys/src/lib.rs independently of tools/gen_rust_sys.py (which writes
lib.rs) and fails on a missing or extra symbol, a different argument count, a different scalar type or a different
pointer constness; also that every status / form / op constant has the header's value, that the built library exports
exactly the declared symbols, and that the safe wrappers (rust/batch.rs) only call symbols that are declared.
CPU only: no compute call."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cheetah_hip.h")
LIB_RS = os.path.join(ROOT, "rust", "cheetah-hip-sys", "src", "lib.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize", "int": "c_int",
           "double": "f64", "float": "f32", "void": "c_void", "char": "c_char"}


def c_prototypes():
    with open(HEADER) as f:
        text = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_0-9 ]+?\**)\s*(cheetah_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.groups()
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            toks = a.replace("*", " * ").split()
            toks = toks[:-1]                     # drop the parameter name
            stars = toks.count("*")
            const = "const" in toks
            base = [t for t in toks if t not in ("*", "const")][0]
            params.append((SCALARS.get(base, base), stars, const))
        rt = ret.replace("*", " * ").split()
        protos[name] = ((SCALARS.get([t for t in rt if t not in ("*", "const")][0]), rt.count("*"), "const" in rt), params)
    return protos, text


def rust_prototypes():
    with open(LIB_RS) as f:
        text = f.read()
    block = text[text.index('extern "C" {'):]
    protos = {}
    for m in re.finditer(r"pub fn (cheetah_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.groups()

        def ty(t):
            t = t.strip()
            stars = t.count("*mut") + t.count("*const")
            const = t.startswith("*const")
            base = t.replace("*mut", "").replace("*const", "").strip()
            return (base, stars, const)
        params = [ty(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        protos[name] = (ty(ret) if ret else ("c_void", 0, False), params)
    return protos, text


def test_every_header_symbol_is_declared_with_the_same_signature():
    c, _ = c_prototypes()
    r, _ = rust_prototypes()
    assert len(c) >= 80
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (cret, cparams) in c.items():
        rret, rparams = r[name]
        assert len(cparams) == len(rparams), name
        assert cret == rret, (name, cret, rret)
        for i, (cp, rp) in enumerate(zip(cparams, rparams)):
            assert cp == rp, (name, i, cp, rp)
    assert any(n.endswith("_dev") for n in r) and "cheetah_multiply_batch_dev_noalloc" in r


def test_constants_have_the_header_values():
    _, ctext = c_prototypes()
    _, rtext = rust_prototypes()
    n = 0
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}", ctext, flags=re.S):
        value = -1
        for it in [x.strip() for x in m.group(1).split(",") if x.strip()]:
            if "=" in it:
                k, v = [x.strip() for x in it.split("=")]
                value = int(v, 0)
            else:
                k, value = it, value + 1
            mm = re.search(r"pub const %s: \w+ = (-?\d+);" % k, rtext)
            assert mm and int(mm.group(1)) == value, k
            n += 1
    assert n >= 40
    assert re.search(r"pub const CHEETAH_NOALLOC_MAX_N: usize = %d;" % (1 << 20), rtext)
    assert re.search(r"pub const CHEETAH_MAX_MANY: usize = 64;", rtext)


def test_generated_file_is_current():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]).returncode == 0


def test_library_exports_exactly_the_declared_symbols():
    lib = os.path.join(ROOT, "cheetah_amd", "lib", "libcheetah_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T cheetah_" in ln}
    r, _ = rust_prototypes()
    assert exported == set(r), (sorted(exported - set(r)), sorted(set(r) - exported))


def test_safe_wrappers_call_only_declared_symbols():
    r, _ = rust_prototypes()
    with open(os.path.join(ROOT, "rust", "batch.rs")) as f:
        used = set(re.findall(r"sys::(cheetah_\w+)", f.read())) - {"cheetah_ctx", "cheetah_basepoint_table"}
    assert used and used <= set(r), sorted(used - set(r))
    for must in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "rust", "cheetah-hip-sys", must))
