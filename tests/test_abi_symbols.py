"""CPU-only: the C-ABI library is built for gfx950, loads, and exports exactly the symbols
include/sumcheck_hip.h declares; computing entry points fail loudly without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT, has_gpu, load_package


HEADER = open(os.path.join(ROOT, "include", "sumcheck_hip.h")).read()


def header_functions():
    text = open(os.path.join(ROOT, "include", "sumcheck_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)) - {"sc_allreduce_fn", "sc_allgather_fn", "sc_draw_fn"})


def test_library_exports_every_declared_symbol():
    pkg = load_package()
    pkg.build()
    lib = pkg.load()
    declared = header_functions()
    assert len(declared) >= 35
    assert sorted(pkg._lib.SIGNATURES) == declared, "ctypes stub and header disagree"
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (sc_[a-z0-9_]+)", out))
    assert exported == set(declared), exported ^ set(declared)


def test_code_object_is_gfx950():
    pkg = load_package()
    blob = open(pkg._lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"pass_kernel" in blob


def test_no_product_dependency_on_oracle():
    """the product tree must not import, link or mention the oracle"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "thaler-study_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "sc_oracle" not in text and "pyref" not in text and "import oracle" not in text, f


def test_host_field_helpers_need_no_gpu():
    pkg = load_package()
    lib = pkg.load()
    f = pkg._lib.ScField()
    assert lib.sc_field_from_modulus(pkg.GOLDILOCKS, ctypes.byref(f)) == 0
    assert f.r_mod_p == 0xFFFFFFFF and f.r2_mod_p == 0xFFFFFFFE00000001
    assert lib.sc_field_from_modulus(4, ctypes.byref(f)) == 1
    F = pkg.Field(389)
    assert lib.sc_field_to_mont(F.ref(), 5) == F.from_int(5)
    assert lib.sc_field_from_mont(F.ref(), F.from_int(77)) == 77
    e = (ctypes.c_uint64 * 3)(F.from_int(3), F.from_int(10), F.from_int(23))   # 3 + 4x + 3x^2
    c = (ctypes.c_uint64 * 3)()
    assert lib.sc_interpolate_quadratic(F.ref(), e, c) == 0
    assert [F.to_int(x) for x in c] == [3, 4, 3]


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    pkg = load_package()
    with pytest.raises(pkg.SumcheckHipError) as ei:
        pkg.Context(pkg.Field(pkg.GOLDILOCKS))
    assert ei.value.code == 2 and "no CPU path" in str(ei.value)


def test_rust_sys_crate_declares_the_same_abi():
    """rust/sumcheck-hip-sys is source only (no Rust toolchain here): at least its declarations must be exactly
    the header's entry points, and every FFI name the safe crate calls must be declared"""
    sys_rs = open(os.path.join(ROOT, "rust", "sumcheck-hip-sys", "src", "lib.rs")).read()
    declared = sorted(set(re.findall(r"pub fn (sc_[a-z0-9_]+)\s*\(", sys_rs)))
    assert declared == header_functions(), set(declared) ^ set(header_functions())
    safe = open(os.path.join(ROOT, "rust", "sumcheck-hip", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(sc_[a-z0-9_]+)\s*\(", safe))
    assert used and used <= set(declared), used - set(declared)
    # every method of the SumCheckPolynomial trait is implemented for the three device-backed polynomials
    for ty in ("GpuG", "GpuW", "GpuTriangleG"):
        m = re.search(r"impl<T: MontConfig<1>> SumCheckPolynomial<F64<T>> for %s<T> \{(.*?)\n\}\n" % ty, safe, flags=re.S)
        assert m, ty
        for meth in ("fn evaluate", "fn fix_variables", "fn to_univariate", "fn num_vars", "fn to_evaluations", "fn native_engine"):
            assert meth in m.group(1), (ty, meth)


def _header_struct_fields(name):
    """field names of `typedef struct <name> { ... } <name>;` in the header, in order (comments stripped)"""
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S)
    assert m, name
    out = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        out.append(first.split()[-1].split("[")[0])
        out += [r.strip().split("[")[0] for r in rest]
    return out


def test_structs_that_cross_the_abi_have_the_same_fields_everywhere():
    """sc_field, sc_launch_record, sc_plan_options, sc_plan_step: the header's fields, in order, are the ctypes binding's and the
    Rust -sys crate's (a struct that grows in one place only would shift every field behind it)"""
    pkg = load_package()
    L = pkg._lib
    sys_rs = open(os.path.join(ROOT, "rust", "sumcheck-hip-sys", "src", "lib.rs")).read()
    for cname, ctype in (("sc_plan_options", L.ScPlanOptions), ("sc_plan_step", L.ScPlanStep), ("sc_launch_record", L.ScLaunchRecord)):
        fields = _header_struct_fields(cname)
        assert [f[0] for f in ctype._fields_] == fields, (cname, fields)
        m = re.search(r"pub struct %s \{(.*?)\n\}" % cname, sys_rs, flags=re.S)
        assert m, cname
        assert re.findall(r"pub (\w+):", m.group(1)) == fields, (cname, fields)
    # every plan action / launch kind of the header has a name in the binding
    for prefix, table in (("SC_PLAN_", L.PLAN_ACTIONS), ("SC_KIND_", L.KIND_NAMES)):
        values = {int(v) for _, v in re.findall(r"#define (%s\w+) (\d+)" % prefix, HEADER)}
        assert values <= set(table) | {1}, (prefix, values - set(table))


@pytest.mark.skipif(not os.path.isdir("/root/reference/sum-check-protocol"), reason="the reference tree is only present in the build container")
def test_round_engine_patch_applies_to_the_reference(tmp_path):
    """rust/patches/sum-check-protocol-round-engine.patch is the ONLY edit the reference needs: it must apply
    cleanly to sum-check-protocol/src/lib.rs and leave every other line untouched"""
    import shutil
    dst = tmp_path / "sum-check-protocol" / "src"
    dst.mkdir(parents=True)
    shutil.copy("/root/reference/sum-check-protocol/src/lib.rs", dst / "lib.rs")
    patch = os.path.join(ROOT, "rust", "patches", "sum-check-protocol-round-engine.patch")
    out = subprocess.run(["patch", "-p1", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    new = (dst / "lib.rs").read_text().splitlines()
    old = open("/root/reference/sum-check-protocol/src/lib.rs").read().splitlines()
    import difflib
    removed = [l for l in difflib.unified_diff(old, new, lineterm="", n=0) if l.startswith("-") and not l.startswith("---")]
    # the only lines that go away: the struct's closing brace moves, and Prover::new's one-line c_1
    assert len(removed) <= 4, removed
    text = "\n".join(new)
    for needle in ("pub trait RoundEngine<F: Field>", "fn hypercube_sum(&self) -> F", "fn native_engine(&self) -> Option<Box<dyn RoundEngine<F>>>",
                   "engine: Option<Box<dyn RoundEngine<F>>>"):
        assert needle in text


def test_rccl_library_selection_and_the_test_double():
    """SC_RCCL_LIBRARY names the RCCL build the product dlopen()s - that file or nothing.  tests/rccl_double (the stand-in that
    lets the RCCL plane's N > 1 control flow run between processes on one GPU) exports the seven entry points the product
    resolves, and serves sc_comm_unique_id through the product without a GPU; a path that cannot be loaded is an error that
    says so, never a silent fall back to librccl."""
    import subprocess
    import sys
    double = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")
    assert os.path.exists(double), "build it: __graft_entry__.build()"
    syms = subprocess.run(["nm", "-D", "--defined-only", double], capture_output=True, text=True, check=True).stdout
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclAllReduce", "ncclAllGather", "ncclGetErrorString"):
        assert (" T " + name) in syms, name
    # nothing in the product names the double
    for dirpath, _, files in os.walk(os.path.join(ROOT, "thaler-study_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".hpp", ".h")):
                assert "rccl_double" not in open(os.path.join(dirpath, f), errors="ignore").read(), os.path.join(dirpath, f)
    code = ("import sys, ctypes; sys.path.insert(0, %r); import __graft_entry__ as g; pkg = g.load_package(); lib = pkg._lib.load();"
            "buf = (ctypes.c_uint8 * 128)(); rc = lib.sc_comm_unique_id(buf); print(rc, bytes(buf).split(b'\\0')[0][:16], lib.sc_last_error(None))" % ROOT)
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SC_RCCL_LIBRARY=double), timeout=300)
    assert ok.returncode == 0 and ok.stdout.startswith("0 b'/sc_rccl_double_"), (ok.stdout, ok.stderr[-500:])
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SC_RCCL_LIBRARY="/nonexistent/librccl.so"), timeout=300)
    assert bad.returncode == 0 and bad.stdout.startswith("3 ") and "SC_RCCL_LIBRARY=/nonexistent/librccl.so" in bad.stdout, (bad.stdout, bad.stderr[-500:])
