"""CPU-side checks of the drop-in boundary: the shared library exports every
symbol include/vatl_hip.h declares, and the ctypes table mirrors the header.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vatl_hip.h")


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_0-9]+\s*\*?)\s*(vatl_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arglist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, arglist)
    return protos


@pytest.fixture(scope="module")
def built_lib():
    import vatl_hip
    if not os.path.exists(vatl_hip.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("vatl_build", os.path.join(ROOT, "vatl4pose-wacv2024_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return vatl_hip


def test_header_declares_the_path():
    p = _prototypes()
    for need in ("vatl_conv2d_fwd", "vatl_deconv4x4s2_fwd", "vatl_decode_argmax_affine", "vatl_thc_pairs", "vatl_localpeak_mean",
                 "vatl_hybrid_ae_wpu", "vatl_masked_mse_fwd_bwd", "vatl_adamw_step", "vatl_version", "vatl_last_error"):
        assert need in p


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib.LIB_PATH)
    for name in _prototypes():
        assert hasattr(lib, name), f"{name} declared in vatl_hip.h but not exported"
    assert lib.vatl_version() == 100


def test_library_exports_nothing_the_header_does_not_declare(built_lib):
    """`nm -D`: the dynamic symbol table's C-named `vatl_*` functions are EXACTLY the header's declarations — no undeclared tuning hook
    (round 5 shipped `vatl_tune_wgrad_blocks`, a summation-order knob, and `vatl_crop_tune_px` that way: they are internal C++ symbols
    with hidden visibility now, reachable only through vatl_tune_set of the profiling variant)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TW" and ln.split()[-1].startswith("vatl_")}
    assert exported == set(_prototypes()), exported ^ set(_prototypes())
    hooks = [ln for ln in out.splitlines() if "tune_wgrad_blocks" in ln or "crop_tune_px" in ln]
    assert not hooks, hooks


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.startswith("const char") and "(" not in decl and decl.endswith("*") else ctypes.c_void_p
    base = decl.split()[0] if decl.split()[0] != "const" else decl.split()[1]
    return {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "int64_t": ctypes.c_int64}[base]


def test_ctypes_table_mirrors_header(built_lib):
    protos = _prototypes()
    assert set(protos) == set(built_lib.SIGNATURES), set(protos) ^ set(built_lib.SIGNATURES)
    for name, (ret, args) in protos.items():
        res, argtypes = built_lib.SIGNATURES[name]
        assert len(args) == len(argtypes), name
        for decl, ct in zip(args, argtypes):
            want = _ctype_of(decl)
            assert ct is want, (name, decl, ct, want)
        if ret.replace(" ", "") == "constchar*":
            assert res is ctypes.c_char_p
        else:
            assert res is _ctype_of(ret), name


def test_route_names_of_the_meter_match_the_header(built_lib):
    """vatl_flop_meter_routes: the header's VATL_ROUTE_NAMES, the Python tuple and the library's route count agree (host-only call)."""
    src = open(HEADER).read()
    m = re.search(r"#define VATL_ROUTE_NAMES((?:\s*\\?\s*\"[a-z0-9_,]+\")+)", src)
    names = tuple("".join(re.findall(r'"([a-z0-9_,]+)"', m.group(1))).split(","))
    assert names == built_lib.ROUTE_NAMES
    counts = (ctypes.c_int64 * 32)()
    assert built_lib.lib().vatl_flop_meter_routes(counts, 32) == len(names) and not any(counts)


def test_conv_cout_pad_is_host_only(built_lib):
    assert [built_lib.conv_cout_pad(c) for c in (17, 32, 40, 64, 96, 256, 2048)] == [32, 32, 64, 64, 128, 256, 2048]


def test_cpu_tensors_are_refused(built_lib):
    import torch
    with pytest.raises(built_lib.VatlError):
        built_lib.decode(torch.zeros(1, 17, 64, 48), torch.zeros(1, 4))


def test_product_library_has_no_wrong_result_ablations(monkeypatch):
    """The profiling ablations (schedule variants 10..13, "no epilogue" / "one k-tile") compute wrong results by construction:
    the shipped library does not contain them — even with VATL_ALLOW_ABLATION=1 the knobs are refused (they exist only in the
    -DVATL_ABLATION variant, build.py --ablation)."""
    import vatl_hip as vh
    monkeypatch.setenv("VATL_ALLOW_ABLATION", "1")
    lib = vh.lib()
    for knob, value in ((0, 10), (0, 11), (0, 12), (0, 13), (4, 1), (6, 1), (6, 2), (17, 1)):
        assert lib.vatl_tune_set(knob, value) != 0
        assert b"VATL_ABLATION" in lib.vatl_last_error()
    assert lib.vatl_tune_set(0, 4) == 0


# include/vatl_hip.h's knob table: (knob, default, another accepted value)
PRODUCT_KNOBS = ((0, 4, 2), (1, 0, 1), (5, 0, 128), (7, 1, 4), (8, 1, 0), (10, 2, 1), (18, 2048, 0), (21, 2, 1), (22, 8, 0), (24, 2, 0), (25, 1, 0))


def test_knob_surface_is_frozen():
    """vatl_tune_set of the shipped library accepts exactly the bit-identical route selectors the header tabulates; the knobs that change
    the summation order (3, 9, 12, 19, 23) and the performance experiments (2, 16) are not in it (host-only calls: no GPU needed)."""
    import vatl_hip as vh
    lib = vh.lib()
    src = open(HEADER).read()
    table = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r"^ \*\s+(\d+)\s+(\d+)\s+[0-9]", src, flags=re.M)}
    assert table == {k: d for k, d, _ in PRODUCT_KNOBS}, table
    try:
        for knob, default, other in PRODUCT_KNOBS:
            assert lib.vatl_tune_set(knob, other) == 0, knob
    finally:
        for knob, default, _ in PRODUCT_KNOBS:
            assert lib.vatl_tune_set(knob, default) == 0, knob
    for knob, value in ((2, 50), (3, 512), (4, 0), (6, 0), (9, 1), (12, 0), (16, 4), (19, 2048), (23, 1), (11, 0), (26, 0), (0, 1), (0, 3)):
        assert lib.vatl_tune_set(knob, value) != 0, (knob, value)
        assert b"not a product knob" in lib.vatl_last_error()


# the environment variables the package may read, each documented in INTEGRATION.md ("Process-global state and switches")
ALLOWED_ENV = {"VATL_HIP_LIB", "VATL_DIST_BACKEND", "VATL_SPAWN", "VATL_WORKER_PARENT", "VATL_WORKER_CLASS",
               "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PYTHONPATH", "HIPCC"}


def test_no_undocumented_environment_switch():
    """Every os.environ / getenv read under the package names a variable of ALLOWED_ENV, and INTEGRATION.md documents each of them:
    no route, kernel or precision choice hides behind an environment variable (round-4 verdict: 29 such reads)."""
    pkg = os.path.join(ROOT, "vatl4pose-wacv2024_amd")
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    literal = r"environ(?:\.get\(|\.setdefault\(|\[)\s*[\"']([A-Za-z0-9_]+)[\"']"
    seen = set()
    for base, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d not in ("build", "build_ablation", "__pycache__")]
        for f in files:
            path = os.path.join(base, f)
            if f.endswith(".py"):
                src = open(path).read()
                reads = re.findall(literal, src)
                # every other mention of the environment must be a whole-environment hand-over to a child process, never a computed name
                rest = re.sub(literal, "", src)
                rest = rest.replace("dict(os.environ,", "").replace("os.environ.update(", "")
                assert not re.search(r"\bos\.environ\b|\bgetenv\b|\bputenv\b", rest), f"{path}: environment access that is not a literal, documented name"
            elif f.endswith((".hip", ".h")):
                src = re.sub(r"#ifdef VATL_ABLATION.*?#(?:else|endif)", "", open(path).read(), flags=re.S)   # the profiling variant reads VATL_ALLOW_ABLATION
                reads = re.findall(r"getenv\(\s*\"([A-Za-z0-9_]+)\"", src)
                assert src.count("getenv") == len(reads), path
            else:
                continue
            for name in reads:
                assert name in ALLOWED_ENV, f"{path} reads ${name}: not a documented switch"
                seen.add(name)
    for name in sorted(seen):
        if name.startswith("VATL_"):
            assert f"`{name}`" in doc, f"{name} is read by the package but INTEGRATION.md does not document it"
    assert "VATL_HIP_LIB" in seen


def test_the_oracle_is_test_infrastructure_only():
    """Nothing that ships or measures the product reaches into oracle/: the package never mentions it, the tools do not import it (tools/make_golden.py, the
    script that made the committed fixtures, is the one exception), bench.py imports it inside `cpu_baseline` only and __graft_entry__.py inside `smoke` only."""
    import ast
    imp = re.compile(r"^\s*(?:from\s+oracle\b|import\s+oracle\b)", re.M)
    for base, dirs, files in os.walk(os.path.join(ROOT, "vatl4pose-wacv2024_amd")):
        dirs[:] = [d for d in dirs if d not in ("build", "build_ablation", "__pycache__")]
        for f in files:
            if f.endswith(".py"):
                assert not imp.search(open(os.path.join(base, f)).read()), f"{base}/{f} imports the oracle"
    for base, dirs, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith(".py") and f != "make_golden.py":
                assert not imp.search(open(os.path.join(base, f)).read()), f"tools/{f} imports the oracle"
    for script, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree = ast.parse(open(os.path.join(ROOT, script)).read())
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
            for node in (ast.walk(fn) if isinstance(fn, ast.FunctionDef) else fn.body):
                names = [a.name for a in node.names] if isinstance(node, ast.Import) else ([node.module or ""] if isinstance(node, ast.ImportFrom) else [])
                if any(n == "oracle" or n.startswith("oracle.") for n in names):
                    assert isinstance(fn, ast.FunctionDef) and fn.name == allowed, f"{script}: oracle imported outside {allowed}()"
