"""CPU: the C-ABI library loads without a GPU and exports every symbol include/slam_hip.h declares
(no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "slam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from slam_llm_amd import lib
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib.raw(), n), f"{n} declared in include/slam_hip.h but not exported by libslamhip.so"
    assert lib.raw().slam_abi_version() == lib.ABI_VERSION == 2
    assert lib.raw().slam_target_arch() == b"gfx950"


def test_binding_table_matches_header():
    from slam_llm_amd import lib
    declared = set(header_functions()) - {"slam_last_error", "slam_abi_version", "slam_target_arch"}
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    # argument counts agree with the header prototypes
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "slam_hip.h")).read(), flags=re.S)
    for name, argtypes in lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\);", src, flags=re.S)
        assert m, name
        nargs = 0 if m.group(1).strip() in ("", "void") else m.group(1).count(",") + 1
        assert nargs == len(argtypes), f"{name}: header has {nargs} args, binding has {len(argtypes)}"


def test_errors_are_reported_not_swallowed():
    """argument validation happens before any launch, so it is checkable without a GPU"""
    import ctypes
    import pytest
    from slam_llm_amd import lib
    with pytest.raises(lib.SlamHipError, match="null operand"):
        lib.call("slam_gemm_bf16_nt", None, 64, None, 64, None, 64, 8, 8, 64, None, None, 0, 0, 0, 1.0, 0, 0, None)
    one = ctypes.c_void_p(16)
    with pytest.raises(lib.SlamHipError, match="multiple of 64"):
        lib.call("slam_gemm_bf16_nt", one, 40, one, 40, one, 8, 8, 8, 40, None, None, 0, 0, 0, 1.0, 0, 0, None)
    with pytest.raises(lib.SlamHipError, match="head_dim"):
        lib.call("slam_attn_fwd", one, 64, one, 64, None, one, 64, one, 64, None, None, 1, 8, 8, 64, 64, 1, 1, 32, 0, 1.0, None, None, None, None, 0, 0, 0.0, 0, None)
    with pytest.raises(lib.SlamHipError, match="Vt must be given"):      # neither the row-major V nor its transposed copy
        lib.call("slam_attn_fwd", one, 64, one, 64, None, None, 0, one, 64, None, None, 1, 8, 8, 64, 64, 1, 1, 64, 0, 1.0, None, None, None, None, 0, 0, 0.0, 0, None)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under slam_llm_amd/ may import or execute it"""
    pkg = os.path.join(ROOT, "slam_llm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
