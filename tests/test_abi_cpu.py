"""CPU: the C-ABI library loads without a GPU, exports every function include/hetmogp_hip.h declares, the ctypes
binding covers exactly that set, and -- with no device -- every compute entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hetmogp_hip.h")
LIB = os.path.join(ROOT, "hetmogp_amd", "libhetmogp_hip.so")


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmogp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return LIB


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    names = declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "libhetmogp_hip.so does not export %s" % n
    lib.hmogp_abi_version.restype = ctypes.c_int
    header_version = int(re.search(r"#define HMOGP_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.hmogp_abi_version() == header_version == 8


def test_ctypes_binding_matches_header(built):
    from hetmogp_amd import _lib
    assert sorted(_lib.EXPORTS) == declared()
    # struct layouts: field order / count of the three ABI structs as declared in the header
    src = open(HEADER).read()
    for cname, cls in (("hmogp_config", _lib.Config), ("hmogp_params", _lib.Params), ("hmogp_outputs", _lib.Outputs)):
        end = src.index("} %s;" % cname)
        body = src[src.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = [re.findall(r"(\w+)\s*;", line)[0] for line in body.split("\n") if ";" in line]
        assert fields == [f[0] for f in cls._fields_], cname


def test_no_cpu_fallback_without_device(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the loud-failure path is exercised on CPU-only hosts")
    from hetmogp_amd import _lib, engine
    with pytest.raises(_lib.HetMOGPError) as ei:
        engine.Engine([("Gaussian", {})], Q=1, M=4, P=1)
    assert ei.value.code == _lib.E_NO_DEVICE
    with pytest.raises(_lib.HetMOGPError):
        engine.gemm(np.eye(2), np.eye(2))
    with pytest.raises(_lib.HetMOGPError):
        engine.var_exp("Bernoulli", np.zeros(3), np.zeros((3, 1)), np.ones((3, 1)))


def test_product_path_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under hetmogp_amd/ may import or execute it."""
    pkg = os.path.join(ROOT, "hetmogp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "svmogp_oracle" not in txt and "likelihoods_oracle" not in txt, f
