"""The C-ABI libraries load and export every symbol include/*.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import multithreading_string_matching_amd as K
from multithreading_string_matching_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_gpu_abi_exports_every_declared_symbol():
    K.build()
    names = declared("kmpgpu.h", "kmpgpu_")
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.GPU_SO)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.GPU_API) == names                 # the Python binding table is complete


def test_host_abi_exports_every_declared_symbol():
    K.build()
    names = declared("kmphost.h", "kmp_")
    names = [n for n in names if n not in ("kmp_alloc_fn", "kmp_free_fn")]
    lib = ctypes.CDLL(_lib.HOST_SO)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.HOST_API) == sorted(names)


def test_binaries_exist():
    K.build()
    for b in ("serial", "openmp_data", "openmp_task"):
        assert os.access(os.path.join(_lib.BINDIR, b), os.X_OK)
