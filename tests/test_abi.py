"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports every symbol
include/nvf_hip.h declares (no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

from nvfpcc_amd import _lib
from nvfpcc_amd.build import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="nvf_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvf_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build()
    assert os.path.isfile(path)
    h = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/nvf_hip.h but not exported"


def test_python_prototypes_cover_the_header():
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    assert _lib.lib().nvf_version() >= 100


def test_workspace_queries_need_no_gpu():
    h = _lib.lib()
    assert h.nvf_wgrad_workspace(16, 8, 8, 4, 32, 32, 32) >= 8 * 8 * 64 * 4
    assert h.nvf_channel_sum_workspace(8) > 0
    assert h.nvf_gdn_bwd_workspace(8) > 0
    assert h.nvf_reduce_workspace() > 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from nvfpcc_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.maxpool2(torch.zeros(1, 1, 4, 4, 4))


def test_codec_library_exports_its_header():
    from nvfpcc_amd.build import build_codec
    h = ctypes.CDLL(build_codec())
    names = declared_symbols("nvf_codec.h")
    assert names == ["nvf_ac_decode", "nvf_ac_encode", "nvf_codec_version"]
    for n in names:
        assert hasattr(h, n)
