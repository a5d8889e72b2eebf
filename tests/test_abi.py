"""CPU: the C-ABI library loads and exports every symbol include/gdmae_hip.h declares (no compute calls)."""
import os
import re

from gdmae_hip import lib as L

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    txt = open(os.path.join(REPO, "include", "gdmae_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gdmae_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    names = _declared()
    assert len(names) >= 20
    assert sorted(L.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    lib = L.load()
    for n in _declared():
        assert hasattr(lib, n), n
    assert lib.gdmae_abi_version() == 1
    assert lib.gdmae_target_arch() == b"gfx950"
    assert lib.gdmae_voxelize_workspace_bytes(1000, 2, 10, 10, 1) > 0
