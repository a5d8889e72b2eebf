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


def test_attention_lse_query_follows_the_implementation_switch():
    """gdmae_window_attention_levels_writes_lse (host-only): the forward leaves log-sum-exp rows exactly when the cooperative bf16 path
    runs - bf16 rows, levels of 16 / 32 / 64 tokens, head dim 16 / 32, H % 4 == 0, implementation 0 - so that the Python autograd stub
    hands `out` / `lse` to the backward only then (the backward must not consult the switch again)."""
    lib = L.load()
    T = L.host_i32([16, 32, 64])
    assert lib.gdmae_window_attention_levels_writes_lse(1, 3, T, 256, 8) == 1
    assert lib.gdmae_window_attention_levels_writes_lse(0, 3, T, 256, 8) == 0          # fp32 rows: exact-fp32 kernels, no lse
    assert lib.gdmae_window_attention_levels_writes_lse(1, 3, L.host_i32([16, 48, 64]), 256, 8) == 0
    assert lib.gdmae_window_attention_levels_writes_lse(1, 3, T, 256, 2) == 0          # H % 4
    for impl, want in ((3, 0), (1, 0), (2, 0), (0, 1)):
        assert lib.gdmae_set_attention_impl(impl) == 0
        assert lib.gdmae_window_attention_levels_writes_lse(1, 3, T, 128, 8) == want, impl
