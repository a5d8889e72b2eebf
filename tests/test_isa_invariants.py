"""Properties of the GENERATED code that this round's speed-ups rest on ("what the ISA said"): checked on the device
assembly hipcc produces for gfx950 - no GPU needed.  A control-flow join with loads in flight, or a select right behind a load, makes
the compiler drain the load counter (`s_waitcnt vmcnt(0)`) between the prefetch of chunk c + 2 and the products of chunk c of the
grouped weight-gradient kernel; a register-starved build would show up as scratch traffic."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "gd-mae_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result"]


def _isa(src, tmp_path):
    out = str(tmp_path / (src.replace(".hip", ".s")))
    subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", "-o", out, src], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    kernels, name = {}, None
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif line.startswith(".Lfunc_end"):
            name = None
        elif name is not None:
            kernels[name].append(line.rstrip("\n"))
    return kernels


def _loop_lines(lines):
    """instructions of the blocks the assembler annotates as part of a loop"""
    inloop, out = False, []
    for l in lines:
        if l.startswith(".LBB"):
            inloop = "Loop" in l
        elif inloop:
            out.append(l.strip())
    return out


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_grouped_weight_gradient_loop_keeps_its_loads_in_flight(tmp_path):
    k = _isa("dw_grouped.hip", tmp_path)
    plain = [n for n in k if "k_dw_groupedILi0E" in n]
    gathered = [n for n in k if "k_dw_groupedILi1E" in n]
    assert len(plain) == 1 and len(gathered) == 1
    for name in plain + gathered:
        body = _loop_lines(k[name])
        assert sum(l.startswith("v_mfma") for l in body) >= 32 and sum(l.startswith("global_load") for l in body) >= 16
        drains = [l for l in body if l.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", l)]
        assert not drains, (name, drains)
        assert not any("scratch_" in l for l in k[name]), name


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_bf16_rounding_is_the_hardware_instruction(tmp_path):
    """every bf16 epilogue of the fused layer kernels converts with v_cvt_pk_bf16_f32; the LayerNorm reductions use DPP row
    operations / permlane swaps, not ds_bpermute shuffles"""
    k = _isa("layer_fused.hip", tmp_path)
    fwd = sorted(n for n in k if "k_layer_fwdILi256E" in n)
    assert len(fwd) == 2          # <256, false> (product) and <256, true> (next layer's in-projection on board: GDMAE_QKV_RIDES=1)
    for name in fwd:
        text = "\n".join(k[name])
        assert text.count("v_cvt_pk_bf16_f32") >= 32
        assert "ds_bpermute" not in text and "ds_swizzle" not in text
        assert text.count("_dpp") + text.count("v_permlane") >= 48
