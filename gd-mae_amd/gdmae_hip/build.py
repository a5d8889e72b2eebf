"""Build libgdmae_hip.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box with the snapshot).

    python -m gdmae_hip.build            # from gd-mae_amd/

`-ffp-contract=off` keeps the bit-exact integer/voxel arithmetic free of FMA contraction (explicit fmaf
is still used in the floating-point kernels); fp32 division/sqrt stay correctly rounded (HIP default).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.abspath(os.path.join(HERE, "..", "csrc"))
LIB = os.path.join(CSRC, "libgdmae_hip.so")
SOURCES = ["capi.hip", "voxelize.hip", "mask.hip", "partition.hip", "segment.hip", "attention.hip", "attention_mfma.hip", "attention_t32.hip", "attention_t16.hip", "attention_coop.hip", "layernorm.hip", "decoder.hip", "chamfer.hip",
           "optim.hip", "input_pipeline.hip", "gemm.hip", "gemm_f32.hip", "encoder_layer.hip", "conv_block.hip", "vfe_fused.hip", "vfe_layer2.hip", "conv_tiles.hip", "conv_dense.hip", "tok_gemm.hip", "layer_fused.hip", "layer_v3.hip", "rows_gemm.hip", "dw_grouped.hip", "center_head.hip", "iou3d_nms.hip", "plan.hip", "spconv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wno-unused-result"]


# per-file additions.  -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs instead of AGPRs; the attention kernels read
# every accumulator element with VALU right after the MFMA, and the AGPR form costs one v_accvgpr_read per element
# (551 -> 177 of ~5000 instructions in the T = 64 backward)
EXTRA_FLAGS = {"attention_t32.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "attention_t16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "attention_coop.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "vfe_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "vfe_layer2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm.h"), os.path.join(CSRC, "conv_tiles.h"), os.path.join(CSRC, "dw_grouped.h"), os.path.join(CSRC, "tok_tiles.h"), os.path.join(CSRC, "layer_tail.h"), os.path.abspath(os.path.join(CSRC, "..", "..", "include", "gdmae_hip.h"))]
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        own = [os.path.join(CSRC, "attn_tiles.h"), os.path.join(CSRC, "attn16_wave.h")] if s in ("attention_coop.hip", "attention_t16.hip") else []
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps + own):
            cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {s} failed ---\n{out.decode()}\n")
        elif verbose and out.strip():
            print(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    # relink when anything was recompiled, the library is missing, or an object is newer than it (objects travel to the
    # GPU box separately; an interrupted link must not leave a stale library behind)
    if force or procs or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-L/opt/rocm/lib", "-lhipblaslt"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
