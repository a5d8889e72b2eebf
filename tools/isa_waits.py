"""Audit of the device ISA of one source file: per kernel, the loops (blocks the assembler annotates `in Loop:`) and the load-counter
waits inside them.  A `s_waitcnt vmcnt(0)` inside a loop that also issues global loads usually means the loads are drained before
the arithmetic they were meant to overlap with (a control-flow join with loads in flight, or a select on just-loaded registers).
   python tools/isa_waits.py <file.hip> [extra hipcc flags...]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = "/tmp/_isa_%s.s" % os.path.basename(src).replace(".hip", "")
extra = sys.argv[2:]
if src.split("/")[-1] in ("attention_t32.hip", "attention_t16.hip", "vfe_fused.hip", "vfe_layer2.hip"):
    extra += ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result"] + extra + ["-S", "--cuda-device-only", "-o", out, src]
subprocess.run(cmd, cwd=os.path.join(REPO, "gd-mae_amd", "csrc"), check=True, stderr=subprocess.DEVNULL)
kern = None; inloop = False; stats = {}
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern = m.group(1); stats[kern] = dict(loads=0, v0=0, vN=0, mfma=0, lines=[]); inloop = False; continue
    if kern is None: continue
    if line.startswith(".Lfunc_end"): kern = None; continue
    if re.match(r"^\.LBB", line): inloop = "Loop" in line
    if not inloop: continue
    t = line.strip()
    st = stats[kern]
    if t.startswith(("global_load", "buffer_load")): st["loads"] += 1
    elif t.startswith("v_mfma"): st["mfma"] += 1
    elif t.startswith("s_waitcnt") and "vmcnt(" in t:
        n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
        if n == 0: st["v0"] += 1
        else: st["vN"] += 1
for k, st in stats.items():
    if st["loads"] == 0: continue
    try: name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    except OSError: name = k
    print("%-90s in loops: %3d loads, %3d mfma, vmcnt(0) x %d, vmcnt(n>0) x %d" % (name[:90], st["loads"], st["mfma"], st["v0"], st["vN"]))
