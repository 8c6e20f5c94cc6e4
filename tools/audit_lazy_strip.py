#!/usr/bin/env python3
"""Audit of the strip-form trailing update's ISA (test infrastructure, run by hand after every edit of gptq_lazy.hip or a compiler change).

gptq_lazy_update_v4_kernel waits with COUNTED `s_waitcnt vmcnt(16)` at the top of a tile: that is only right while (a) the interior
path of a tile issues exactly 16 store instructions, (b) the compiler adds no vector-memory wait or load of its own to the kernel (every
load is an LDS-DMA request issued from asm) and (c) nothing spills to scratch.

usage: tools/audit_lazy_strip.py   (compiles neural_compressor_amd/csrc/gptq_lazy.hip with -save-temps into /tmp)
"""
import os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = "/tmp/lazy_audit"; os.makedirs(tmp, exist_ok=True)
src = os.path.join(root, "neural_compressor_amd", "csrc", "gptq_lazy.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-c", src, "-o", os.path.join(tmp, "lazy.o"),
                       "-save-temps=obj"], cwd=tmp)
asm = open(os.path.join(tmp, "gptq_lazy-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
m = re.search(r"^(_ZN\S*gptq_lazy_update_v4_kernel\S*):[^\n]*\n", asm, re.M)
assert m, "kernel not found"
body = asm[m.end():asm.find(".end_amdhsa_kernel", m.end())]
lines = body.split("\n")
inasm, own_waits, asm_waits = False, [], []
for l in lines:
    if "#ASMSTART" in l: inasm = True
    elif "#ASMEND" in l: inasm = False
    elif "s_waitcnt" in l and "vmcnt" in l:
        (asm_waits if inasm else own_waits).append(l.strip())
stores = [l for l in lines if "global_store_dword" in l]
loads = [l for l in lines if re.search(r"\bglobal_load_dword", l) and "lds" not in l]
dma = [l for l in lines if "global_load_lds" in l]
vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
# the interior path: the run of unguarded stores (no v_cmp / exec juggling between them) must be exactly 16 long
runs, run = [], 0
for l in lines:
    s = l.strip()
    if "global_store_dword" in s:
        run += 1
    elif s.startswith(("s_and_saveexec", "s_cbranch", "s_or_b64 exec", "v_cmp")) or re.match(r"^\.?[\w$]+:", s):
        if run: runs.append(run)
        run = 0
if run: runs.append(run)
print(f"VGPRs {vgprs}, scratch {scratch} B, LDS-DMA requests {len(dma)}, register-destination global loads {len(loads)}, stores {len(stores)} (runs {runs})")
print(f"vmcnt waits inside asm statements: {sorted(set(asm_waits))}; compiler-inserted: {own_waits}")
ok = scratch == 0 and not loads and not own_waits and 16 in runs and vgprs <= 256
print("OK" if ok else "FAIL")
sys.exit(0 if ok else 1)
