#!/usr/bin/env python3
"""Audit of the direct-to-register dequant-GEMM's ISA (test infrastructure, run by hand after every edit of gemm_d2r.hip).

The kernel keeps its 256 accumulators in literally named AGPRs that the compiler does not know about (gemm_d2r.hip), so three
things must hold in the emitted code: no scratch, no compiler-generated v_accvgpr_* (only the asm statements may touch AGPRs),
and -- for speed -- only a handful of instructions between consecutive MFMAs of the K-loop.

usage: tools/audit_d2r.py [--dump N]   (compiles neural_compressor_amd/csrc/gemm_d2r.hip with -save-temps into /tmp)
"""
import collections, os, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = "/tmp/d2r_audit"; os.makedirs(tmp, exist_ok=True)
src = os.path.join(root, "neural_compressor_amd", "csrc", "gemm_d2r.hip")
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-w", "-c", src,
                       "-o", os.path.join(tmp, "d2r.o"), "-save-temps=obj"] + extra, cwd=tmp)
asm = open(os.path.join(tmp, "gemm_d2r-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else 0
bad = 0
pos = 0
while True:
    i = asm.find("woq_gemm_w4_d2r_kernel", pos)
    if i < 0:
        break
    j = asm.find(":", i)
    name = asm[asm.rfind("\n", 0, i) + 1:j]
    if not name.startswith("_Z") or "\n" in name:
        pos = i + 1
        continue
    end = asm.find(".Lfunc_end", j)
    body = asm[j:end]
    pos = end
    lines = body.split("\n")
    inasm, acc_outside = False, 0
    for l in lines:
        if "#ASMSTART" in l: inasm = True
        elif "#ASMEND" in l: inasm = False
        elif not inasm and "accvgpr" in l: acc_outside += 1
    ins = [l.strip() for l in lines if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    mf = [k for k, l in enumerate(ins) if l.startswith("v_mfma")]
    gaps = collections.Counter(mf[k + 1] - mf[k] - 1 for k in range(len(mf) - 1))
    scratch = body.count("scratch_")
    ok = acc_outside == 0 and scratch == 0
    bad += not ok
    print(f"{name[:70]}: {len(mf)} MFMAs, compiler v_accvgpr {acc_outside}, scratch ops {scratch} -> {'OK' if ok else 'FAIL'}")
    print("  instructions between consecutive MFMAs (count: occurrences):", dict(sorted(gaps.items())))
    if dump and mf:
        print("\n".join("    " + l for l in ins[mf[0] - 3: mf[0] + dump]))
sys.exit(1 if bad else 0)
