#!/usr/bin/env python3
"""spmm_sweep_kernel keeps its gather ring (v14-v15 temporaries, v16-v27) and its accumulators (v28-v127) in registers it names by hand inside inline asm;
the compiler must stay below v14 everywhere else.  This compiles gf_sweep.hip to ISA and checks every instruction OUTSIDE the asm
blocks.  Run by tests/test_host_logic.py::test_sweep_kernel_register_map (needs hipcc; no GPU).   usage: check_sweep_isa.py [file.s]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT = 14


def isa_text():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    pk = os.path.join(ROOT, "graph-neural-networks_amd")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "sweep.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pk, "csrc"),
                        "-S", "--cuda-device-only", os.path.join(pk, "csrc", "gf_sweep.hip"), "-o", out], check=True, capture_output=True)
        return open(out).read()


def main():
    txt = isa_text()
    m = re.search(r"^(_Z\w*spmm_sweep_kernel\w*):\s*;.*?$(.*)", txt, re.S | re.M)
    assert m, "kernel not found"
    in_asm, bad, n = False, [], 0
    for line in m.group(2).splitlines():
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0]
        n += 1
        # registers the COMPILER chose print as vN / v[lo:hi]; the kernel's hand-named ones as v[N] -- inside the asm blocks too, the
        # compiler-chosen operands (inputs, temporaries) must stay below the limit
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", code)]
        regs += [int(hi) for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
        if any(r >= LIMIT for r in regs):
            bad.append(code)
    spill = re.search(r"\.vgpr_spill_count:\s*(\d+)", txt)
    print(f"spmm_sweep_kernel: {n} compiler instructions checked, {len(bad)} touch v{LIMIT}+; vgpr spills: {spill.group(1) if spill else '?'}")
    for b in bad[:20]:
        print("   ", b)
    return 1 if bad or (spill and int(spill.group(1)) != 0) else 0


if __name__ == "__main__":
    sys.exit(main())
