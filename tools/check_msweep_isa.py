#!/usr/bin/env python3
"""spmm_msweep_kernel names its registers by hand: the asm body owns v24-v255 and a0-a255 (listed as clobbers), the compiler may keep
live values only in v0-v23.  This check compiles gf_msweep.hip to ISA and fails if, for any instantiation,
  * the register allocator spilled (.vgpr_spill_count / .sgpr_spill_count != 0) or the kernel uses scratch (.private_segment_fixed_size != 0),
  * a compiler-emitted instruction (anything outside the ;;#ASMSTART ... ;;#ASMEND regions) names a vector register above v23 or any
    accumulator register -- the body keeps state there between (batch entry, hop) passes,
  * the kernel does not get the whole register file (.vgpr_count 512, .agpr_count 256: one wave per SIMD is what the image's geometry assumes).
The instantiations with phases of compiler code around the asm bodies (last template argument != 0: hub rows, the layout pre-phase; they re-zero
the accumulators before every body) are exempt from the register rule and may spill a few registers in their cold paths (time-out, trace, row-table
copy); they must still own the whole register file, and their spills are bounded (<= 48 each).
Run by tests/test_host_logic.py::test_msweep_isa_register_contract (needs hipcc; no GPU)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = os.path.join(ROOT, "graph-neural-networks_amd")
LIMIT = int(os.environ.get("MS_ISA_LIMIT", "24"))                                        # first vector register the asm body owns


def is_hub(name):
    a = re.findall(r"Li(\d+)E", name.split("spmm_msweep_kernelI", 1)[1])
    return len(a) >= 6 and a[5] != "0"


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PK, "csrc"),
                        "-S", "--cuda-device-only", os.path.join(PK, "csrc", "gf_msweep.hip"), "-o", out], check=True, capture_output=True)
        txt = open(out).read()
    bad = nk = 0
    for m in re.finditer(r"^(_Z\w*spmm_msweep_kernel\w*):\s*;.*?$(.*?)s_endpgm", txt, re.S | re.M):
        nk += 1
        name, body = m.group(1), m.group(2)
        if is_hub(name):
            continue
        in_asm = False
        for line in body.splitlines():
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            s = t.split(";")[0].strip()
            if in_asm or not s or s.startswith("."):
                continue
            for r in re.finditer(r"\b([va])\[?(\d+)(?::(\d+))?\]?", s.split(None, 1)[1] if " " in s else ""):
                hi = int(r.group(3) or r.group(2))
                if r.group(1) == "a" or hi >= LIMIT:
                    bad += 1
                    print(f"{name[:60]}: compiler-emitted instruction touches a register of the asm body: {s}")
    seen = 0
    for blk in txt[txt.index("amdhsa.kernels:"):].split("\n  - ")[1:]:
        nm = re.search(r"\.name:\s+(_Z\w*spmm_msweep_kernel\w*)", blk)
        if not nm:
            continue
        name = nm.group(1)
        seen += 1
        f = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", blk)}
        hub = is_hub(name)
        lim = 48 if hub else 0
        if f.get("vgpr_spill_count", 0) > lim or f.get("sgpr_spill_count", 0) > lim or (f.get("private_segment_fixed_size", 0) and not hub) or f.get("vgpr_count") != 512 or f.get("agpr_count") != 256:
            bad += 1
            print(f"{name[:60]}: {f}")
    print(f"gf_msweep.hip: {nk} kernel bodies scanned, {seen} metadata blocks checked, {bad} violations")
    return 1 if bad or nk == 0 or seen != nk else 0


if __name__ == "__main__":
    sys.exit(main())
