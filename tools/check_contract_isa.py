#!/usr/bin/env python3
"""gf_contract.hip / gf_gradw.hip issue their output stores through inline asm (the compiler then counts loads only in its waitcnt
pass).  The hazard recogniser does not look into inline asm, so a store whose data registers are written directly by an MFMA would
miss the XDL-write -> VMEM-read wait states.  Today every stored value passes through a compiler-visible VALU instruction first (the
ReLU select / `* one`); this check compiles both files to ISA and fails if the instruction that last wrote a store's data register (in program order) is an MFMA.  Run by tests/test_host_logic.py::test_asm_stores_never_read_mfma_results (needs hipcc; no GPU)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = os.path.join(ROOT, "graph-neural-networks_amd")


def regs(tok):
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def check(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PK, "csrc"),
                        "-S", "--cuda-device-only", os.path.join(PK, "csrc", src), "-o", out], check=True, capture_output=True)
        txt = open(out).read()
    bad = nk = ns = 0
    for m in re.finditer(r"^(_Z\w+):\s*;.*?$(.*?)s_endpgm", txt, re.S | re.M):
        body = m.group(2)
        if "v_mfma" not in body:
            continue
        nk += 1
        writer = {}                               # register -> mnemonic of the instruction that wrote it last (program order)
        in_asm = False
        for line in body.splitlines():
            if line.strip().startswith(";;#ASMSTART"):
                in_asm = True
            elif line.strip().startswith(";;#ASMEND"):
                in_asm = False
            s = line.strip().split(";")[0].strip()
            if not s or s.startswith(".") or s.startswith(";"):
                continue
            parts = s.split(None, 1)
            op = parts[0]
            ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
            if op.startswith("global_store") or op.startswith("buffer_store"):
                if not in_asm:
                    continue                      # the compiler's own stores: its hazard recogniser sees them
                ns += 1
                data = regs(ops[1]) if len(ops) > 1 else set()
                hit = [r for r in data if writer.get(r, "").startswith("v_mfma")]
                if hit:
                    bad += 1
                    print(f"{src}: {m.group(1)[:70]}: a store's data register was last written by an MFMA: {s}")
                continue
            if ops and (op.startswith("v_") or op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("ds_read")
                        or op.startswith("scratch_load")):
                for r in regs(ops[0]):
                    writer[r] = op
    print(f"{src}: {nk} MFMA kernels, {ns} asm stores checked, {bad} read MFMA results directly")
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(f) for f in ("gf_contract.hip", "gf_gradw.hip")) else 0)
