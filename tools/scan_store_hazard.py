#!/usr/bin/env python3
"""Static scan of the gfx950 ISA of the product kernels for the store-data hazard found at the end of round 3 (DESIGN s4.3):
a vector-memory store of more than 64 bits whose data VGPRs are written again before TWO wait states have passed.  On gfx950 the
last lanes of each 16-lane row are read late: with one wait state -- all the compiler (ROCm 7.2) inserts, and none for a buffer
store with an SGPR soffset -- tools/store_valu_hazard.hip still finds wrong dwords; with two it finds none.

    python tools/scan_store_hazard.py [file.s ...]      (default: compiles csrc/spdy_kernels.hip and csrc/spdy_step.hip to ISA)
Reports every (kernel, store, offending instruction) with fewer than WAIT wait states in between."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT = int(os.environ.get("SCAN_WAIT", "2"))

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

def dest_regs(op, args):
    """VGPRs an instruction writes (first operand for VALU / loads; both for the swaps)."""
    if op.startswith(("v_cmp", "v_nop", "v_readfirstlane", "v_readlane")):
        return set()
    if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")):
        return regs(args[0]) | regs(args[1])
    if op.startswith("v_"):
        return regs(args[0]) if args else set()
    return set()

def scan(path):
    kernel, found, window = None, [], []          # window: (data regs, wait states elapsed, text, line no)
    nstores = 0
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith((".", "//")):
            continue
        if s.endswith(":"):
            if not s.startswith(".L"):
                kernel = s[:-1]; window = []
            continue
        parts = s.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        # hazards first: does this instruction write data of a store still inside its window?
        d = dest_regs(op, args)
        for data, ws, text, l0 in window:
            if d & data and ws < WAIT:
                found.append((kernel, l0, text, ln, s, ws))
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):      # what follows in the file is not what follows in time
            window = []
            continue
        # advance wait states
        step = (int(args[0]) + 1) if op == "s_nop" and args else 1
        window = [(data, ws + step, text, l0) for data, ws, text, l0 in window if ws + step < WAIT]
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx([34])", op)
        if m:
            nstores += 1
            datatok = args[0] if m.group(1) == "buffer" else args[1]
            window.append((regs(datatok), 0, s, ln))
    return nstores, found

def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix="scan_isa_")
        for src in ("spdy_kernels", "spdy_step"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "--cuda-device-only", "--no-gpu-bundle-output",
                                   "-S", os.path.join(ROOT, "speedy.f90_amd", "csrc", src + ".hip"), "-o", out])
            files.append(out)
    total = 0
    for f in files:
        n, found = scan(f)
        total += len(found)
        print("%s: %d stores of more than 64 bits, %d with a data register rewritten within %d wait state(s)" % (os.path.basename(f), n, len(found), WAIT))
        per = {}
        for k, l0, text, ln, s, ws in found:
            per.setdefault(k, []).append((l0, text, ln, s, ws))
        for k, lst in per.items():
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
            print("  %s: %d" % (name, len(lst)))
            for l0, text, ln, s, ws in lst[:3]:
                print("      line %d: %s   -> %d wait state(s) later, line %d: %s" % (l0, text[:70], ws, ln, s[:60]))
    return 1 if total else 0

if __name__ == "__main__":
    sys.exit(main())
