#!/usr/bin/env python3
"""Static scan of the gfx950 ISA of every kernel for two patterns that cost this project time:
  * store ... s_waitcnt vmcnt(0) ... store  -- stores count in vmcnt on gfx9: a wait the compiler put in front of a store block
    (for a load the block's value hangs on) also waits for the acknowledgement of the store before it: serial round trips;
  * a returning atomic followed at once by s_waitcnt vmcnt(0) -- `atomicAdd` through the compiler's atomic optimiser.
and for one pattern that is WRONG (exit status 1, checked by tests/test_host.py):
  * a returning atomic written as inline asm (between ;APP / ;NO_APP, so its result register is invisible to the compiler's
    wait insertion) whose destination VGPR is read or written by any instruction before the next full `s_waitcnt vmcnt(0)`.
    python tools/isa_scan.py [file.hip ...]        (needs hipcc; no GPU)"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep-q-learning_amd", "csrc")


def dem(n):
    try:
        return subprocess.run(["c++filt", n.replace("DF16b", "Dh")], capture_output=True, text=True).stdout.strip().replace("half", "bf16")[:72] or n
    except Exception:
        return n


def inflight_atomic_hazards(raw_lines):
    """raw_lines: the kernel's assembly lines, comments kept (";APP" / ";NO_APP" bracket inline asm). Returns a list of
    (atomic line, offending line) for inline-asm returning atomics whose destination register is touched while in flight."""
    bad, in_asm, pending = [], False, []          # pending: [(dest register regex, atomic text)]
    for l in raw_lines:
        t = l.strip()
        if t.startswith(";") and ("ASMSTART" in t or t.startswith(";APP")):
            in_asm = True; continue
        if t.startswith(";") and ("ASMEND" in t or t.startswith(";NO_APP")):
            in_asm = False; continue
        if not t or t.startswith(";"):
            continue
        code = t.split(";")[0]
        if "s_waitcnt" in code and "vmcnt(0)" in code:
            pending = []; continue
        for rx, atext in pending:
            if rx.search(code):
                bad.append((atext, code.strip()))
        m = re.match(r"(global|flat|buffer)_atomic_\w+\s+(v\d+|v\[\d+:\d+\])\s*,", code)
        if m and in_asm and re.search(r"\b(sc0|glc)\b", code):
            d = m.group(2)
            regs = [d] if "[" not in d else [f"v{i}" for i in range(int(d[2:].split(":")[0]), int(d.split(":")[1][:-1]) + 1)]
            for r_ in regs:
                n = int(r_[1:])
                # the register by itself, or inside a range v[a:b] that covers it
                pending.append((re.compile(r"\b%s\b|v\[(%s)\]" % (r_, "|".join(f"{a}:{b}" for a in range(max(0, n - 15), n + 1) for b in range(n, n + 16) if b > a))), code.strip()))
    return bad


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    status = 0
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            if f.endswith(".s"):
                tmp = open(f)
            else:
              subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only", "-o", tmp.name, f],
                           check=True, stderr=subprocess.DEVNULL)
            s = open(tmp.name).read()
        for m in re.finditer(r"^(\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if not name.startswith("_Z") and not name.startswith("k_"):
                continue
            for atext, off in inflight_atomic_hazards(body.split("\n")):
                status = 1
                print(f"HAZARD {os.path.basename(f)} {dem(name)}: result of `{atext}` touched in flight by `{off}`")
            lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
            seq, seen, pending, near = 0, False, False, 0
            last_store = -10 ** 9
            for i, l in enumerate(lines):
                if l.startswith(("global_store", "flat_store", "buffer_store")):
                    if pending:
                        seq += 1; pending = False
                    seen = True; last_store = i
                elif "s_waitcnt" in l and "vmcnt(0)" in l and seen:
                    pending = True; seen = False
                    near += i - last_store <= 32
            atom = sum(1 for i, l in enumerate(lines) if l.startswith("global_atomic") and " sc0" in l and any("vmcnt(0)" in x for x in lines[i + 1:i + 6]))
            if seq >= 3 or atom:
                print(f"{os.path.basename(f):18s} {dem(name):74s} store->vmcnt(0)->store: {seq:3d} ({near} within 32 instrs)   returning atomic waited at once: {atom}")


    return status


if __name__ == "__main__":
    sys.exit(main())
