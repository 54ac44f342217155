#!/usr/bin/env python3
"""Static scan of the gfx950 ISA of every kernel for two patterns that cost this project time:
  * store ... s_waitcnt vmcnt(0) ... store  -- stores count in vmcnt on gfx9: a wait the compiler put in front of a store block
    (for a load the block's value hangs on) also waits for the acknowledgement of the store before it: serial round trips;
  * a returning atomic followed at once by s_waitcnt vmcnt(0) -- `atomicAdd` through the compiler's atomic optimiser.
    python tools/isa_scan.py [file.hip ...]        (needs hipcc; no GPU)"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep-q-learning_amd", "csrc")


def dem(n):
    try:
        return subprocess.run(["c++filt", n.replace("DF16b", "Dh")], capture_output=True, text=True).stdout.strip().replace("half", "bf16")[:72] or n
    except Exception:
        return n


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only", "-o", tmp.name, f],
                           check=True, stderr=subprocess.DEVNULL)
            s = open(tmp.name).read()
        for m in re.finditer(r"^(\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if not name.startswith("_Z") and not name.startswith("k_"):
                continue
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


if __name__ == "__main__":
    main()
