#!/usr/bin/env python
"""Where does a kernel wait for ONE load at a time?  Reads the gfx950 assembly hipcc leaves behind with -save-temps and, per kernel, lists the
vector-memory loads and how many of them are followed by `s_waitcnt vmcnt(0)` before the next load is issued -- the signature of loads that
were written to be in flight together but come out of the compiler as one memory round trip after the other (a load inside `if (lane has
work)`: the PHI of the loaded value is resolved by moves at the end of the predicated block, which need the data; or a load sunk into the
conditional block that consumes it).  Round 4 found the kNN kernel's four candidate loads and its two home-slot probes serialised this way.

    python tools/isa_load_chains.py <file.s> [kernel-substring] [--min-serial N]

<file.s>: `hipcc --offload-arch=gfx950 --cuda-device-only -S -O3 ... x.hip -o x.s` (or the *-hip-amdgcn-*.s that -save-temps leaves).  --min-serial N
lists only kernels with at least N single-load-then-wait sequences (pointer chases through descriptors account for two or three in every batched
kernel).  tests/test_isa_loads.py holds the groups the hot kernels must keep.
"""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line.rstrip("\n"))
            if line.strip().startswith(".end_amdhsa_kernel") or line.startswith("\t.section"):
                pass
    if name:
        yield name, body


def analyse(body):
    loads = waits0 = 0
    serial = 0   # loads whose next vector-memory event is a vmcnt(0) wait (nothing else issued in between)
    pending = False
    runs, run = [], 0
    for ln in body:
        t = ln.strip()
        if t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            loads += 1
            run += 1
            pending = True
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            waits0 += 1
            if pending and run == 1:
                serial += 1
            if run:
                runs.append(run)
            run = 0
            pending = False
        elif t.startswith(".section") or t.startswith(".amdhsa_kernel"):
            break  # (the kernel's code is over; an s_endpgm is not the end -- early exits have their own)
    if run:
        runs.append(run)
    return loads, waits0, serial, runs


def main():
    argv = list(sys.argv[1:])
    min_serial = 0
    if "--min-serial" in argv:
        k = argv.index("--min-serial")
        min_serial = int(argv[k + 1])
        del argv[k:k + 2]
    path = argv[0]
    pat = argv[1] if len(argv) > 1 else ""
    for name, body in kernels(path):
        if pat and pat not in name:
            continue
        if not any(".amdhsa_kernel" in b or "s_endpgm" in b for b in body):
            continue
        loads, waits0, serial, runs = analyse(body)
        if loads == 0 or serial < min_serial:
            continue
        short = re.sub(r"^_ZN3lio\d+", "", name)[:70]
        print(f"{short:72s} loads {loads:3d}  vmcnt(0) waits {waits0:3d}  single-load-then-wait {serial:3d}  groups {runs}")


if __name__ == "__main__":
    main()
