#!/usr/bin/env python3
"""Static check of a kernel's ISA (`hipcc -S --cuda-device-only`) for the hazard of hand-issued asynchronous loads: between
an inline-asm `global_load_*` and the next `s_waitcnt vmcnt(..)` on the fall-through path no instruction may READ the
load's destination registers - the compiler believes the value exists when the asm statement ends, so a copy it inserts
there (a PHI move when the old value of the variable is still live below the load) reads garbage, and the data that arrives
later lands in a register that holds something else by then.
    python tools/async_load_check.py build/isa/x.s ['<kernel name fragment>']      (no fragment: every kernel of the file)
Scan of the fall-through path, following unconditional branches (conditional ones are not taken); the scan of a load
ends at the next vmcnt wait.  Prints every suspect line."""
import re
import sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else None       # no pattern: every kernel of the file
lines = open(path).read().split("\n")
LOAD = ("global_load", "buffer_load", "flat_load")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(name, body):
    in_asm = False
    bad = n_loads = 0
    for i, l in enumerate(body):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not (in_asm and s.startswith(LOAD)) or " lds" in s or "_lds_" in s.split()[0]:
            continue                        # (LDS-DMA loads write no vector register)
        n_loads += 1
        dst = regs(s.split()[1].rstrip(","))
        j, steps = i + 1, 0
        while j < len(body) and steps < 600:
            t = body[j].strip()
            steps += 1
            if t.startswith("s_waitcnt") and "vmcnt" in t:
                break
            if t.startswith(("s_endpgm", "s_setpc")):
                break
            if t.startswith("s_branch"):                           # follow the unconditional branch
                lab = t.split()[1] + ":"
                j = next((k for k, u in enumerate(body) if u.startswith(lab)), len(body))
                continue
            if not t or t.startswith((";", ".")) or t.startswith("s_"):
                j += 1
                continue
            ops = [o.strip().rstrip(",") for o in t.split()[1:]]
            srcs = set()
            for o in ops[1:]:                      # first operand = destination (stores aside)
                srcs |= regs(o)
            if t.startswith(("global_store", "ds_write", "scratch_store")):
                srcs |= regs(ops[0])
            if srcs & dst:
                bad += 1
                print("  %s line %d: load -> v%s read before its wait at line %d:  %s" % (name[:60], i, sorted(dst)[:1], j, t))
                break
            j += 1
    return n_loads, bad


starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and (pat is None or pat in l)]
tot = [0, 0, 0]
for st in starts:
    end = next((i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm")), len(lines))
    n, b = check(lines[st].split(":")[0], lines[st:end])
    if n:
        tot[0] += 1; tot[1] += n; tot[2] += b
        if pat is not None or b:
            print("%s: %d inline-asm loads, %d read before a vmcnt wait" % (lines[st].split(":")[0][:90], n, b))
print("%s: %d kernels with inline-asm register loads, %d loads, %d read before their wait" % (path, tot[0], tot[1], tot[2]))
