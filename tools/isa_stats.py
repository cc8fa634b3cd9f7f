#!/usr/bin/env python3
"""Instruction mix of one kernel from `hipcc -S --cuda-device-only` output, per basic block, with an issue-cycle estimate
(gfx950 rates of tools/valu_rate_bench.hip: fp32 add/mul/fma, v_mov, v_and, v_add_u32 2 cycles per wave; other VALU 4;
transcendentals and f64 8; LDS / VMEM / SALU counted, not priced).
  python tools/isa_stats.py build/isa/advect.s 'sl_advect_bwd_row64ILi2' [--blocks]"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
show_blocks = "--blocks" in sys.argv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
CHEAP = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_add_u32",
         "v_mac_f32", "v_subrev_f32", "v_fmaak_f32", "v_fmamk_f32", "v_pk_")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_exp_", "v_log_")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_mfma", "v_smfmac")):
            return "mfma", 0
        if "f64" in op:
            return "valu_f64", 8
        if op.startswith(TRANS):
            return "valu_trans", 8
        if op.startswith(CHEAP):
            return "valu_cheap", 2
        return "valu_other", 4
    if op.startswith("ds_"):
        return ("lds_atomic" if ("add" in op or "max" in op or "min" in op or "cmpst" in op) else "lds"), 0
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0
    if op.startswith("s_waitcnt"):
        return "waitcnt", 0
    if op.startswith("s_barrier"):
        return "barrier", 0
    if op.startswith("s_"):
        return "salu", 0
    return "other", 0


blocks = []
cur = ["entry", {}, 0]
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith((";", ".", "//")):
        if re.match(r"^\.LBB\S+:", s):
            blocks.append(cur)
            cur = [s.split(":")[0], {}, 0]
        continue
    op = s.split()[0]
    k, cyc = classify(op)
    cur[1][k] = cur[1].get(k, 0) + 1
    cur[2] += cyc
blocks.append(cur)
tot = {}
for name, d, cyc in blocks:
    for k, v in d.items():
        tot[k] = tot.get(k, 0) + v
print("kernel", pat, "instructions by class:", dict(sorted(tot.items())))
print("VALU issue cycles (all blocks once):", sum(b[2] for b in blocks))
big = sorted(blocks, key=lambda b: -b[2])[:8 if show_blocks else 3]
for name, d, cyc in big:
    print(f"  block {name}: valu cycles {cyc}  {dict(sorted(d.items()))}")
m = re.search(r"; NumVgprs: (\d+)", "\n".join(lines[end:end + 60]))
for l in lines[end:end + 80]:
    if any(t in l for t in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "NumSgprs")):
        print("  ", l.strip())
