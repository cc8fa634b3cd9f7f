#!/usr/bin/env python3
"""Registers, scratch and LDS of kernels straight from the built object (no 5-minute `hipcc -S`):
  python tools/kernel_meta.py build/obj/gemm.o [name substring ...]"""
import os, re, subprocess, sys, tempfile
obj, pats = sys.argv[1], sys.argv[2:]
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={d}/fat.bin", obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={d}/fat.bin", f"--output={d}/k.co", "--unbundle"])
    t = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{d}/k.co"]).decode()
for blk in t.split("- .agpr_count:")[1:]:
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m or (pats and not any(p in m.group(1) for p in pats)):
        continue
    g = lambda k: (re.findall(k + r":\s+(\d+)", blk) or ["?"])[0]
    print("%-78s vgpr %3s scratch %4s spills %3s lds(static) %6s" % (m.group(1)[:78], g(r"\.vgpr_count"), g(r"\.private_segment_fixed_size"),
                                                                       g(r"\.vgpr_spill_count"), g(r"\.group_segment_fixed_size")))
