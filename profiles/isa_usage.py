#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel (VGPRs, SGPRs, scratch, occupancy, LDS).
usage: hipcc ... -c x.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> usage.txt; isa_usage.py usage.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
    name = b.split("\n", 1)[0].strip()
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if flt and flt not in name:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    row = (name[:90], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
    print("%-90s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %2s LDS %s" % row)
