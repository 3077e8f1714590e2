#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (make asm)."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "resource_usage.txt").read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    short = re.sub(r"^_ZN3dfk\d+", "", name)[:44]
    print("%-46s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %2s LDS %6s" % (
        short, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"),
        g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
