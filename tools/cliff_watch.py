#!/usr/bin/env python3
"""tools/cliff_watch.py LOG -- per-pass k_count rates of a DFK_TRACE=1 bench log: every pass's G instances/s, the passes
below 0.9 x the median of their place in the step, and what the trace says around them."""
import re, sys, statistics
rates, lines = [], open(sys.argv[1], errors="replace").read().splitlines()
for i, l in enumerate(lines):
    m = re.search(r"k_count: (\d+) instances in ([\d.]+) ms \(([\d.]+) G instances/s\)", l)
    if m and int(m.group(1)) > 10**9: rates.append((i, int(m.group(1)), float(m.group(2)), float(m.group(3))))
if not rates: sys.exit("no k_count lines")
med = statistics.median(r[3] for r in rates)
slow = [r for r in rates if r[3] < 0.9 * med]
print(f"{len(rates)} main passes, median {med:.1f} G instances/s, min {min(r[3] for r in rates):.1f}, slow (< 0.9 x median): {len(slow)}")
for i, n, ms, g in slow:
    print(f"--- line {i}: {n} instances in {ms} ms = {g} G/s")
    for l in lines[max(0, i - 12):i + 3]: print("    " + l[:200])
