#!/usr/bin/env python
"""Place the address of a 'Memory access fault' line among the '[alloc] <ptr> <bytes>' lines of the same log
(RPDE_LOG_ALLOC=1): which allocation holds it, or which ones end / start next to it."""
import re
import sys

txt = open(sys.argv[1], errors="replace").read()
allocs = [(int(p, 16), int(n)) for p, n in re.findall(r"\[alloc\] (0x[0-9a-f]+) (\d+)", txt)]
m = re.search(r"Memory access fault.*?address (0x[0-9a-f]+)", txt)
last = re.findall(r"\[launch\] ([^\n]*?)(?: \.\.\.|$)", txt)
print(f"{sys.argv[1]}: {len(allocs)} allocations, last launch: {last[-1] if last else '?'}")
if not m:
    sys.exit("no fault line")
a = int(m.group(1), 16)
inside = [(p, n, i) for i, (p, n) in enumerate(allocs) if p <= a < p + n]
print(f"fault address {a:#x}")
for p, n, i in inside:
    print(f"  inside allocation #{i}: {p:#x} + {n} (offset {a - p})")
below = sorted(((a - (p + n), p, n, i) for i, (p, n) in enumerate(allocs) if p + n <= a))[:3]
above = sorted(((p - a, p, n, i) for i, (p, n) in enumerate(allocs) if p > a))[:2]
for d, p, n, i in below:
    print(f"  {d} bytes behind the end of allocation #{i}: {p:#x} + {n}")
for d, p, n, i in above:
    print(f"  {d} bytes in front of allocation #{i}: {p:#x} + {n}")
