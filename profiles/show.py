#!/usr/bin/env python3
"""Pretty-print the kernel table of a profiles/summarize_rocprof.py summary: short name, calls, total/avg/min/max us, %."""
import sys
for l in open(sys.argv[1]):
    if l.startswith('#'):
        if 'per-dispatch' in l:
            break
        continue
    parts = l.rsplit('|', 6)
    print(parts[0][:78].ljust(80), *[p.strip().rjust(10) for p in parts[1:]])
