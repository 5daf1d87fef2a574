#!/usr/bin/env python3
"""merge_tuned.py SHIPPED.csv EXTRA.csv: appends the (op, shape) lines of EXTRA that SHIPPED does not have yet."""
import sys

ship, extra = sys.argv[1], sys.argv[2]
have = set()
lines = open(ship).read().splitlines()
for ln in lines:
    p = ln.split(",")
    if p[0] != "Validator":
        have.add((p[0], p[1]))
added = 0
for ln in open(extra).read().splitlines():
    p = ln.split(",")
    if p[0] != "Validator" and (p[0], p[1]) not in have:
        lines.append(ln)
        have.add((p[0], p[1]))
        added += 1
open(ship, "w").write("\n".join(lines) + "\n")
print("added", added, "entries")
