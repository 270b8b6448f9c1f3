#!/usr/bin/env python3
"""Summarise a rocprofv3 results DB (rocpd sqlite) per kernel: calls/step, ms/step, avg us, share.
usage: prof_summary.py run_results.db STEPS [--md]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = re.sub(r"\(.*", "", n) if len(n) > 90 else n
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms = {tot/1e6/steps:.2f} ms/step over {steps} steps; {len(rows)/steps:.0f} launches/step")
print("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|")
for n, a in sorted(agg.items(), key=lambda x: -x[1][1])[:45]:
    print(f"| `{n[:100]}` | {a[0]/steps:.1f} | {a[1]/1e6/steps:.3f} | {a[1]/a[0]/1e3:.1f} | {100*a[1]/tot:.2f} |")
