#!/usr/bin/env python3
"""Bubbles at the step boundaries of a rocprofv3 --kernel-trace CSV of bench.py: for every denoise step (delimited by the
cfg_euler launch that ends it) the chip-idle time before its first launch, the gaps (> 3 us) among its first 12 launches, the
step's span and its total idle time.     python tools/step_bubbles.py trace.csv"""
import csv, re, sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", re.sub(r"void |ttg::|\(anonymous namespace\)::", "", r["Kernel_Name"]))[:48]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "cfg_euler" in r[2]]
for a, b in zip(marks[:-1], marks[1:]):
    step = rows[a + 1: b + 1]
    prev_end = max(e for _, e, _ in rows[max(0, a - 40): a + 1])
    end_max, idle, head = prev_end, 0.0, []
    for i, (s, e, name) in enumerate(step):
        gap = max(0, s - end_max) / 1e3
        idle += gap
        if i < 12 and gap > 3:
            head.append(f"{gap:.0f}us before #{i} {name}")
        end_max = max(end_max, e)
    print(f"step span {(step[-1][1] - prev_end) / 1e6:7.3f} ms  idle {idle / 1e3:6.3f} ms  launches {len(step)}  | " + "; ".join(head))
