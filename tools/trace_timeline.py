#!/usr/bin/env python3
"""Timeline of the LAST denoise step in a rocprofv3 --kernel-trace CSV: one line per launch in start order --
start (us from the step's first launch), duration, gap since every earlier launch had ended (idle chip), how many other launches
overlap it, workgroups, kernel.    python tools/trace_timeline.py trace.csv > timeline.txt"""
import csv, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void |ttg::", "", name)
    return re.sub(r"\(.*", "", name)[:64]

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
rows.sort()
marks = [i for i, r in enumerate(rows) if "cfg_euler" in r[2]]
rows = rows[marks[-2] + 1: marks[-1] + 1]
t0 = rows[0][0]
end_max = rows[0][0]
print(f"{'start us':>9s} {'dur us':>7s} {'idle':>6s} {'ovl':>3s} {'blocks':>6s}  kernel")
for i, (s, e, name, blocks) in enumerate(rows):
    idle = max(0, s - end_max) / 1e3
    ovl = sum(1 for (s2, e2, _, _) in rows[max(0, i - 12): i + 12] if s2 < e and e2 > s) - 1
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {idle:6.1f} {ovl:3d} {blocks:6d}  {short(name)}")
    end_max = max(end_max, e)
