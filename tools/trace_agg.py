#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / mean us / total ms over the LAST `--last` fraction of
the trace (the graph replays), plus the busy/idle split of that window.  python tools/trace_agg.py trace.csv [--steps N]"""
import csv, sys, re, collections

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:70]

def main():
    path = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])), int(r["Workgroup_Size_X"])))
    rows.sort()
    # the timed replays are the tail: find the Euler-step kernel launches and keep the last `steps` of them
    marks = [i for i, r in enumerate(rows) if "cfg_euler" in r[2]]
    if len(marks) > steps:
        rows = rows[marks[-steps - 1] + 1: marks[-1] + 1]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    agg = collections.OrderedDict()
    for s, e, name, grid, wg in rows:
        a = agg.setdefault((short(name), grid[0] // wg, grid[1], grid[2]), [0, 0])
        a[0] += 1; a[1] += e - s
    busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
    for s, e, *_ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = (t1 - t0) / 1e6
    print(f"window {wall/steps:.2f} ms/step over {steps} step(s): {len(rows)//steps} launches/step, GPU busy {busy/1e6/steps:.2f} ms/step, "
          f"idle {wall/steps - busy/1e6/steps:.2f} ms/step, sum of kernel durations {sum(a[1] for a in agg.values())/1e6/steps:.2f} ms/step")
    print(f"{'kernel':70s} {'blocks':>18s} {'n/step':>6s} {'us':>8s} {'ms/step':>8s}")
    for (name, gx, gy, gz), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{name:70s} {str((gx, gy, gz)):>18s} {n/steps:6.1f} {t/n/1e3:8.1f} {t/1e6/steps:8.3f}")

if __name__ == "__main__":
    main()
