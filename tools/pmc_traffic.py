#!/usr/bin/env python3
"""HBM-side bytes per launch of the MFMA kernels from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs
of tools/pmc_step.py).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of a wide coalesced
read stream (MI355X_MICROARCH.md, HBM section); both count L2 <-> fabric requests, Infinity-Cache hits included.
python tools/pmc_traffic.py <fetch_dir> <write_dir> <key e.g. vgl_lo> <out_traffic.json> <out_raw.json>"""
import csv, glob, json, os, re, sys, collections


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(gemm_pp_kernel|gemm_w320h_kernel|gemm_w320_kernel|gemm_kernel|attn_kernel|attn_pipe_kernel|attn8_kernel|tattn_kernel|splitk_epilogue_kernel)<[^>]*>", r["Kernel_Name"])
            if not m:
                continue
            a = acc[m.group(0)]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def main():
    fetch_dir, write_dir, key, out, raw_out = sys.argv[1:6]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    raw, traffic = {}, {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        if k not in write:
            continue
        b = (2.0 * fetch[k][1] + write[k][1]) * 1024.0
        raw[k] = {"launches_profiled": fetch[k][0], "fetch_kb_raw": fetch[k][1], "write_kb": write[k][1], "hbm_bytes_per_launch": b}
        traffic[k] = int(round(b))
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc["_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace --kernel-include-regex "
                   "'gemm_kernel|attn_kernel') -- python tools/pmc_step.py <mode> <res>; tools/pmc_traffic.py: per launch = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950 FETCH_SIZE reads half of a wide coalesced stream, "
                   "MI355X_MICROARCH.md section HBM; counts L2<->fabric requests, Infinity-Cache hits included)")
    doc[key] = traffic
    commit = os.environ.get("TT_COMMIT", "")
    if not commit:
        try:
            import subprocess
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = ""
    doc["_measured_at_commit"] = commit or "unknown (no .git on the GPU box: see the committing revision of this file)"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    doc["_kernel_source_sha16"] = bench.csrc_hash()        # bench.py reports `roofline.traffic` only for the same kernel sources
    json.dump(doc, open(out, "w"), indent=1)
    json.dump(raw, open(raw_out, "w"), indent=1)
    for k, v in traffic.items():
        print(f"{k:60s} {raw[k]['launches_profiled']:5d} launches  {v/1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
