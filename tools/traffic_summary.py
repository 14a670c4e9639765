#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh into profiles/traffic_rNN.json.

rocprofv3 reports both counters in KiB.  MI355X_MICROARCH.md (section HBM): on gfx950
FETCH_SIZE reads exactly half the bytes of a wide coalesced streaming read and other
access patterns are uncalibrated, WRITE_SIZE reads 16-byte streaming stores exactly.
The probe therefore launches a calibration kernel with known traffic in front of every workload
(stand-alone post_process on a 4096^2 RGBA frame: 268435456 B read, 268435456 B written) and the
read-side factor measured on it is applied to the reprojection kernels.  The dispatches are
attributed to workloads by their ORDER (tools/traffic_probe.py writes it): the calibration
launches separate the groups — warm-up groups are skipped, a measured group is the sum of ALL its dispatches (window
launches, fill launches, merged multi-output launches) divided by its frames."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (kernel_source_sha: the stamp bench.py checks before it trusts this file)

src_dir, out_path = sys.argv[1], sys.argv[2]
KNOWN = 4096 * 4096 * 4 * 4
order = json.load(open(os.path.join(src_dir, "order.json")))
SKIP = ("build_tables_kernel", "build_xsep_kernel", "synth_fill_kernel", "checksum_kernel", "__amd_rocclr", "geo_build_lists_kernel")


def dispatches(counter, pass_dir=None, scale=1024.0):
    """[(dispatch id, kernel name, value)] in dispatch order (FETCH_SIZE / WRITE_SIZE: KiB -> bytes)."""
    rows = {}
    for f in glob.glob(os.path.join(src_dir, pass_dir or counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                i = int(r["Dispatch_Id"])
                rows[i] = (r["Kernel_Name"], rows.get(i, ("", 0.0))[1] + float(r["Counter_Value"]) * scale)
    return [(i, k, v) for i, (k, v) in sorted(rows.items()) if not any(s in k for s in SKIP)]


def per_workload(counter, pass_dir=None, scale=1024.0):
    """calibration average, {workload: (bytes per frame, kernel names, frames averaged)}"""
    groups, cal, cur = [], [], None
    for _i, k, v in dispatches(counter, pass_dir, scale):
        if "post_process_kernel" in k:
            cal.append(v)
            cur = []
            groups.append(cur)
        elif cur is not None:
            cur.append((k, v))
    out = {}
    for o, g in zip(order, groups):
        if o["workload"] is None:  # a warm-up group
            continue
        frames = o["frames"]
        if not g or len(groups) != len(order):
            out[o["workload"]] = None  # the dispatch sequence is not what the probe says: do not guess
            continue
        out[o["workload"]] = (sum(v for _k, v in g) / frames, sorted({k for k, _v in g}), frames, len(g) / frames)
    return (sum(cal) / len(cal) if cal else None), out


cal_f, fetch = per_workload("FETCH_SIZE")
cal_w, write = per_workload("WRITE_SIZE")
read_factor = KNOWN / cal_f if cal_f else 2.0
write_factor = KNOWN / cal_w if cal_w else 1.0
result = {"_kernel_source_sha": bench.kernel_source_sha(),
          "_calibration": {"kernel": "post_process_kernel on a 4096^2 RGBA frame", "known_bytes_each_way": KNOWN, "FETCH_SIZE_bytes_raw": cal_f,
                           "WRITE_SIZE_bytes_raw": cal_w, "read_factor": read_factor, "write_factor": write_factor},
          "_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/traffic_probe.py; KiB -> "
                     "bytes; reads scaled by the factor measured on the calibration kernel (guide: 2.0 for 16 B/lane "
                     "streams on gfx950), writes taken as reported; per FRAME of the workload (one launch; six for the cubemap)"}
# Cross-check of the read side (optional third pass, tools/collect_traffic.sh): the L2's read requests to the fabric by
# size, 32 / 64 / 128 bytes each.  No streaming-pattern factor is involved, so it also holds for the gathers.
by_size = {}
if glob.glob(os.path.join(src_dir, "RDREQ", "**", "*counter_collection.csv"), recursive=True):
    parts = {n: per_workload(n, "RDREQ", 1.0) for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}

    def request_bytes(get):
        vals = {n: get(parts[n]) for n in parts}
        if any(v is None for v in vals.values()):
            return None
        n_all, n32, n64, n128 = (vals[n] for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
        return {"requests": n_all, "of_32B": n32, "of_64B": n64, "of_128B": n128,
                # requests the three size counters do not cover are taken as 64-byte ones
                "bytes": 32.0 * n32 + 64.0 * n64 + 128.0 * n128 + 64.0 * max(0.0, n_all - n32 - n64 - n128)}

    cal_rq = request_bytes(lambda p: p[0])
    result["_calibration"]["read_requests_by_size"] = cal_rq
    for o in order:
        if o["workload"] is None:
            continue
        rq = request_bytes(lambda p, wl=o["workload"]: (p[1].get(wl) or (None,))[0])
        if rq:
            by_size[o["workload"]] = rq
for o in order:
    wl = o["workload"]
    if wl is None:
        continue
    f, w = fetch.get(wl), write.get(wl)
    if not f or not w:
        continue
    rd, wr = f[0] * read_factor, w[0]
    result[wl] = {"kernels": f[1], "frames_averaged": f[2], "launches_per_frame": f[3], "FETCH_SIZE_bytes_raw": f[0],
                  "WRITE_SIZE_bytes_raw": w[0], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    if wl in by_size:
        result[wl]["read_requests_by_size"] = by_size[wl]
json.dump(result, open(out_path, "w"), indent=1)
print(json.dumps(result, indent=1))
