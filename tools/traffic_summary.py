#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh.

rocprofv3 reports both counters in KiB.  MI355X_MICROARCH.md (section HBM): on gfx950
FETCH_SIZE reads exactly half the bytes of a wide coalesced streaming read and other
access patterns are uncalibrated, WRITE_SIZE reads 16-byte streaming stores exactly.
The probe therefore runs a calibration kernel with known traffic (stand-alone
post_process on a 4096^2 RGBA frame: 268435456 B read, 268435456 B written) and the
read-side factor measured on it is applied to the reprojection kernels."""
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


def per_kernel(counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src_dir, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


fetch, nf = per_kernel("FETCH_SIZE")
write, nw = per_kernel("WRITE_SIZE")
cal = [k for k in fetch if "post_process_kernel" in k]
read_factor, write_factor = 2.0, 1.0
calibration = None
if cal:
    k = cal[0]
    read_factor = KNOWN / fetch[k]
    write_factor = KNOWN / write[k] if k in write and write[k] else 1.0
    calibration = {"kernel": k, "known_bytes_each_way": KNOWN, "FETCH_SIZE_bytes_raw": fetch[k],
                   "WRITE_SIZE_bytes_raw": write.get(k), "read_factor": read_factor, "write_factor": write_factor}
names = {"reproject_bicubic_win_kernel<0, 1,": "fisheye_to_rect_bicubic",
         "reproject_bicubic_win_kernel<0, 3,": "equirect_to_rect_bicubic",
         "reproject_tile_kernel<1, 3, 1, 4>": "equirect_to_fisheye_bilinear",
         "reproject_tile_kernel<0, 3, 0, 4>": "equirect_to_rect_nearest",
         "reproject_bicubic_win_kernel<4, 0,": "rect_to_equirect_bicubic"}
result = {"_kernel_source_sha": bench.kernel_source_sha(),
          "_calibration": calibration,
          "_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/traffic_probe.py; KiB -> "
                     "bytes; reads scaled by the factor measured on the calibration kernel (guide: 2.0 for 16 B/lane "
                     "streams on gfx950), writes taken as reported"}
for k in fetch:
    for pat, wl in names.items():
        if pat in k:
            rd = fetch[k] * read_factor
            wr = write.get(k, float("nan"))
            result[wl] = {"kernel": k, "launches_averaged": nf[k], "FETCH_SIZE_bytes_raw": fetch[k],
                          "WRITE_SIZE_bytes_raw": write.get(k), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                          "hbm_bytes_per_launch": rd + wr}
json.dump(result, open(out_path, "w"), indent=1)
print(json.dumps(result, indent=1))
