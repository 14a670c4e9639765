#!/usr/bin/env python3
"""Copies the generated roofline table (profiles/<round>_roofline.md: the table and the CPU-baseline line) into DESIGN.md between
the roofline markers.  usage: update_design.py [round: r05]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = open(os.path.join(ROOT, "profiles", f"{rnd}_roofline.md")).read()
keep = []
for line in src.splitlines():
    if line.startswith("## rocprofv3"):
        break
    if line.startswith("# Roofline table"):
        continue
    keep.append(line)
block = "\n".join(keep).strip() + f"\n\n(The kernel-trace listing by launch shape: `profiles/{rnd}_roofline.md`.)"
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
new = re.sub(r"<!-- roofline:begin -->.*?<!-- roofline:end -->", "<!-- roofline:begin -->\n" + block.replace("\\", "\\\\") + "\n<!-- roofline:end -->", text, flags=re.S)
assert new != text or block in text, "markers not found"
open(path, "w").write(new)
print(f"DESIGN.md: roofline block updated from profiles/{rnd}_roofline.md ({len(keep)} lines)")
