#!/usr/bin/env python3
"""Regenerate DESIGN.md's table of environment knobs (between the KNOBS markers) from the one table in
utmos_amd/csrc/utmos_hip.hip (g_knobs) and the places the fields are used:  python3 tools/knob_table.py"""
import glob
import os
import re

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "utmos_amd/csrc/utmos_hip.hip")).read()
knobs = re.findall(r'UTM_KNOB_[ID]\("(\w+)", (\w+), ([^)]+)\)', src)
files = sorted(glob.glob(os.path.join(root, "utmos_amd/csrc/*.h"))) + [os.path.join(root, "utmos_amd/csrc/utmos_hip.hip")]
rows = []
for env, field, dflt in knobs:
    uses = []
    for f in files:
        for i, ln in enumerate(open(f), 1):
            if re.search(r"(tune|tn)\." + field + r"\b", ln) and "UTM_KNOB" not in ln:
                uses.append(f"{os.path.basename(f)}:{i}")
    rows.append(f"| `{env}` | {dflt.strip()} | {', '.join(uses)} |")
table = ("| knob | default | where the value is used (every knob is READ in one place: `read_tune`, `utmos_hip.hip`, at `utm_ctx_create` and `utm_reset`) |\n"
         "|---|---|---|\n" + "\n".join(rows))
path = os.path.join(root, "DESIGN.md")
text = open(path).read()
a, b = "<!-- KNOBS BEGIN -->", "<!-- KNOBS END -->"
text = text[:text.index(a) + len(a)] + "\n" + table + "\n" + text[text.index(b):]
open(path, "w").write(text)
print(f"{len(rows)} knobs")
