#!/usr/bin/env python3
"""Copy a recorded GPU parity run (gpurun_out/parity_r06.json, written by tests/parity_record.py) into
tests/golden/parity_bounds.json (what the asserts are tightened against) and profiles/parity_r06.json (the record)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "parity_r06.json")
rec = json.load(open(src))
# MERGE: a gpurun call starts with an empty gpurun_out/, so a partial test run records only its own cases
for dst in (os.path.join(ROOT, "tests", "golden", "parity_bounds.json"), os.path.join(ROOT, "profiles", "parity_r06.json")):
    try:
        old = json.load(open(dst))
    except (OSError, ValueError):
        old = {}
    old.update(rec)
    json.dump(old, open(dst, "w"), indent=1, sort_keys=True)
    print(f"{dst}: {len(rec)} cases merged, {len(old)} in all")
