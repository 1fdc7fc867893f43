#!/usr/bin/env python3
"""Copy a recorded GPU parity run (gpurun_out/parity_r03.json, written by tests/parity_record.py) into
tests/golden/parity_bounds.json (what the asserts are tightened against) and profiles/parity_r03.json (the record)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "parity_r03.json")
rec = json.load(open(src))
json.dump(rec, open(os.path.join(ROOT, "tests", "golden", "parity_bounds.json"), "w"), indent=1, sort_keys=True)
shutil.copy(src, os.path.join(ROOT, "profiles", "parity_r03.json"))
print(f"{len(rec)} cases")
