#!/usr/bin/env python3
"""one line per case of tools/byedge_ab.py's output (stdin): mean launch times of both forms"""
import json
import sys

for line in sys.stdin:
    d = json.loads(line)
    keep = {k: round(v[0], 4) for k, v in d.items() if k.endswith("_ms")}
    keep.update({k: v for k, v in d.items() if k.startswith("equal")})
    print(d["kind"], keep)
