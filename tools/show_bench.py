#!/usr/bin/env python3
"""Pretty-prints the interesting part of a bench.py JSON line (stdin or file)."""
import json
import sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for line in src:
    if line.startswith("{"):
        j = json.loads(line)
        r = j["roofline"]
        print(f"value={j['value']} {j['unit']} ms/step={j['ms_per_step']} dom={r['kernel']} frac={r['frac']} pipeline_frac={r['pipeline_frac_hbm']}")
        print("  stages:", " ".join(f"{k}={v}" for k, v in r["stage_ms_per_step"].items()))
        if "cpu_baseline" in j:
            print("  cpu:", j["cpu_baseline"]["value"], j["cpu_baseline"]["unit"])
