#!/usr/bin/env python3
"""HBM traffic of the dominant kernel per bench step, from the two PMC pass summaries of tools/pmc_passes.sh:

    hbm_traffic.py <tag>_pmc_FETCH_SIZE_summary.json <tag>_pmc_WRITE_SIZE_summary.json [<bench line json>] > profiles/rNN_hbm_traffic.json

One pass = ONE pack of the library in the profiled process (bench.py --steps 1 --warmup 0 --no-cli --no-pe --no-cpu-baseline).  FETCH_SIZE
counts 128-byte requests as 64 bytes on gfx950 (MI355X_MICROARCH.md, HBM section): doubled here, as pmc_summary.py does.
"""
import json, sys

fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
bench = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
coders = [k for k in fetch if k.startswith("fs_encode_streams")]
fb = sum(fetch[k]["FETCH_SIZE"]["sum"] for k in coders) * 1024.0 * 2.0
wb = sum(write[k]["WRITE_SIZE"]["sum"] for k in coders if k in write) * 1024.0
launches = sum(fetch[k]["FETCH_SIZE"]["launches"] for k in coders)
out = {"kernel": " + ".join(sorted(coders)) + " (forms of the same coder)",
       "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-pe (one pass per counter: tools/pmc_passes.sh; one pack of the library per pass)",
       "launches_in_pass": launches, "fetch_bytes_per_step_corrected_x2": fb, "write_bytes_per_step": wb, "hbm_bytes_per_step": fb + wb,
       "other_kernels_bytes_per_step": {k: {"fetch_x2": fetch[k]["FETCH_SIZE"]["sum"] * 2048.0, "write": write.get(k, {}).get("WRITE_SIZE", {}).get("sum", 0.0) * 1024.0} for k in fetch if k not in coders}}
if bench:
    alg = bench["roofline"]["algorithmic_bytes_per_launch"] * bench["roofline"]["launches"] / bench["steps"]
    out["algorithmic_bytes_per_step"] = alg
    out["traffic_over_algorithmic"] = (fb + wb) / alg
print(json.dumps(out, indent=1))
