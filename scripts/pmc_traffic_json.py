"""profiles/rNN_pmc_traffic.json from a scripts/gpu_prof_pmc.sh summary: mean FETCH_SIZE / WRITE_SIZE (KiB per dispatch)
per kernel, the table bench.py replays in `roofline.traffic` (with its provenance).

    python scripts/pmc_traffic_json.py gpurun_out/prof_cfg2_fp32/summary.txt profiles/r03_pmc_traffic.json [commit] [command]

`_meta` records the commit the profiled library was built from (PTX_COMMIT or argv[3]: the GPU box has no .git) and the
profiled command, so bench.py's `roofline.traffic_source` can say which build the replayed counters belong to.
"""
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
out, sect = {}, None
for line in open(src):
    m = re.match(r"## counters: pmc_(\w+)/", line)
    if m:
        sect = m.group(1)
        continue
    if line.startswith("## "):
        sect = None
    if sect in ("FETCH_SIZE", "WRITE_SIZE") and not line.startswith("kernel") and line.strip():
        m = re.match(r"(.{62})\s+(\d+)\s+([0-9.e+-]+)\s+([0-9.e+-]+)", line)
        if m:
            row = out.setdefault(m.group(1).strip(), {})
            row[sect + "_KiB"] = float(m.group(3))
            row["calls"] = int(m.group(2))           # dispatches behind the mean (bench.py weights template instantiations by it)
out["_meta"] = {"commit": sys.argv[3] if len(sys.argv) > 3 else os.environ.get("PTX_COMMIT"),
                "command": sys.argv[4] if len(sys.argv) > 4 else None, "source": src}
json.dump(out, open(dst, "w"), indent=1)
print(len(out), "kernels ->", dst)
