"""Merge tuned-tile tables dumped by bench.py (PTX_TUNED_OUT) into pretorched-x_amd/tuned_gfx950.json.

    python scripts/merge_tuned.py [--new-only] gpurun_out/tuned_*.json
--new-only: adopt only problems the table on disk does not hold yet (a dump of a full re-tune on another box would
otherwise churn entries that differ by measurement noise).  Only entries a dump changed relative to the table on disk are adopted (later files win among those); entries whose
tile name the current build does not have are dropped."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pretorched_x_amd import engine, _lib        # noqa: E402

lib = _lib.lib()
names = {lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())}
names |= {lib.ptx_conv3d_chain_config_name(i).decode() for i in range(lib.ptx_conv3d_chain_num_configs())}     # "chain:" keys
names |= {"chain", "pair"}                                                                                       # "alt:" keys
names |= {"igemm", "lanes", "program", "launches"} | set(engine.BODY_SHAPES)                                     # "body:" / "lanes:" / "prog:" keys
path = os.path.join(ROOT, "pretorched-x_amd", "tuned_gfx950.json")
table = json.load(open(path))
n0 = len(table)
base = dict(table)
new_only = "--new-only" in sys.argv[1:]
for f in [a for a in sys.argv[1:] if a != "--new-only"]:
    # every dump is the FULL in-memory table of its run (the shipped table + what that run tuned): adopt only the entries
    # a run changed or added, so one run's stale copy of another run's problems cannot clobber them
    for k, v in json.load(open(f)).items():
        if base.get(k) != v and not (new_only and k in base):
            table[k] = v
table = {k: v for k, v in table.items() if v[0] in names}
engine._tuned = {k: (str(v[0]), int(v[1])) for k, v in table.items()}
engine.save_tuned_table(path)
print("%d -> %d entries" % (n0, len(table)))
