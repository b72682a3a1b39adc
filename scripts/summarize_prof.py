"""Condense rocprofv3 (rocpd SQLite) outputs into per-kernel tables: kernel stats + PMC means.

    python scripts/summarize_prof.py gpurun_out/prof [last_n_dispatches_per_kernel]
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    """Kernel name in <= 60 characters WITHOUT losing template arguments: the conv template's argument list
    <BM, BN, BK, WM, WN, MT, KTAIL, K22, DMA, NSTAGE, F16, X3, KWR, CHAIN, REPI> is written without blanks and with t / f for
    the booleans (the chained / row-major-epilogue variants differ only in its last two entries)."""
    name = name.replace("void ptx::", "").replace("ptx::", "").replace("(ptx::ConvArgs)", "").replace("(ConvArgs)", "")
    name = name.replace("conv_igemm_kernel", "conv_igemm")
    if name.startswith("conv_igemm<"):
        name = name.replace(", ", ",").replace("true", "t").replace("false", "f")
    return name[:60]


for f in sorted(glob.glob(os.path.join(root, "trace*", "*.db"))):
    db = sqlite3.connect(f)
    print("## kernel trace:", os.path.relpath(f, root))
    rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), "
                           "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                           "order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    print("%-62s %6s %11s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "pct", "vgpr", "agpr", "lds"))
    for r in rows[:30]:
        print("%-62s %6d %11.1f %10.2f %10.2f %6.2f %5d %5d %7d" % (short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                 100.0 * r[2] / tot, r[5], r[6], r[7]))

for f in sorted(glob.glob(os.path.join(root, "pmc_*", "*.db"))):
    db = sqlite3.connect(f)
    print("\n## counters:", os.path.relpath(f, root))
    acc = defaultdict(lambda: defaultdict(list))
    for k, c, v, dur in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        acc[short(k)][c].append(v)
        acc[short(k)]["duration_us"].append(dur / 1e3)
    names = sorted({c for v in acc.values() for c in v})
    print("%-62s %6s " % ("kernel (mean per dispatch)", "calls") + " ".join("%20s" % n[:20] for n in names))
    order = sorted(acc.items(), key=lambda kv: -sum(kv[1]["duration_us"]))
    for k, v in order[:24]:
        n = max(len(x) for x in v.values())
        print("%-62s %6d " % (k, n) + " ".join("%20.5g" % (sum(v.get(c, [0.0])) / max(len(v.get(c, [1])), 1)) for c in names))
