#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x) rocpd SQLite result: per-kernel stats (== `--stats`) and PMC sums.

usage: tools/rocpd_summary.py <results.db> [--skip N | --last N]   ->  markdown table on stdout
`--skip N` ignores the first N dispatches of every kernel (warm-up launches); `--last N` keeps only the last N dispatches of every
kernel (the timed region of a bench run that tuned the placement of its planes first: the probes ran the same kernels elsewhere).
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    c = sqlite3.connect(db)
    rows = c.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.workgroup_size_y, "
                     "k.arch_vgpr_count, k.sgpr_count, d.group_segment_size, d.id, d.event_id "
                     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k on d.kernel_id = k.id order by d.start").fetchall()
    per = defaultdict(list)
    meta = {}
    for name, st, en, gx, gy, wx, wy, vg, sg, lds, did, eid in rows:
        short = name.split("(")[0]
        per[short].append((en - st, eid))
        meta[short] = (gx, gy, wx, wy, vg, sg, lds)
    pmc = defaultdict(lambda: defaultdict(list))
    try:
        for eid, pname, val in c.execute("select e.event_id, p.name, e.value from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id"):
            pmc[eid][pname].append(val)
    except sqlite3.Error:
        pass
    if last:
        per = {k: v[-last:] for k, v in per.items()}
        skip = 0
    total = sum(sum(d for d, _ in v[skip:]) for v in per.values())
    print(f"source: {db}  (dispatches: {len(rows)}, " + (f"last {last} per kernel" if last else f"skipped first {skip} per kernel") + ")\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | vgpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, v in sorted(per.items(), key=lambda kv: -sum(d for d, _ in kv[1][skip:])):
        d = [x for x, _ in v[skip:]]
        if not d:
            continue
        gx, gy, wx, wy, vg, sg, lds = meta[name]
        print(f"| {name} | {len(d)} | {sum(d) / 1e6:.3f} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {max(d) / 1e3:.1f} | "
              f"{100.0 * sum(d) / max(total, 1):.1f} | {gx}x{gy} | {wx}x{wy} | {vg} | {sg} | {lds} |")
    if pmc:
        names = sorted({n for e in pmc.values() for n in e})
        print("\nPMC (average per dispatch; summed over XCDs/instances as rocprofv3 reports them)\n")
        print("| kernel | " + " | ".join(names) + " |")
        print("|---|" + "---|" * len(names))
        for name, v in per.items():
            acc = defaultdict(list)
            for _, eid in v[skip:]:
                for n, vals in pmc.get(eid, {}).items():
                    acc[n].append(sum(vals))
            if acc:
                print(f"| {name} | " + " | ".join(f"{sum(acc[n]) / max(len(acc[n]), 1):.4g}" if acc[n] else "-" for n in names) + " |")


if __name__ == "__main__":
    main()
