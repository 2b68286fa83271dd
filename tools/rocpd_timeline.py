#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 rocpd database: start (us, relative), duration, queue, kernel, grid.
usage: tools/rocpd_timeline.py <results.db> [N=80] [name-regex] [skip_last=0]"""
import re
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.{q} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k "
                 "on d.kernel_id = k.id order by d.start").fetchall()
skip_last = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = [r for r in rows if not pat or pat.search(r[0])]
rows = rows[:len(rows) - skip_last][-n:]
t0 = rows[0][1]
prev_end = t0
for name, st, en, gx, qid in rows:
    short = re.sub(r"_ZN2wx\d+(k_[a-z_]+)I?(.*?)E+v.*", r"\1<\2>", name.split("(")[0])[:44]
    print(f"{(st - t0) / 1e3:10.1f} us  +{(st - prev_end) / 1e3:7.1f} gap  {(en - st) / 1e3:8.1f} us  q{qid}  {short:44s} grid {gx}")
    prev_end = max(prev_end, en)
