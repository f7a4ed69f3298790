"""Kernel averages out of a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME writes DIR/NAME_results.db
when no --output-format is given): python scripts/rocprof_top.py DIR/NAME_results.db [rows] [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
flt = sys.argv[3] if len(sys.argv) > 3 else ""
q = "select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0 from kernels group by name order by sum(end-start) desc"
n = 0
for name, cnt, avg, mn, mx in db.execute(q):
    if flt and flt not in name:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{short:60s} n={cnt:5d} avg={avg:8.1f} min={mn:8.1f} max={mx:8.1f} us")
    n += 1
    if n >= rows:
        break
