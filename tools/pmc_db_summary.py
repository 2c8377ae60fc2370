"""Summarise rocprofv3 --pmc results (rocpd .db files): per kernel name, mean of each counter over its dispatches.
   python tools/pmc_db_summary.py <dir-with-db-files>...   (test / measurement tooling)"""
import glob
import os
import sqlite3
import sys

for d in sys.argv[1:]:
    for db in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
        print(f"== {os.path.basename(d.rstrip('/'))}")
        for k, cn, n, v in rows:
            if k.startswith("__amd_rocclr"):
                continue
            print(f"{k[:100]:100s} {cn:30s} n={n:4d} mean={v:.6g}")
