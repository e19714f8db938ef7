"""Sum of every rocprofv3 --pmc counter over ALL dispatches of a run (results .db, run ON the GPU box).
usage: python tools/pmc_total.py <results.db>   ->  one line per counter: name total dispatches"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
for name, tot, n in c.execute("select counter_name, sum(value), count(*) from counters_collection group by counter_name"):
    print(name, f"{tot:.6g}", n)
