"""Full kernel names (+ calls, mean us, vgpr, agpr, lds) from a rocprofv3 rocpd database."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
for name, n, avg, vg, ag, lds in cur.execute(
        "select name, count(*), avg(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels "
        "group by name order by 3 desc").fetchall():
    if n >= 4 and avg > 50:
        print(f"{n:4d} x {avg:9.1f} us  vgpr {vg} agpr {ag} lds {lds}  {name}")
