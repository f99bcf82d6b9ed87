"""Summarise a rocprofv3 rocpd database (kernel trace) into a small text table for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats.txt "<command>"
"""
import sqlite3
import sys

db_path, out_path = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
sources = open(sys.argv[4]).read().splitlines() if len(sys.argv) > 4 else []   # `sha256sum` lines of what was profiled
cur = sqlite3.connect(db_path).cursor()
rows = cur.execute(
    "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
    "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"
).fetchall()
total = sum(r[2] for r in rows)
with open(out_path, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n# command: {cmd}\n# total kernel time {total:.2f} ms\n")
    for line in sources:   # the build these rows describe (tools/profile_round.sh)
        f.write(f"# sha256 {line}\n")
    f.write(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel\n")
    for name, n, ms, avg, mn, mx, vg, ag, lds in rows:
        if ms / total < 0.0005:
            continue
        f.write(f"{ms:10.2f} {100 * ms / total:6.2f} {n:6d} {avg:10.1f} {mn:9.1f} {mx:10.1f} {vg or 0:5d} {ag or 0:5d} {lds or 0:7d}  {name[:110]}\n")
print(open(out_path).read())
