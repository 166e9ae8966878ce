"""Per-kernel summary of a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats
writes <name>_results.db on this image).  usage: python profiles/summarize_rocpd.py x.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                 "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':84s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>5s} vgpr agpr lds")
for r in rows:
    print(f"{r[0][:84]:84s} {r[1]:6d} {r[2]:9.2f} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100*r[2]/tot:5.1f} "
          f"{r[6]} {r[7]} {r[8]}")
