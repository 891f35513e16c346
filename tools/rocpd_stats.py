"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd SQLite database --
the same table `rocprofv3 --stats` prints.  usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    q = (f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         f"from kernels group by {name} order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{100.0*r[2]/tot:.2f},{r[4]},{r[5]}")
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
