"""Kernel timeline (name, start offset, duration, gap to the previous kernel's end; microseconds) of the LAST n dispatches in a
rocprofv3 rocpd SQLite database.  usage: python tools/rocpd_timeline.py <results.db> <out.csv> [n]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    rows = list(cur.execute(f"select {name}, start, end from kernels order by start"))[-n:]
    t0, prev = rows[0][1], rows[0][1]
    with open(sys.argv[2], "w") as f:
        f.write("name,start_us,dur_us,gap_us\n")
        for nm, s, e in rows:
            short = nm.split("(")[0].split("::")[-1][:40]
            f.write(f"{short},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{(s - prev) / 1e3:.1f}\n")
            prev = e


if __name__ == "__main__":
    main()
