"""Per-kernel PMC summary from a rocprofv3 rocpd database: for every (kernel, counter) the number of
dispatches, mean counter value and mean duration.  usage: python tools/rocpd_pmc.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    q = ("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
         "group by name, counter_name order by 5*3 desc")
    lines = ["Name,Counter,Dispatches,MeanValue,MeanDurationNs"]
    for r in cur.execute(q):
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]:.1f}")
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
