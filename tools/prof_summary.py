"""Turn rocprofv3 output (rocpd .db or csv dir) into a small text summary for profiles/.
usage: python tools/prof_summary.py <dir> <out.txt> [title]"""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def from_db(path, out):
    con = sqlite3.connect(path)
    cur = con.cursor()
    out.write("kernel,calls,total_us,avg_us,percent\n")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        out.write(f'"{name}",{calls},{total:.1f},{avg:.2f},{pct:.2f}\n')
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" in tabs:
        out.write("\n# PMC (sum over dispatches / mean per dispatch)\nkernel,counter,dispatches,mean_per_dispatch\n")
        q = ("select k.name, c.counter_name, count(*), avg(c.value) from counters_collection c "
             "join kernels k on k.dispatch_id = c.dispatch_id group by k.name, c.counter_name")
        try:
            for r in cur.execute(q):
                out.write(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.6g}\n')
        except sqlite3.Error as e:
            out.write(f"# counters query failed: {e}; tables: {tabs}\n")


def main():
    d, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    with open(dst, "w") as out:
        from pmc_summary import head_stamp  # same directory

        out.write(f"# {title}\n# measured at commit {head_stamp()}\n")
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        for p in dbs:
            out.write(f"# source: {os.path.basename(p)}\n")
            from_db(p, out)
    print(open(dst).read())


if __name__ == "__main__":
    main()
