"""Summarise rocprofv3 rocpd databases (ROCm 7.2 writes SQLite, not CSV).

    python tools/rocpd_summary.py stats  <kt_results.db>  [out.csv]    # per-kernel calls / total / avg / min / max (ns), % of kernel time
    python tools/rocpd_summary.py pmc    <p_results.db>   <name-substr> # per-dispatch counter values of the kernels whose name contains substr
"""
import csv, re, sqlite3, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def stats(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    table = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")]
    for n, k, t, a, mn, mx in rows:
        table.append((short(n), k, int(t), round(a, 1), int(mn), int(mx), round(100.0 * t / tot, 3)))
    if out:
        with open(out, "w", newline="") as f:
            csv.writer(f).writerows(table)
    for r in table[:28]:
        print(",".join(str(v) for v in r))
    print(f"# {len(rows)} kernels, {sum(r[1] for r in rows)} dispatches, {tot / 1e6:.2f} ms of kernel time")


def pmc(db, sub):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, value, end-start from counters_collection where kernel_name like ? order by id", (f"%{sub}%",)).fetchall()
    for n, cn, v, d in rows:
        print(f"{short(n)[:70]:70s} {cn} = {v:.1f}   ({d} ns)")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        pmc(sys.argv[2], sys.argv[3])
