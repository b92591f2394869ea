"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (CSV-ish text).
Usage: python tools/rocpd_summary.py <results.db> [top_n] [min_start_frac]
min_start_frac drops the first fraction of the timeline (model build / warm-up)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    if not rows:
        print("no kernels")
        return
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    cut = t0 + frac * (t1 - t0)
    agg = {}
    for n, s, e in rows:
        if s < cut:
            continue
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("# kernels after cut: total GPU time %.3f ms over %d dispatches; window %.1f ms" % (tot / 1e6, sum(a[0] for a in agg.values()), (t1 - cut) / 1e6))
    print("%-90s %8s %12s %10s %10s %10s %6s" % ("name", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-90s %8d %12.1f %10.1f %10.1f %10.1f %6.2f" % (n[:90], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))


if __name__ == "__main__":
    main()
