"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table
(count, total/avg/min/max ms, % of GPU kernel time) — the `--stats` view, as text."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>11s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"{n[:90]:90s} {c:7d} {s/1e6:11.3f} {a/1e6:10.4f} {mn/1e6:10.4f} {mx/1e6:10.4f} {100*s/total:6.2f}")
    lines.append(f"{'TOTAL kernel time':90s} {sum(r[1] for r in rows):7d} {total/1e6:11.3f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
