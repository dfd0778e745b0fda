"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, skip_first=0, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        short = name.split('(')[0]
        d = stats.setdefault(short, [0, 0.0, 1e30, 0.0])
        dur = (e - s)
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    total = sum(v[1] for v in stats.values())
    print(f'# {path}: {len(rows)} dispatches, total kernel time {total / 1e6:.3f} ms')
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for name, (n, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'| {name[:90]} | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} |')


if __name__ == '__main__':
    main(sys.argv[1])
