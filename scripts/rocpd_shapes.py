"""Per-(kernel, grid) breakdown of a rocprofv3 rocpd kernel trace: where the time goes, wide vs narrow dispatches."""
import collections
import sqlite3
import sys


def main(path, pairs, top=45):
    db = sqlite3.connect(path)
    rows = db.execute("select name,grid_x,grid_y,grid_z,workgroup_x,workgroup_y,duration from kernels order by start").fetchall()
    tot = sum(r[6] for r in rows)
    g = collections.defaultdict(lambda: [0, 0])
    wide = narrow = 0
    for n, gx, gy, gz, wx, wy, d in rows:
        wgs = (gx // wx) * (gy // max(wy, 1)) * gz
        nm = n.split('(')[0].replace('void ', '').replace('geotr::', '')[:44]
        k = (nm, gx // wx, gy // max(wy, 1), gz)
        g[k][0] += 1
        g[k][1] += d
        if wgs >= 256:
            wide += d
        else:
            narrow += d
    print(f'{len(rows)} dispatches ({len(rows) / pairs:.0f}/pair), kernel time {tot / 1e6 / pairs:.3f} ms/pair: '
          f'wide(>=256 WGs) {wide / 1e6 / pairs:.3f}, narrow {narrow / 1e6 / pairs:.3f}')
    for k, (c, d) in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{k[0]:46s} grid {k[1]:5d}x{k[2]:5d}x{k[3]:3d}  n/pair {c / pairs:5.1f}  avg {d / c / 1e3:7.1f} us  '
              f'{d / 1e3 / pairs:7.1f} us/pair  {100 * d / tot:4.1f}%')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
