"""What exactly is wrong in a failing wave of scripts/packed_fp32_mfma_hazard.hip (VICTIM=indices SHOW=1)?  (profiles/r03_concurrency_hazard.md 4e)

Input: the program's output with its "wrong: pair (i, j) = wave w lane l, component c: got g, idle-GPU result r" lines.  The program's inputs
are regenerated here (std::mt19937(11) + uniform_real_distribution<float> = numpy's RandomState(11) raw draws / 2^32), every reported
"idle-GPU result" is recomputed on the CPU, and for every failing wave the ONE quantity is solved for that explains all of its wrong
values: the z component of (neighbour 1 - point i) -- and compared with candidates built from the kernel's inputs.

usage: python scripts/hazard_wrong_value_analysis.py profiles/r03_hazard_standalone_runs.txt
"""
import re
import sys

import numpy as np

N, FACTOR = 251, 180.0 / (15.0 * np.pi)


def points():
    draws = np.random.RandomState(11).randint(0, 2 ** 32, size=N * 3, dtype=np.uint64).astype(np.uint32)
    u = draws.astype(np.float32) / np.float32(4294967296.0)
    u = np.where(u >= 1, np.nextafter(np.float32(1), np.float32(0)), u)
    return (np.float32(3.0) * u).astype(np.float32).reshape(N, 3).astype(np.float64)


def angle(r, a):
    return np.arctan2(np.linalg.norm(np.cross(r, a), axis=-1), (r * a).sum(-1)) * FACTOR


def main():
    pts = points()
    waves, victim = {}, None
    for line in open(sys.argv[1]):
        v = re.search(r'"victim": "([^"]+)"', line)
        if v:
            victim = v[1]  # the "wrong:" lines follow the JSON line of their run
        if victim != 'indices only':
            continue
        m = re.search(r'pair \(i (\d+), j (\d+)\) = wave (\d+) lane (\d+), component (\d+): got ([-\d.e]+), idle-GPU result ([-\d.e]+)', line)
        if m:  # the indices-only victim: component = which of the four indices
            waves.setdefault(int(m[3]), []).append((int(m[1]), int(m[2]), int(m[4]), int(m[5]), float(m[6]), float(m[7])))
    print(f'{sum(map(len, waves.values()))} wrong values in {len(waves)} waves; lanes {sorted({o[2] for w in waves.values() for o in w})}; '
          f'components {sorted({o[3] for w in waves.values() for o in w})} (0 = distance, 1..3 = angles to the nearest, 2nd, 3rd neighbour)')
    for w, obs in sorted(waves.items()):
        i = obs[0][0]
        a = np.array([pts[o[1]] - pts[i] for o in obs])
        got, want = np.array([o[4] for o in obs]), np.array([o[5] for o in obs])
        d = ((pts - pts[i]) ** 2).sum(1)
        d[i] = np.inf
        r0, r1, r2 = np.argsort(d)[:3]
        rx, ry, rz = pts[r1] - pts[i]
        recomputed = np.abs(angle(np.array([[rx, ry, rz]]), a) - want).max()
        grid = np.linspace(-8, 8, 16001)  # one unknown: the z component the wave used for (neighbour 1 - point i)
        err = np.array([np.abs(angle(np.array([[rx, ry, z]]), a) - got).max() for z in grid])
        z = grid[err.argmin()]
        for _ in range(40):  # refine
            fine = np.linspace(z - 2e-3, z + 2e-3, 81)
            e = np.array([np.abs(angle(np.array([[rx, ry, t]]), a) - got).max() for t in fine])
            z = fine[e.argmin()]
            if e.min() < 2e-6:
                break
        candidates = {'neighbour1.z (the minuend alone: the subtraction of point_i.z did not happen)': pts[r1][2],
                      'neighbour1.z - point_i.y (op_sel ignored)': pts[r1][2] - pts[i][1], 'neighbour1.z - point_i.x': pts[r1][2] - pts[i][0],
                      'neighbour2.z - point_i.z (the other half of the pair)': pts[r2][2] - pts[i][2], 'neighbour0.z - point_i.z': pts[r0][2] - pts[i][2]}
        name, value = min(candidates.items(), key=lambda kv: abs(kv[1] - z))
        print(f'wave {w:4d} (point i = {i:3d}, {len(obs)} lanes): idle-GPU results recomputed to {recomputed:.1e}; all wrong values are explained to '
              f'{e.min():.1e} by z = {z:.5f} instead of {rz:.5f}  ==  {name} = {value:.5f}')


if __name__ == '__main__':
    main()
