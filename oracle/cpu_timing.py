"""TEST INFRASTRUCTURE ONLY -- times oracle/model_oracle.forward on the host CPU in a CHILD process with a fixed torch thread count
(bench.py's cpu_baseline leg).  A child, because a thread count that oversubscribes the host (torch CPU ops of this size thrash far
below a 256-core box's core count) cannot be interrupted from inside; the parent kills it after its time budget.

    python -m oracle.cpu_timing <blob.pt> <threads> <warmup> <reps>     -> one JSON line {"threads", "times_s", "median_s"}
`blob.pt` = torch.save({'sd': state_dict, 'cfg': oracle config dict, 'data': [oracle input dicts]})."""
import json
import sys
import time


def main():
    path, threads, warmup, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    torch.set_num_threads(threads)
    from oracle import model_oracle as mo
    blob = torch.load(path, weights_only=False)
    sd, cfg, datas = blob['sd'], blob['cfg'], blob['data']
    times = []
    for i in range(warmup + reps):
        data = datas[i % len(datas)]
        t0 = time.perf_counter()
        mo.forward(sd, cfg, data)
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
        print(json.dumps({'threads': threads, 'times_s': times, 'median_s': sorted(times)[len(times) // 2] if times else None,
                          'partial': i + 1 < warmup + reps}), flush=True)


if __name__ == '__main__':
    main()
