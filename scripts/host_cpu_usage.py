"""CPU time the bench process burns while the GPU works (do the lane threads sleep or spin in their waits?).
usage: python scripts/host_cpu_usage.py [bench.py args]"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t0 = time.time()
p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-fp32-mode'] + sys.argv[1:],
                   capture_output=True, text=True)
wall = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
line = json.loads(p.stdout.strip().splitlines()[-1])
region = line['steps'] * line['ms_per_step'] / 1e3
print(json.dumps({'pairs_per_s': line['value'], 'timed_region_s': round(region, 2), 'process_wall_s': round(wall, 2),
                  'user_s': round(ru.ru_utime, 2), 'sys_s': round(ru.ru_stime, 2),
                  'cpus_busy_over_process': round((ru.ru_utime + ru.ru_stime) / wall, 2),
                  'voluntary_ctx_switches': ru.ru_nvcsw, 'involuntary_ctx_switches': ru.ru_nivcsw}))
