import os, sys, time, importlib.util
sys.argv = ['bench.py']
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print('os.cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'usable', b.usable_cores())
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
from synthetic import recipes as synth
t = time.time(); r = b.cpu_baseline(synth.CONFIGS['llava15_7b'], 1087, 128, 2); print({k: r[k] for k in ('value', 'dtype', 'cores', 'by_dtype')}, 'wall', time.time() - t)
