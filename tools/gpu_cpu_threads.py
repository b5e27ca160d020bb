"""How the CPU checker's rollout (bench.py's cpu_baseline leg) scales with OpenMP threads on the GPU box's host:
cgroup limits, affinity, and the time of one N=1024 humanoidrun rollout per thread count (each in its own process:
OMP_NUM_THREADS is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = r'''
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/model-based-diffusion_amd")
from oracle.oracle import Oracle
from mbd_hip.model import Model
b = Oracle("f32_omp")
m = Model.from_json(open(sys.argv[1] + "/model-based-diffusion_amd/assets/compiled/humanoidrun.json").read())
ms = m.to_struct()
s = b.forward(ms, m.init_q, np.zeros(m.qd_size(), np.float32))
Y = np.zeros((50, 17), np.float32)
y = b.sample(b.prng_key(3), 1, 1024, 50, 17, 0, 1024, 0.5, Y)
b.rollout(ms, s, y)
ts, tr = [], []
for _ in range(5):
    t = time.perf_counter(); b.sample(b.prng_key(3), 1, 1024, 50, 17, 0, 1024, 0.5, Y); ts.append(time.perf_counter() - t)
    t = time.perf_counter(); b.rollout(ms, s, y); tr.append(time.perf_counter() - t)
print(f"sample {1e3 * min(ts):8.2f} ms   rollout {1e3 * min(tr):8.2f} ms (median {1e3 * sorted(tr)[2]:8.2f})")
'''
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
os.system("lscpu | grep -i 'model name\\|socket\\|thread\\|core(s)\\|numa node(s)'")
for n in (8, 16, 32, 64, 128, 256):
    for extra in ({}, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"}, {"OMP_WAIT_POLICY": "active"}):
        env = dict(os.environ, OMP_NUM_THREADS=str(n), **extra)
        r = subprocess.run([sys.executable, "-c", ONE, ROOT], env=env, capture_output=True, text=True)
        print(f"threads {n:4d} {extra}: {r.stdout.strip()} {r.stderr.strip()[-200:]}")
