"""What the GPU box's host gives a process: logical CPUs, cgroup CPU quota, and how a CPU-bound loop scales over threads."""
import os, time, threading, ctypes, sys
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node' | head -12")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle, datagen
bs = 4 << 20
blk = [datagen.block(i, bs) for i in range(10)]
for jobs in (1, 8, 16, 32, 64, 128):
    nb = max(10, jobs * 2)
    s = np.concatenate([blk[i % 10] for i in range(nb)])
    t = time.time(); k = oracle.compress("BWT+RANK+ZRLT", "ANS0", bs, s, jobs=jobs); t1 = time.time()
    print("oracle encode jobs=%d: %.1f MB/s total, %.2f MB/s per thread" % (jobs, len(s) / (t1 - t) / 1e6, len(s) / (t1 - t) / 1e6 / jobs), flush=True)
