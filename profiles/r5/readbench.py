import os, time, threading, numpy as np
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
d = "/dev/shm/rb"; os.makedirs(d, exist_ok=True)
n = 64
a = np.random.rand(480000).astype(np.float32)
for i in range(n): a.tofile("%s/%d.bin" % (d, i))
bufs = [bytearray(1920000) for _ in range(16)]
def work(ids):
    for i in ids:
        with open("%s/%d.bin" % (d, i), "rb", buffering=0) as f: f.readinto(bufs[i % 16])
for nt in (1, 2, 3, 4, 8):
    t0 = time.perf_counter()
    for rep in range(4):
        th = [threading.Thread(target=work, args=(range(k, n, nt),)) for k in range(nt)]
        [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("threads %d: %.1f us per file, %.1f GB/s" % (nt, 1e6 * dt / (4 * n), 4 * n * 1.92e6 / dt / 1e9))
import shutil; shutil.rmtree(d)
