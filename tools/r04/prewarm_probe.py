"""Does the chip's clock state decide the 20-step figure?  128 pairs of 16 kbp, 5 warm-up + 20 timed passes (the driver's arguments), after an idle
period and a pre-warm of the given length (untimed passes of the same batch).  MI355X_MICROARCH.md, 'DVFS give-back': steady clocks need ~2 s of load."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import csa_amd
from csa_amd.synth import config4_tasks
csa_amd.init(device=0)
tasks = config4_tasks(0, 128, 16384)
cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)


def trial(prewarm_s, idle_s, steps=20):
    pb = csa_amd.PairBatch(tasks)
    pb.sync()
    time.sleep(idle_s)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm_s:
        for _ in range(16):
            pb.run()
        pb.sync()
    for _ in range(5):
        pb.run()
    pb.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        pb.run()
    pb.sync()
    dt = time.perf_counter() - t0
    pb.close()
    return cells * steps / dt / 1e9


for idle in (0.0, 10.0):
    for pre in (0.0, 0.05, 0.5, 2.0):
        vals = [trial(pre, idle) for _ in range(3)]
        print("idle %4.0f s  prewarm %.2f s: %s" % (idle, pre, " ".join("%.0f" % v for v in vals)), flush=True)
print("sustained, 1200 steps: %.0f" % trial(0.0, 0.0, steps=1200), flush=True)
