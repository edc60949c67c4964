#!/usr/bin/env python3
"""How long the CPU oracle jobs of tests/test_gpu_fuzz.py take when they run ahead in worker processes (no GPU work):
per job the time at which its result is ready; the box's CPU count and quota.   python scripts/oracle_ahead_time.py [workers]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases            # noqa: E402
import test_gpu_fuzz as T    # noqa: E402

if __name__ == "__main__":
    try:
        print("cpus", os.cpu_count(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError:
        pass
    plan = []
    T._geometry_case(None, plan)
    T._chromosome_cases(None, plan)
    T._genome_cases(None, plan)
    order = sorted(plan, key=lambda j: -(j[0] * max(j[1], 400)))
    t0 = time.time()
    fuzz_cases.start_ahead(order, workers=int(sys.argv[1]) if len(sys.argv) > 1 else None)
    print("submitted %d jobs in %.1f s" % (len(order), time.time() - t0), flush=True)
    futs = dict(fuzz_cases._AHEAD)
    pending = set(futs)
    while pending:
        for j in list(pending):
            if futs[j].done():
                pending.discard(j)
                print("n %5d dpx %4d ready at %6.1f s" % (j[0], j[1], time.time() - t0), flush=True)
        time.sleep(0.5)
    fuzz_cases.stop_ahead()
