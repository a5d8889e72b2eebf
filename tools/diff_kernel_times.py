"""Per-kernel time per step of two rocprofv3 --kernel-trace runs of tools/phase_times.py (sqlite output), and their difference:
which kernels pay for a change of mode.  usage: diff_kernel_times.py A_results.db B_results.db [steps]"""
import collections
import sqlite3
import sys


def load(path, steps):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if r[0].startswith("k_adam")]
    lo, hi = idx[-steps - 1], idx[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in rows[lo + 1:hi + 1]:
        key = n.split("(")[0][:60]
        agg[key][0] += 1
        agg[key][1] += (e - s) / 1e3
    return {k: (v[0] / steps, v[1] / steps) for k, v in agg.items()}


steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
a, b = load(sys.argv[1], steps), load(sys.argv[2], steps)
print("kernel us/step  A %.1f  B %.1f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
d = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k) for k in set(a) | set(b)), reverse=True)
for diff, k in d[:30] + d[-8:]:
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    print(f"{diff:8.1f}  {k[:52]:52s} A {x[0]:5.1f}x {x[1]:7.1f}   B {y[0]:5.1f}x {y[1]:7.1f}")
