"""What of a step does not scale with the frames (VERDICT r5 task 7b): two one-step kernel traces of the same build on the same box, 8 and
4 frames per step (tools/quick_trace.sh b8 / b4 --batch-per-gpu 4); per kernel name t8, t4 and the part that would remain at zero frames,
fixed = 2 t4 - t8 (a kernel whose time is proportional to the frames has fixed = 0; one that does not depend on them fixed = t).
usage: python tools/scaling_split.py <step_trace_8frames.csv> <step_trace_4frames.csv> [out.csv]"""
import collections, csv, sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        n = r["name"].split("(")[0].replace("void ", "")
        n = n.split("<")[0]
        agg[n][0] += 1
        agg[n][1] += float(r["dur_us"])
    return agg


a8, a4 = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in sorted(set(a8) | set(a4)):
    n8, t8 = a8.get(k, [0, 0.0])
    n4, t4 = a4.get(k, [0, 0.0])
    rows.append((k, n8, t8, n4, t4, 2 * t4 - t8))
rows.sort(key=lambda r: -r[5])
T8, T4 = sum(r[2] for r in rows), sum(r[4] for r in rows)
print(f"kernel time per step: 8 frames {T8:.0f} us in {sum(r[1] for r in rows)} launches, 4 frames {T4:.0f} us in {sum(r[3] for r in rows)}; "
      f"fixed part 2 t4 - t8 = {2 * T4 - T8:.0f} us ({100 * (2 * T4 - T8) / T4:.0f} % of the 4-frame step's kernel time)")
print(f"{'kernel':42s} {'n8':>4s} {'t8 us':>8s} {'n4':>4s} {'t4 us':>8s} {'fixed us':>9s}")
for r in rows[:40]:
    print(f"{r[0][:42]:42s} {r[1]:4d} {r[2]:8.1f} {r[3]:4d} {r[4]:8.1f} {r[5]:9.1f}")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_8_frames", "us_8_frames", "launches_4_frames", "us_4_frames", "fixed_us_2t4_minus_t8"])
        for r in rows:
            w.writerow([r[0], r[1], round(r[2], 1), r[3], round(r[4], 1), round(r[5], 1)])
