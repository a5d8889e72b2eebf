"""Compact one training step out of a rocprofv3 kernel-trace CSV: per dispatch (start offset us, duration us,
grid, kernel name) for the second-to-last k_adam-delimited step (the last step of a bench run
prefetches no next batch, so its trace would lack the geometry-plan kernels of the side stream).  Usage: trace_step.py <kernel_trace.csv> <out.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam")]
a, b = adam[-3] + 1, adam[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,dur_us,grid,wg,name,queue\n")
    for r in rows[a:b]:
        f.write("%.1f,%.1f,%s,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                        r["Grid_Size_X"], r["Workgroup_Size_X"], r["Kernel_Name"][:70].replace(",", ";") + "," + str(r.get("Queue_Id", ""))))
print("step dispatches:", b - a, "span ms:", (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e6)
