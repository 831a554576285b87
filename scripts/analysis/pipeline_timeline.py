"""Timeline of the two-frames-in-flight pipeline from a rocprofv3 --kernel-trace csv: for a few steady-state steps, when each kernel of the cull
chain runs relative to the shade launches (us).  usage: pipeline_timeline.py <kernel_trace.csv> [first shade launch to show] [launches]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
shades = [i for i, r in enumerate(rows) if r[2].startswith("k2_shade")]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(shades) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t0 = rows[shades[first]][0]
t1 = rows[shades[first + n]][0]
for s, e, name in rows:
    if s >= t0 - 1000 and s < t1:
        print(f"{(s - t0) / 1e3:9.2f} .. {(e - t0) / 1e3:9.2f}  ({(e - s) / 1e3:7.2f})  {name}")
per = [(rows[shades[i + 1]][0] - rows[shades[i]][0]) / 1e3 for i in range(first, first + 20) if i + 1 < len(shades)]
print("shade start-to-start periods (us):", [round(x, 1) for x in per])
