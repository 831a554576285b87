"""Summarise rocprofv3 counter_collection CSVs: per kernel, mean of each counter, and per-wave ratios."""
import csv, glob, collections, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "shade"
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/pmc_*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name'].split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in tot.items():
    if pat not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = m.get('SQ_WAVES', 1)
    print(k, "waves", int(w))
    for c in sorted(m):
        print(f"   {c:26s} {m[c]:16.0f}   per wave {m[c] / w:10.1f}")
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_INSTS_VALU' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        print(f"   kernel cycles/XCD {cyc:.0f}; VALU issue share (instr*4 / (1024 SIMD * cycles)) = {m['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.2f}")
