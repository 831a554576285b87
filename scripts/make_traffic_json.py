"""profiles/<round>/traffic.json from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of the same command.
Usage: python scripts/make_traffic_json.py <dir with fetch pass> <dir with write pass> <out.json> <config>"""
import csv, glob, json, sys, collections


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --frames-in-flight 1   (and a second, separate pass with --pmc WRITE_SIZE)",
    "correction": "MI355X_MICROARCH.md, HBM section: counters are in KB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (check: the depth pass reads the 33.2 MB depth image)",
    "caveat": "the x2 on FETCH_SIZE is calibrated for 16-byte-per-lane STREAMING loads only (MI355X_MICROARCH.md, HBM section: other access widths and WRITE_SIZE are uncalibrated); for kernels dominated by 4-16-byte GATHERS -- k1_tile_cull's light records, the K3 texel look-ups of k2_shade_csm* -- the figure is an upper bound (a 64-byte request tallied once would be counted twice), and Infinity-Cache hits are included: it bounds fabric requests, not DRAM bytes",
    "config": sys.argv[4],
    "kernels": {},
}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("k"):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
