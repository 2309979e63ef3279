"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/pmc_fetch_write_summary.json: average KB per launch per kernel. bench.py reads that file
for roofline.traffic.  Usage: python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = r["Dispatch_Id"]
            per_dispatch[key] += float(r["Counter_Value"])  # summed over XCDs / instances
            names[key] = r["Kernel_Name"]
        for key, v in per_dispatch.items():
            name = re.sub(r"\(.*", "", names[key]).strip()
            acc[name][0] += 1
            acc[name][1] += v
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(fetch) | set(write)):
    n = fetch.get(name, write.get(name))[0]
    out[name] = {"launches": n,
                 "FETCH_SIZE_KB_avg": fetch[name][1] / max(fetch[name][0], 1) if name in fetch else 0.0,
                 "WRITE_SIZE_KB_avg": write[name][1] / max(write[name][0], 1) if name in write else 0.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for name, r in sorted(out.items(), key=lambda kv: -kv[1]["FETCH_SIZE_KB_avg"] * kv[1]["launches"])[:14]:
    print(f"{name[:60]:60s} n={r['launches']:5d} fetch {r['FETCH_SIZE_KB_avg'] / 1e3:10.1f} MB write {r['WRITE_SIZE_KB_avg'] / 1e3:10.1f} MB")
