"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (run ON the GPU box; keeps outputs small).

    python tools/pmc_summary.py OUT.json name1=file1.csv name2=file2.csv ...

For every kernel name: launches and the per-launch mean of every counter.  FETCH_SIZE / WRITE_SIZE are
reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
streams (MI355X_MICROARCH.md, HBM section), so `hbm_read_bytes = 2 * FETCH_SIZE * 1024` is added.
"""
import collections
import csv
import json
import re
import sys

out_path = sys.argv[1]
result = {}
for spec in sys.argv[2:]:
    tag, path = spec.split("=", 1)
    disp = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "aurora::" not in name:
                continue
            short = re.sub(r"\(.*", "", name.replace("void ", "").replace("aurora::(anonymous namespace)::", ""))
            d = disp.setdefault(r["Dispatch_Id"], {"kernel": short})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    for d in disp.values():
        a = agg.setdefault(d["kernel"], collections.Counter())
        a["launches"] += 1
        for k, v in d.items():
            if k != "kernel":
                a[k] += v
    for k, a in agg.items():
        n = a.pop("launches")
        e = result.setdefault(k, {})
        e.setdefault("launches", {})[tag] = n
        for c, v in a.items():
            e[c + "_per_launch"] = v / n
            if c == "FETCH_SIZE":
                e["hbm_read_bytes_per_launch"] = 2.0 * v * 1024 / n
            if c == "WRITE_SIZE":
                e["hbm_write_bytes_per_launch"] = v * 1024 / n
json.dump(result, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "bytes" in kk or kk == "launches"} for k, v in result.items()}, indent=1)[:3000])
