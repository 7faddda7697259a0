#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes into one JSON: per kernel, the per-launch average of every counter (summed over the XCD
instances of a dispatch).  Usage: python tools/pmc_summary.py <out.json> <pass dir> [<pass dir> ...] [--match substring]"""
import collections, csv, glob, json, os, sys


def main():
    args = sys.argv[1:]
    match = None
    if "--match" in args:
        i = args.index("--match"); match = args[i + 1]; del args[i: i + 2]
    out, dirs = args[0], args[1:]
    res = collections.defaultdict(dict)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
            meta = {}
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if match and match not in k:
                    continue
                per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                meta[k] = {x: r.get(x) for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if x in r}
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for (k, _), cs in per.items():
                for c, v in cs.items():
                    agg[k][c].append(v)
            for k, cs in agg.items():
                res[k].setdefault("launch", meta.get(k, {}))
                for c, vs in cs.items():
                    res[k][c] = sum(vs) / len(vs)
                res[k]["launches_seen"] = max(res[k].get("launches_seen", 0), max(len(v) for v in cs.values()))
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k[:90], {c: (round(x, 1) if isinstance(x, float) else x) for c, x in v.items() if c != "launch"})


if __name__ == "__main__":
    main()
