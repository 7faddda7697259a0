"""Fold rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs of the same command) into
profiles/<name>.json: average HBM bytes per launch of every fz:: kernel.

  python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> "<command that was profiled>"

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half the bytes of wide coalesced reads, WRITE_SIZE
is exact -> hbm_bytes_corrected = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KB)."""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "fz::" in r["Kernel_Name"]:
                per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])   # summed over XCDs / instances
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for k, v in per_dispatch.items():
            acc[names[k]].append(v)
    return acc


def main():
    fdir, wdir, out, cmd = sys.argv[1:5]
    F, W = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `%s`. Units: KB per launch (average over the "
                    "launches of the run). Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports exactly half the bytes of wide coalesced "
                    "reads on gfx950: hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024. Narrow accesses (4-B gathers/stores of the "
                    "sort kernels, 8-B fp64 accesses) are uncalibrated: treat those rows as indicative." % cmd}
    for k in sorted(set(F) | set(W)):
        f = sum(F[k]) / len(F[k]) if F.get(k) else 0.0
        w = sum(W[k]) / len(W[k]) if W.get(k) else 0.0
        name = k[5:] if k.startswith("void ") else k
        res[name] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": max(len(F.get(k, [])), len(W.get(k, []))),
                     "hbm_bytes_corrected": (2 * f + w) * 1024}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_corrected"] / 1e6, 1) for k, v in res.items() if k != "_note"}, indent=0))


if __name__ == "__main__":
    main()
