"""Aggregate rocprofv3 --pmc counter_collection CSVs (one per pass) into a small per-kernel JSON + markdown table."""
import collections, csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "moss" not in k:
            continue
        short = k.split("(anonymous namespace)::")[-1].split("(")[0]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, c in res.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
        # (16 B/lane) reads -> doubled.  Our reads are 16-byte record gathers and dwordx4 streams, i.e. that access class.
        c["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in res.items()}, indent=1))
