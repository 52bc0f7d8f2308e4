"""Longest HIP API calls of a rocprofv3 --hip-trace run (which call a one-off 40-90 ms host stall sits in).
usage: python scripts/long_hip_calls.py <dir with *_hip_api_trace.csv> [min_ms]"""
import csv, glob, sys
d = sys.argv[1]; min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    long = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r) for r in rows]
    long.sort(key=lambda x: -x[0])
    print(f, len(rows), "calls")
    for dur, r in long[:40]:
        if dur < min_ms * 1e6:
            break
        print(f"  {dur / 1e6:9.3f} ms  at {(int(r['Start_Timestamp']) - t0) / 1e9:8.3f} s  {r['Function']}")
