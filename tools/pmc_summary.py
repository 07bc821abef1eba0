#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into profiles/<round>/<name>_pmc.json.

usage: pmc_summary.py <kernel substring> <fetch csv> <write csv> <out json>
HBM-side traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for
gfx950: FETCH_SIZE (KB) reports half of a wide (16 B/lane) coalesced read stream -> doubled;
WRITE_SIZE (KB) is exact for 16-B streaming stores. Infinity-Cache hits are included in both."""
import csv
import json
import sys


def avg(path, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    kernel, fcsv, wcsv, out = sys.argv[1:5]
    f, nf = avg(fcsv, "FETCH_SIZE", kernel)
    w, nw = avg(wcsv, "WRITE_SIZE", kernel)
    res = {"kernel": kernel, "launches_sampled": [nf, nw], "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
           "fetch_bytes_corrected": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
           "traffic_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE x1; units KB -> bytes x1024"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
