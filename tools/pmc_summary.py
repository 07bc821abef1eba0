#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into profiles/<round>/<name>_pmc.json.

usage: pmc_summary.py <out json> <kernel substring>[,<kernel substring>...] <fetch csv> <write csv> [<busy csv>]
HBM-side traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for
gfx950: FETCH_SIZE (KB) reports half of a wide (16 B/lane) coalesced read stream -> doubled;
WRITE_SIZE (KB) is exact for 16-B streaming stores. Infinity-Cache hits are included in both.
The optional third pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE + timestamps) gives the MFMA-busy fraction and
the clock the chip held: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; busy = MFMA_BUSY / (clock cycles x 256 CUs x 4 SIMDs)."""
import csv
import json
import sys


def rows(path, kernel):
    return [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]


def avg(rs, counter):
    vals = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def one(kernel, fcsv, wcsv, bcsv):
    f, nf = avg(rows(fcsv, kernel), "FETCH_SIZE")
    w, nw = avg(rows(wcsv, kernel), "WRITE_SIZE")
    res = {"kernel": kernel, "launches_sampled": [nf, nw], "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
           "fetch_bytes_corrected": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
           "traffic_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE x1; units KB -> bytes x1024"}
    if bcsv:
        rb = rows(bcsv, kernel)
        mf, _ = avg(rb, "SQ_VALU_MFMA_BUSY_CYCLES")
        sq, _ = avg(rb, "SQ_BUSY_CYCLES")
        gui, n = avg(rb, "GRBM_GUI_ACTIVE")
        dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rb if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
        if gui and dur:
            d = sum(dur) / len(dur)                      # ns, under the counter pass (profiled runs hold a lower clock)
            cycles = gui / 8.0
            res.update({"SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CYCLES": sq, "GRBM_GUI_ACTIVE": gui, "duration_ns_profiled": d,
                        "clock_ghz": round(cycles / d, 3),
                        "mfma_busy_frac": round(mf / (cycles * 256 * 4), 4) if mf else None,
                        "busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)"})
    return res


def main():
    out, kernels, fcsv, wcsv = sys.argv[1:5]
    bcsv = sys.argv[5] if len(sys.argv) > 5 else None
    ks = kernels.split(";") if ";" in kernels else kernels.split(",")      # (";" when a kernel name contains ", ")
    res = [one(k, fcsv, wcsv, bcsv) for k in ks]
    res = res[0] if len(res) == 1 else {"kernels": res}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
