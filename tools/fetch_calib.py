#!/usr/bin/env python3
"""Summarise tools/fetch_calib.sh: per calibration kernel, every collected counter averaged over its launches, and
the ratio known bytes / (FETCH_SIZE * 1024) resp. (WRITE_SIZE * 1024) -- the factor a raw counter has to be
multiplied with for that access width.  Writes <dir>/summary.json and prints a table."""
import collections
import csv
import glob
import json
import sys


def main(d):
    known = json.load(open(d + "/bytes.json"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k in known:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, b in known.items():
        c = {name: sum(v) / len(v) for name, v in agg[k].items()}
        row = dict(known_read_bytes=b["read"], known_write_bytes=b["write"], counters=c)
        if c.get("FETCH_SIZE") and b["read"]:
            row["read_bytes_per_FETCH_SIZE_KiB"] = b["read"] / (c["FETCH_SIZE"] * 1024.0)
        if c.get("WRITE_SIZE") and b["write"]:
            row["write_bytes_per_WRITE_SIZE_KiB"] = b["write"] / (c["WRITE_SIZE"] * 1024.0)
        if c.get("TCC_EA0_RDREQ_sum") and b["read"]:
            row["read_bytes_per_RDREQ"] = b["read"] / c["TCC_EA0_RDREQ_sum"]
        if c.get("TCC_EA0_WRREQ_sum") and b["write"]:
            row["write_bytes_per_WRREQ"] = b["write"] / c["TCC_EA0_WRREQ_sum"]
        if "TCC_EA0_RDREQ_128B_sum" in c and "TCC_EA0_RDREQ_sum" in c:
            n128, n64, n32 = c["TCC_EA0_RDREQ_128B_sum"], c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c["TCC_EA0_RDREQ_32B_sum"]
            row["read_bytes_by_request_size"] = 128 * n128 + 64 * n64 + 32 * n32
            row["requests_unclassified"] = c["TCC_EA0_RDREQ_sum"] - n128 - n64 - n32
        out[k] = row
    json.dump(out, open(d + "/summary.json", "w"), indent=1)
    for k, row in out.items():
        print("%-16s read %12d write %12d  x_fetch %s  x_write %s  B/RDREQ %s  B/WRREQ %s" % (
            k, row["known_read_bytes"], row["known_write_bytes"],
            "%.3f" % row["read_bytes_per_FETCH_SIZE_KiB"] if "read_bytes_per_FETCH_SIZE_KiB" in row else "-",
            "%.3f" % row["write_bytes_per_WRITE_SIZE_KiB"] if "write_bytes_per_WRITE_SIZE_KiB" in row else "-",
            "%.1f" % row["read_bytes_per_RDREQ"] if "read_bytes_per_RDREQ" in row else "-",
            "%.1f" % row["write_bytes_per_WRREQ"] if "write_bytes_per_WRREQ" in row else "-"))
        if "read_bytes_by_request_size" in row:
            print("      bytes by request size %d (%.3f of known), unclassified requests %d" % (
                row["read_bytes_by_request_size"], row["read_bytes_by_request_size"] / max(row["known_read_bytes"], 1),
                row["requests_unclassified"]))
        print("     ", {n: round(v) for n, v in sorted(row["counters"].items())})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fetch_calib")
