"""Summarise rocprofv3 --pmc passes (one counter_collection.csv per pass, --output-format csv) into a JSON keyed by kernel.
usage: python tools/pmc_summary.py out.json note pass1_counter_collection.csv [pass2_counter_collection.csv ...]
Per kernel and counter: average per launch and launch count.  When both FETCH_SIZE and WRITE_SIZE are present (units: KB),
hbm_bytes_per_launch_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced stream (MI355X_MICROARCH.md; calibrated in round 1 on gemv_rows_kernel, which reads a 4.295 GB matrix once)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def kernel_sources_sha256():
    """Hash of the library's sources (gp_algos_amd/csrc/*.hip, *.h, sorted by name): written into the summary so that bench.py can
    tell a summary of another build from one of the code it runs (same function in bench.py)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gp_algos_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    if not m:
        return name[:60]
    targs = (m.group(2) or "").replace(" ", "")
    return m.group(1) + targs


def main():
    out, note, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    for k, cs in agg.items():
        e = {}
        for c, vals in cs.items():
            e[c + "_avg_per_launch"] = sum(vals) / len(vals)
            e["launches_" + c] = len(vals)
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            e["hbm_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_avg_per_launch"] + e["WRITE_SIZE_avg_per_launch"]) * 1024.0
        kernels[k] = e
    json.dump({"note": note, "kernel_sources_sha256": kernel_sources_sha256(), "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0)):
        print("%-44s %s" % (k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()}))


if __name__ == "__main__":
    main()
