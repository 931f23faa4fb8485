"""Turn two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace, same
bench.py command) into the per-kernel HBM traffic table bench.py reports as roofline.traffic.

usage: pmc_traffic.py <key e.g. medium/b1/t128> <fetch_dir> <write_dir> <out.json> ["command string"] [bench_full.json]

With a bench_full.json of the same workload (bench.py's full result object: roofline.kernels[*].algorithmic_bytes_per_launch,
from the engine's level-2 profile rows) every kernel also gets its algorithmic bytes and the ratio traffic / algorithmic.

hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are KiB; the factor 2 is
the gfx950 correction of MI355X_MICROARCH.md (section HBM). That note calibrates 16-byte-per-lane streams; the
kernels here read 4 bytes per lane, for which the guide gives no calibration, so the read side is an upper
bound (WRITE_SIZE alone reproduces the algorithmic output bytes of the fused MRF kernels exactly)."""
import collections, csv, glob, json, os, re, sys


def norm(name):
    # the engine's level-2 profile rows carry the same spelling (full template arguments, no spaces)
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("pe::", "").replace(" ", "")


def collect(d, counter):
    tot, calls = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = norm(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                calls[k] += 1
    return {k: tot[k] / calls[k] for k in tot}, calls


def main():
    key, fd, wd, out = sys.argv[1:5]
    fetch, calls = collect(fd, "FETCH_SIZE")
    write, _ = collect(wd, "WRITE_SIZE")
    table = {}
    for k in sorted(fetch, key=lambda k: -fetch[k] * calls[k]):
        w = write.get(k, 0.0)
        table[k] = {"launches": calls[k], "fetch_kib_raw": round(fetch[k], 1), "write_kib_raw": round(w, 1),
                    "hbm_bytes_per_launch": int((2 * fetch[k] + w) * 1024)}
    if len(sys.argv) > 6 and os.path.exists(sys.argv[6]):
        full = json.load(open(sys.argv[6]))
        kern = (full.get("roofline") or {}).get("kernels") or {}
        for k, row in table.items():
            a = (kern.get(k) or {}).get("algorithmic_bytes_per_launch")
            row["algorithmic_bytes_per_launch"] = None if not a else int(a)
            row["traffic_over_algorithmic"] = None if not a else round(row["hbm_bytes_per_launch"] / a, 2)
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc[key] = {"command": sys.argv[5] if len(sys.argv) > 5 else "", "kernels": table}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{out}: {key}: {len(table)} kernels")


if __name__ == "__main__":
    main()
