"""tools/pmc_traffic.py TAG DIR_FETCH DIR_WRITE [bench args] — fold two rocprofv3 counter_collection
CSVs into HBM bytes per launch per kernel (see tools/pmc_traffic.sh)."""
import collections, csv, glob, json, os, re, sys

tag, dfetch, dwrite = sys.argv[1:4]
bargs = sys.argv[4:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ellp_amd.build import engine_source_hash  # noqa: E402


def arg(name, default):
    return bargs[bargs.index(name) + 1] if name in bargs else default


def fold(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            mk = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
            k = mk.group(0) if mk else r["Kernel_Name"][:40]
            a = acc[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fe, wr = fold(dfetch, "FETCH_SIZE"), fold(dwrite, "WRITE_SIZE")
out = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 400 "
                   "--warmup 50. Counter units are KiB. gfx950 correction (MI355X_MICROARCH.md, HBM section): "
                   "FETCH_SIZE reports 1/2 of a wide coalesced read stream, so read bytes = 2*FETCH_SIZE*1024; "
                   "WRITE_SIZE is exact.",
       "workload": f"m={arg('--m', '2000')} n={arg('--n', '5000')} {arg('--solver', 'primal')}",
       "engine_source_hash": engine_source_hash(), "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    nf, sf = fe.get(k, [0, 0.0])
    nw, sw = wr.get(k, [0, 0.0])
    if max(nf, nw) < 10:
        continue
    rd = 2.0 * 1024.0 * sf / max(nf, 1)
    wb = 1024.0 * sw / max(nw, 1)
    out["kernels"][k] = {"launches": nf, "FETCH_SIZE_KiB_avg": round(sf / max(nf, 1), 1),
                         "WRITE_SIZE_KiB_avg": round(sw / max(nw, 1), 1), "hbm_read_bytes_per_launch": int(rd),
                         "hbm_write_bytes_per_launch": int(wb), "hbm_bytes_per_launch": int(rd + wb)}
print(json.dumps(out, indent=1))
