"""profiles/r04_pmc_decode.json from the two counter passes of tools/profile_all.sh step 3 (measurement helper).
usage: pmc_json.py <pmc_fetch.txt> <pmc_write.txt> <out.json>   (the per-kernel means tools/pmc_summary.py printed, in KB)
Per kernel of the decode: HBM-side bytes per launch = FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md for wide coalesced
reads, calibrated in round 1 with a 1 GiB copy: profiles/r01_pmc_traffic.md) + WRITE_SIZE, from SEPARATE --pmc passes."""
import json, re, sys


def means(path):
    out = {}
    for line in open(path):
        f = line.split()
        if len(f) < 7 or f[1] != "mean":
            continue
        name = line.split("total", 1)[1].strip()
        m = re.search(r"dctts::(\w+)", name)
        if not m:
            continue
        k = m.group(1)
        n = int(f[2])
        if k in out:                                   # several instantiations of one template: launch-weighted mean
            v0, n0 = out[k]
            out[k] = ((v0 * n0 + float(f[0]) * n) / (n0 + n), n0 + n)
        else:
            out[k] = (float(f[0]), n)
    return out


fetch, write = means(sys.argv[1]), means(sys.argv[2])
if "xcone_kernel" not in fetch or "xcone_kernel" not in write:
    sys.exit("no xcone_kernel rows in the counter summaries")
res = {"_note": "FETCH_SIZE x2 + WRITE_SIZE per launch (KB raw values beside it); the memory-side counters include Infinity-Cache hits; workload: "
                "tools/decode_only.py 40 (B = 32, eager, decode mode 3).  Counter collection serialises dispatches across queues, so under --pmc the two "
                "decode streams meet through events and the passenger workgroups are a launch of their own (ROCPROF_COUNTER_COLLECTION -> DCTTS_SYNC_VALUES=0)."}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    if k not in write:
        continue
    res[k] = {"launches": fetch[k][1], "fetch_size_kb_raw": round(fetch[k][0], 1), "fetch_correction": 2.0, "write_size_kb_raw": round(write[k][0], 1),
              "hbm_bytes_per_launch": int(fetch[k][0] * 2.0 * 1024 + write[k][0] * 1024)}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[3]).read())
