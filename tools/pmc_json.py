"""profiles/r03_pmc_xgroup.json from the two counter passes of tools/profile_all.sh step 3 (measurement helper).
usage: pmc_json.py <pmc_fetch.txt> <pmc_write.txt> <out.json>   (the per-kernel means tools/pmc_summary.py printed, in KB)"""
import json, sys
def mean_of(path, key):
    for line in open(path):
        if key in line:
            f = line.split()
            return float(f[0]), int(f[2])
    return None, 0
fetch, n = mean_of(sys.argv[1], "xgroup_kernel")
write, _ = mean_of(sys.argv[2], "xgroup_kernel")
if fetch is None or write is None:
    sys.exit("no xgroup_kernel rows in the counter summaries")
B, d = 32, 256
lay = 4.0 * (d * 2 * d + 3 * B * 2 * d + 2 * B * 64 + B * d + 4 * d)          # bench.py: algorithmic bytes of one layer
json.dump({
    "kernel": "xgroup_kernel (a run of newest-row highway layers of the decode as one launch: AudioDec 6 layers / AudioEnc 10 layers; mean over both), B=32",
    "launches": n, "fetch_size_kb_raw": fetch, "fetch_correction": 2.0, "write_size_kb_raw": write, "write_correction": 1.0,
    "hbm_bytes_per_launch": int(fetch * 2.0 * 1024 + write * 1024), "mean_layers_per_launch": 8.0,
    "algorithmic_bytes_per_launch_8_layers": lay * 8,
    "note": "FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md (calibrated in round 1 with a 1 GiB copy: profiles/r01_pmc_traffic.md); the "
            "memory-side counters include Infinity-Cache hits; tools/decode_only.py 40 (eager, DM=3; counter collection serialises dispatches across "
            "queues, so the two decode streams meet through events and the passenger workgroups are a launch of their own: ROCPROF_COUNTER_COLLECTION "
            "-> DCTTS_SYNC_VALUES=0), separate --pmc passes.  bench.py times the AudioEnc runs only (10 layers): scale by 10/8 to compare."
}, open(sys.argv[3], "w"), indent=1)
print(open(sys.argv[3]).read())
