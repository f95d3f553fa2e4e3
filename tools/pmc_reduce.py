"""Reduce the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) to fabric bytes per GEMM launch.
Usage: python tools/pmc_reduce.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [min_us] [tcc hit/miss csv]
Population (round 3): launches of the PLAIN instantiation gemm256_kernel<0,0,0> only, >= min_us under the counter pass (150 us keeps the
333 launches per step >= 0.1 TFLOP that bench.py's roofline object averages over and drops the 17 small ones).
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports half the bytes of wide coalesced reads on gfx950 -> x 1024 x 2;
WRITE_SIZE x 1024 (uncalibrated). Both are fabric-side (L2 <-> Infinity Cache / HBM) counters: Infinity-Cache hits are included."""
import csv, json, os, re, sys, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mla_amd import hip   # the library the counters were collected on (same tree, same call): its gemm256 source id goes into the record

PLAIN = r"gemm256_kernel<0, ?0, ?0(, ?(true|false))?>"     # the plain instantiation only: the population bench.py's algorithmic_bytes_per_launch is over


def load(path, counter, min_us):
    n = 0; tot = 0.0; dur = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not re.search(PLAIN, r["Kernel_Name"]):
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if us < min_us:
            continue
        n += 1; tot += float(r["Counter_Value"]); dur += us
    return n, tot / max(n, 1), dur / max(n, 1)

min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 300.0
nf, fkb, fus = load(sys.argv[1], "FETCH_SIZE", min_us)
nw, wkb, wus = load(sys.argv[2], "WRITE_SIZE", min_us)
hit_rate = None
if len(sys.argv) > 5:
    _, hit, _ = load(sys.argv[5], "TCC_HIT_sum", min_us)
    _, miss, _ = load(sys.argv[5], "TCC_MISS_sum", min_us)
    hit_rate = hit / max(hit + miss, 1.0)
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 1 --warmup 1`, gemm256_kernel<0,0,0> launches >= {min_us:.0f} us",
       "tcc_hit_rate": hit_rate,
       "collected": datetime.date.today().isoformat(),
       "gemm_source_id": hip.gemm_source_id(),
       "corrections": "FETCH_SIZE x 1024 B x 2 (gfx950 counts 128-B requests as 64 B); WRITE_SIZE x 1024 B (uncalibrated)",
       "launches": nf, "avg_launch_us": fus, "fetch_size_kb_avg": fkb, "write_size_kb_avg": wkb,
       "hbm_read_bytes_per_launch": fkb * 1024 * 2, "hbm_write_bytes_per_launch": wkb * 1024,
       "hbm_bytes_per_launch": fkb * 1024 * 2 + wkb * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
