"""Per-kernel counter table from four rocprofv3 --pmc passes over `python bench.py --steps 1 --warmup 1` (tools/collect_counters.sh):
  pass 1  SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass 2  TCC_HIT_sum TCC_MISS_sum        pass 3  FETCH_SIZE        pass 4  WRITE_SIZE
Formulas (rocprofv3's own derived-metric expressions, `rocprofv3 -L` on the box; gfx950 corrections from MI355X_MICROARCH.md):
  MfmaUtil %  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)        VALUBusy % = SQ_ACTIVE_INST_VALU / (256 CUs x GRBM_GUI_ACTIVE)
  LDS conflict % = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of LDS-array cycles that are conflict replays)
  wait % = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave-cycles parked in s_waitcnt / s_barrier)        L2 hit % = TCC_HIT / (TCC_HIT + TCC_MISS)
  fabric read = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests as 64 B), write = WRITE_SIZE KiB x 1024; Infinity-Cache hits included
Counter passes perturb timing (clock 1.9 vs 2.0 GHz): durations here are for the ratios only; see the kernel-stats CSV for times.
Usage: python tools/pmc_table.py <pass1.csv> <pass2.csv> <pass3.csv> <pass4.csv> [out.json [steps_in_the_pmc_run]]
out.json (round 6): the rows as data -- kernel, dispatches per step, us, MfmaUtil % -- plus the identity of the library the counters were
collected on; bench.py's roofline.step_weighted_mfma_util reads the newest profiles/r*_pmc_table.json."""
import collections, csv, hashlib, json, os, re, sys

FAMILIES = [("gemm256<0,0,0> plain (>= 150 us)", r"gemm256_kernel<0, ?0, ?0(, ?(true|false))?>", 150.0),
            ("gemm256<0,0,1> gate|up+SwiGLU", r"gemm256_kernel<0, ?0, ?1(, ?(true|false))?>", 150.0),
            ("gemm256<0,0,2> dact+SwiGLU bwd", r"gemm256_kernel<0, ?0, ?2(, ?(true|false))?>", 150.0),
            ("attn_fwd", r"attn_fwd_kernel", 50.0), ("attn_bwd_dq", r"attn_bwd_dq_kernel", 50.0), ("attn_bwd_dkv", r"attn_bwd_dkv_kernel", 50.0),
            ("attn_bwd (one pass)", r"attn_bwd_kernel", 50.0), ("attn_bwd merged launch (dQ + dK.dV blocks)", r"attn_bwd_merged_kernel", 50.0),
            ("adamw", r"adamw_vec4_kernel", 50.0), ("rmsnorm_bwd", r"rmsnorm_bwd_kernel", 20.0), ("tile_transpose", r"tile_transpose_kernel", 10.0)]


def load(path):
    """{family: {counter: [per-dispatch (sum over instances, max over instances)]}, '_us': [...]}"""
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        fam = next((f for f, pat, _ in FAMILIES if re.search(pat, name)), None)
        if fam is None:
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if us < next(m for f, _, m in FAMILIES if f == fam):
            continue
        d = per[fam][r["Dispatch_Id"]]
        v = float(r["Counter_Value"])
        s, m = d.get(r["Counter_Name"], (0.0, 0.0))
        d[r["Counter_Name"]] = (s + v, max(m, v))
        d["_us"] = (us, us)
    out = {}
    for fam, disp in per.items():
        agg = collections.defaultdict(list)
        for d in disp.values():
            for k, v in d.items():
                agg[k].append(v)
        out[fam] = agg
    return out


def mean(xs):
    return sum(xs) / max(len(xs), 1)


p1, p2, p3, p4 = (load(a) for a in sys.argv[1:5])
print(__doc__.split("Usage")[0])
hdr = f"{'kernel':34s} {'n':>4s} {'us(pmc)':>8s} {'clk GHz':>7s} {'MfmaUtil%':>9s} {'VALUBusy%':>9s} {'LDSconf%':>8s} {'wait%':>6s} {'L2hit%':>6s} {'rd GB':>7s} {'wr GB':>7s} {'fabric TB/s':>11s}"
print(hdr)
json_rows = []
steps_in_run = int(sys.argv[6]) if len(sys.argv) > 6 else 2          # collect_counters.sh: --steps 1 --warmup 1
for fam, _, _ in FAMILIES:
    a = p1.get(fam)
    if not a:
        continue
    us = mean([u for u, _ in a["_us"]])
    gui_s, gui_m = mean([s for s, _ in a["GRBM_GUI_ACTIVE"]]), mean([m for _, m in a["GRBM_GUI_ACTIVE"]])
    gui = gui_m
    if gui / (us * 1e3) > 4.0:          # a single reduced value that is a sum over the 8 XCCs
        gui = gui_s / 8.0
    sm = lambda k: mean([s for s, _ in a[k]]) if a.get(k) else float("nan")   # noqa: E731
    mfma = 100.0 * sm("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024.0)
    valu = 100.0 * sm("SQ_ACTIVE_INST_VALU") / (256.0 * gui)
    conf = 100.0 * sm("SQ_LDS_BANK_CONFLICT") / max(sm("SQ_LDS_IDX_ACTIVE"), 1.0)
    wait = 100.0 * sm("SQ_WAIT_ANY") / max(sm("SQ_WAVE_CYCLES"), 1.0)
    b = p2.get(fam, {})
    hit = mean([s for s, _ in b.get("TCC_HIT_sum", [])]) if b.get("TCC_HIT_sum") else float("nan")
    miss = mean([s for s, _ in b.get("TCC_MISS_sum", [])]) if b.get("TCC_MISS_sum") else float("nan")
    c, d = p3.get(fam, {}), p4.get(fam, {})
    rd = mean([s for s, _ in c.get("FETCH_SIZE", [])]) * 1024 * 2 if c.get("FETCH_SIZE") else float("nan")
    wr = mean([s for s, _ in d.get("WRITE_SIZE", [])]) * 1024 if d.get("WRITE_SIZE") else float("nan")
    us3 = mean([u for u, _ in c["_us"]]) if c.get("_us") else us
    json_rows.append(dict(kernel=fam, n_per_step=len(a["_us"]) / steps_in_run, us=round(us, 2), clk_ghz=round(gui / (us * 1e3), 3), mfma_util_pct=round(mfma, 2),
                          l2_hit_pct=round(100 * hit / (hit + miss), 2) if hit == hit else None, fabric_read_gb=round(rd / 1e9, 4) if rd == rd else None,
                          fabric_write_gb=round(wr / 1e9, 4) if wr == wr else None))
    print(f"{fam:34s} {len(a['_us']):4d} {us:8.1f} {gui / (us * 1e3):7.2f} {mfma:9.1f} {valu:9.1f} {conf:8.1f} {wait:6.1f} "
          f"{100 * hit / (hit + miss):6.1f} {rd / 1e9:7.3f} {wr / 1e9:7.3f} {(rd + wr) / (us3 * 1e-6) / 1e12:11.2f}")

if len(sys.argv) > 5:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    lib_path = os.environ.get("MLA_HIP_LIB") or os.path.join(root, "mla_amd", "libmla_hip.so")
    lib_id = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()[:16]
    try:
        from mla_amd import hip
        gid = hip.gemm_source_id()
    except Exception:   # noqa: BLE001
        gid = None
    json.dump(dict(rows=json_rows, steps_in_pmc_run=steps_in_run, library_id=lib_id, gemm_source_id=gid,
                   note="MfmaUtil % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) per kernel family; us = mean dispatch duration under the counter pass"),
              open(sys.argv[5], "w"), indent=1)
