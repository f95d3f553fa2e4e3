"""Reduce a rocm-smi sampler log (tools/collect_counters.sh: one JSON object per line, every 0.2 s) next to an un-profiled bench run:
socket power and shader clock while the training step runs. Usage: python tools/power_summary.py <samples.jsonl> <bench.json>"""
import json, re, statistics, sys
rows = []
for line in open(sys.argv[1]):
    try:
        c = json.loads(line).get("card0", {})
    except Exception:
        continue
    pw = next((float(v) for k, v in c.items() if "power" in k.lower()), None)
    sk = next((v for k, v in c.items() if k.lower().startswith("sclk clock speed")), None)
    m = re.search(r"(\d+)", sk or "")
    if pw is not None and m:
        rows.append((pw, int(m.group(1))))
bench = json.load(open(sys.argv[2]))
print(f"bench: {bench['ms_per_step']} ms/step, {bench['value']} samples/s over {bench['steps']} steps (un-profiled, sampler beside it)")
print(f"{len(rows)} rocm-smi samples at 0.2 s (model build, warm-up and the timed steps)")
busy = [r for r in rows if r[0] > 0.6 * max(p for p, _ in rows)]      # samples taken while the step loop runs
for name, sel in (("all samples", rows), ("while stepping (power > 60 % of the maximum seen)", busy)):
    if not sel:
        continue
    pw, sk = [p for p, _ in sel], [s for _, s in sel]
    print(f"{name}: n={len(sel)}  power W min/median/max = {min(pw):.0f} / {statistics.median(pw):.0f} / {max(pw):.0f}   "
          f"sclk MHz min/median/max = {min(sk)} / {int(statistics.median(sk))} / {max(sk)}")
print("sclk histogram while stepping (MHz bucket of 100: samples):",
      dict(sorted({b: sum(1 for _, s in busy if s // 100 * 100 == b) for b in {s // 100 * 100 for _, s in busy}}.items())))
print("dense bf16 MFMA peak scales with sclk: 2.5 PFLOP/s is the figure at the 2.4 GHz boost clock")
