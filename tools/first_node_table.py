#!/usr/bin/env python
"""Reduce the bench lines `tools/first_node_run.sh` left in <outdir> to one table + SCALE_first_node.json in the shape of the driver's
SCALE_rNN.json: per N the whole-job value, ms/step, weak-scaling efficiency = value(N) / (N * value(1)), the slowest / fastest rank and
how long the compute stream sat behind reduce-scatter / all-gather events; then the knob sweep at the largest N relative to its default."""
import glob
import json
import os
import re
import sys


def load(path):
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except (OSError, ValueError):
        return None


def main(out):
    runs = {}
    for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
        tag = os.path.basename(f)[len("bench_"):-len(".json")]
        d = load(f)
        if d:
            runs[tag] = d
    base = runs.get("n1")
    rows, scale = [], {"metric": None, "unit": None, "runs": []}
    print(f"{'run':<16} {'N':>2} {'ms/step':>9} {'value':>9} {'eff':>6} {'rank ms min..max':>20} {'rs wait ms':>11} {'ag wait ms':>11}  knobs")
    for tag, d in sorted(runs.items(), key=lambda kv: (kv[1]["n_gpus"], kv[0])):
        n = d["n_gpus"]
        eff = d["value"] / (n * base["value"]) if base else float("nan")
        pr = d.get("per_rank") or {}
        rs = max(pr.get("rs_event_wait_ms_per_step") or [0.0])
        ag = max(pr.get("gather_event_wait_ms_per_step") or [0.0])
        knobs = d.get("collective_knobs") or {}
        m = re.match(r"n\d+$", tag)
        print(f"{tag:<16} {n:>2} {d['ms_per_step']:>9.2f} {d['value']:>9.3f} {eff:>6.3f} "
              f"{str(pr.get('step_ms_min', '')) + '..' + str(pr.get('step_ms_max', '')):>20} {rs:>11.2f} {ag:>11.2f}  {knobs if not m else ''}")
        if m:
            scale["metric"], scale["unit"] = d["metric"], d["unit"]
            scale["runs"].append({"n_gpus": n, "value": d["value"], "ms_per_step": d["ms_per_step"], "scaling": d["scaling"],
                                  "efficiency_vs_n1": round(eff, 4), "per_rank": pr or None, "roofline_frac": (d.get("roofline") or {}).get("frac")})
        rows.append(tag)
    if not rows:
        print("no bench_*.json lines found in", out)
        return 1
    json.dump(scale, open(os.path.join(out, "SCALE_first_node.json"), "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/first_node"))
