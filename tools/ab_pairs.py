"""Alternating same-box A/B pairs of `python bench.py` (round 6; replaces the one-shot tools/r5_call_*.sh lease scripts): runs
[A, B] x N back to back in ONE box, prints ms_per_step of every run, the paired differences B - A and their mean with a 95 % confidence
interval (Student t over the N pairs). A and B are given as 'ENV=VAL ... :: bench flags' strings.
    python tools/ab_pairs.py --pairs 5 --a ":: --eager-lm-head" --b "::" [--steps 8] [--config 1] [--out gpurun_out/x.txt]"""
import argparse, json, math, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T95 = {2: 12.706, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365, 9: 2.306, 10: 2.262}


def run(spec, steps, warmup, config):
    envs, _, flags = spec.partition("::")
    env = dict(os.environ)
    for kv in envs.split():
        k, _, v = kv.partition("=")
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), "--config", str(config), "--no-cpu-baseline",
           "--no-secondary", "--no-gemm-profile"] + flags.split()
    cp = subprocess.run(cmd, capture_output=True, text=True, env=env)
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    if cp.returncode != 0 or not line:
        raise SystemExit(f"bench failed: {cp.stderr[-800:]}")
    j = json.loads(line[-1])
    return j["ms_per_step"], (j.get("box") or {})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=5)
    ap.add_argument("--a", required=True)
    ap.add_argument("--b", required=True)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lines = [f"A: bench.py {args.a!r}    B: bench.py {args.b!r}    config {args.config}, {args.steps} timed steps after {args.warmup} warm-up, alternating A B A B ... in one box"]
    diffs = []
    for i in range(args.pairs):
        (a, ba), (b, bb) = run(args.a, args.steps, args.warmup, args.config), run(args.b, args.steps, args.warmup, args.config)
        diffs.append(b - a)
        lines.append(f"pair {i + 1}: A {a:8.2f} ms   B {b:8.2f} ms   B - A {b - a:+7.2f} ms   (sclk {ba.get('sclk_mhz_median')} / {bb.get('sclk_mhz_median')} MHz, "
                     f"{ba.get('socket_power_w_median')} / {bb.get('socket_power_w_median')} W, random-MFMA {ba.get('mfma_random_pflops')} / {bb.get('mfma_random_pflops')} PFLOP/s)")
        print(lines[-1], flush=True)
    mean = statistics.mean(diffs)
    if len(diffs) > 1:
        half = T95.get(len(diffs), 2.0) * statistics.stdev(diffs) / math.sqrt(len(diffs))
        lines.append(f"paired difference B - A over {len(diffs)} pairs: mean {mean:+.2f} ms, 95 % CI [{mean - half:+.2f}, {mean + half:+.2f}] ms"
                     f" -> {'excludes 0' if (mean - half) * (mean + half) > 0 else 'INCLUDES 0'}")
    else:
        lines.append(f"paired difference B - A: {mean:+.2f} ms (one pair)")
    print(lines[-1])
    if args.out:
        with open(args.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
