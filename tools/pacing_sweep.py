#!/usr/bin/env python
"""Pacing sweep of the in-system fused mapper: the reference's two-process system on NOISY synthetic sequences, untouched and
with SURVEY 8(f)'s rows applied (`--fused`) under different iteration budgets per tracked frame (`GSICP_FUSED_ITERS_PER_FRAME`, 0 = free-run) and
the "free-run but keep the trackable Gaussians' geometry" experiment (`GSICP_FUSED_FREEZE_TRACKABLE`).  One JSON document on stdout / --out.

    python tools/pacing_sweep.py --out gpurun_out/r05_fused_pacing_sweep.json [--quick]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SEQUENCES = {
    "replica_noisy_fast": ["--synthetic", "300", "--noise", "--speed", "2", "--jitter", "0.003"],
    "tum_noisy": ["--synthetic", "200", "--shape", "tum", "--noise"],
    "replica_clean": ["--synthetic", "300"],
    "replica_noisy_fast_30fps": ["--synthetic", "300", "--noise", "--speed", "2", "--jitter", "0.003", "--limit30"],
}


def one(seq, fused, env_extra, cache, timeout=400):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_reference_slam.py"), "--cache", cache, "--timeout", str(timeout - 40)] + SEQUENCES[seq]
    if fused:
        cmd.append("--fused")
    env = dict(os.environ)
    env.update(env_extra)
    t0 = time.time()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        res = json.loads(lines[-1]) if lines else {"status": "failed", "stderr": p.stderr[-1500:]}
    except subprocess.TimeoutExpired:
        res = {"status": "timeout"}
    keep = ("status", "system_fps", "ate_rmse_cm", "ate_true_rmse_cm", "ate_max_cm", "psnr", "ssim", "fused_mapper", "wall_s", "frames", "entry")
    out = {k: res.get(k) for k in keep if k in res}
    out.update(sequence=seq, variant="fused" if fused else "untouched", env=env_extra, harness_wall_s=round(time.time() - t0, 1))
    if res.get("status") != "measured":
        out["error"] = res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--cache", default="/tmp/gsicp_cache")
    ap.add_argument("--quick", action="store_true", help="one run per configuration, fewer budgets")
    ap.add_argument("--budget-s", type=float, default=900.0, help="stop starting new runs after this many seconds")
    ap.add_argument("--plan", default="v1", help="v1: budgets 1/2/4/8 + free-run + the freeze experiment; v2: the confirmation runs after v1")
    a = ap.parse_args()
    k = lambda v: {"GSICP_FUSED_ITERS_PER_FRAME": str(v)}      # noqa: E731
    geom = dict(k(0), GSICP_FUSED_FREEZE_TRACKABLE="geom")
    plan = []
    if a.plan == "v2":
        plan = [("replica_noisy_fast", True, k(2)), ("replica_noisy_fast", True, k(4)), ("replica_noisy_fast", True, geom), ("replica_noisy_fast", True, geom),
                ("replica_noisy_fast", False, {}), ("replica_noisy_fast_30fps", True, geom), ("replica_noisy_fast_30fps", True, k(2)),
                ("tum_noisy", True, geom), ("tum_noisy", True, geom), ("tum_noisy", True, k(2)), ("replica_clean", True, geom)]
    for seq in (() if a.plan == "v2" else ("replica_noisy_fast", "tum_noisy")):
        plan.append((seq, False, {}))
        for v in ((1, 2, 4) if (a.quick or seq == "tum_noisy") else (1, 2, 4, 8)):
            plan.append((seq, True, k(v)))
        plan.append((seq, True, k(0)))
        if seq == "replica_noisy_fast" and not a.quick:
            plan.append((seq, False, {}))               # second untouched run: run-to-run spread of the bar itself
            plan.append((seq, True, dict(k(0), GSICP_FUSED_FREEZE_TRACKABLE="xyz")))
            plan.append((seq, True, dict(k(0), GSICP_FUSED_FREEZE_TRACKABLE="geom")))
    if a.plan != "v2":
        plan.append(("replica_clean", True, k(2)))
    t0 = time.time()
    runs = []
    for seq, fused, env in plan:
        if time.time() - t0 > a.budget_s:
            runs.append({"sequence": seq, "variant": "fused" if fused else "untouched", "env": env, "status": "skipped (time budget)"})
            continue
        r = one(seq, fused, env, a.cache)
        runs.append(r)
        fm = r.get("fused_mapper") or {}
        sys.stderr.write(f"{seq:20s} {r['variant']:9s} {json.dumps(env):70s} fps {r.get('system_fps')} ate {r.get('ate_rmse_cm')} / {r.get('ate_true_rmse_cm')} "
                         f"psnr {r.get('psnr')} iters {fm.get('iterations')} gpu_ms {fm.get('gpu_median_ms_per_iteration')}\n")
        if a.out:
            with open(a.out, "w") as fh:
                json.dump({"what": __doc__.strip().split("\n\n")[0], "sequences": SEQUENCES, "runs": runs}, fh, indent=1)
    doc = {"what": __doc__.strip().split("\n\n")[0], "sequences": SEQUENCES, "runs": runs, "total_s": round(time.time() - t0, 1)}
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
