#!/usr/bin/env python
"""Turns the drop-in call traces of a reference run (tools/run_reference_slam.py --trace DIR) into a per-frame time budget of the
tracker process and an iteration budget of the mapper process.  Time INSIDE gicp.* / raster.* calls is this repo's; the gaps are the
reference's own host code (image conversion, numpy, torch CPU ops, shared-memory hand-off, waiting for the other process)."""
import glob
import json
import os
import sys

import numpy as np


def load(path):
    ev = []
    for ln in open(path):
        n, a, b = ln.split()
        ev.append((n, float(a), float(b)))
    return ev


def main(d):
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "*.trace"))):
        ev = load(f)
        names = {e[0] for e in ev}
        if "gicp.align" in names:
            al = [i for i, e in enumerate(ev) if e[0] == "gicp.align"]
            frames = []
            for k in range(1, len(al) - 1):          # frame = [set_input_source enter, next set_input_source enter)
                i0 = max(j for j in range(al[k]) if ev[j][0] == "gicp.set_input_source")
                i1 = max(j for j in range(al[k + 1]) if ev[j][0] == "gicp.set_input_source")
                seg = ev[i0:i1]
                inside = sum(b - a for _, a, b in seg)
                total = ev[i1][1] - ev[i0][1]
                key = any(n == "gicp.get_source_rotationsq" for n, _, _ in seg)
                tkey = any(n == "gicp.set_input_target" for n, _, _ in seg)
                per = {}
                for n, a, b in seg:
                    per[n] = per.get(n, 0.0) + (b - a)
                frames.append(dict(total=total, inside=inside, key=key, tkey=tkey, per=per))
            def stat(sel):
                fr = [f for f in frames if sel(f)]
                if not fr:
                    return None
                per = {}
                for f in fr:
                    for n, v in f["per"].items():
                        per[n] = per.get(n, 0.0) + v
                return dict(frames=len(fr), ms_per_frame=round(1e3 * np.mean([f["total"] for f in fr]), 3),
                            median_ms=round(1e3 * float(np.median([f["total"] for f in fr])), 3),
                            ms_inside_dropin=round(1e3 * np.mean([f["inside"] for f in fr]), 3),
                            dropin_ms_by_call={n: round(1e3 * v / len(fr), 3) for n, v in sorted(per.items())})
            out["tracker"] = dict(plain_frames=stat(lambda f: not f["key"]), mapping_keyframes=stat(lambda f: f["key"] and not f["tkey"]),
                                  tracking_keyframes=stat(lambda f: f["tkey"]), all=stat(lambda f: True))
        elif "refglue.fused_mapping_iteration" in names:     # the FUSED system (tools/run_reference_slam.py --fused): one call per mapper iteration
            it = [e for e in ev if e[0] == "refglue.fused_mapping_iteration"]
            if len(it) > 3:
                dt = np.diff([e[1] for e in it])
                out["mapper"] = dict(iterations=len(it), median_ms_per_iteration=round(1e3 * float(np.median(dt)), 3),
                                     mean_ms_per_iteration=round(1e3 * float(dt.mean()), 3), p90_ms_per_iteration=round(1e3 * float(np.quantile(dt, 0.9)), 3),
                                     call_ms=round(1e3 * float(np.median([b - a for _, a, b in it])), 3), variant="fused: one hipGraph replay per iteration")
        elif "raster.forward" in names:
            fw = [e for e in ev if e[0] == "raster.forward"]
            bw = [e for e in ev if e[0] == "raster.backward"]
            if len(fw) > 3:
                dt = np.diff([e[1] for e in fw])
                out["mapper"] = dict(iterations=len(fw), median_ms_per_iteration=round(1e3 * float(np.median(dt)), 3),
                                     mean_ms_per_iteration=round(1e3 * float(dt.mean()), 3),
                                     forward_call_ms=round(1e3 * float(np.median([b - a for _, a, b in fw])), 3),
                                     backward_call_ms=round(1e3 * float(np.median([b - a for _, a, b in bw])), 3) if bw else None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
