#!/usr/bin/env python3
"""A/B runs of bench.py for kernel work: one line per (config, knobs) with the rate and the per-family launch durations /
fractions of the HBM peak from the full record.
   python tools/ab_bench.py OUT.jsonl "3:" "3:mac3=0" "1:mac3=0,subsets=1" ...      (config:knobs[:extra bench args])"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = sys.argv[1]
    for spec in sys.argv[2:]:
        parts = spec.split(":")
        cfg, knobs = parts[0], parts[1] if len(parts) > 1 else ""
        extra = parts[2].split() if len(parts) > 2 else []
        steps = "8" if cfg == "3" else "6"
        with tempfile.NamedTemporaryFile(suffix=".json") as tf:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--side", "0", "--cpu-seconds", "0", "--steps", steps,
                   "--warmup", "2", "--full-out", tf.name] + (["--tune", knobs] if knobs else []) + (["--lockstep", "1"] if cfg == "5" else []) + extra
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
            if r.returncode != 0:
                rec = {"spec": spec, "error": (r.stdout + r.stderr)[-600:]}
            else:
                full = json.load(open(tf.name))
                fam = {k: [round(v["avg_launch_ms"], 4), v["frac"], v.get("frac_per_launch")] for k, v in (full.get("roofline_all") or {}).items()}
                rec = {"spec": spec, "value": full["value"], "ms_per_step": full["ms_per_step"], "probe": (full.get("probe") or {}).get("rms_error"),
                       "subsets": full["config"]["subsets"],
                       "call_us[p50,p99,max]": [(full.get("call_us") or {}).get(k) for k in ("p50", "p99", "max")], "families[avg_ms, frac(union), frac_per_launch]": fam}
        print(json.dumps(rec), flush=True)
        with open(out, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
