#!/bin/bash
# bench.py: the default line and the two-rank code path on one GPU (gloo), summarised
cd "$GRAFT_REPO_ROOT"; R=gpurun_out/r6; mkdir -p $R
python3 bench.py --gpus 2 --debug-single-device --steps 10 --warmup 3 2>$R/bench_two_ranks.err | tail -1 > $R/bench_two_ranks.json
python3 - <<PY
import json
try:
    d = json.load(open("$R/bench_two_ranks.json"))
    print("two ranks:", d["value"], "evals/s;", "secondary:", {k: d.get("secondary", {}).get(k) for k in ("value", "first_call_s", "error", "ll_grad_evaluations")})
except Exception as e:
    print("two ranks failed:", e); print(open("$R/bench_two_ranks.err").read()[-1500:])
PY
python3 bench.py 2>$R/bench_default.err | tail -1 > $R/bench_default.json
python3 - <<PY
import json
try:
    d = json.load(open("$R/bench_default.json"))
    print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
    print({k: v for k, v in d["secondary"].items() if k in ("value", "first_call_s", "first_over_steady")})
    print("mcmc", d["secondary_mcmc"].get("value"))
    s = d["secondary_stim"]; print({k: s[k] for k in s if "map" in k or k in ("value", "ms_per_eval")})
    print(d["secondary_narrow_shard"]); print("cpu", d["cpu_baseline"]["value"])
except Exception as e:
    print("default failed:", e); print(open("$R/bench_default.err").read()[-1500:])
PY
