#!/bin/bash
# Headline launch (uint4 g128 blob of 4096x11008) under settings of the OQ_RTN_RING* knobs, one process each (the knobs are
# read once per process); bench.py checks every run against the reference's digests.
#   bash scripts/lab_ring.sh out_dir "RING=0" "RING=1 RING_WAVES=10" ...
out=$1; shift
mkdir -p "$out"
i=0
for setting in "$@"; do
    envs=""
    for kv in $setting; do envs="$envs OQ_RTN_$kv"; done
    log="$out/run_$i.json"
    env $envs python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 30 > "$log" 2> "$out/run_$i.err" || { echo "FAILED: $setting"; tail -5 "$out/run_$i.err"; }
    python - "$log" "$setting" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:45s} launch_us {r['launch_us']:7.2f} frac {r['frac']:.4f} verified {d.get('verified_vs_reference_digest')} kernel {r.get('kernel')}", flush=True)
except Exception as e:
    print(sys.argv[2], "no line:", e, flush=True)
PY
    i=$((i+1))
done
