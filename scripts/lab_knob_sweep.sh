#!/bin/bash
# bash scripts/lab_knob_sweep.sh out.log layout shapes "KNOB=a KNOB2=b" "..." : one process per setting (scripts/lab_rtn_shapes.py)
out=$1; lay=$2; shapes=$3; shift 3
: > "$out"
for s in "$@"; do
    envs=""; for kv in $s; do envs="$envs OQ_RTN_$kv"; done
    echo "== $lay $s" >> "$out"
    env $envs timeout -k 10 300 python scripts/lab_rtn_shapes.py --layout "$lay" --reps 400 --trials 2 --shapes "$shapes" >> "$out" 2>/dev/null || exit 1
done
python - "$out" <<'PY'
import json, re, sys, collections
res = collections.defaultdict(dict)
key = None
for l in open(sys.argv[1]):
    m = re.match(r"== (\S+) (.*)", l)
    if m:
        key = m.group(2).strip() or "(default)"
    elif l.startswith("{"):
        d = json.loads(l)
        res[d["shape"]].setdefault(key, []).append(min(d["us"]))
for shp, v in res.items():
    print(shp)
    for k, x in v.items():
        print(f"    {k:32s}", " ".join(f"{t:7.2f}" for t in x))
PY
