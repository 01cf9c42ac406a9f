#!/bin/bash
# factor + loop of a wave beside the Hessians of the next one (side streams) against everything in sequence, one box
for flags in "" "--overlap --factor-streams 1" "--overlap --factor-streams 2" ""; do
  python3 bench_gptq.py --no-cpu-baseline --hessian-methods "" $flags 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$flags]', d['seconds'], flush=True)"
done
