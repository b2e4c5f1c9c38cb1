#!/bin/bash
# A/B runs of bench.py on ONE box: every line is "<label> <ms_per_step>" (env switches in front of the label).
#   gpurun -- 'bash tools/ab_bench.sh aasvc "" S2SVC_PROLOGUE_OVERLAP=0 "S2SVC_AAS_FBRANCH=0 S2SVC_FS_PREFETCH=0"'
wl=$1; shift
for envs in "$@" "$1"; do
  ms=$(env $envs python bench.py --workload $wl --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'], d['final_losses'].get('grad_norm'))")
  echo "[$wl] '${envs}' $ms"
done
