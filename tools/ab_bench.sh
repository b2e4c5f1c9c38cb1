#!/bin/bash
# A/B runs of bench.py on ONE box: every line is "[workload] '<env switches>' <ms_per_step> <grad_norm>".
#   gpurun -- 'bash tools/ab_bench.sh aasvc "" S2SVC_GEMM_W8=0 "S2SVC_AAS_FBRANCH=0 S2SVC_FS_PREFETCH=0"'
#   EXTRA="--split-backward" bash tools/ab_bench.sh aasvc ...      (further bench.py flags)
wl=$1; shift
for envs in "$@" "$1"; do
  ms=$(env $envs python bench.py --workload $wl --no-cpu-baseline --no-extras --steps 100 --warmup 10 ${EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'], d['final_losses'].get('grad_norm'))")
  echo "[$wl${EXTRA:+ $EXTRA}] '${envs}' $ms"
done
