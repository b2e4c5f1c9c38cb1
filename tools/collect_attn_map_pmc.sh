#!/bin/bash
# Counters of the attention-map kernels (csrc/attn_map.hip) inside the Transformer-TTS step: HBM-side bytes and LDS bank conflicts, each
# in its own pass with --kernel-trace only (eager launches: every dispatch is its own counter sample).
#   gpurun --timeout 900 -- 'bash tools/collect_attn_map_pmc.sh'      -> gpurun_out/profiles/attn_map_pmc.txt
set -u
cd "$(dirname "$0")/.."
R=$PWD
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
db() { ls "$1"/*.db "$1"/*/*.db 2>/dev/null | head -1; }
run() {  # run <dir> <counters...>
  local d=$1; shift
  rm -rf "$d"
  (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d "$d" -o r -- python "$R/bench.py" --workload tts --no-graph --no-cpu-baseline --no-extras --steps 6 --warmup 2 > "$d.log" 2>&1)
  python tools/rocpd_pmc.py "$(db "$d")" attn_map
}
{
  echo "# attn_map_kernel<NW, MODE, DK2> inside bench.py --workload tts --no-graph (B = 8: 32 (utterance, head) pairs; T = 151 text / 320 frames, d_k = 96)"
  echo "# HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)"
  run /tmp/pmc_am_fetch FETCH_SIZE
  run /tmp/pmc_am_write WRITE_SIZE
  run /tmp/pmc_am_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
} > "$OUT/attn_map_pmc.txt" 2>&1
cat "$OUT/attn_map_pmc.txt"
