#!/bin/bash
# Re-collect every artefact under profiles/ on a GPU box (run from the repo root; results land in gpurun_out/profiles/,
# copy the ones to keep into profiles/ with the round prefix).  Counters run in their own passes, with --kernel-trace only.
#   gpurun --timeout 2700 -- 'bash tools/collect_profiles.sh'
#   SECTIONS="bench steps" bash tools/collect_profiles.sh     (only some of: bench dp micro steps roofline busy)
set -u
cd "$(dirname "$0")/.."
R=$PWD
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
db() { ls "$1"/*.db "$1"/*/*.db 2>/dev/null | head -1; }
prof() {  # prof <dir> <rocprofv3 args...> -- <command...>
  local d=$1; shift
  rm -rf "$d"
  (cd /tmp && rocprofv3 "$@" > "$d.log" 2>&1)
}

want() { [[ " ${SECTIONS:-all} " == *" all "* || " ${SECTIONS:-all} " == *" $1 "* ]]; }

if want bench; then
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --workload aasvc > "$OUT/bench_aasvc.json" 2>> "$OUT/bench.err"
python bench.py --workload tts --no-extras > "$OUT/bench_tts.json" 2>> "$OUT/bench.err"
fi
if want dp; then
# the data-parallel step at world size 1: mechanism alone (--split-backward: marks, no collectives) and with the FORCED exchange, per stage mode
for wl in vtn aasvc; do
  for mode in flush marks graphs; do
    python bench.py --workload $wl --split-backward --stage-mode $mode --no-cpu-baseline --no-extras > "$OUT/bench_${wl}_split_${mode}.json" 2>> "$OUT/bench.err"
    python bench.py --workload $wl --force-dist --stage-mode $mode --no-cpu-baseline --no-extras > "$OUT/bench_${wl}_force_dist_${mode}.json" 2>> "$OUT/bench.err"
  done
done
python bench.py --workload aasvc --split-backward --stage-mode flush --min-bucket-mb 1 --no-cpu-baseline --no-extras > "$OUT/bench_aasvc_split_flush_min1.json" 2>> "$OUT/bench.err"
python bench.py --workload aasvc --split-backward --stage-mode graphs --dp-decoder-stages 3 --no-cpu-baseline --no-extras > "$OUT/bench_aasvc_split_graphs_h3.json" 2>> "$OUT/bench.err"
python bench.py --workload aasvc --split-backward --stage-mode graphs --no-cpu-baseline --no-extras --stage-times > "$OUT/bench_aasvc_split_backward.json" 2>> "$OUT/bench.err"
fi
if want micro; then
python tools/bench_decode.py > "$OUT/bench_decode.json" 2>> "$OUT/bench.err"
python tools/kernel_code_sizes.py > "$OUT/kernel_code_sizes.txt" 2>&1
(for wl in vtn aasvc tts; do python tools/aten_in_step.py --workload $wl; done) > "$OUT/aten_in_step.txt" 2>&1
python tools/gemm_bench.py > "$OUT/gemm_bench.txt" 2>&1
python tools/gemm8_bench.py > "$OUT/gemm8_bench.txt" 2>&1
python tools/bench_frontend.py --cpu > "$OUT/bench_frontend.json" 2>> "$OUT/bench.err"
(python tools/bench_trainer.py --workload vtn | tail -1; python tools/bench_trainer.py --workload aasvc --steps 40 | tail -1) > "$OUT/bench_trainer.json" 2>> "$OUT/bench.err"

fi
if want steps; then
# per-kernel statistics + one-step timelines of the three workloads
prof /tmp/prof_step --kernel-trace --stats -d /tmp/prof_step -o vtn -- python "$R/bench.py" --no-cpu-baseline --no-extras --steps 24 --warmup 3
python tools/rocpd_stats.py "$(db /tmp/prof_step)" > "$OUT/vtn_train_bf16_kernel_stats.txt" 2>&1
python tools/rocpd_timeline.py "$(db /tmp/prof_step)" 3 > "$OUT/vtn_train_bf16_timeline.txt" 2>&1
prof /tmp/prof_aas --kernel-trace --stats -d /tmp/prof_aas -o aas -- python "$R/bench.py" --workload aasvc --no-cpu-baseline --steps 24 --warmup 3
python tools/rocpd_stats.py "$(db /tmp/prof_aas)" > "$OUT/aasvc_train_bf16_kernel_stats.txt" 2>&1
python tools/rocpd_timeline.py "$(db /tmp/prof_aas)" 3 > "$OUT/aasvc_train_bf16_timeline.txt" 2>&1
prof /tmp/prof_tts --kernel-trace --stats -d /tmp/prof_tts -o tts -- python "$R/bench.py" --workload tts --no-cpu-baseline --no-extras --steps 24 --warmup 3
python tools/rocpd_stats.py "$(db /tmp/prof_tts)" > "$OUT/tts_train_bf16_kernel_stats.txt" 2>&1
python tools/rocpd_timeline.py "$(db /tmp/prof_tts)" 3 > "$OUT/tts_train_bf16_timeline.txt" 2>&1
prof /tmp/prof_dec --kernel-trace --stats -d /tmp/prof_dec -o dec -- python "$R/tools/bench_decode.py" --iters 2
python tools/rocpd_stats.py "$(db /tmp/prof_dec)" > "$OUT/decode_kernel_stats.txt" 2>&1
python tools/rocpd_timeline.py "$(db /tmp/prof_dec)" 5 decode_emit_advance > "$OUT/decode_step_timeline.txt" 2>&1

fi
if want roofline; then
# dominant kernel of the headline workload (and of AAS-VC): timing + counters
for wl in vtn aasvc; do
  prof /tmp/prof_roof_$wl --kernel-trace --stats -d /tmp/prof_roof_$wl -o r -- python "$R/bench.py" --roofline-only --workload $wl
  python tools/rocpd_stats.py "$(db /tmp/prof_roof_$wl)" > "$OUT/roofline_${wl}_kernel_stats.txt" 2>&1
  prof /tmp/prof_fetch_$wl --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch_$wl -o r -- python "$R/bench.py" --roofline-only --workload $wl
  python tools/rocpd_pmc.py "$(db /tmp/prof_fetch_$wl)" gemm_8ph > "$OUT/roofline_${wl}_pmc_fetch.txt" 2>&1
  prof /tmp/prof_write_$wl --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write_$wl -o r -- python "$R/bench.py" --roofline-only --workload $wl
  python tools/rocpd_pmc.py "$(db /tmp/prof_write_$wl)" gemm_8ph > "$OUT/roofline_${wl}_pmc_write.txt" 2>&1
done
prof /tmp/prof_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/prof_sq1 -o r -- python "$R/bench.py" --roofline-only
python tools/rocpd_pmc.py "$(db /tmp/prof_sq1)" gemm_8ph > "$OUT/roofline_vtn_pmc_sq.txt" 2>&1
prof /tmp/prof_sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/prof_sq2 -o r -- python "$R/bench.py" --roofline-only
python tools/rocpd_pmc.py "$(db /tmp/prof_sq2)" gemm_8ph >> "$OUT/roofline_vtn_pmc_sq.txt" 2>&1
# the same two SQ passes for the GEMM that carries the AAS-VC step (4096 x 1536 x 1536, 256 x 128 tiles)
prof /tmp/prof_sq3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/prof_sq3 -o r -- python "$R/bench.py" --roofline-only --workload aasvc
python tools/rocpd_pmc.py "$(db /tmp/prof_sq3)" gemm_8ph > "$OUT/roofline_aasvc_pmc_sq.txt" 2>&1
prof /tmp/prof_sq4 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/prof_sq4 -o r -- python "$R/bench.py" --roofline-only --workload aasvc
python tools/rocpd_pmc.py "$(db /tmp/prof_sq4)" gemm_8ph >> "$OUT/roofline_aasvc_pmc_sq.txt" 2>&1
fi
if want busy; then
# MFMA-pipe busy per GEMM-shaped kernel over WHOLE steps (eager launches: every dispatch is its own counter sample)
HDR="# MFMA-pipe busy fraction per GEMM-shaped kernel over WHOLE training steps (eager launches of bench.py --no-graph under\n# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace; tools/step_mfma_busy.py).\n# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg us x 2400 cycles/us); wait/wave = SQ_WAIT_ANY / SQ_WAVE_CYCLES."
printf "$HDR\n" > "$OUT/step_mfma_busy.txt"
for wl in vtn aasvc; do
  prof /tmp/prof_busy_$wl --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d /tmp/prof_busy_$wl -o r -- python "$R/bench.py" --workload $wl --no-graph --no-cpu-baseline --no-extras --steps 8 --warmup 2
  python tools/step_mfma_busy.py $wl "$(db /tmp/prof_busy_$wl)" >> "$OUT/step_mfma_busy.txt" 2>&1
  python tools/step_mfma_busy.py $wl "$(db /tmp/prof_busy_$wl)" by-grid >> "$OUT/step_mfma_busy_by_grid.txt" 2>&1
done
python tools/gemm_bench.py --filter "vtn wgrad grouped w8" > "$OUT/w8_grouped_bench.txt" 2>&1
(python tools/bench_trainer.py --workload aasvc --accum 8 --batch 2 --steps 64 | tail -1) > "$OUT/bench_trainer_accum8.json" 2>> "$OUT/bench.err"
fi
ls -la "$OUT"
