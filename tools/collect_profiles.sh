#!/bin/bash
# Re-collect every artefact under profiles/ on a GPU box (run from the repo root; results land in gpurun_out/profiles/,
# copy the ones to keep into profiles/).  Counters run in their own passes, with --kernel-trace only.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'
set -u
cd "$(dirname "$0")/.."
R=$PWD
OUT=$R/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
db() { ls "$1"/*.db "$1"/*/*.db 2>/dev/null | head -1; }

python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python tools/bench_aasvc.py 2>/dev/null | tail -1 > "$OUT/bench_aasvc_vc2.json"
python tools/bench_decode.py 2>/dev/null | tail -1 > "$OUT/bench_decode_c5.json"
python tools/gemm_bench.py > "$OUT/gemm_bench.txt" 2>&1

rm -rf /tmp/prof_step /tmp/prof_roof /tmp/prof_fetch /tmp/prof_write /tmp/prof_sq1 /tmp/prof_sq2
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_step -o vtn -- python "$R/bench.py" --no-cpu-baseline --steps 24 --warmup 3 > /tmp/prof_step.log 2>&1)
python tools/rocpd_stats.py "$(db /tmp/prof_step)" > "$OUT/vtn_train_bf16_kernel_stats.txt" 2>&1
python tools/rocpd_timeline.py "$(db /tmp/prof_step)" 3 > "$OUT/vtn_train_bf16_timeline.txt" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_roof -o r -- python "$R/bench.py" --roofline-only > /tmp/prof_roof.log 2>&1)
python tools/rocpd_stats.py "$(db /tmp/prof_roof)" > "$OUT/roofline_kernel_stats.txt" 2>&1
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch -o r -- python "$R/bench.py" --roofline-only > /tmp/prof_fetch.log 2>&1)
python tools/rocpd_pmc.py "$(db /tmp/prof_fetch)" gemm_glds > "$OUT/roofline_pmc_fetch.txt" 2>&1
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write -o r -- python "$R/bench.py" --roofline-only > /tmp/prof_write.log 2>&1)
python tools/rocpd_pmc.py "$(db /tmp/prof_write)" gemm_glds > "$OUT/roofline_pmc_write.txt" 2>&1
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/prof_sq1 -o r -- python "$R/bench.py" --roofline-only > /tmp/prof_sq1.log 2>&1)
python tools/rocpd_pmc.py "$(db /tmp/prof_sq1)" gemm_glds > "$OUT/roofline_pmc_sq.txt" 2>&1
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/prof_sq2 -o r -- python "$R/bench.py" --roofline-only > /tmp/prof_sq2.log 2>&1)
python tools/rocpd_pmc.py "$(db /tmp/prof_sq2)" gemm_glds >> "$OUT/roofline_pmc_sq.txt" 2>&1
ls -la "$OUT"
