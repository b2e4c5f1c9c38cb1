set -u
R=$PWD; OUT=$R/gpurun_out/profiles; mkdir -p $OUT; export TMPDIR=/tmp
db() { ls "$1"/*.db "$1"/*/*.db 2>/dev/null | head -1; }
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_tts -o tts -- python "$R/bench.py" --workload tts --no-cpu-baseline --no-extras --steps 24 --warmup 3 > /tmp/prof_tts.log 2>&1)
python tools/rocpd_stats.py "$(db /tmp/prof_tts)" > "$OUT/tts_train_bf16_kernel_stats.txt" 2>&1
python tools/rocpd_timeline.py "$(db /tmp/prof_tts)" 3 > "$OUT/tts_train_bf16_timeline.txt" 2>&1
head -40 $OUT/tts_train_bf16_kernel_stats.txt
