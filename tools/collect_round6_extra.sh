#!/bin/bash
# Round-6 additions to the tracked profile set (via gpurun, after tools/collect_round.sh r06): one-step kernel traces + kernel stats of the
# non-headline configurations (BASELINE configs[2] per-GPU batch, configs[3], configs[4]) and the 8-vs-4-frame scaling split.
set -u
OUT=$PWD/gpurun_out/profiles_r06
mkdir -p $OUT
bash tools/quick_trace.sh r6b8 > $OUT/trace_b8.txt 2>&1
bash tools/quick_trace.sh r6c4 --batch-per-gpu 4 > $OUT/trace_c4.txt 2>&1
bash tools/quick_trace.sh r6d --config D > $OUT/trace_d.txt 2>&1
bash tools/quick_trace.sh r6e --config E > $OUT/trace_e.txt 2>&1
cp gpurun_out/trace_r6c4/step_trace.csv $OUT/r06_step_trace_4frames.csv
cp gpurun_out/trace_r6c4/kernel_stats.csv $OUT/r06_kernel_stats_4frames.csv
cp gpurun_out/trace_r6d/step_trace.csv $OUT/r06_configD_step_trace.csv
cp gpurun_out/trace_r6d/kernel_stats.csv $OUT/r06_configD_kernel_stats.csv
cp gpurun_out/trace_r6e/step_trace.csv $OUT/r06_configE_step_trace.csv
cp gpurun_out/trace_r6e/kernel_stats.csv $OUT/r06_configE_kernel_stats.csv
python tools/scaling_split.py gpurun_out/trace_r6b8/step_trace.csv gpurun_out/trace_r6c4/step_trace.csv $OUT/r06_scaling_split_8_vs_4_frames.csv > $OUT/r06_scaling_split.txt
# library symbols in any of the traces (hipBLASLt: Cijk_*, MIOpen: miopen* / naive_conv / Im2Col ...): must be empty
grep -l -i "Cijk_\|miopen\|im2col\|rocblas" $OUT/r06_config?_kernel_stats.csv $OUT/r06_kernel_stats_4frames.csv > $OUT/r06_library_symbols.txt 2>/dev/null
echo "files with library kernel symbols: $(wc -l < $OUT/r06_library_symbols.txt)" >> $OUT/r06_library_symbols.txt
ls -la $OUT
