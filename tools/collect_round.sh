#!/bin/bash
# Round artifacts on the GPU box (via gpurun): bench line, rocprofv3 kernel stats + one-step trace, PMC traffic passes,
# SQ counters of the hand-written MFMA kernels.  Usage: bash tools/collect_round.sh r02
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/profiles_$TAG
mkdir -p $OUT
if [ -z "${SKIP_BENCH:-}" ]; then python bench.py --legs full > $OUT/bench_full.log 2>&1; fi
if [ -z "${SKIP_BENCH:-}" ]; then tail -1 $OUT/bench_full.log > $OUT/${TAG}_bench_n1.json; fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -f csv -T -d /tmp/prof_kt -- python /root/repo/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-also > /tmp/kt.log 2>&1
python /root/repo/tools/trace_step.py $(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_step_trace.csv
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  rocprofv3 --pmc $C -f csv -T -d /tmp/prof_$C -- python /root/repo/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-also > /tmp/pmc_$C.log 2>&1
  python /root/repo/tools/pmc_by_kernel.py $(find /tmp/prof_$C -name "*counter_collection.csv" | head -1) $C $OUT/${TAG}_pmc_${C}_by_kernel.csv
done
bash /root/repo/tools/pmc_kernels.sh $OUT/${TAG}_pmc_sq_mfma_kernels.csv "k_conv3x3_tiles|k_tok_gemm|k_tok_ffn|k_layer_|k_ln2_bwd_top|k_attn_|k_dw_grouped|k_spconv|k_v2_|k_vfe1" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
python /root/repo/tools/class_shares.py $OUT/${TAG}_step_trace.csv $OUT/class_shares.json "profiles/${TAG}_step_trace.csv (rocprofv3 --kernel-trace of bench.py, one optimizer step incl. the side-stream geometry plan of the next batch)"
python /root/repo/tools/step_traffic.py $OUT $TAG > $OUT/${TAG}_step_traffic.json
ls -la $OUT
