#!/bin/bash
# Collect the round's committed evidence on a GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# Writes gpurun_out/<tag>_*: bench JSON, rocprofv3 kernel stats (single-stream and overlapped), PMC traffic / MFMA-busy reductions.
# PMC passes run alone (--pmc with --kernel-trace only), the program itself after `--`.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-sustain"
python3 $ROOT/bench.py --steps 30 --warmup 10 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err || exit 1
echo "bench done"
UDASEG_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_serial -- $BENCH > $OUT/${TAG}_prof_serial.log 2>&1 || exit 1
cp $(find $OUT/${TAG}_prof_serial -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_serial.csv
echo "serial stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_ovl -- $BENCH > $OUT/${TAG}_prof_ovl.log 2>&1 || exit 1
cp $(find $OUT/${TAG}_prof_ovl -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_overlapped.csv
python3 $ROOT/tools/trace_neighbors.py $OUT/${TAG}_prof_ovl > $OUT/${TAG}_step_gaps.txt 2>&1
python3 $ROOT/tools/ddp_tail.py $(find $OUT/${TAG}_prof_ovl -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_step_tail.txt 2>&1
python3 $ROOT/tools/step_timeline.py $(find $OUT/${TAG}_prof_ovl -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_step_timeline.txt 2>&1
echo "overlapped stats done"
UDASEG_SERIAL=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- $BENCH > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 1
echo "pmc fetch done"
UDASEG_SERIAL=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- $BENCH > $OUT/${TAG}_pmc_write.log 2>&1 || exit 1
echo "pmc write done"
python3 $ROOT/tools/pmc_traffic.py $(find $OUT/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1) \
        $(find $OUT/${TAG}_pmc_write -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic.json || exit 1
UDASEG_SERIAL=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv \
        -d $OUT/${TAG}_pmc_mfma -- $BENCH > $OUT/${TAG}_pmc_mfma.log 2>&1 || exit 1
python3 $ROOT/tools/pmc_mfma.py $(find $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_mfma_util.json || exit 1
echo "pmc mfma done"
UDASEG_SERIAL=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -- $BENCH > $OUT/${TAG}_pmc_sq.log 2>&1 || exit 1
python3 $ROOT/tools/pmc_sq.py $(find $OUT/${TAG}_pmc_sq -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_sq.txt || exit 1
rm -rf $OUT/${TAG}_pmc_sq
echo "pmc sq done"
# in-kernel timelines of the two dominant split kernels, the 1x1 kernels stand-alone, and the one-GPU rehearsal of the RCCL path
python3 $ROOT/tools/wgrad_timeline.py > $OUT/${TAG}_wgrad_timeline.txt 2>&1
python3 $ROOT/tools/f3_timeline.py > $OUT/${TAG}_f3_timeline.txt 2>&1
python3 $ROOT/tools/up_probe.py > $OUT/${TAG}_up_probe.txt 2>&1
python3 $ROOT/tools/gemm1x1_probe.py 2>&1 | grep "M=" > $OUT/${TAG}_gemm1x1_probe.txt
UDASEG_GEMM_1X1=0 python3 $ROOT/tools/gemm1x1_probe.py 2>&1 | grep "M=" > $OUT/${TAG}_gemm1x1_probe_stream_only.txt
python3 $ROOT/tools/x3_bias_check.py 2>&1 | grep "l2" > $OUT/${TAG}_mfma_bias.txt
UDASEG_F3_SIGNS=0 python3 $ROOT/tools/x3_bias_check.py 2>&1 | grep "halo" > $OUT/${TAG}_mfma_bias_nosigns.txt
echo "timelines done"
# the bf16 legs (cfg 3, cfg 5): single-stream kernel statistics
UDASEG_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_bf16_cfg3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 \
        --no-cpu-baseline --no-roofline --no-also --no-sustain --workload adversarial --dtype bf16 > $OUT/${TAG}_prof_bf16_cfg3.log 2>&1 || exit 1
cp $(find $OUT/${TAG}_prof_bf16_cfg3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bf16_cfg3.csv
UDASEG_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_bf16_cfg5 -- python3 $ROOT/bench.py --steps 5 --warmup 2 \
        --no-cpu-baseline --no-roofline --no-also --no-sustain --encoder resnet50 --size 768 --dtype bf16 > $OUT/${TAG}_prof_bf16_cfg5.log 2>&1 || exit 1
cp $(find $OUT/${TAG}_prof_bf16_cfg5 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_bf16_cfg5.csv
echo "bf16 stats done"
# bf16 legs: HBM-side traffic and matrix-pipe busy per kernel symbol (separate PMC passes, program directly after --)
for CFG in cfg3 cfg5; do
  if [ $CFG = cfg3 ]; then WL="--workload adversarial --dtype bf16"; TXT="r18-Unet + discriminator adversarial iteration 8+8x512x512 bf16 (BASELINE cfg 3)";
  else WL="--encoder resnet50 --size 768 --dtype bf16"; TXT="r50-Unet 8x3x768x768 bf16 train step (BASELINE cfg 5 per-GPU work)"; fi
  B16="python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-also --no-sustain $WL"
  UDASEG_SERIAL=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pf_$CFG -- $B16 > $OUT/${TAG}_pf_$CFG.log 2>&1 || exit 1
  UDASEG_SERIAL=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pw_$CFG -- $B16 > $OUT/${TAG}_pw_$CFG.log 2>&1 || exit 1
  python3 $ROOT/tools/pmc_traffic.py $(find $OUT/${TAG}_pf_$CFG -name "*counter_collection.csv" | head -1) \
          $(find $OUT/${TAG}_pw_$CFG -name "*counter_collection.csv" | head -1) $OUT/${TAG}_bf16_${CFG}_pmc_traffic.json "$TXT" > $OUT/${TAG}_bf16_${CFG}_pmc_traffic.txt || exit 1
  UDASEG_SERIAL=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv \
          -d $OUT/${TAG}_pm_$CFG -- $B16 > $OUT/${TAG}_pm_$CFG.log 2>&1 || exit 1
  python3 $ROOT/tools/pmc_mfma.py $(find $OUT/${TAG}_pm_$CFG -name "*counter_collection.csv" | head -1) $OUT/${TAG}_bf16_${CFG}_pmc_mfma_util.json "$TXT" || exit 1
  UDASEG_SERIAL=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/${TAG}_pl2_$CFG -- $B16 > $OUT/${TAG}_pl2_$CFG.log 2>&1
  python3 $ROOT/tools/pmc_l2.py $(find $OUT/${TAG}_pl2_$CFG -name "*counter_collection.csv" | head -1) $OUT/${TAG}_bf16_${CFG}_pmc_l2.json "$TXT"
  rm -rf $OUT/${TAG}_pf_$CFG $OUT/${TAG}_pw_$CFG $OUT/${TAG}_pm_$CFG $OUT/${TAG}_pl2_$CFG
  echo "bf16 pmc $CFG done"
done
rm -rf $OUT/${TAG}_prof_bf16_cfg3 $OUT/${TAG}_prof_bf16_cfg5
rm -rf $OUT/${TAG}_prof_serial $OUT/${TAG}_prof_ovl $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma
ls -la $OUT/${TAG}_*
