cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-also --no-sustain"
for i in 1 2; do
  $B > gpurun_out/r4_ddp_plain_$i.json 2>/dev/null
  UDASEG_DDP_REHEARSE=1 $B > gpurun_out/r4_ddp_reh_$i.json 2>/dev/null
done
for blk in 96 144; do UDASEG_WGRAD_F3_BLOCKS=$blk UDASEG_DDP_REHEARSE=1 $B > gpurun_out/r4_ddp_reh_b$blk.json 2>/dev/null; UDASEG_WGRAD_F3_BLOCKS=$blk $B > gpurun_out/r4_ddp_plain_b$blk.json 2>/dev/null; done
UDASEG_DDP_REHEARSE=2 $B > gpurun_out/r4_ddp_hooks_only.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
UDASEG_DDP_REHEARSE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ddptrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-also --no-sustain > $GRAFT_REPO_ROOT/gpurun_out/r4_ddp_trace.log 2>&1
cp $(find /tmp/ddptrace -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r4_ddp_kernel_trace.csv
ls -la $GRAFT_REPO_ROOT/gpurun_out/r4_ddp_*
