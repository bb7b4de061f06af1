cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_1x1.py tests/test_gpu_halo.py -m gpu -x -q 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-also --no-sustain --encoder resnet50 --size 768 --dtype bf16"
UDASEG_GEMM_1X1=0 $B > gpurun_out/r4_cfg5_gemm0.json 2> gpurun_out/r4_cfg5_gemm0.err; echo rc=$?
$B --layer-table > gpurun_out/r4_cfg5_gemm1.json 2> gpurun_out/r4_cfg5_gemm1.err; echo rc=$?
UDASEG_GEMM_1X1_MAXM=400000 $B > gpurun_out/r4_cfg5_gemm2.json 2> gpurun_out/r4_cfg5_gemm2.err; echo rc=$?
UDASEG_FUSE_BN_APPLY=0 $B > gpurun_out/r4_cfg5_gemm3.json 2> gpurun_out/r4_cfg5_gemm3.err; echo rc=$?
UDASEG_FUSE_BN_APPLY=0 UDASEG_GEMM_1X1_MAXM=400000 $B > gpurun_out/r4_cfg5_gemm4.json 2> gpurun_out/r4_cfg5_gemm4.err; echo rc=$?
