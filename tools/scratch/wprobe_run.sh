#!/bin/bash
# timing-only decomposition of the fp32 weight-gradient kernel (build/libudaseg_wprobe.so: conv_wgrad.hip with runtime probe bits:
# 1 no global loads after the first tile, 2 no LDS stores, 4 no barriers, 8 no epilogue (atomics), 16 no MFMAs / fragment reads)
for p in 0 1 3 7 8 15 16 24; do
  UDASEG_WGRAD_PROBE=$p UDASEG_LIB=$PWD/build/libudaseg_wprobe.so timeout -k 10 120 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-also --no-sustain > gpurun_out/wprobe_$p.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/wprobe_$p.json").read().strip().splitlines()[-1])
k=d["roofline"]["all_conv_kernels"]["by_kernel"]
print("probe $p", d["value"], {n[:34]: (v["ms_per_step"], v["tflops"]) for n,v in k.items() if "wgrad_kernel<" in n and "small" not in n})
PY
done
