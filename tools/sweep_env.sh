#!/bin/bash
# usage: tools/sweep_env.sh OUT VAR "v1 v2 ..." [legs: h 3 5]   -- one bench line per (value, leg), appended to gpurun_out/OUT
out=gpurun_out/$1; var=$2; vals=$3; legs=${4:-"3 5"}
for v in $vals; do for leg in $legs; do
  case $leg in
    h) a="--dtype fp32";;
    3) a="--dtype bf16 --encoder resnet18 --size 512 --workload adversarial";;
    5) a="--dtype bf16 --encoder resnet50 --size 768";;
  esac
  env $var=$v timeout -k 10 240 python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-sustain > gpurun_out/sweep_tmp.json 2> gpurun_out/sweep_err.log || exit 1
  python - "$var=$v" "$leg" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/sweep_tmp.json").read().strip().splitlines()[-1])
ts = d["roofline"].get("time_split", {})
bn = ts.get("batchnorm_passes", {})
ck = d["roofline"]["all_conv_kernels"]["by_kernel"]
print(sys.argv[1], "leg", sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms  serial", ts.get("serial_step_ms"), " bn_ms", bn.get("ms"),
      " conv:", {k.replace("conv_", "").replace("_kernel", "")[:26]: v["ms_per_step"] for k, v in list(ck.items())[:12]})
PY
done; done
cat $out
