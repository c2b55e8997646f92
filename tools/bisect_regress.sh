#!/usr/bin/env bash
# one gpurun call: every library (and, for three commits, that commit's Python host code too) on the two regressed numbers
mkdir -p gpurun_out/bisect
for lib in var_so/lib_8751b1f.so var_so/lib_604f30b.so var_so/lib_d5f948b.so var_so/lib_b211421.so var_so/lib_90ef8ec.so var_so/lib_cc95aad.so var_so/lib_4afbc78.so bblean_amd/libbbhip.so; do
  echo "#### current python, $lib"
  BBHIP_LIBRARY=$PWD/$lib timeout 600 python tools/bisect_regress.py 2>&1 | grep "==\|Error\|error" 
done
for c in 8751b1f 604f30b cc95aad; do
  echo "#### python + library of $c"
  BB_ROOT=$PWD/var_so/t_$c BBHIP_LIBRARY=$PWD/var_so/t_$c/bblean_amd/libbbhip.so timeout 600 python tools/bisect_regress.py 2>&1 | grep "==\|Error\|error"
done
echo "#### VMM probe"
hipcc --offload-arch=gfx950 -o /tmp/vmm_probe tools/probe/vmm_probe.cpp && timeout 120 /tmp/vmm_probe g 2>&1 | tail -n 80
