#!/bin/bash
# Which kernel-selection change moved the 24-image smoke loss (-0.014923 in rounds 2-3, -0.014511 in round 4)?
# Runs __graft_entry__.smoke() under the instrumented library's A/B switches (IIC_HIP_LIB=dbg IIC_DEBUG=..., iic_amd/_lib.py).  -> gpurun_out/smoke_bisect.txt
mkdir -p gpurun_out
OUT=gpurun_out/smoke_bisect.txt
: > $OUT
for E in "" "iic_debug_bd_ms=4" "iic_debug_enable_pw=0" "iic_debug_bd_ms=4,iic_debug_enable_pw=0" "iic_debug_bd_w1=0" "iic_debug_enable_p64=0"; do
  echo "== IIC_DEBUG='$E'" >> $OUT
  for rep in 1 2; do
    IIC_HIP_LIB=dbg IIC_DEBUG="$E" python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke ok\|Error\|assert" | tail -2 >> $OUT
  done
done
cat $OUT
