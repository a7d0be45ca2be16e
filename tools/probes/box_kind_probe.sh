#!/bin/bash
# Which kind of box is this?  (on most boxes of the pool the staggered Winograd first round is worth +4.5 % at
# batch 1, on the others -- which run the unstaggered library ~7 % faster -- it costs 3 %.)  Prints the firmware versions, clocks and
# power state the runtime reports next to an interleaved A/B of the two forms, to correlate over several leases.
cd "$(dirname "$0")/../.."
echo "== $(date -u +%FT%TZ) $(hostname)"
rocm-smi --showfw 2>/dev/null | grep -i -E "MEC|CP|SMC|SDMA|RLC|PSP|VBIOS|IMU" | head -20
rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -30
cat /sys/module/amdgpu/version 2>/dev/null
uname -r
M4D_STAGGER_AUTOTUNE=0 tools/ab_bench.sh 3 "M4D_WINO6_STAGGER_US=0" "M4D_WINO6_STAGGER_US=9" 2>&1 | tail -2
rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power" | head -8
