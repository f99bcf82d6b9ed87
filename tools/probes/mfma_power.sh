#!/bin/bash
# clock and power beside pure MFMA streams of the two bf16 shapes (tools/probes/mfma_power.hip)
cd "$(dirname "$0")/../.." || exit 1
BIN=${1:-/tmp/mfma_power}
sample() { for i in $(seq 1 ${1:-4}); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" | sed -e "s/.*sclk clock level: [^(]*(/sclk /" -e "s/Mhz)//" -e "s/.*Power (W): /W /" | tr "\n" " "; echo; sleep 1; done; }
for form in 0 1 0 1; do
  $BIN $form 7 & P=$!; sleep 2.5; sample 4; wait $P
done
