#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box: tools/probes/fetch_calib under rocprofv3 --pmc.
#   tools/fetch_calib.sh  ->  gpurun_out/r02_fetch_calibration.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/calib
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/calib -o c -- $ROOT/tools/probes/fetch_calib > $OUT/r02_fetch_calibration.txt 2>&1
F=$(find /tmp/calib -name "*counter_collection.csv" | head -1)
python - "$F" >> $OUT/r02_fetch_calibration.txt <<'PY'
import csv, sys
tot = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "fetch_calib" in r["Kernel_Name"]:
        tot[r["Kernel_Name"]] = tot.get(r["Kernel_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k[:60]:60s} FETCH_SIZE = {v:.0f} KiB = {v * 1024 / 2**31:.3f} x the 2 GiB actually read")
PY
cat $OUT/r02_fetch_calibration.txt | grep -v "^W\|^E\|^I" | tail -8
