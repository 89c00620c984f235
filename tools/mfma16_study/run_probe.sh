#!/bin/bash
# run on the GPU box: every case file under build/mfma16 -> gpurun_out/$(basename ${1:-build/mfma16})/<name>.out
set -e
mkdir -p gpurun_out/$(basename ${1:-build/mfma16})
for f in ${1:-build/mfma16}/*.bin; do
  n=$(basename "$f" .bin)
  timeout -k 10 120 tools/mfma16_study/mfma16_probe "$f" "gpurun_out/$(basename ${1:-build/mfma16})/$n.out"
done
ls -la gpurun_out/$(basename ${1:-build/mfma16}) | tail -30
