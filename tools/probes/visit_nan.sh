#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 600 python3 tools/probes/nan_probe.py 7868 > gpurun_out/fuzz/nan.log 2>&1
cut -c1-330 gpurun_out/fuzz/nan.log | tail -40
