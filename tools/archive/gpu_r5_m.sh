#!/bin/bash
mkdir -p gpurun_out
timeout 240 python tools/tier_a_hostcost.py maicity > gpurun_out/tier_a_hostcost.log 2>&1
tail -70 gpurun_out/tier_a_hostcost.log
