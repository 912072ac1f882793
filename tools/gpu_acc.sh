#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/nd_accuracy.py acrobot 40 2>&1 | tail -12 | tee gpurun_out/nd_accuracy_acrobot.log
timeout 200 python tools/nd_accuracy.py mini_cheetah 40 2>&1 | tail -12 | tee gpurun_out/nd_accuracy.log
timeout 200 python tools/nd_accuracy.py hopper 50 2>&1 | tail -12 | tee gpurun_out/nd_accuracy_hopper.log
