#!/bin/bash
# Development aid: the whole -m gpu suite under a timeout, log into gpurun_out/
out=gpurun_out/gpu_suite.log
timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q > $out 2>&1; echo "rc=$?" >> $out
tail -15 $out
