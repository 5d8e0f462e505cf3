#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nmt.py -x -q > gpurun_out/nmt.log 2>&1
tail -5 gpurun_out/nmt.log
python -m pytest tests/test_gpu_fullsize_decode.py -x -q -k "timeout or timed" > gpurun_out/dec.log 2>&1
tail -3 gpurun_out/dec.log
