#!/bin/bash
timeout 1200 python tools/step_repro_soak.py --iters 300 2>&1 | grep -v amdgpu.ids | tail -12
