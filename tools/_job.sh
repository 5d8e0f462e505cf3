#!/bin/bash
python tools/_dbg.py 2>&1 | grep -v amdgpu.ids | tail -20
