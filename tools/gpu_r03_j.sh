#!/bin/bash
O=gpurun_out/r03j
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests_full.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/gpu_tests_full.log | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke
