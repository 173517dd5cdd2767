#!/bin/bash
cd /root/repo
for s in 10 12 14 0 16 3; do timeout 300 tools/bin/w2d_bench 5 $s "w2d 64,w2d 32" 2>&1 | grep -v "^$" | awk '/^==/ {print} /w2d 64   |w2d 32   |chain|w2d 64 time|\[time\]|mismatch/ {print}' | cut -c1-260; done
