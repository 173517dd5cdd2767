#!/bin/bash
cd /root/repo
for s in 10 12 14 7 15 0 1 18; do timeout 300 tools/bin/w2d_bench 5 $s "w2d 64 defer,w2d 32 defer,dfr2,w2d 64 ,w2d 32 " 2>&1 | grep -v "^$" | awk '/^==/ {print} /w2d 64   |w2d 32   |defer  |dfr2|mismatch/ {print}' | cut -c1-150; done
