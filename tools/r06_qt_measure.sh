#!/bin/bash
# round 6: quadtree latency at BASELINE configs[3]'s shape (one 1920 x 1080 / 4000-feature stereo frame) and at configs[1]'s, product build + timing build
out=gpurun_out/r06/qt_$1
mkdir -p $out
export MORB_W=1920 MORB_H=1080 MORB_NF=4000
python tools/latency_b1.py > $out/latency_c4.txt 2>&1
python tools/fast_phases.py 1 > $out/phases_c4_b1.txt 2>&1
unset MORB_W MORB_H MORB_NF
python tools/latency_b1.py > $out/latency_c2.txt 2>&1
python tools/fast_phases.py 1 > $out/phases_c2_b1.txt 2>&1
tail -n +1 $out/latency_c4.txt $out/phases_c4_b1.txt $out/latency_c2.txt | grep -v amdgpu.ids
