#!/usr/bin/env python3
"""Extract the 256x4 rBRIEF sampling-pattern CONSTANTS (data, not code) from the
reference (src/ORBextractor.cc:147-404, `bit_pattern_31_`) into a plain number
table shared by the oracle and the HIP kernels.  Run once in the build container
(the reference is not present on the GPU box); the output is committed.

    python tools/extract_pattern.py /root/reference > morb_slam_amd/csrc/orb_pattern.inc
"""
import re, sys
src = open(sys.argv[1] + "/src/ORBextractor.cc").read()
m = re.search(r"bit_pattern_31_\[256 \* 4\] = \{(.*?)\};", src, re.S)
body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
vals = [int(v) for v in re.findall(r"-?\d+", body)]
assert len(vals) == 1024, len(vals)
print("// 256 test pairs (x0,y0,x1,y1) of the 31x31 rBRIEF pattern; data extracted by")
print("// tools/extract_pattern.py from the reference table bit_pattern_31_ (ORBextractor.cc:147-404).")
for i in range(0, 1024, 16):
    print(",".join(str(v) for v in vals[i:i + 16]) + ",")
