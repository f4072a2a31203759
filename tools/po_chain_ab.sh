# A/B of the edge-order chain's two forms (run through gpurun after: python tools/ab_build.py po2v1 optimizer.hip -DMORB_PO2_CHAIN_ONE_WAIT)
for i in 1 2 3; do
 python tools/pose_opt_modes.py 2>/dev/null | grep exact | sed 's/^/default /'
 MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_po2v1.so python tools/pose_opt_modes.py 2>/dev/null | grep exact | sed 's/^/one-wait /'
done
python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-90
MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_po2v1.so python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-90
python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-90
MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_po2v1.so python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-90
