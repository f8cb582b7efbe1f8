#!/bin/bash
# Real-data end-to-end numbers with the UNTOUCHED reference driving this repo's drop-in packages (BASELINE configs[2], [3]):
# runs gs_icp_slam_unlimit.py on Replica room0 / office0 and TUM fr1_desk if they are on this machine, with the flags of
# replica.sh / tum.sh, and prints the statistics the reference prints (System FPS, ATE RMSE, PSNR).  Prints "not measured" for
# every dataset that is absent — never an estimate.
#   usage: bash tools/run_reference.sh [DATA_ROOT]      (default DATA_ROOT: ./dataset, the reference's own default)
DATA=${1:-dataset}
HERE=$(cd "$(dirname "$0")/.." && pwd)
REF=${GSICP_REFERENCE:-/root/reference}
[ -d "$REF" ] || REF=$HERE/oracle/_ref/refpy
for scene in room0 office0; do
  python "$HERE/tools/run_reference_slam.py" --dataset "$DATA/Replica/$scene" --config "$REF/configs/Replica/caminfo.txt" --timeout 3600
done
# TUM: the harness reads the dataset type off the config's third line and passes tum.sh's flags [REF tum.sh:135-142]; utils/traj_utils.py uses
# np.unicode_ (removed in NumPy 2), which tests/refstubs/sitecustomize.py restores by environment — the file is not edited.  The
# reference's TUM loader branch is exercised on a synthetic TUM-layout sequence by tests/test_reference_slam_gpu.py.
python "$HERE/tools/run_reference_slam.py" --dataset "$DATA/TUM/rgbd_dataset_freiburg1_desk" --config "$REF/configs/TUM/rgbd_dataset_freiburg1_desk.txt" --timeout 3600
