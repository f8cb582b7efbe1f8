mkdir -p gpurun_out/r03e
python tools/run_reference_slam.py --synthetic 120 --limit30 --timeout 300 --log gpurun_out/r03e/rep120.log > gpurun_out/r03e/rep120.json 2> gpurun_out/r03e/rep120.err; tail -c 600 gpurun_out/r03e/rep120.json; echo
python tools/run_reference_slam.py --synthetic 60 --shape tum --noise --limit30 --timeout 300 --log gpurun_out/r03e/tum60.log > gpurun_out/r03e/tum60.json 2> gpurun_out/r03e/tum60.err; tail -c 600 gpurun_out/r03e/tum60.json; echo
python -m pytest tests -m gpu -x -q --deselect tests/test_reference_slam_gpu.py 2>&1 | tail -15 > gpurun_out/r03e/pytest.log; tail -8 gpurun_out/r03e/pytest.log
