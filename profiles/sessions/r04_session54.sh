#!/bin/bash
# round 4, GPU session 54: thin products on generation 1 by shape -- parity, the thin products without the override, the plans that end in a thin block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_residency.py -x -q -m gpu > $O/s54_pytest.log 2>&1
tail -3 $O/s54_pytest.log
for shape in "464 66000 66000" "232 33000 33000" "848 50000 50000" "1000 33000 33000" "1000 16384 16384" "464 16384 16384" "464 65536 4096" "1699 50021 70017"; do
  timeout 300 python tools/prof_product.py $shape 20 >> $O/s54_thin.log 2>&1
done
grep shape $O/s54_thin.log | sed 's/pass bytes.*leaf /leaf /' | sed 's/, C checksum.*//'
timeout 900 python tools/depth_model_sweep.py 66000,66000,66000 33000,33000,33000 50000,50000,50000 65664,65664,65664 > $O/s54_sweep.log 2>&1
cut -c1-330 $O/s54_sweep.log
