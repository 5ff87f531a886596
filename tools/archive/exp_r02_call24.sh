#!/bin/bash
# Statistics on the device (HIP build) + the C host + the snapshot path after the h5lite change
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_statistics.py tests/test_snapshots.py tests/test_c_host.py -m gpu -q -x 2>&1 | tail -4
