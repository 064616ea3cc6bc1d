#!/usr/bin/env python3
"""Extract the /tf stream of the reference's recorded bag (test/test_multiple_targets.bag) into
tests/golden/multiple_targets_tf.npz with target_estimation_amd/rosbag_tf.py.  The fixture is data
(stamps, frame names, poses); the bag itself is not copied.  Run where /root/reference exists."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from target_estimation_amd import rosbag_tf  # noqa: E402

BAG = "/root/reference/test/test_multiple_targets.bag"

if __name__ == "__main__":
    tr = rosbag_tf.read_tf(BAG)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "multiple_targets_tf.npz"), **rosbag_tf.to_arrays(tr))
    print("wrote %d transforms" % len(tr))
