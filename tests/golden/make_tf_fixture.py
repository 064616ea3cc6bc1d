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
    stats = {}
    tr = rosbag_tf.read_tf(BAG, stats=stats)
    # what the bag's own index section says (written by the ROS recorder; a second path through the file)
    declared, conn_count, chunk_count = rosbag_tf.declared_counts(BAG)
    assert declared.get("/tf") == stats["messages"], (declared, stats)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "multiple_targets_tf.npz"), **rosbag_tf.to_arrays(tr),
                        tf_messages_declared_by_the_bag_index=np.int64(declared["/tf"]), tf_messages_decoded=np.int64(stats["messages"]),
                        bag_conn_count=np.int64(conn_count), bag_chunk_count=np.int64(chunk_count))
    print("wrote %d transforms of %d /tf messages (the bag's index declares %d; %d connections, %d chunks)" % (
        len(tr), stats["messages"], declared["/tf"], conn_count, chunk_count))
