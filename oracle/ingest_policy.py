"""Pure-Python restatement of the reference's ROS ingest policy -- TEST INFRASTRUCTURE ONLY.

class Measurement (include/target_estimation/target_manager_ros.hpp:74-134), the /tf callback's id
parsing (src/target_manager_ros.cpp:26-39, utils.hpp:273-313) and RosTargetManager::update
(src/target_manager_ros.cpp:41-92), driving one CPU-oracle target per id.  ROS transport is replaced
by explicit push()/tick(now) calls."""
import numpy as np

from .oracle import OracleTarget


def get_id(s):
    """utils.hpp:302-313: 'xxx_id' -> id, exactly two '_'-separated tokens."""
    parts = s.split("_")
    if len(parts) != 2:
        return None
    digits = ""
    for ch in parts[1]:
        if ch.isdigit() or (ch in "+-" and not digits):
            digits += ch
        else:
            break
    try:
        return int(digits)
    except ValueError:
        return None


class Mailbox:                                   # class Measurement
    def __init__(self):
        self.new_meas = True                     # :78-82
        self.last_meas_time = 0.0
        self.stamp = 0.0
        self.pose = np.zeros(7)

    def update(self, stamp, pose):               # :96-114
        if stamp > self.stamp:
            self.new_meas = True
            self.last_meas_time = stamp
        else:
            self.new_meas = False
        self.stamp = stamp
        self.pose = np.array(pose, dtype=np.float64)


class RefIngest:
    def __init__(self, model, Q, R, P0, expiration_time=1000.0, token="target"):
        self.model, self.Q, self.R, self.P0 = model, Q, R, P0
        self.expiration_time = expiration_time
        self.token = token
        self.t = 0.0
        self.mail = {}
        self.targets = {}

    def push(self, id, stamp, pose):
        self.mail.setdefault(id, Mailbox()).update(stamp, pose)

    def push_named(self, frame, stamp, pose):    # measurementCallBack, one transform
        if self.token not in frame:
            return 0
        id = get_id(frame)
        if id is None:
            return -1
        self.push(id, stamp, pose)
        return 1

    def tick(self, dt, now):                     # RosTargetManager::update
        for id in sorted(self.mail):
            mb = self.mail[id]
            last = mb.last_meas_time
            if mb.new_meas:                      # Measurement::read does not clear the flag
                if id not in self.targets:
                    self.targets[id] = OracleTarget(self.model, self.Q, self.R, self.P0, mb.pose, dt, self.t)
                self.targets[id].add_measurement(dt, mb.pose)
            elif id in self.targets:
                self.targets[id].update(dt)
            if last > 0.0 and (now - last) >= self.expiration_time:
                del self.mail[id]
                self.targets.pop(id, None)
        ids = sorted(self.targets)
        poses = np.array([self.targets[i].pose()[0] for i in ids]).reshape(-1, 7)
        self.t += dt
        return np.array(ids, dtype=np.uint32), poses
