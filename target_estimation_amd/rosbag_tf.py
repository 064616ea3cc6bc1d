"""Minimal reader for `/tf` streams in rosbag v2 files (uncompressed chunks), no ROS needed.

The reference records multi-target data as a bag of `tf2_msgs/TFMessage` on `/tf`
(test/test_multiple_targets.bag: frames `target_0..2` seen from a camera frame) and its node
consumes exactly that stream (src/target_manager_ros.cpp:15-39).  This module extracts the
transforms so that recorded data can be replayed through the measurement ingest
(target_estimation_amd.manager.MeasurementIngest) without ROS.

Format notes (rosbag 2.0): `#ROSBAG V2.0\\n`, then records `<header_len><header><data_len><data>`;
header fields are `<len><name>=<value>`; op 0x05 = chunk (its data is a sequence of records),
0x07 = connection (topic, type), 0x02 = message data (conn id, receive time).  A TFMessage is
`uint32 n` x TransformStamped{ Header{uint32 seq, time stamp, string frame_id}, string
child_frame_id, Vector3 translation, Quaternion rotation(x, y, z, w) } in little-endian ROS
serialisation.
"""
import struct

import numpy as np

OP_MSG, OP_BAG_HEADER, OP_CHUNK, OP_CHUNK_INFO, OP_CONN = 0x02, 0x03, 0x05, 0x06, 0x07


def _fields(header):
    out, pos = {}, 0
    while pos < len(header):
        (ln,) = struct.unpack_from("<I", header, pos)
        pos += 4
        name, _, value = header[pos:pos + ln].partition(b"=")
        out[name.decode()] = value
        pos += ln
    return out


def _records(buf, pos=0, end=None):
    end = len(buf) if end is None else end
    while pos + 4 <= end:
        (hl,) = struct.unpack_from("<I", buf, pos)
        header = buf[pos + 4:pos + 4 + hl]
        pos += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, pos)
        data = buf[pos + 4:pos + 4 + dl]
        pos += 4 + dl
        yield _fields(header), data


def _string(buf, pos):
    (ln,) = struct.unpack_from("<I", buf, pos)
    return buf[pos + 4:pos + 4 + ln].decode(errors="replace"), pos + 4 + ln


def parse_tf_message(data):
    """-> list of (stamp_sec, frame_id, child_frame_id, pose7 [x y z qx qy qz qw])."""
    (n,) = struct.unpack_from("<I", data, 0)
    pos, out = 4, []
    for _ in range(n):
        _seq, sec, nsec = struct.unpack_from("<III", data, pos)
        pos += 12
        frame, pos = _string(data, pos)
        child, pos = _string(data, pos)
        vals = struct.unpack_from("<7d", data, pos)
        pos += 56
        # toSec(), include/target_estimation/utils.hpp:59-62
        out.append((float(sec) + 1e-9 * float(nsec), frame, child, np.array(vals, dtype=np.float64)))
    if pos != len(data):   # the record's data_len was written by the recorder: the parse must consume it exactly
        raise ValueError("TFMessage of %d bytes parsed to %d" % (len(data), pos))
    return out


def declared_counts(path):
    """What the bag says about itself, read from its INDEX section (not from the chunks read_tf walks): the bag header
    record (op 0x03: index_pos, conn_count, chunk_count), the connection records behind index_pos and the chunk-info
    records (op 0x06), whose data lists (connection id, message count) pairs per chunk.  Returns
    {topic: messages the recorder counted}, conn_count, chunk_count."""
    buf = open(path, "rb").read()
    magic = b"#ROSBAG V2.0\n"
    if not buf.startswith(magic):
        raise ValueError("not a rosbag v2 file: %s" % path)
    first = next(_records(buf, len(magic)))[0]
    if first.get("op", b"\xff")[0] != OP_BAG_HEADER:
        raise ValueError("the first record is not the bag header")
    index_pos = struct.unpack("<Q", first["index_pos"])[0]
    conn_count = struct.unpack("<I", first["conn_count"])[0]
    chunk_count = struct.unpack("<I", first["chunk_count"])[0]
    topics, counts, chunks = {}, {}, 0
    for fields, data in _records(buf, index_pos):
        op = fields.get("op", b"\xff")[0]
        if op == OP_CONN:
            topics[struct.unpack("<I", fields["conn"])[0]] = fields["topic"].decode()
        elif op == OP_CHUNK_INFO:
            chunks += 1
            n = struct.unpack("<I", fields["count"])[0]
            for k in range(n):
                conn, cnt = struct.unpack_from("<II", data, 8 * k)
                counts[conn] = counts.get(conn, 0) + cnt
    if len(topics) != conn_count or chunks != chunk_count:
        raise ValueError("index section: %d connections / %d chunk infos, header says %d / %d" % (len(topics), chunks, conn_count, chunk_count))
    per_topic = {}
    for conn, cnt in counts.items():
        per_topic[topics[conn]] = per_topic.get(topics[conn], 0) + cnt
    return per_topic, conn_count, chunk_count


def read_tf(path, topic="/tf", stats=None):
    """All transforms of `topic`, in bag order: list of dicts with keys
    recv_time, stamp, frame_id, child_frame_id, pose (np.ndarray[7]).  stats (a dict) receives
    messages = the number of TFMessage records decoded."""
    buf = open(path, "rb").read()
    if not buf.startswith(b"#ROSBAG V2.0\n"):
        raise ValueError("not a rosbag v2 file: %s" % path)
    conns, out = {}, []

    def handle(fields, data):
        op = fields.get("op", b"\xff")[0]
        if op == OP_CONN:
            conn = struct.unpack("<I", fields["conn"])[0]
            info = _fields(data)
            conns[conn] = (fields["topic"].decode(), info.get("type", b"").decode())
        elif op == OP_MSG:
            conn = struct.unpack("<I", fields["conn"])[0]
            tp, ty = conns.get(conn, ("", ""))
            if tp == topic and ty == "tf2_msgs/TFMessage":
                sec, nsec = struct.unpack("<II", fields["time"])
                recv = float(sec) + 1e-9 * float(nsec)
                if stats is not None:
                    stats["messages"] = stats.get("messages", 0) + 1
                for stamp, frame, child, pose in parse_tf_message(data):
                    out.append(dict(recv_time=recv, stamp=stamp, frame_id=frame, child_frame_id=child, pose=pose))
        elif op == OP_CHUNK:
            if fields.get("compression", b"none") != b"none":
                raise ValueError("compressed chunks are not supported (%r)" % fields.get("compression"))
            for f2, d2 in _records(data):
                handle(f2, d2)

    for fields, data in _records(buf, len(b"#ROSBAG V2.0\n")):
        handle(fields, data)
    return out


def to_arrays(transforms):
    """Column arrays (for storing as a fixture): recv_time, stamp [n], pose [n,7], and the frame names
    as fixed-width byte strings."""
    n = len(transforms)
    return dict(recv_time=np.array([t["recv_time"] for t in transforms]),
                stamp=np.array([t["stamp"] for t in transforms]),
                pose=np.array([t["pose"] for t in transforms]).reshape(n, 7),
                frame_id=np.array([t["frame_id"] for t in transforms], dtype="S64"),
                child_frame_id=np.array([t["child_frame_id"] for t in transforms], dtype="S64"))


def replay(transforms, ingest, dt, on_tick=None):
    """Feed recorded transforms to a MeasurementIngest-like object at the node's loop period dt
    (src/target_node.cpp:32-44): every message is pushed when its receive time has passed, then one
    tick runs.  A malformed target frame name drops the rest of its message, as the reference's
    callback does (src/target_manager_ros.cpp:34-35)."""
    if not transforms:
        return 0
    t, i, ticks = transforms[0]["recv_time"], 0, 0
    end = transforms[-1]["recv_time"]
    skip_recv = None
    while t <= end + dt:
        while i < len(transforms) and transforms[i]["recv_time"] <= t:
            tr = transforms[i]
            i += 1
            if skip_recv == (tr["recv_time"], tr["stamp"]):
                continue
            if ingest.push_named(tr["child_frame_id"], tr["stamp"], tr["pose"]) < 0:
                skip_recv = (tr["recv_time"], tr["stamp"])
        result = ingest.tick(dt, t)
        if on_tick is not None:
            on_tick(ticks, t, result)
        ticks += 1
        t += dt
    return ticks
