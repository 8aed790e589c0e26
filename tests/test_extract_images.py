"""tools/extract_images.py — counterpart of the reference's extractimage.py (rosbag -> frame%06d.png), without ROS.
No bag ships with the reference and rosbag is not installed, so the reader is exercised on bags written here, record by
record, after the published v2.0 layout (op codes 3 / 5 / 7 / 2, chunk compression none and bz2, an index record to skip):
PARITY UNPINNED against a real ROS-written bag."""
import bz2
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import extract_images as ex  # noqa: E402


def field(k, v):
    b = k.encode() + b"=" + v
    return struct.pack("<I", len(b)) + b


def record(fields, data):
    h = b"".join(field(k, v) for k, v in fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def image_msg(seq, img, encoding, stamp):
    h, w = img.shape[:2]
    step = img.strides[0]
    frame_id = b"cam"
    return (struct.pack("<III", seq, stamp, 5) + struct.pack("<I", len(frame_id)) + frame_id + struct.pack("<II", h, w) +
            struct.pack("<I", len(encoding)) + encoding.encode() + bytes([0]) + struct.pack("<I", step) +
            struct.pack("<I", img.nbytes) + img.tobytes())


def write_bag(path, topics, compression):
    """topics: {topic: (encoding, [images])}; messages interleaved over the topics, two chunks."""
    conn_recs, msgs = [], []
    for cid, (topic, (enc, imgs)) in enumerate(topics.items()):
        ch = field("topic", topic.encode()) + field("type", b"sensor_msgs/Image") + field("md5sum", b"060021388200f6f0f447d0fcd9c64743")
        conn_recs.append(record([("op", b"\x07"), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], ch))
        for k, im in enumerate(imgs):
            msgs.append((k, cid, record([("op", b"\x02"), ("conn", struct.pack("<I", cid)), ("time", struct.pack("<II", 100 + k, cid))],
                                        image_msg(k, im, enc, 100 + k))))
    msgs.sort(key=lambda m: (m[0], m[1]))
    half = len(msgs) // 2
    out = b"#ROSBAG V2.0\n" + record([("op", b"\x03"), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", len(topics))),
                                      ("chunk_count", struct.pack("<I", 2))], b" " * 64)
    for part in (msgs[:half], msgs[half:]):
        body = b"".join(conn_recs) + b"".join(m[2] for m in part)
        data = bz2.compress(body) if compression == "bz2" else body
        out += record([("op", b"\x05"), ("compression", compression.encode()), ("size", struct.pack("<I", len(body)))], data)
        out += record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", 0)), ("count", struct.pack("<I", 0))], b"")
    open(path, "wb").write(out)


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_extracts_the_requested_topic_in_order(tmp_path, compression):
    from PIL import Image
    rng = np.random.default_rng(3)
    left = [rng.integers(0, 255, (24, 40), dtype=np.uint8) for _ in range(5)]
    right = [rng.integers(0, 255, (24, 40, 3), dtype=np.uint8) for _ in range(5)]
    bag = str(tmp_path / "stereo.bag")
    write_bag(bag, {"/stereo/left/image_raw": ("mono8", left), "/stereo/right/image_raw": ("bgr8", right)}, compression)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_images.py"), bag, str(tmp_path / "left"), "/stereo/left/image_raw"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "Wrote image 4" in r.stdout
    assert ex.main([bag, str(tmp_path / "right"), "/stereo/right/image_raw"]) == 0
    for k in range(5):
        assert np.array_equal(np.asarray(Image.open(tmp_path / "left" / ("frame%06d.png" % k))), left[k])
        # bgr8 data -> imwrite -> the PNG holds R,G,B: reading it back as RGB gives the channels reversed
        assert np.array_equal(np.asarray(Image.open(tmp_path / "right" / ("frame%06d.png" % k)))[..., ::-1], right[k])
    assert len(os.listdir(tmp_path / "left")) == 5


def test_wrong_topic_lists_the_topics_and_rgb8_is_written_swapped(tmp_path, capsys):
    from PIL import Image
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 255, (8, 12, 3), dtype=np.uint8)]
    bag = str(tmp_path / "b.bag")
    write_bag(bag, {"/cam/rgb": ("rgb8", imgs)}, "none")
    assert ex.main([bag, str(tmp_path / "none"), "/nope"]) == 0
    assert "nothing written" in capsys.readouterr().out
    assert ex.main([bag, str(tmp_path / "rgb"), "/cam/rgb"]) == 0
    # passthrough + imwrite: an rgb8 topic lands on disk with red and blue swapped, as with the reference script
    assert np.array_equal(np.asarray(Image.open(tmp_path / "rgb" / "frame000000.png"))[..., ::-1], imgs[0])


def test_extracted_frames_feed_the_cli_reader(tmp_path):
    """The PNGs come out in the layout and format tools/svo_cli.cpp reads (8-bit, non-interlaced, frame%06d.png)."""
    bag = str(tmp_path / "b.bag")
    img = (np.arange(30 * 50) % 251).astype(np.uint8).reshape(30, 50)
    write_bag(bag, {"/l": ("mono8", [img, img])}, "none")
    assert ex.main([bag, str(tmp_path / "left"), "/l"]) == 0
    png = open(tmp_path / "left" / "frame000001.png", "rb").read()
    assert png[:8] == b"\x89PNG\r\n\x1a\n" and png[24] == 8 and png[25] == 0 and png[28] == 0       # depth 8, gray, no interlace
