#!/usr/bin/env python3
"""Counterpart of the reference's extractimage.py: turn one image topic of a ROS bag into the
`frame%06d.png` folder the CLI reads — without ROS.  The reference needs rosbag, cv_bridge and cv2 (none in this image);
this reads the bag format (v2.0) directly: records = header fields + data; chunks (op 5, compression none / bz2) hold
connection records (op 7: topic, type) and message records (op 2: connection id, time, payload); sensor_msgs/Image payloads are
deserialised by hand (std_msgs/Header, height, width, encoding, is_bigendian, step, data).

    python tools/extract_images.py <bag_file> <output_dir> <image_topic>        # the reference's argument order

Like the reference (`desired_encoding="passthrough"` + cv2.imwrite), the pixel data is written as it arrives and imwrite's
channel convention applies: 3-channel data is taken as B,G,R (so an `rgb8` topic ends up with red and blue swapped on disk,
exactly as with the reference script).  mono8 / mono16 / bgr8 / rgb8 / bgra8 / rgba8 encodings."""
import bz2
import os
import struct
import sys
import zlib


def _fields(hdr):
    out, o = {}, 0
    while o + 4 <= len(hdr):
        (n,) = struct.unpack_from("<I", hdr, o)
        o += 4
        k, _, v = hdr[o:o + n].partition(b"=")
        out[k.decode()] = v
        o += n
    return out


def _records(buf, o=0, end=None):
    end = len(buf) if end is None else end
    while o + 8 <= end:
        (hl,) = struct.unpack_from("<I", buf, o)
        hdr = _fields(buf[o + 4:o + 4 + hl])
        o += 4 + hl
        (dl,) = struct.unpack_from("<I", buf, o)
        yield hdr, buf[o + 4:o + 4 + dl]
        o += 4 + dl


def read_messages(path, topics=None):
    """Yields (topic, msg_type, time_ns, payload) for every message of the bag, in file order (rosbag.Bag.read_messages)."""
    buf = open(path, "rb").read()
    magic = b"#ROSBAG V2.0\n"
    if not buf.startswith(magic):
        raise ValueError("%s is not a ROS bag v2.0" % path)
    conns = {}

    def walk(data):
        for hdr, body in _records(data):
            op = hdr.get("op", b"\xff")[0]
            if op == 0x05:                                            # chunk
                comp = hdr.get("compression", b"none")
                if comp == b"none":
                    inner = body
                elif comp == b"bz2":
                    inner = bz2.decompress(body)
                else:
                    raise ValueError("chunk compression %r is not supported (none / bz2)" % comp)
                yield from walk(inner)
            elif op == 0x07:                                          # connection
                (cid,) = struct.unpack("<I", hdr["conn"])
                ch = _fields(body)
                conns[cid] = (hdr["topic"].decode(), ch.get("type", b"").decode())
            elif op == 0x02:                                          # message data
                (cid,) = struct.unpack("<I", hdr["conn"])
                secs, nsecs = struct.unpack("<II", hdr["time"])
                topic, mtype = conns.get(cid, ("?", "?"))
                if topics is None or topic in topics:
                    yield topic, mtype, secs * 1000000000 + nsecs, body

    yield from walk(buf[len(magic):])


def decode_image(payload):
    """sensor_msgs/Image -> (height, width, encoding, is_bigendian, step, data bytes, stamp_ns)."""
    o = 0
    seq, secs, nsecs = struct.unpack_from("<III", payload, o); o += 12
    (n,) = struct.unpack_from("<I", payload, o); o += 4 + n            # frame_id
    height, width = struct.unpack_from("<II", payload, o); o += 8
    (n,) = struct.unpack_from("<I", payload, o); o += 4
    encoding = payload[o:o + n].decode(); o += n
    is_be = payload[o]; o += 1
    (step,) = struct.unpack_from("<I", payload, o); o += 4
    (n,) = struct.unpack_from("<I", payload, o); o += 4
    return height, width, encoding, is_be, step, payload[o:o + n], secs * 1000000000 + nsecs


def write_png(path, rows, width, height, channels, depth=8):
    """rows: `height` byte strings of width * channels * depth / 8 bytes, PNG sample order (big-endian for 16 bit)."""
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[channels]

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + bytes(r) for r in rows)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def image_to_png(path, msg):
    height, width, enc, is_be, step, data, _ = msg
    layouts = {"mono8": (1, 1), "8UC1": (1, 1), "bgr8": (3, 1), "rgb8": (3, 1), "8UC3": (3, 1), "bgra8": (4, 1), "rgba8": (4, 1), "8UC4": (4, 1),
               "mono16": (1, 2), "16UC1": (1, 2)}
    if enc not in layouts:
        raise ValueError("encoding %r is not supported" % enc)
    ch, bps = layouts[enc]
    rows = []
    for y in range(height):
        r = data[y * step:y * step + width * ch * bps]
        if bps == 2 and not is_be:                                    # PNG is big-endian
            r = bytes(b for i in range(0, len(r), 2) for b in (r[i + 1], r[i]))
        if ch >= 3:                                                   # cv2.imwrite takes the buffer as B,G,R(,A) whatever the topic said
            px = [r[i:i + ch] for i in range(0, len(r), ch)]
            r = b"".join(p[2:3] + p[1:2] + p[0:1] + p[3:] for p in px)
        rows.append(r)
    write_png(path, rows, width, height, ch, 8 * bps)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        print(__doc__)
        return 2
    bag_file, output_dir, topic = argv
    print("Extract images from %s on topic %s into %s" % (bag_file, topic, output_dir))
    os.mkdir(output_dir)                                              # like the reference: fails if it exists
    count, seen = 0, set()
    for t, mtype, _, payload in read_messages(bag_file):
        seen.add(t)
        if t != topic:
            continue
        image_to_png(os.path.join(output_dir, "frame%06i.png" % count), decode_image(payload))
        print("Wrote image %i" % count)
        count += 1
    if count == 0:
        print("nothing written, make sure your topic is valid?")
        print(sorted(seen))
    return 0


if __name__ == "__main__":
    sys.exit(main())
