""".vox volume files (reference io.clj:9-33).

Layout: ASCII magic ``VOXEL`` (5 B), three big-endian int32 resolutions
(java.io.DataOutputStream.writeInt), one byte element size (=1), then the raw
volume bytes, x fastest.  Header = 18 bytes, so a 512^3 file is the "134MB" the
reference README quotes.
"""
import struct

import numpy as np

MAGIC = b"VOXEL"
HEADER_SIZE = 18


def save_volume(path, res, voxels):
    """Write ``voxels`` (uint8/int8, len rx*ry*rz).  ``res`` is an int (the
    reference writes it three times, io.clj:12-15) or an (rx, ry, rz) triple."""
    if isinstance(res, (int, np.integer)):
        res = (int(res),) * 3
    rx, ry, rz = (int(r) for r in res)
    data = np.ascontiguousarray(voxels).view(np.uint8).reshape(-1)
    if data.size != rx * ry * rz:
        raise ValueError(f"volume has {data.size} bytes, header says {rx}x{ry}x{rz}")
    with open(path, "wb") as out:
        out.write(MAGIC)
        out.write(struct.pack(">iii", rx, ry, rz))
        out.write(struct.pack("b", 1))
        out.write(data.tobytes())


def load_volume(path):
    """-> (uint8 array of rx*ry*rz, (rx, ry, rz)).  Like the reference the
    magic is skipped rather than validated strictly; a wrong magic or a short
    file raises ValueError instead of rendering garbage."""
    with open(path, "rb") as f:
        head = f.read(HEADER_SIZE)
        if len(head) < HEADER_SIZE or head[:5] != MAGIC:
            raise ValueError(f"{path}: not a VOXEL volume file")
        rx, ry, rz = struct.unpack(">iii", head[5:17])
        elem = head[17]
        if elem != 1 or min(rx, ry, rz) <= 0:
            raise ValueError(f"{path}: unsupported header res=({rx},{ry},{rz}) elem={elem}")
        n = rx * ry * rz
        vox = np.fromfile(f, dtype=np.uint8, count=n)
    if vox.size != n:
        raise ValueError(f"{path}: truncated volume ({vox.size} of {n} bytes)")
    return vox, (rx, ry, rz)
