"""Point-cloud file helpers of the JRDB loader.

The reference star-imports ``load_pointcloud`` and ``get_lidar_with_sweeps`` from ``data/utils/utils.py``
(dataloader.py:8, 121-129), a module that is NOT in the reference repository.  PARITY UNPINNED for both; what is built here:

* ``load_pointcloud(url)`` -- reads a Point Cloud Data file (the public PCD format, v0.7: ASCII header, ``DATA ascii`` /
  ``binary`` / ``binary_compressed`` body) and returns (N, 4) float32 rows [x, y, z, intensity] (intensity 0 when the file has
  no such field).
* ``get_lidar_with_sweeps(pc, num_points)`` -- a cloud of exactly ``num_points`` rows: a random subset without replacement
  (rows kept in their original order) when the cloud is larger, the whole cloud plus randomly repeated rows when it is smaller
  (the convention of pcdet's ``sample_points``, data_processor.py:181-211).  ``num_points <= 0`` returns the cloud as is.
"""
import struct

import numpy as np

_NP_TYPES = {("F", 4): "f4", ("F", 8): "f8", ("I", 1): "i1", ("I", 2): "i2", ("I", 4): "i4", ("I", 8): "i8",
             ("U", 1): "u1", ("U", 2): "u2", ("U", 4): "u4", ("U", 8): "u8"}


def _lzf_decompress(data, out_len):
    """LZF (the codec of ``DATA binary_compressed``): literal runs and back references, as published with liblzf."""
    out = bytearray(out_len)
    i, o, n = 0, 0, len(data)
    while i < n:
        ctrl = data[i]
        i += 1
        if ctrl < 32:                                             # literal run of ctrl + 1 bytes
            run = ctrl + 1
            out[o:o + run] = data[i:i + run]
            i += run
            o += run
        else:                                                     # back reference
            length = ctrl >> 5
            if length == 7:
                length += data[i]
                i += 1
            ref = o - ((ctrl & 0x1f) << 8) - data[i] - 1
            i += 1
            for _ in range(length + 2):                           # may overlap its own output
                out[o] = out[ref]
                o += 1
                ref += 1
    if o != out_len:
        raise ValueError("corrupt binary_compressed PCD body")
    return bytes(out)


def read_pcd(url):
    """-> dict field name -> (N,) array (fields with COUNT > 1 come back as (N, COUNT))."""
    with open(url, "rb") as f:
        raw = f.read()
    head, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        head[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = head["FIELDS"]
    sizes = [int(v) for v in head["SIZE"]]
    types = head["TYPE"]
    counts = [int(v) for v in head.get("COUNT", ["1"] * len(fields))]
    n = int(head["POINTS"][0]) if "POINTS" in head else int(head["WIDTH"][0]) * int(head["HEIGHT"][0])
    dtype = np.dtype([(name, "<" + _NP_TYPES[(t, s)], (c,) if c > 1 else ()) for name, t, s, c in zip(fields, types, sizes, counts)])
    kind = head["DATA"][0].lower()
    if kind == "ascii":
        rows = np.loadtxt(raw[pos:].decode("ascii").splitlines(), dtype=np.float64, ndmin=2) if n else np.zeros((0, sum(counts)))
        out, col = {}, 0
        for name, t, s, c in zip(fields, types, sizes, counts):
            a = rows[:n, col:col + c].astype(_NP_TYPES[(t, s)])
            out[name] = a[:, 0] if c == 1 else a
            col += c
        return out
    if kind == "binary":
        arr = np.frombuffer(raw, dtype=dtype, count=n, offset=pos)
        return {name: arr[name] for name in fields}
    if kind == "binary_compressed":                               # field-major (structure of arrays) after decompression
        comp_len, out_len = struct.unpack_from("<II", raw, pos)
        body = _lzf_decompress(raw[pos + 8:pos + 8 + comp_len], out_len)
        out, at = {}, 0
        for name, t, s, c in zip(fields, types, sizes, counts):
            a = np.frombuffer(body, dtype="<" + _NP_TYPES[(t, s)], count=n * c, offset=at)
            at += n * c * s
            out[name] = a if c == 1 else a.reshape(n, c)
        return out
    raise ValueError("unknown PCD DATA kind %r" % kind)


def write_pcd(url, points, fields=("x", "y", "z", "intensity"), data="binary"):
    """points (N, len(fields)) float32 -> a PCD v0.7 file (``data`` = 'binary' | 'ascii')."""
    points = np.ascontiguousarray(points, dtype="<f4")
    n, c = points.shape
    assert c == len(fields)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\n"
            "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (" ".join(fields), " ".join(["4"] * c), " ".join(["F"] * c),
                                                               " ".join(["1"] * c), n, n, data))
    with open(url, "wb") as f:
        f.write(head.encode("ascii"))
        if data == "binary":
            f.write(points.tobytes())
        elif data == "ascii":
            for row in points:
                f.write((" ".join(repr(float(v)) for v in row) + "\n").encode("ascii"))
        else:
            raise ValueError(data)


def load_pointcloud(url):
    """-> (N, 4) float32 [x, y, z, intensity]."""
    f = read_pcd(url)
    n = len(f["x"])
    pc = np.zeros((n, 4), dtype=np.float32)
    pc[:, 0], pc[:, 1], pc[:, 2] = f["x"], f["y"], f["z"]
    for name in ("intensity", "i"):
        if name in f:
            pc[:, 3] = f[name]
            break
    return pc


def get_lidar_with_sweeps(pc, num_points):
    """pc (N, C) -> (num_points, C) (see the module docstring); draws from numpy's global generator, as the loader's other
    random steps do."""
    n = pc.shape[0]
    if num_points is None or num_points <= 0 or n == num_points or n == 0:
        return pc
    if n > num_points:
        keep = np.sort(np.random.choice(n, num_points, replace=False))
        return pc[keep]
    extra = np.random.choice(n, num_points - n, replace=(num_points - n) > n)
    return np.concatenate([pc, pc[extra]], axis=0)
