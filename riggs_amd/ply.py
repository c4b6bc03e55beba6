"""Minimal PLY codec for the Gaussian checkpoints (``point_cloud.ply``) the reference writes with the third-party
``plyfile`` package (/root/reference/scene/gaussian_model.py:253-272 ``save_ply``, :279-336 ``load_ply``; ``plyfile`` is
a pip dependency that is not vendored, so parity is anchored on the published PLY format and the reference's call sites:
one ``vertex`` element, every property ``float`` (numpy 'f4'), ``binary_little_endian 1.0`` as ``PlyData([el]).write``
emits on little-endian hosts).  The reader accepts the general single-element case (any scalar property types, ascii or
binary, either byte order) so files from upstream 3DGS tooling load too."""
from __future__ import annotations

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def write_vertex_ply(path: str, names, columns: np.ndarray) -> None:
    """``columns`` (N, len(names)) float32 -> binary little-endian PLY with one float property per name."""
    columns = np.ascontiguousarray(columns, dtype="<f4")
    if columns.ndim != 2 or columns.shape[1] != len(names):
        raise ValueError("columns must be (N, %d)" % len(names))
    header = ["ply", "format binary_little_endian 1.0", "element vertex %d" % columns.shape[0]]
    header += ["property float %s" % n for n in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(columns.tobytes())


def read_vertex_ply(path: str):
    """Returns (names, structured array) of the ``vertex`` element."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if in_vertex:
                    raise ValueError("only single-element (vertex) PLY files are supported")
                if tok[1] != "vertex":
                    raise ValueError("first element must be 'vertex', got %r" % tok[1])
                count, in_vertex = int(tok[2]), True
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError("list properties are not supported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError("incomplete PLY header")
        if fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float64, ndmin=2)[:count]
            arr = np.empty(count, dtype=[(n, t) for n, t in props])
            for i, (n, _) in enumerate(props):
                arr[n] = data[:, i]
        else:
            bo = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, bo + t) for n, t in props])
            arr = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
    return [n for n, _ in props], arr
