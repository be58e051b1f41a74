"""Minimal `plyfile` stand-in (test infrastructure, see README.md): binary little-endian PLY with scalar properties only --
what scene/dataset_readers.py:112-135 and scene/gaussian_model.py:374-430, 486-530 read and write."""
import numpy as np

_PLY2NP = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
           "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}
_NP2PLY = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint"}


class PlyProperty:
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = tuple(PlyProperty(n, data.dtype[n].str.lstrip("<|=")) for n in data.dtype.names)

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements):
        self.elements = list(elements)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def write(self, path):
        with open(path, "wb") as f:
            f.write(b"ply\nformat binary_little_endian 1.0\n")
            for e in self.elements:
                f.write(("element %s %d\n" % (e.name, len(e.data))).encode())
                for n in e.data.dtype.names:
                    f.write(("property %s %s\n" % (_NP2PLY[e.data.dtype[n].str.lstrip("<|=")], n)).encode())
            f.write(b"end_header\n")
            for e in self.elements:
                f.write(np.ascontiguousarray(e.data.astype(e.data.dtype.newbyteorder("<"))).tobytes())

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"ply"
            fmt = f.readline().split()
            assert fmt[1] == b"binary_little_endian", "only binary little-endian PLY is supported by this stand-in"
            elements, cur = [], None
            while True:
                line = f.readline().decode().strip()
                if line == "end_header":
                    break
                tok = line.split()
                if tok[0] == "element":
                    cur = [tok[1], int(tok[2]), []]
                    elements.append(cur)
                elif tok[0] == "property":
                    assert tok[1] != "list", "list properties are not supported by this stand-in"
                    cur[2].append((tok[2], "<" + _PLY2NP[tok[1]]))
            out = []
            for name, count, props in elements:
                dt = np.dtype(props)
                out.append(PlyElement(name, np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count).copy()))
        return PlyData(out)
