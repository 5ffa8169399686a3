"""Minimal `plyfile`-compatible PLY reader / writer -- just the surface the reference uses, so that
its unmodified Python (`scene/gaussian_model.py:18,239-311,398-483`, `scene/dataset_readers.py:22,107-130`,
`update_old_ply_format.py`) imports and runs with this directory on PYTHONPATH; the third-party `plyfile` package the
reference depends on is not installed in this image.

Written from the PLY format description (header: `ply` / `format <ascii|binary_little_endian|binary_big_endian> 1.0` /
`comment ...` / `element <name> <count>` / `property <type> <name>` / `end_header`, then the elements' rows in order),
not from plyfile's sources.  Supported: scalar properties of the eight PLY number types in the three encodings; list
properties are read in ascii / binary files into object arrays and written back (the reference never uses them).

    PlyElement.describe(structured_array, name)          -> PlyElement
    PlyData(elements, text=False, byte_order='<')         .write(path_or_file)
    PlyData.read(path_or_file)                            -> PlyData;   data['vertex'], data.elements[i], 'name' in data
    element['x'] / element.data / element.count / element.name / element.properties / 'x' in element

What the reference stores with it (gaussian_model.py:239-311): one element `vertex_<d>` per SH degree d = 0..max with
properties x y z (float, or half bit-cast to short), f_dc_0..2, f_rest_0..3((d+1)^2-1)-1 in channel-major order,
opacity, scale_0..2, rot_0..3 (float | short | uchar codebook indices), plus, when quantised, a 256-row element
`codebook_centers` with one column per codebook (features_dc, features_rest_0..14, opacity, scaling, rotation_re,
rotation_im).
"""
import io

import numpy as np

__all__ = ["PlyData", "PlyElement", "PlyProperty", "PlyListProperty", "PlyParseError"]

# PLY type name (and its sized alias) -> numpy kind+size
_PLY_TO_NP = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
    "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
    "double": "f8", "float64": "f8",
}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
              "f8": "double"}
_FORMATS = {"ascii": None, "binary_little_endian": "<", "binary_big_endian": ">"}


class PlyParseError(Exception):
    pass


class PlyProperty:
    def __init__(self, name, val_dtype):
        self.name = str(name)
        self.val_dtype = _PLY_TO_NP.get(val_dtype, val_dtype)
        if self.val_dtype not in _NP_TO_PLY:
            raise ValueError(f"unsupported property type {val_dtype!r}")

    def dtype(self, byte_order="="):
        return np.dtype(byte_order + self.val_dtype)

    def header_line(self):
        return f"property {_NP_TO_PLY[self.val_dtype]} {self.name}"

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {_NP_TO_PLY[self.val_dtype]!r})"


class PlyListProperty(PlyProperty):
    def __init__(self, name, len_dtype, val_dtype):
        super().__init__(name, val_dtype)
        self.len_dtype = _PLY_TO_NP.get(len_dtype, len_dtype)

    def header_line(self):
        return f"property list {_NP_TO_PLY[self.len_dtype]} {_NP_TO_PLY[self.val_dtype]} {self.name}"


class PlyElement:
    """One PLY element: a name and a structured numpy array with one field per property."""

    def __init__(self, name, properties, count, data=None):
        self.name = str(name)
        self.properties = tuple(properties)
        self.count = int(count)
        self.data = data

    @staticmethod
    def describe(data, name, len_types=None, val_types=None, comments=None):
        if not isinstance(data, np.ndarray) or data.dtype.names is None or data.ndim != 1:
            raise TypeError("describe() needs a one-dimensional structured numpy array")
        props = []
        for field in data.dtype.names:
            dt = data.dtype.fields[field][0]
            if dt.kind == "O":
                props.append(PlyListProperty(field, (len_types or {}).get(field, "u1"),
                                             (val_types or {}).get(field, "i4")))
            else:
                if dt.shape:
                    raise ValueError(f"field {field!r}: only scalar (or list) properties are supported")
                props.append(PlyProperty(field, dt.kind + str(dt.itemsize)))
        return PlyElement(name, props, len(data), data)

    def _has_lists(self):
        return any(isinstance(p, PlyListProperty) for p in self.properties)

    def _row_dtype(self, byte_order):
        return np.dtype([(p.name, p.dtype(byte_order)) for p in self.properties])

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, value):
        self.data[key] = value

    def __contains__(self, name):
        return any(p.name == name for p in self.properties)

    def __len__(self):
        return self.count

    def ply_property(self, name):
        for p in self.properties:
            if p.name == name:
                return p
        raise KeyError(name)

    def header_lines(self):
        return [f"element {self.name} {self.count}"] + [p.header_line() for p in self.properties]

    def __repr__(self):
        return f"PlyElement({self.name!r}, {self.properties!r}, count={self.count})"

    # ---- body encoding ------------------------------------------------------------------------------------------
    def _write(self, stream, text, byte_order):
        if text:
            for row in self.data:
                out = []
                for p in self.properties:
                    v = row[p.name]
                    if isinstance(p, PlyListProperty):
                        out.append(str(len(v)))
                        out.extend(repr(x.item()) if hasattr(x, "item") else repr(x) for x in v)
                    else:
                        out.append(repr(v.item()))
                stream.write((" ".join(out) + "\n").encode("ascii"))
        elif not self._has_lists():
            stream.write(np.ascontiguousarray(self.data.astype(self._row_dtype(byte_order), copy=False)).tobytes())
        else:
            for row in self.data:
                for p in self.properties:
                    if isinstance(p, PlyListProperty):
                        v = np.asarray(row[p.name], dtype=np.dtype(byte_order + p.val_dtype))
                        stream.write(np.array(len(v), dtype=np.dtype(byte_order + p.len_dtype)).tobytes())
                        stream.write(v.tobytes())
                    else:
                        stream.write(np.array(row[p.name], dtype=p.dtype(byte_order)).tobytes())

    def _read(self, stream, text, byte_order):
        native = np.dtype([(p.name, "O" if isinstance(p, PlyListProperty) else p.dtype("="))
                           for p in self.properties])
        if text:
            self.data = np.empty(self.count, dtype=native)
            for k in range(self.count):
                line = stream.readline()
                if not line:
                    raise PlyParseError(f"element {self.name}: early end of file at row {k}")
                tok = line.split()
                at = 0
                for p in self.properties:
                    if isinstance(p, PlyListProperty):
                        n = int(tok[at])
                        self.data[p.name][k] = np.array(tok[at + 1:at + 1 + n], dtype=np.dtype(p.val_dtype)
                                                        if p.val_dtype[0] != "f" else float).astype(p.val_dtype)
                        at += 1 + n
                    else:
                        self.data[p.name][k] = (float if p.val_dtype[0] == "f" else int)(tok[at])
                        at += 1
        elif not self._has_lists():
            dt = self._row_dtype(byte_order)
            raw = stream.read(dt.itemsize * self.count)
            if len(raw) != dt.itemsize * self.count:
                raise PlyParseError(f"element {self.name}: early end of file")
            self.data = np.frombuffer(raw, dtype=dt, count=self.count).astype(native)
        else:
            self.data = np.empty(self.count, dtype=native)
            for k in range(self.count):
                for p in self.properties:
                    if isinstance(p, PlyListProperty):
                        ld = np.dtype(byte_order + p.len_dtype)
                        n = int(np.frombuffer(stream.read(ld.itemsize), dtype=ld)[0])
                        vd = np.dtype(byte_order + p.val_dtype)
                        self.data[p.name][k] = np.frombuffer(stream.read(vd.itemsize * n), dtype=vd).astype(p.val_dtype)
                    else:
                        d = p.dtype(byte_order)
                        self.data[p.name][k] = np.frombuffer(stream.read(d.itemsize), dtype=d)[0]


class PlyData:
    """A PLY file: an ordered list of elements (+ comments)."""

    def __init__(self, elements=(), text=False, byte_order="=", comments=(), obj_info=()):
        if byte_order == "=":
            byte_order = "<" if np.little_endian else ">"
        if byte_order not in ("<", ">"):
            raise ValueError("byte_order must be '<', '>' or '='")
        self.elements = list(elements)
        self.text = bool(text)
        self.byte_order = byte_order
        self.comments = list(comments)
        self.obj_info = list(obj_info)
        names = [e.name for e in self.elements]
        if len(set(names)) != len(names):
            raise ValueError("two elements with the same name")

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name):
        return any(e.name == name for e in self.elements)

    def __len__(self):
        return len(self.elements)

    def __iter__(self):
        return iter(self.elements)

    @property
    def header(self):
        fmt = "ascii" if self.text else ("binary_little_endian" if self.byte_order == "<" else "binary_big_endian")
        lines = ["ply", f"format {fmt} 1.0"]
        lines += [f"comment {c}" for c in self.comments]
        lines += [f"obj_info {c}" for c in self.obj_info]
        for e in self.elements:
            lines += e.header_lines()
        lines.append("end_header")
        return "\n".join(lines)

    def write(self, stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            f.write((self.header + "\n").encode("ascii"))
            for e in self.elements:
                e._write(f, self.text, self.byte_order)
        finally:
            if own:
                f.close()

    @staticmethod
    def read(stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            if f.readline().strip() != b"ply":
                raise PlyParseError("not a PLY file (missing 'ply' magic line)")
            fmt, elements, comments, obj_info = None, [], [], []
            while True:
                raw = f.readline()
                if not raw:
                    raise PlyParseError("early end of file inside the header")
                line = raw.decode("ascii", errors="replace").strip()
                if not line:
                    continue
                key, _, rest = line.partition(" ")
                if key == "format":
                    name, version = rest.split()
                    if name not in _FORMATS or version != "1.0":
                        raise PlyParseError(f"unknown format line: {line}")
                    fmt = name
                elif key == "comment":
                    comments.append(rest)
                elif key == "obj_info":
                    obj_info.append(rest)
                elif key == "element":
                    ename, count = rest.split()
                    elements.append(PlyElement(ename, [], int(count)))
                elif key == "property":
                    if not elements:
                        raise PlyParseError("property before any element")
                    tok = rest.split()
                    e = elements[-1]
                    if tok[0] == "list":
                        prop = PlyListProperty(tok[3], tok[1], tok[2])
                    else:
                        if tok[0] not in _PLY_TO_NP:
                            raise PlyParseError(f"unknown property type {tok[0]!r}")
                        prop = PlyProperty(tok[1], tok[0])
                    e.properties = e.properties + (prop,)
                elif key == "end_header":
                    break
                else:
                    raise PlyParseError(f"unknown header line: {line}")
            if fmt is None:
                raise PlyParseError("missing format line")
            text, order = fmt == "ascii", _FORMATS[fmt]
            for e in elements:
                e._read(f, text, order or "=")
            return PlyData(elements, text=text, byte_order=order or "=", comments=comments, obj_info=obj_info)
        finally:
            if own:
                f.close()

    def __repr__(self):
        return f"PlyData({self.elements!r}, text={self.text}, byte_order={self.byte_order!r})"


def _self_test():
    a = np.zeros(3, dtype=[("x", "f4"), ("n", "u1")])
    a["x"] = [1.5, -2.0, 3.25]
    a["n"] = [1, 2, 255]
    buf = io.BytesIO()
    PlyData([PlyElement.describe(a, "vertex")]).write(buf)
    buf.seek(0)
    b = PlyData.read(buf)["vertex"].data
    assert (a == b).all()


if __name__ == "__main__":
    _self_test()
    print("ok")
