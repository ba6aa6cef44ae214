"""Self-contained Feather V2 (Arrow IPC file) reader / writer for flat tables of primitive columns.

Why: the wire format of the leaderboard (save_zip.py:56-81, tools/test/save_zip_gt.py:64-108) is one Feather file
per sweep, written by the reference through pandas -> pyarrow.  The MI355X box image has neither pyarrow nor
network access, so this module speaks the container itself:

  * ``read_table(bytes) -> {column: numpy array}`` reads what pandas/pyarrow write (including their default
    LZ4-frame buffer compression, decoded by ``himo_lz4_frame_decompress`` in libhimo_amd.so) and what this
    module writes;
  * ``write_table({column: array}) -> bytes`` writes an uncompressed Feather V2 file that ``pandas.read_feather``
    / ``pyarrow`` read back with the same column names and dtypes.

Supported column types: int8..int64, uint8..uint64, float32/float64, bool; no nulls, no nesting, no dictionaries
-- everything the HiMo zips contain.  Format reference: the Arrow columnar specification (IPC file format,
Message / Schema / RecordBatch / Footer flatbuffers); flatbuffers are parsed and built by the ~100 lines below.
"""
from __future__ import annotations

import ctypes
import struct

import numpy as np

MAGIC = b"ARROW1"
_CONT = 0xFFFFFFFF
# Type union ids (Schema.fbs)
_T_INT, _T_FLOAT, _T_BOOL = 2, 3, 6
_V5 = 4                      # MetadataVersion.V5


# ------------------------------------------------------------------------------------------------------------
# flatbuffer reading
# ------------------------------------------------------------------------------------------------------------
class _Table:
    def __init__(self, buf, pos):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        self.vt_len = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, field):
        o = 4 + 2 * field
        if o >= self.vt_len:
            return 0
        return struct.unpack_from("<H", self.buf, self.vt + o)[0]

    def scalar(self, field, fmt, default=0):
        o = self._off(field)
        return struct.unpack_from("<" + fmt, self.buf, self.pos + o)[0] if o else default

    def _indirect(self, field):
        o = self._off(field)
        if not o:
            return None
        p = self.pos + o
        return p + struct.unpack_from("<I", self.buf, p)[0]

    def table(self, field):
        p = self._indirect(field)
        return None if p is None else _Table(self.buf, p)

    def string(self, field):
        p = self._indirect(field)
        if p is None:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n]).decode()

    def vector(self, field):
        """(start, length) of a vector's elements."""
        p = self._indirect(field)
        if p is None:
            return 0, 0
        return p + 4, struct.unpack_from("<I", self.buf, p)[0]

    def table_vector(self, field):
        start, n = self.vector(field)
        out = []
        for i in range(n):
            p = start + 4 * i
            out.append(_Table(self.buf, p + struct.unpack_from("<I", self.buf, p)[0]))
        return out


def _root(buf, pos):
    return _Table(buf, pos + struct.unpack_from("<I", buf, pos)[0])


def _field_dtype(f: _Table):
    tt = f.scalar(2, "B")
    t = f.table(3)
    if tt == _T_FLOAT:
        prec = t.scalar(0, "h")
        return {1: np.dtype("<f4"), 2: np.dtype("<f8"), 0: np.dtype("<f2")}[prec]
    if tt == _T_INT:
        bits, signed = t.scalar(0, "i"), t.scalar(1, "?", False)
        return np.dtype(f"<{'i' if signed else 'u'}{bits // 8}")
    if tt == _T_BOOL:
        return np.dtype(bool)
    raise NotImplementedError(f"Arrow type id {tt} of column {f.string(0)!r} is not supported")


def _decompress(raw: memoryview) -> bytes:
    """One compressed Arrow buffer: int64 uncompressed length (-1 = stored as is) + LZ4 frame."""
    n = struct.unpack_from("<q", raw, 0)[0]
    if n == -1:
        return bytes(raw[8:])
    from . import _lib
    lib = _lib.load()
    src = bytes(raw[8:])
    dst = ctypes.create_string_buffer(max(int(n), 1))
    got = lib.himo_lz4_frame_decompress(src, len(src), dst, int(n))
    if got != n:
        raise ValueError(f"LZ4 frame decode failed ({got} of {n} bytes)")
    return dst.raw[:n]


def read_table(data: bytes) -> dict:
    """Feather V2 bytes -> ordered {column name: 1-D numpy array}."""
    buf = memoryview(data)
    if bytes(buf[:6]) != MAGIC or bytes(buf[-6:]) != MAGIC:
        raise ValueError("not a Feather V2 / Arrow IPC file (the reference writes V2; V1 'FEA1' files are not supported)")
    flen = struct.unpack_from("<i", buf, len(buf) - 10)[0]
    fstart = len(buf) - 10 - flen
    footer = _root(buf, fstart)
    schema = footer.table(1)
    fields = schema.table_vector(1)
    names = [f.string(0) for f in fields]
    dtypes = [_field_dtype(f) for f in fields]
    for f in fields:
        if f.table(4) is not None:
            raise NotImplementedError("dictionary-encoded columns are not supported")
    parts = {n: [] for n in names}
    bstart, nb = footer.vector(3)                               # recordBatches: [Block{offset:long, metaDataLength:int, bodyLength:long}]
    for i in range(nb):
        off, meta_len, _pad, body_len = struct.unpack_from("<qiiq", buf, bstart + 24 * i)
        p = off
        if struct.unpack_from("<I", buf, p)[0] == _CONT:
            p += 4
        p += 4                                                  # metadata size
        msg = _root(buf, p)
        if msg.scalar(1, "B") != 3:                              # MessageHeader.RecordBatch
            continue
        rb = msg.table(2)
        length = rb.scalar(0, "q")
        bufs_start, n_bufs = rb.vector(2)
        comp = rb.table(3)
        if comp is not None and comp.scalar(0, "b") != 0:
            raise NotImplementedError("only LZ4_FRAME buffer compression is supported (ZSTD found)")
        body = off + meta_len
        if n_bufs != 2 * len(names):
            raise NotImplementedError("nested or variable-width columns are not supported")
        for c, (name, dt) in enumerate(zip(names, dtypes)):
            boff, blen = struct.unpack_from("<qq", buf, bufs_start + 16 * (2 * c + 1))   # [validity, data] per column
            raw = buf[body + boff: body + boff + blen]
            payload = _decompress(raw) if (comp is not None and blen > 0) else bytes(raw)
            if dt == np.dtype(bool):
                bits = np.unpackbits(np.frombuffer(payload, dtype=np.uint8), bitorder="little")[:length]
                parts[name].append(bits.astype(bool))
            else:
                parts[name].append(np.frombuffer(payload, dtype=dt, count=length).copy())
    return {n: (np.concatenate(parts[n]) if parts[n] else np.empty(0, dt)) for n, dt in zip(names, dtypes)}


# ------------------------------------------------------------------------------------------------------------
# flatbuffer building (back to front, like the reference builders)
# ------------------------------------------------------------------------------------------------------------
class _Builder:
    def __init__(self):
        self.b = bytearray()          # holds the buffer REVERSED in time: we prepend by building from the end

    # the buffer grows toward lower addresses; offsets are measured from the END of the final buffer
    def _size(self):
        return len(self.b)

    def _prepend(self, data: bytes):
        self.b[0:0] = data

    def _align(self, n, extra=0):
        pad = (-(len(self.b) + extra)) % n
        if pad:
            self._prepend(b"\x00" * pad)

    def string(self, s: str) -> int:
        raw = s.encode() + b"\x00"
        self._align(4, len(raw) + 4)
        self._prepend(raw)
        self._prepend(struct.pack("<I", len(raw) - 1))
        return self._size()

    def struct_vector(self, fmt: str, rows, align=8) -> int:
        body = b"".join(struct.pack("<" + fmt, *r) for r in rows)
        self._align(align, len(body))                     # elements aligned; the length word sits right before them
        self._prepend(body)
        self._align(4)
        self._prepend(struct.pack("<I", len(rows)))
        return self._size()

    def offset_vector(self, offsets) -> int:
        self._align(4, 4 * len(offsets) + 4)
        for k, o in reversed(list(enumerate(offsets))):
            # element k will live at position (size_after + ...) ; write relative offset = distance to target
            self._prepend(struct.pack("<I", 0))            # placeholder, patched below
        base = self._size()
        for k, o in enumerate(offsets):
            pos_from_end = base - 4 * k                     # this element's own offset from the end
            rel = pos_from_end - o
            struct.pack_into("<I", self.b, 4 * k, rel)
        self._prepend(struct.pack("<I", len(offsets)))
        return self._size()

    def table(self, fields) -> int:
        """fields: list of (kind, value) in field-id order; kind in {"b","B","h","i","q","?"} scalars, "o" offset
        (value = offset from the end returned by a previous call, or None), "n" = absent."""
        # layout of the table body: [soffset to vtable][fields sorted by size desc]
        items = [(i, k, v) for i, (k, v) in enumerate(fields) if k != "n" and not (k == "o" and v is None)]
        sizes = {"b": 1, "B": 1, "?": 1, "h": 2, "H": 2, "i": 4, "o": 4, "q": 8}
        order = sorted(items, key=lambda t: -sizes[t[1]])
        body_len = 4 + sum(sizes[k] for _, k, _ in order)
        maxal = max([4] + [sizes[k] for _, k, _ in order])
        # place fields with natural alignment relative to table start
        pos, slots = 4, {}
        for i, k, v in order:
            sz = sizes[k]
            pos = (pos + sz - 1) // sz * sz
            slots[i] = (pos, k, v)
            pos += sz
        body_len = (pos + 3) // 4 * 4
        self._align(maxal, body_len)
        body = bytearray(body_len)
        self._prepend(bytes(body))
        table_end = self._size()                            # offset (from end) of the table START
        for i, (p, k, v) in slots.items():
            if k == "o":
                field_from_end = table_end - p
                struct.pack_into("<I", self.b, p, field_from_end - v)
            else:
                struct.pack_into("<" + k, self.b, p, v)
        n_fields = len(fields)
        vt = bytearray(4 + 2 * n_fields)
        struct.pack_into("<HH", vt, 0, len(vt), body_len)
        for i, (p, k, v) in slots.items():
            struct.pack_into("<H", vt, 4 + 2 * i, p)
        if len(vt) % 4:
            self._prepend(b"\x00" * (4 - len(vt) % 4))      # keep the table 4-aligned after the vtable is added
        self._prepend(bytes(vt))
        vt_from_end = self._size()
        # soffset at table start: table_pos - vtable_pos (positive when the vtable precedes the table)
        struct.pack_into("<i", self.b, self._size() - table_end, vt_from_end - table_end)
        return table_end

    def finish(self, root: int) -> bytes:
        self._align(8, 4)
        self._prepend(struct.pack("<I", self._size() + 4 - root))
        return bytes(self.b)


def _type_of(dt: np.dtype):
    dt = np.dtype(dt)
    if dt == np.dtype(bool):
        return _T_BOOL, None
    if dt.kind == "f":
        return _T_FLOAT, {2: 0, 4: 1, 8: 2}[dt.itemsize]
    if dt.kind in "iu":
        return _T_INT, (dt.itemsize * 8, dt.kind == "i")
    raise TypeError(f"unsupported column dtype {dt}")


def _schema(b: _Builder, names, dtypes) -> int:
    fields = []
    for name, dt in zip(names, dtypes):
        tt, info = _type_of(dt)
        if tt == _T_FLOAT:
            t = b.table([("h", info)])
        elif tt == _T_INT:
            t = b.table([("i", info[0]), ("?", info[1])])
        else:
            t = b.table([])
        n = b.string(name)
        kids = b.offset_vector([])                              # readers expect a (possibly empty) children vector
        fields.append(b.table([("o", n), ("?", False), ("B", tt), ("o", t), ("n", None), ("o", kids), ("n", None)]))
    fv = b.offset_vector(fields)
    return b.table([("h", 0), ("o", fv), ("n", None), ("n", None)])


def _message(header_type: int, build_header, body_len: int) -> bytes:
    b = _Builder()
    h = build_header(b)
    root = b.table([("h", _V5), ("B", header_type), ("o", h), ("q", body_len), ("n", None)])
    fb = b.finish(root)
    pad = (-len(fb)) % 8
    return struct.pack("<II", _CONT, len(fb) + pad) + fb + b"\x00" * pad


_FRAMING_CACHE: dict = {}


def _framing(names: tuple, dtypes: tuple, length: int):
    """(bytes before the record batch body, bytes after it, body length, [(offset, nbytes) of each column's data in the body]) of
    a table of fixed-width columns: everything in a Feather file but the column values depends on the schema and the row
    count alone, so a stream of equally shaped sweeps builds it once per row count (cached: the last 64 shapes)."""
    key = (names, tuple(np.dtype(d).str for d in dtypes), int(length))
    got = _FRAMING_CACHE.get(key)
    if got is not None:
        return got
    buffers, nodes, spans, at = [], [], [], 0
    for dt in dtypes:
        dt = np.dtype(dt)
        nodes.append((length, 0))
        buffers.append((at, 0))                               # validity bitmap: absent (no nulls)
        nbytes = (length + 7) // 8 if dt == np.dtype(bool) else length * dt.itemsize
        buffers.append((at, nbytes))
        spans.append((at, nbytes))
        at += nbytes + (-nbytes) % 8
    body_len = at
    head = bytearray(MAGIC + b"\x00\x00")
    head += _message(1, lambda b: _schema(b, names, dtypes), 0)

    def rb_header(b: _Builder):
        bv = b.struct_vector("qq", buffers)
        nv = b.struct_vector("qq", nodes)
        return b.table([("q", length), ("o", nv), ("o", bv), ("n", None)])

    rb_msg = _message(3, rb_header, body_len)
    rb_off = len(head)
    head += rb_msg
    fb = _Builder()
    sch = _schema(fb, names, dtypes)
    blocks = fb.struct_vector("qiiq", [(rb_off, len(rb_msg), 0, body_len)])
    dicts = fb.struct_vector("qiiq", [])
    root = fb.table([("h", _V5), ("o", sch), ("o", dicts), ("o", blocks), ("n", None)])
    footer = fb.finish(root)
    tail = footer + struct.pack("<i", len(footer)) + MAGIC
    got = (bytes(head), bytes(tail), body_len, spans)
    if len(_FRAMING_CACHE) >= 64:
        _FRAMING_CACHE.pop(next(iter(_FRAMING_CACHE)))
    _FRAMING_CACHE[key] = got
    return got


def write_table(columns: dict) -> bytes:
    """{name: 1-D array} -> Feather V2 bytes (single record batch, uncompressed, 8-byte aligned buffers)."""
    names = list(columns)
    arrays = []
    for n in names:
        a = np.asarray(columns[n])
        if a.ndim != 1:
            raise ValueError(f"column {n!r} must be 1-D")
        arrays.append(a)
    length = len(arrays[0]) if arrays else 0
    if any(len(a) != length for a in arrays):
        raise ValueError("all columns must have the same length")
    head, tail, body_len, spans = _framing(tuple(names), tuple(a.dtype for a in arrays), length)
    out = np.zeros(len(head) + body_len + len(tail), np.uint8)
    out[:len(head)] = np.frombuffer(head, np.uint8)
    for a, (off, nbytes) in zip(arrays, spans):
        dst = out[len(head) + off:len(head) + off + nbytes]
        if a.dtype == np.dtype(bool):
            dst[:] = np.packbits(a.astype(np.uint8), bitorder="little")
        else:
            dst.view(a.dtype.newbyteorder("<"))[:] = a          # ONE pass over the column, strided or not, straight into place
    out[len(head) + body_len:] = np.frombuffer(tail, np.uint8)
    return out.tobytes()


def write_matrix(values: np.ndarray, names) -> np.ndarray:
    """The same file as ``write_table({names[j]: values[:, j]})`` for an (n, k) array of one fixed-width dtype, as a uint8 array
    (a buffer for ``file.write`` / ``ZipFile.writestr`` without a further copy): the framing comes from the cache, each column
    is gathered from the row-major rows straight into its place.  The per-sweep encoder of ``save_zip``."""
    v = np.asarray(values)
    if v.ndim != 2 or v.shape[1] != len(names) or v.dtype == np.dtype(bool):
        raise ValueError("write_matrix takes an (n, len(names)) array of a numeric dtype")
    n, k = v.shape
    head, tail, body_len, spans = _framing(tuple(names), (v.dtype,) * k, n)
    out = np.empty(len(head) + body_len + len(tail), np.uint8)
    out[:len(head)] = np.frombuffer(head, np.uint8)
    le = v.dtype.newbyteorder("<")
    for j, (off, nbytes) in enumerate(spans):
        lo = len(head) + off
        out[lo:lo + nbytes].view(le)[:] = v[:, j]
        pad = (-nbytes) % 8
        if pad:
            out[lo + nbytes:lo + nbytes + pad] = 0
    out[len(head) + body_len:] = np.frombuffer(tail, np.uint8)
    return out
