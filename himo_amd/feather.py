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
    dtypes = [a.dtype for a in arrays]
    body = bytearray()
    buffers, nodes = [], []
    for a in arrays:
        nodes.append((length, 0))
        buffers.append((len(body), 0))                        # validity bitmap: absent (no nulls)
        if a.dtype == np.dtype(bool):
            payload = np.packbits(a.astype(np.uint8), bitorder="little").tobytes()
        else:
            payload = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False)).tobytes()
        buffers.append((len(body), len(payload)))
        body += payload + b"\x00" * ((-len(payload)) % 8)

    out = bytearray(MAGIC + b"\x00\x00")
    schema_msg = _message(1, lambda b: _schema(b, names, dtypes), 0)
    out += schema_msg

    def rb_header(b: _Builder):
        bv = b.struct_vector("qq", buffers)
        nv = b.struct_vector("qq", nodes)
        return b.table([("q", length), ("o", nv), ("o", bv), ("n", None)])

    rb_msg = _message(3, rb_header, len(body))
    rb_off = len(out)
    out += rb_msg + body

    fb = _Builder()
    sch = _schema(fb, names, dtypes)
    blocks = fb.struct_vector("qiiq", [(rb_off, len(rb_msg), 0, len(body))])
    dicts = fb.struct_vector("qiiq", [])
    root = fb.table([("h", _V5), ("o", sch), ("o", dicts), ("o", blocks), ("n", None)])
    footer = fb.finish(root)
    out += footer + struct.pack("<i", len(footer)) + MAGIC
    return bytes(out)
