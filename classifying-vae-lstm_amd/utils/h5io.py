"""Keras-2.0.0-compatible weight files (.h5) without h5py.

The reference checkpoints with `model.save_weights(<run>.h5)` (utils/model_utils.py:138) and
restores with `model.load_weights` (cl_vae/model.py:238, cl_vrnn/model.py:281).  The layout
Keras writes: root attributes `layer_names` (fixed-length byte strings,
model.layers order), `backend`, `keras_version`; one group per layer with attribute
`weight_names`; datasets named by TF variable name ("<layer>/kernel:0"), i.e. full path
/<layer>/<layer>/kernel:0, float32.

h5py is not installable here, so this module implements the small subset of the HDF5 file format
those files use: superblock v0, version-1 object headers, symbol-table groups (v1 B-tree + local
heap + SNOD), contiguous little-endian float datasets, fixed-length string attributes.  The
writer emits exactly that subset; the reader additionally follows object-header continuation
blocks and multi-node group B-trees so that files produced by h5py/Keras load too.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b'\x89HDF\r\n\x1a\n'
LEAF_K = 64          # up to 128 links per symbol-table node => one node per group
INTERNAL_K = 16


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


# --------------------------------------------------------------------------- #
# writer
# --------------------------------------------------------------------------- #
def _dt_f32():
    return struct.pack('<BBBBI', 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)


def _dt_str(n):
    return struct.pack('<BBBBI', 0x13, 0x01, 0x00, 0x00, n)       # fixed-length, null-padded, ASCII


def _ds_simple(shape):
    return struct.pack('<BBBBI', 1, len(shape), 0, 0, 0) + b''.join(struct.pack('<Q', int(d)) for d in shape)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack('<HHBBBB', mtype, len(data), flags, 0, 0, 0) + data


def _attr_msg(name, dt, ds, payload):
    nm = name.encode() + b'\0'
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + payload
    return _msg(0x000C, body)


def _attr_strings(name, values):
    """1-D array of fixed-length byte strings (what h5py writes for a numpy 'S' array)."""
    vals = [v.encode() if isinstance(v, str) else bytes(v) for v in values]
    n = max([len(v) for v in vals] + [1])
    payload = b''.join(v.ljust(n, b'\0') for v in vals)
    return _attr_msg(name, _dt_str(n), _ds_simple((len(vals),)), payload)


def _attr_scalar_string(name, value):
    v = value.encode() if isinstance(value, str) else bytes(value)
    return _attr_msg(name, _dt_str(max(len(v), 1)), struct.pack('<BBBBI', 1, 0, 0, 0, 0), v or b'\0')


def _object_header(msgs):
    body = b''.join(msgs)
    return struct.pack('<BBHII', 1, 0, len(msgs), 1, len(body)) + b'\0\0\0\0' + body


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)            # superblock + root symbol-table entry, patched at the end

    def alloc(self, data):
        self.buf += b'\0' * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, dtype='<f4')
        daddr = self.alloc(arr.tobytes() if arr.size else b'')
        msgs = [_msg(0x0001, _ds_simple(arr.shape)), _msg(0x0003, _dt_f32(), flags=1),
                _msg(0x0005, struct.pack('<BBBB', 2, 2, 2, 0)),
                _msg(0x0008, struct.pack('<BBQQ', 3, 1, daddr if arr.size else UNDEF, arr.nbytes))]
        return self.alloc(_object_header(msgs))

    def group(self, links, attr_msgs=()):
        """links: {name: object header address}.  Returns (header address, btree address, heap address)."""
        names = sorted(links, key=lambda s: s.encode())
        if len(names) > 2 * LEAF_K:
            raise ValueError("too many links in one group")
        heap_data = bytearray(8)            # offset 0: the empty string
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b'\0')
        heap_data_addr = self.alloc(bytes(heap_data))
        heap_addr = self.alloc(b'HEAP' + struct.pack('<BBBBQQQ', 0, 0, 0, 0, len(heap_data), 1, heap_data_addr))
        snod = bytearray(b'SNOD' + struct.pack('<BBH', 1, 0, len(names)))
        for n in names:
            snod += struct.pack('<QQII', offs[n], links[n], 0, 0) + b'\0' * 16
        snod += b'\0' * (40 * (2 * LEAF_K - len(names)))
        snod_addr = self.alloc(bytes(snod))
        tree = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, 1 if names else 0, UNDEF, UNDEF))
        if names:
            tree += struct.pack('<QQQ', 0, snod_addr, offs[names[-1]])
        tree += b'\0' * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(tree))
        btree_addr = self.alloc(bytes(tree))
        hdr = self.alloc(_object_header([_msg(0x0011, struct.pack('<QQ', btree_addr, heap_addr))] + list(attr_msgs)))
        return hdr, btree_addr, heap_addr

    def finish(self, root):
        hdr, btree, heap = root
        self.buf += b'\0' * (-len(self.buf) % 8)
        sb = SIG + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack('<QQII', 0, hdr, 1, 0) + struct.pack('<QQ', btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def save_keras_weights(path, layers, backend='tensorflow', keras_version='2.0.0'):
    """layers: [(layer_name, [weight short names], [arrays])] in model.layers order (weight-less layers too)."""
    w = _Writer()
    root_links = {}
    for lname, wnames, arrays in layers:
        full = ['%s/%s:0' % (lname, wn) for wn in wnames]
        inner = {('%s:0' % wn): w.dataset(a) for wn, a in zip(wnames, arrays)}
        links = {}
        if inner:
            links[lname] = w.group(inner)[0]           # "/<layer>/<layer>/kernel:0"
        root_links[lname] = w.group(links, [_attr_strings('weight_names', full)])[0]
    root = w.group(root_links, [_attr_strings('layer_names', [l[0] for l in layers]),
                                _attr_scalar_string('backend', backend),
                                _attr_scalar_string('keras_version', keras_version)])
    with open(path, 'wb') as f:
        f.write(w.finish(root))


# --------------------------------------------------------------------------- #
# reader
# --------------------------------------------------------------------------- #
class _Reader:
    def __init__(self, data):
        self.d = data
        if data[:8] != SIG:
            raise ValueError("not an HDF5 file")
        ver = data[8]
        if ver not in (0, 1):
            raise ValueError("unsupported HDF5 superblock version %d (expected the v0/v1 format Keras 2.0 wrote)" % ver)
        if data[13] != 8 or data[14] != 8:
            raise ValueError("only 8-byte offsets/lengths are supported")
        off = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from('<Q', data, off)[0]
        self.root_hdr = struct.unpack_from('<Q', data, off + 32 + 8)[0]

    # object header v1 -> list of (type, bytes)
    def messages(self, addr):
        d = self.d
        ver, _, nmsg, _, size = struct.unpack_from('<BBHII', d, addr)
        if ver != 1:
            raise ValueError("unsupported object header version %d" % ver)
        out = []
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from('<HHB', d, pos)
                body = d[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                       # continuation
                    caddr, clen = struct.unpack_from('<QQ', body, 0)
                    blocks.append((caddr + self.base, clen))
                out.append((mtype, body))
        return out

    @staticmethod
    def _dtype(body):
        cls = body[0] & 0x0F
        size = struct.unpack_from('<I', body, 4)[0]
        if cls == 1:
            if body[1] & 1:
                raise ValueError("big-endian floats are not supported")
            return np.dtype('<f%d' % size), 8 + 12
        if cls == 3:
            return np.dtype('S%d' % size), 8
        if cls == 0:
            signed = (body[1] >> 3) & 1
            return np.dtype('<%s%d' % ('i' if signed else 'u', size)), 8 + 4
        if cls == 9 and (body[1] & 0x0F) == 1:
            return 'vlen_str', 8          # variable-length string: (length, global-heap address, index) per element
        raise ValueError("unsupported datatype class %d" % cls)

    def _gheap_object(self, addr, index):
        """object `index` of the global heap collection at `addr` (variable-length string storage)."""
        d = self.d
        addr += self.base
        if d[addr:addr + 4] != b'GCOL':
            raise ValueError("bad global heap collection")
        size = struct.unpack_from('<Q', d, addr + 8)[0]
        pos, end = addr + 16, addr + size
        while pos + 16 <= end:
            idx, _ref, _res, osz = struct.unpack_from('<HHIQ', d, pos)
            if idx == 0:
                break
            if idx == index:
                return bytes(d[pos + 16:pos + 16 + osz])
            pos += 16 + (osz + 7) // 8 * 8
        raise ValueError("global heap object %d not found" % index)

    @staticmethod
    def _shape(body):
        ver, rank, flags = body[0], body[1], body[2]
        off = 8 if ver == 1 else 4
        return tuple(struct.unpack_from('<Q', body, off + 8 * i)[0] for i in range(rank))

    def attrs(self, addr):
        out = {}
        for mtype, body in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = body[0]
            if ver == 1:
                _, _, nsz, dsz, ssz = struct.unpack_from('<BBHHH', body, 0)
                pos = 8
                pad = lambda n: (n + 7) // 8 * 8
            elif ver in (2, 3):
                _, _, nsz, dsz, ssz = struct.unpack_from('<BBHHH', body, 0)
                pos = 8 + (1 if ver == 3 else 0)
                pad = lambda n: n
            else:
                continue
            name = bytes(body[pos:pos + nsz]).split(b'\0')[0].decode()
            pos += pad(nsz)
            dt, _ = self._dtype(body[pos:pos + dsz])
            pos += pad(dsz)
            shape = self._shape(body[pos:pos + ssz])
            pos += pad(ssz)
            n = int(np.prod(shape)) if shape else 1
            if isinstance(dt, str):           # variable-length strings (what h5py >= 3 writes for a list of bytes)
                vals = []
                for i in range(n):
                    ln, ga, gi = struct.unpack_from('<IQI', body, pos + 16 * i)
                    vals.append(self._gheap_object(ga, gi)[:ln] if ln else b'')
                arr = np.array(vals, dtype=object)
                out[name] = arr.reshape(shape) if shape else arr[0]
                continue
            arr = np.frombuffer(bytes(body[pos:pos + n * dt.itemsize]), dtype=dt)
            out[name] = arr.reshape(shape) if shape else arr[0]
        return out

    def links(self, addr):
        """{name: object header address} of a symbol-table group."""
        st = [b for t, b in self.messages(addr) if t == 0x0011]
        if not st:
            return {}
        btree, heap = struct.unpack_from('<QQ', st[0], 0)
        d = self.d
        hp = heap + self.base
        if d[hp:hp + 4] != b'HEAP':
            raise ValueError("bad local heap")
        heap_data = struct.unpack_from('<Q', d, hp + 24)[0] + self.base
        out = {}

        def name_at(off):
            s = heap_data + off
            return bytes(d[s:d.index(b'\0', s)]).decode()

        def walk(node):
            node += self.base
            if d[node:node + 4] == b'TREE':
                _ntype, level, used = struct.unpack_from('<BBH', d, node + 4)
                for i in range(used):
                    child = struct.unpack_from('<Q', d, node + 24 + 8 + 16 * i)[0]
                    walk(child)
            elif d[node:node + 4] == b'SNOD':
                nsym = struct.unpack_from('<H', d, node + 6)[0]
                for i in range(nsym):
                    noff, ohdr = struct.unpack_from('<QQ', d, node + 8 + 40 * i)
                    out[name_at(noff)] = ohdr + self.base
            else:
                raise ValueError("bad group node")
        walk(btree)
        return out

    def is_group(self, addr):
        return any(t == 0x0011 for t, _ in self.messages(addr))

    def dataset(self, addr):
        shape, dt, layout = (), None, None
        for t, b in self.messages(addr):
            if t == 0x0001:
                shape = self._shape(b)
            elif t == 0x0003:
                dt, _ = self._dtype(b)
            elif t == 0x0008:
                layout = b
        if dt is None or layout is None:
            raise ValueError("not a dataset")
        n = int(np.prod(shape)) if shape else 1
        ver = layout[0]
        if ver == 3:
            cls = layout[1]
            if cls == 1:
                daddr, _sz = struct.unpack_from('<QQ', layout, 2)
                raw = self.d[daddr + self.base:daddr + self.base + n * dt.itemsize] if n else b''
            elif cls == 0:
                sz = struct.unpack_from('<H', layout, 2)[0]
                raw = layout[4:4 + sz]
            else:
                raise ValueError("chunked datasets are not supported (Keras weights are contiguous)")
        elif ver in (1, 2):
            rank, cls = layout[1], layout[2]
            if cls != 1:
                raise ValueError("only contiguous layout is supported")
            daddr = struct.unpack_from('<Q', layout, 8)[0]
            raw = self.d[daddr + self.base:daddr + self.base + n * dt.itemsize]
        else:
            raise ValueError("unsupported layout version %d" % ver)
        return np.frombuffer(bytes(raw), dtype=dt).reshape(shape).copy()

    def visit_datasets(self, addr, prefix=''):
        out = {}
        for name, a in self.links(addr).items():
            path = prefix + '/' + name if prefix else name
            if self.is_group(a):
                out.update(self.visit_datasets(a, path))
            else:
                out[path] = self.dataset(a)
        return out


def load_keras_weights(path):
    """-> [(layer_name, [arrays in weight_names order])] in `layer_names` order."""
    with open(path, 'rb') as f:
        r = _Reader(f.read())
    root = r.root_hdr + r.base
    ra = r.attrs(root)
    if 'layer_names' not in ra:
        raise ValueError("%s has no layer_names attribute (not a Keras weight file)" % path)
    groups = r.links(root)
    out = []
    for ln in np.atleast_1d(ra['layer_names']):
        ln = ln.decode() if isinstance(ln, bytes) else str(ln)
        g = groups[ln]
        wn = r.attrs(g).get('weight_names', np.array([], dtype='S1'))
        ds = r.visit_datasets(g)
        arrs = []
        for w in np.atleast_1d(wn):
            w = w.decode() if isinstance(w, bytes) else str(w)
            arrs.append(ds[w].astype(np.float32))
        out.append((ln, arrs))
    return out
