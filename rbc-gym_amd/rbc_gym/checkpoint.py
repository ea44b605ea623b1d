"""Checkpoint files: converged flow states used as reset initial conditions.

Replaces the HDF5.jl read in the reference (`initialize_from_checkpoint`, rbc_sim2D.jl:173-186;
writer rbc_sim2D.jl:33-43,64-66).  The reference's files are plain HDF5: root attributes
`num_episodes`, `start_seed` and contiguous, unfiltered little-endian float64 datasets `b`, `u`,
`w` of Julia shape (E, Nx, 1, Nz[+1]) -- which a C-order reader sees as (Nz[+1], 1, Nx, E).

h5py is not available where this package runs, so `_MiniHDF5` below is a dependency-free reader
for exactly that subset of HDF5 (superblock v0, version-1 object headers, symbol-table groups,
contiguous layout, fixed-point/IEEE datatypes).  `.npz` files with arrays b,u,w laid out
(E, Nz[+1], Nx) are accepted as well.
"""
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class _MiniHDF5:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.d = f.read()
        if self.d[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError(f"{path}: not an HDF5 file")
        ver = self.d[8]
        if ver not in (0, 1):
            raise ValueError(f"{path}: HDF5 superblock version {ver} not supported by the built-in reader")
        self.so, self.sl = self.d[13], self.d[14]            # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise ValueError("only 8-byte offsets/lengths supported")
        p = 24 if ver == 0 else 28                            # -> base address, free space, eof, driver
        p += 4 * 8
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self.root_header = struct.unpack_from("<Q", self.d, p + 8)[0]
        self.datasets = {}
        self.attrs = {}
        msgs = self._object_header(self.root_header)
        for mtype, body in msgs:
            if mtype == 0x000C:
                name, val = self._attribute(body)
                self.attrs[name] = val
            elif mtype == 0x0011:                              # old-style group: symbol table
                btree, heap = struct.unpack_from("<QQ", body, 0)
                for name, addr in self._group_entries(btree, heap):
                    self.datasets[name] = addr
            elif mtype == 0x0006:                              # compact group: one Link message per member
                name, addr = self._link(body)
                if addr is not None:
                    self.datasets[name] = addr

    # -- object header (version 1) -------------------------------------------------------------
    def _object_header(self, addr):
        d = self.d
        if d[addr] != 1:
            raise ValueError("only version-1 object headers supported")
        nmsg = struct.unpack_from("<H", d, addr + 2)[0]
        size = struct.unpack_from("<I", d, addr + 8)[0]
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg + 64:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end:
                mtype, msize, _flags = struct.unpack_from("<HHB", d, p)
                body = d[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:                           # continuation
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, clen))
                elif mtype != 0:
                    out.append((mtype, body))
        return out

    # -- groups: v1 B-tree of symbol nodes + local heap ------------------------------------------
    def _heap_data(self, heap):
        d = self.d
        if d[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap")
        return struct.unpack_from("<Q", d, heap + 8 + 16)[0]

    def _group_entries(self, btree, heap):
        d = self.d
        hdata = self._heap_data(heap)
        out = []

        def name_at(off):
            s = hdata + off
            e = d.index(b"\x00", s)
            return d[s:e].decode()

        def walk(node):
            if d[node:node + 4] == b"TREE":
                level = d[node + 5]
                n = struct.unpack_from("<H", d, node + 6)[0]
                p = node + 8 + 16
                for j in range(n):
                    child = struct.unpack_from("<Q", d, p + 8 + j * 16)[0]   # key,child,key,child...
                    walk(child)
                _ = level
            elif d[node:node + 4] == b"SNOD":
                n = struct.unpack_from("<H", d, node + 6)[0]
                p = node + 8
                for j in range(n):
                    noff, haddr = struct.unpack_from("<QQ", d, p + j * 40)
                    out.append((name_at(noff), haddr))
            else:
                raise ValueError("bad group node")

        walk(btree)
        return out

    # -- messages ------------------------------------------------------------------------------
    @staticmethod
    def _link(body):
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]; p += 1
        if flags & 0x04:
            p += 8                                            # creation order
        if flags & 0x10:
            p += 1                                            # name character set
        nlen_size = 1 << (flags & 0x03)
        nlen = int.from_bytes(body[p:p + nlen_size], "little"); p += nlen_size
        name = body[p:p + nlen].decode(); p += nlen
        if ltype != 0:
            return name, None                                 # soft / external links are not followed
        return name, struct.unpack_from("<Q", body, p)[0]

    @staticmethod
    def _dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        p = 8 if ver == 1 else 4
        return tuple(struct.unpack_from("<Q", body, p + 8 * j)[0] for j in range(rank))

    @staticmethod
    def _datatype(body):
        cls = body[0] & 0x0F
        bits0 = body[1]
        size = struct.unpack_from("<I", body, 4)[0]
        if bits0 & 1:
            raise ValueError("big-endian data not supported")
        if cls == 0:
            signed = bool(bits0 & 0x08)
            return np.dtype(f"<{'i' if signed else 'u'}{size}")
        if cls == 1:
            return np.dtype(f"<f{size}")
        raise ValueError(f"datatype class {cls} not supported")

    def _attribute(self, body):
        ver = body[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
        if ver == 1:
            pad = lambda n: (n + 7) & ~7
            p = 8
            name = body[p:p + nsz].split(b"\x00")[0].decode(); p += pad(nsz)
            dt = self._datatype(body[p:p + tsz]); p += pad(tsz)
            shape = self._dataspace(body[p:p + ssz]) if ssz else (); p += pad(ssz)
        elif ver in (2, 3):
            p = 8 if ver == 2 else 9
            name = body[p:p + nsz].split(b"\x00")[0].decode(); p += nsz
            dt = self._datatype(body[p:p + tsz]); p += tsz
            shape = self._dataspace(body[p:p + ssz]) if ssz else (); p += ssz
        else:
            raise ValueError("attribute version not supported")
        n = int(np.prod(shape)) if shape else 1
        val = np.frombuffer(body, dt, n, p)
        return name, (val.reshape(shape) if shape else val[0])

    def read(self, name):
        if name not in self.datasets:
            raise KeyError(name)
        shape = dt = None
        addr = size = None
        for mtype, body in self._object_header(self.datasets[name]):
            if mtype == 0x0001:
                shape = self._dataspace(body)
            elif mtype == 0x0003:
                dt = self._datatype(body)
            elif mtype == 0x0008:
                ver = body[0]
                if ver == 3:
                    if body[1] != 1:
                        raise ValueError(f"dataset {name}: only contiguous layout supported")
                    addr, size = struct.unpack_from("<QQ", body, 2)
                elif ver in (1, 2):
                    rank, lclass = body[1], body[2]
                    if lclass != 1:
                        raise ValueError(f"dataset {name}: only contiguous layout supported")
                    addr = struct.unpack_from("<Q", body, 8)[0]
                    size = None
                else:
                    raise ValueError("layout version not supported")
            elif mtype == 0x000B:
                raise ValueError(f"dataset {name}: filtered data not supported")
        if shape is None or dt is None or addr is None or addr == UNDEF:
            raise ValueError(f"dataset {name}: incomplete header")
        n = int(np.prod(shape))
        if size is not None and size < n * dt.itemsize:
            raise ValueError(f"dataset {name}: storage smaller than its dataspace")
        return np.frombuffer(self.d, dt, n, addr).reshape(shape)


def read_checkpoint(path):
    """-> dict(b=(E,nz,nx), u=(E,nz,nx), w=(E,nz+1,nx) float64, num_episodes, start_seed)."""
    path = str(path)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npz"):
        z = np.load(path)
        out = {k: np.ascontiguousarray(z[k], np.float64) for k in ("b", "u", "v", "w") if k in z.files}
        out["num_episodes"] = int(out["b"].shape[0])
        out["start_seed"] = int(z["start_seed"]) if "start_seed" in z.files else -1
        return out
    h = _MiniHDF5(path)
    out = {}
    names = ("b", "u", "v", "w") if "v" in h.datasets else ("b", "u", "w")
    for k in names:
        a = h.read(k)                       # (Nz[+1], Ny, Nx, E) as stored by the reference's writer (Ny = 1 in 2D)
        if a.ndim != 4:
            raise ValueError(f"{path}: dataset {k} has unexpected shape {a.shape}")
        a = np.moveaxis(a, -1, 0)           # (E, Nz[+1], Ny, Nx)
        if "v" not in h.datasets:
            if a.shape[2] != 1:
                raise ValueError(f"{path}: dataset {k} has unexpected shape {a.shape} (expected a 2D checkpoint)")
            a = a[:, :, 0]
        out[k] = np.ascontiguousarray(a, dtype=np.float64)
    out["num_episodes"] = int(h.attrs.get("num_episodes", out["b"].shape[0]))
    out["start_seed"] = int(h.attrs.get("start_seed", -1))
    return out


def write_checkpoint_npz(path, b, u, w, start_seed=0, v=None):
    """Device-state checkpoint writer (npz container; same arrays as the reference's HDF5 files; pass v for 3D)."""
    extra = {} if v is None else {"v": np.asarray(v, np.float64)}
    np.savez_compressed(path, b=np.asarray(b, np.float64), u=np.asarray(u, np.float64), w=np.asarray(w, np.float64),
                        start_seed=np.int64(start_seed), num_episodes=np.int64(np.asarray(b).shape[0]), **extra)


# ---------------------------------------------------------------------------------------------
# HDF5 writer: the same subset of the format the reference's files use (what HDF5.jl writes at
# rbc_sim2D.jl:36-43,64-66): superblock v0, a symbol-table root group with two scalar int64
# attributes, and contiguous little-endian float64 datasets.  Files written here open with
# libhdf5 (h5py / HDF5.jl / h5dump) and with `_MiniHDF5`.
# ---------------------------------------------------------------------------------------------
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header_v1(msgs):
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


_DT_F64 = struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
_DT_I64 = struct.pack("<B3BI", 0x10, 0x08, 0x00, 0x00, 8) + struct.pack("<HH", 0, 64)


def _dataspace_v1(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _attribute_v1(name, value):
    nm = name.encode() + b"\0"
    ds = _dataspace_v1(())
    return (struct.pack("<BxHHH", 1, len(nm), len(_DT_I64), len(ds)) + _pad8(nm) + _pad8(_DT_I64) + _pad8(ds)
            + struct.pack("<q", int(value)))


def write_hdf5(path, datasets, attrs):
    """datasets: {name: float64 ndarray} stored contiguously in C order under the root group;
    attrs: {name: int} scalar int64 root attributes."""
    names = sorted(datasets)                       # symbol-table entries are ordered by name
    if len(names) > 8:
        raise ValueError("the built-in HDF5 writer holds at most 8 datasets (one symbol-table node)")
    arrays = {k: np.ascontiguousarray(datasets[k], dtype="<f8") for k in names}
    # local heap data segment: "" at offset 0, then the names, each 8-byte aligned
    heap_data, name_off = bytearray(8), {}
    for k in names:
        name_off[k] = len(heap_data)
        heap_data += _pad8(k.encode() + b"\0")
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16        # one free block (next = 1 "none", size 32)
    heap_size = len(heap_data)

    # fixed layout: superblock 0..96 | root header | heap header | heap data | B-tree node | SNOD | dataset headers | data
    sb_size = 8 + 8 + 4 + 4 + 4 * 8 + 40       # = 96
    root_msgs_wo_addr = None
    attr_msgs = [_msg(0x000C, _attribute_v1(k, attrs[k])) for k in attrs]
    root_hdr_size = 16 + len(_msg(0x0011, b"\0" * 16)) + sum(len(m) for m in attr_msgs)
    a_root = sb_size
    a_heap = a_root + root_hdr_size
    a_heap_data = a_heap + 32
    a_btree = a_heap_data + heap_size
    btree_size = 24 + (2 * 16 + 1) * 8 + 2 * 16 * 8            # header + keys + children for K=16
    a_snod = a_btree + btree_size
    snod_size = 8 + 8 * 40                                     # 2K entries, K=4
    a_dset = a_snod + snod_size

    def dataset_header(arr, addr):
        return _object_header_v1([
            _msg(0x0001, _dataspace_v1(arr.shape)),
            _msg(0x0003, _DT_F64, flags=1),
            _msg(0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),       # fill value v2: alloc late, fill if-set, default
            _msg(0x0008, struct.pack("<BBQQ", 3, 1, addr, arr.nbytes)),
        ])

    dset_hdr_size = {k: len(dataset_header(arrays[k], 0)) for k in names}
    a_hdr, pos = {}, a_dset
    for k in names:
        a_hdr[k] = pos
        pos += dset_hdr_size[k]
    a_data = {}
    pos = -(-pos // 2048) * 2048                 # data blocks start on a 2 KiB boundary like the reference's files
    for k in names:
        a_data[k] = pos
        pos += arrays[k].nbytes
    eof = pos

    out = bytearray()
    # superblock v0
    out += b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, 4, 16, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQII", 0, a_root, 1, 0) + struct.pack("<QQ", a_btree, a_heap)   # root symbol-table entry
    assert len(out) == sb_size
    out += _object_header_v1([_msg(0x0011, struct.pack("<QQ", a_btree, a_heap))] + attr_msgs)
    assert len(out) == a_heap
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, heap_size, free_off, a_heap_data)
    out += heap_data
    assert len(out) == a_btree
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
    node += struct.pack("<QQQ", 0, a_snod, name_off[names[-1]])            # key0 | child0 | key1
    out += node + b"\0" * (btree_size - len(node))
    assert len(out) == a_snod
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for k in names:
        snod += struct.pack("<QQII16x", name_off[k], a_hdr[k], 0, 0)
    out += snod + b"\0" * (snod_size - len(snod))
    for k in names:
        assert len(out) == a_hdr[k]
        out += dataset_header(arrays[k], a_data[k])
    out += b"\0" * (a_data[names[0]] - len(out))
    with open(path, "wb") as f:
        f.write(out)
        for k in names:
            f.write(arrays[k].tobytes())


def write_checkpoint(path, b, u, w, start_seed=0, v=None):
    """Write a checkpoint file in the reference's on-disk format (rbc_sim2D.jl:36-43,64-66; 3D: rbc_sim3D.jl writer):
    root attributes num_episodes / start_seed and datasets b,u,[v],w of Julia shape (E, Nx, Ny, Nz[+1]), i.e.
    (Nz[+1], Ny, Nx, E) in C order.  Inputs are (E, nz[+1], nx) for 2D or (E, nz[+1], ny, nx) for 3D.
    A path ending in .npz writes the npz container instead."""
    if str(path).endswith(".npz"):
        return write_checkpoint_npz(path, b, u, w, start_seed=start_seed, v=v)
    fields = {"b": b, "u": u, "w": w}
    if v is not None:
        fields["v"] = v
    out = {}
    for k, a in fields.items():
        a = np.asarray(a, np.float64)
        if a.ndim == 3:
            a = a[:, :, None, :]                  # (E, nz, 1, nx)
        if a.ndim != 4:
            raise ValueError(f"field {k}: expected (E, nz, nx) or (E, nz, ny, nx), got {a.shape}")
        out[k] = np.ascontiguousarray(np.moveaxis(a, 0, -1))      # (nz, ny, nx, E)
    write_hdf5(str(path), out, {"num_episodes": out["b"].shape[-1], "start_seed": int(start_seed)})
