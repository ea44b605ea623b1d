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
