"""Parthenon-style HDF5 dumps (``*.phdf``) of the fields and the photon swarm.

What the reference's tools read from a dump (reference analysis/jhdf.py:32-92,
tst/regression_test.py:361-381, analysis/plot.py:60-95, output block of
inputs/stepdiff_smr.in:86-94): ``Time``, ``NumBlocks``, ``NumDims``, ``MeshBlockSize``,
``BlockBounds`` (from the per-block node coordinates), ``Variables`` / ``Get(name)`` as
``[block, k, j, i]`` arrays of interior cells, and for swarms ``GetSwarm("photons")`` with
``x``, ``y``, ``z`` and ``Get("id")``.  Parthenon's reader (``phdf.py``) is not part of the
reference tree, so compatibility with it is UNVERIFIED; this module writes the layout with
root-level ``/Levels`` and ``/LogicalLocations`` -- the one reference analysis/jhdf.py:91-101 names
(its list of non-variable root entries: Blocks, Info, Input, Levels, Locations, LogicalLocations,
Params, SparseInfo, VolumeLocations) -- and labels it ``OutputFormatVersion = 3`` (Parthenon's
version 4 moved block metadata to ``/Blocks/loc.*``, which this writer does not produce) --

    /Info                attributes: Time, dt, NCycle, NumDims, NumMeshBlocks, MeshBlockSize[3],
                         MaxLevel, IncludesGhost, NGhost, Coordinates, OutputFormatVersion,
                         OutputDatasetNames, NumComponents, ComponentNames, BlocksPerPE,
                         RootGridDomain[9]
    /Input  (attribute File: the deck)      /Params  (empty: the package parameters are not dumped)
    /Blocks/xmin         [nb, 3]        /Levels  [nb]        /LogicalLocations  [nb, 3]
    /Locations/x,y,z     [nb, n+1]   node coordinates      /VolumeLocations/x,y,z  [nb, n] centres
    /<variable>          [nb, nk, nj, ni]                   (e.g. field.jaybenne.energy_tally)
    /<swarm>/counts, offsets  [nb]   /<swarm>/SwarmVars/<name>  [nparticles], grouped by block

-- through the HDF5 C library (``libhdf5``) via ctypes: there is no h5py in this image.  When the
library cannot be found ``available()`` is False and the command line falls back to ``.npz``.
``read_dump`` is the matching minimal reader (what jhdf exposes), used by the tests and by
``python -m jaybenne_amd --check-dump``.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
from typing import Dict, Optional

import numpy as np

_lib = None
_hid = C.c_int64      # hid_t (HDF5 >= 1.10)
_F_ACC_TRUNC, _F_ACC_RDONLY = 0x0002, 0x0000
_P_DEFAULT = 0
_S_ALL = 0
_S_SCALAR = 0


def _find() -> Optional[str]:
    cands = [os.environ.get("JAYBENNE_HDF5_LIB"), ctypes.util.find_library("hdf5"),
             "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so",
             "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "libhdf5.so"]
    for c in cands:
        if not c:
            continue
        try:
            C.CDLL(c)
            return c
        except OSError:
            continue
    return None


def _load():
    global _lib
    if _lib is None:
        path = _find()
        if path is None:
            raise RuntimeError("libhdf5 not found (set JAYBENNE_HDF5_LIB); use the .npz output")
        L = C.CDLL(path)
        L.H5open()
        for name, res, args in [
                ("H5Fcreate", _hid, [C.c_char_p, C.c_uint, _hid, _hid]),
                ("H5Fopen", _hid, [C.c_char_p, C.c_uint, _hid]),
                ("H5Fclose", C.c_int, [_hid]),
                ("H5Gcreate2", _hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
                ("H5Gopen2", _hid, [_hid, C.c_char_p, _hid]),
                ("H5Gclose", C.c_int, [_hid]),
                ("H5Screate_simple", _hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                ("H5Screate", _hid, [C.c_int]),
                ("H5Sclose", C.c_int, [_hid]),
                ("H5Sget_simple_extent_ndims", C.c_int, [_hid]),
                ("H5Sget_simple_extent_dims", C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                ("H5Dcreate2", _hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
                ("H5Dopen2", _hid, [_hid, C.c_char_p, _hid]),
                ("H5Dwrite", C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
                ("H5Dread", C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
                ("H5Dget_space", _hid, [_hid]),
                ("H5Dget_type", _hid, [_hid]),
                ("H5Dclose", C.c_int, [_hid]),
                ("H5Acreate2", _hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]),
                ("H5Aopen", _hid, [_hid, C.c_char_p, _hid]),
                ("H5Awrite", C.c_int, [_hid, _hid, C.c_void_p]),
                ("H5Aread", C.c_int, [_hid, _hid, C.c_void_p]),
                ("H5Aget_space", _hid, [_hid]),
                ("H5Aget_type", _hid, [_hid]),
                ("H5Aclose", C.c_int, [_hid]),
                ("H5Tcopy", _hid, [_hid]),
                ("H5Tset_size", C.c_int, [_hid, C.c_size_t]),
                ("H5Tget_size", C.c_size_t, [_hid]),
                ("H5Tget_class", C.c_int, [_hid]),
                ("H5Tclose", C.c_int, [_hid]),
                ("H5Lexists", C.c_int, [_hid, C.c_char_p, _hid]),
                ("H5Literate", C.c_int, [_hid, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p])]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L.t_f64 = _hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
        L.t_i32 = _hid.in_dll(L, "H5T_NATIVE_INT_g").value
        L.t_i64 = _hid.in_dll(L, "H5T_NATIVE_LLONG_g").value
        L.t_u64 = _hid.in_dll(L, "H5T_NATIVE_ULLONG_g").value
        L.t_str = _hid.in_dll(L, "H5T_C_S1_g").value
        _lib = L
    return _lib


def available() -> bool:
    try:
        _load()
        return True
    except (RuntimeError, OSError, ValueError):
        return False


def _h5type(L, a: np.ndarray):
    if a.dtype == np.float64:
        return L.t_f64
    if a.dtype == np.int32:
        return L.t_i32
    if a.dtype == np.int64:
        return L.t_i64
    if a.dtype == np.uint64:
        return L.t_u64
    raise TypeError(f"no HDF5 type for {a.dtype}")


def _check(h, what):
    if h < 0:
        raise IOError(f"HDF5 call failed: {what}")
    return h


class _Writer:
    def __init__(self, path: str):
        self.L = _load()
        self.f = _check(self.L.H5Fcreate(path.encode(), _F_ACC_TRUNC, _P_DEFAULT, _P_DEFAULT), "H5Fcreate")

    def group(self, name: str):
        g = _check(self.L.H5Gcreate2(self.f, name.encode(), _P_DEFAULT, _P_DEFAULT, _P_DEFAULT), "H5Gcreate2")
        self.L.H5Gclose(g)

    def dataset(self, name: str, a: np.ndarray):
        a = np.ascontiguousarray(a)
        dims = (C.c_uint64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
        sp = _check(self.L.H5Screate_simple(max(a.ndim, 1), dims, None), "H5Screate_simple")
        t = _h5type(self.L, a)
        d = _check(self.L.H5Dcreate2(self.f, name.encode(), t, sp, _P_DEFAULT, _P_DEFAULT, _P_DEFAULT), name)
        if a.size:
            _check(self.L.H5Dwrite(d, t, _S_ALL, _S_ALL, _P_DEFAULT, a.ctypes.data), "H5Dwrite " + name)
        self.L.H5Dclose(d)
        self.L.H5Sclose(sp)

    def attr(self, obj: str, name: str, value):
        L = self.L
        g = _check(L.H5Gopen2(self.f, obj.encode(), _P_DEFAULT), obj)
        if isinstance(value, str):
            raw = value.encode()
            t = L.H5Tcopy(L.t_str)
            L.H5Tset_size(t, max(len(raw), 1))
            sp = L.H5Screate(_S_SCALAR)
            a = _check(L.H5Acreate2(g, name.encode(), t, sp, _P_DEFAULT, _P_DEFAULT), name)
            buf = C.create_string_buffer(raw, max(len(raw), 1))
            L.H5Awrite(a, t, buf)
            L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)
        else:
            arr = np.atleast_1d(np.asarray(value))
            if arr.dtype.kind == "i":
                arr = arr.astype(np.int32) if np.abs(arr).max(initial=0) < 2 ** 31 else arr.astype(np.int64)
            elif arr.dtype.kind == "f":
                arr = arr.astype(np.float64)
            dims = (C.c_uint64 * 1)(arr.size)
            sp = L.H5Screate_simple(1, dims, None) if np.ndim(value) else L.H5Screate(_S_SCALAR)
            t = _h5type(L, arr)
            a = _check(L.H5Acreate2(g, name.encode(), t, sp, _P_DEFAULT, _P_DEFAULT), name)
            L.H5Awrite(a, t, arr.ctypes.data)
            L.H5Aclose(a); L.H5Sclose(sp)
        L.H5Gclose(g)

    def close(self):
        self.L.H5Fclose(self.f)


def write_dump(path: str, mesh, time: float, dt: float, ncycle: int,
               variables: Dict[str, np.ndarray], swarms: Optional[Dict[str, Dict[str, np.ndarray]]] = None,
               input_text: str = "") -> None:
    """``variables``: name -> ``[nblocks, nk, nj, ni]`` WITH ghost zones (as the fields are held);
    the interior is written (IncludesGhost = 0).  ``swarms``: swarm name -> {"blk": global block id
    per particle, "swarm.x": ..., "id": ...}; particles are written grouped by block."""
    w = _Writer(path)
    nb = mesh.nblocks
    sl = mesh.interior()
    nx = [int(v) for v in mesh.nx]
    w.group("Info")
    w.attr("Info", "OutputFormatVersion", 3)   # root-level /Levels, /LogicalLocations (see above)
    w.attr("Info", "Time", float(time))
    w.attr("Info", "dt", float(dt))
    w.attr("Info", "NCycle", int(ncycle))
    w.attr("Info", "NumDims", int(mesh.ndim))
    w.attr("Info", "NumMeshBlocks", int(nb))
    w.attr("Info", "MeshBlockSize", np.array(nx, dtype=np.int32))
    w.attr("Info", "MaxLevel", int(np.max(mesh.blk_level)))
    w.attr("Info", "IncludesGhost", 0)
    w.attr("Info", "NGhost", int(mesh.ng))
    w.attr("Info", "Coordinates", "UniformCartesian")
    w.attr("Info", "OutputDatasetNames", ",".join(variables))
    w.attr("Info", "NumComponents", np.ones(max(len(variables), 1), dtype=np.int32))  # all scalars
    w.attr("Info", "ComponentNames", ",".join(variables))
    w.attr("Info", "BlocksPerPE", np.array([nb], dtype=np.int32))   # written by one process
    root = np.array([mesh.gmin[0], mesh.gmax[0], 1.0, mesh.gmin[1], mesh.gmax[1], 1.0,
                     mesh.gmin[2], mesh.gmax[2], 1.0])
    w.attr("Info", "RootGridDomain", root)
    w.group("Input")
    w.attr("Input", "File", input_text if input_text else "(no input deck recorded)")
    w.group("Params")
    w.group("Blocks")
    w.dataset("Blocks/xmin", np.asarray(mesh.blk_xmin, dtype=np.float64))
    w.dataset("Levels", np.asarray(mesh.blk_level, dtype=np.int32))
    dxs = np.asarray(mesh.blk_dx, dtype=np.float64)
    lloc = np.rint((np.asarray(mesh.blk_xmin) - np.asarray(mesh.gmin)[None, :]) /
                   (dxs * np.array(nx)[None, :])).astype(np.int64)
    lloc[:, mesh.ndim:] = 0
    w.dataset("LogicalLocations", lloc)
    w.group("Locations")
    w.group("VolumeLocations")
    for d, ax in enumerate("xyz"):
        nodes = np.asarray(mesh.blk_xmin)[:, d, None] + np.arange(nx[d] + 1)[None, :] * dxs[:, d, None]
        cents = np.asarray(mesh.blk_xmin)[:, d, None] + (np.arange(nx[d])[None, :] + 0.5) * dxs[:, d, None]
        w.dataset(f"Locations/{ax}", nodes)
        w.dataset(f"VolumeLocations/{ax}", cents)
    for name, arr in variables.items():
        w.dataset(name, np.asarray(arr, dtype=np.float64)[sl])
    for sname, sv in (swarms or {}).items():
        blk = np.asarray(sv["blk"], dtype=np.int64)
        order = np.argsort(blk, kind="stable")
        counts = np.bincount(blk, minlength=nb).astype(np.int64)
        offsets = np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.int64)
        w.group(sname)
        w.group(sname + "/SwarmVars")
        w.dataset(sname + "/counts", counts)
        w.dataset(sname + "/offsets", offsets)
        for vname, arr in sv.items():
            if vname != "blk":
                w.dataset(f"{sname}/SwarmVars/{vname}", np.ascontiguousarray(np.asarray(arr)[order]))
    w.close()


# ------------------------------------------------------------------------------------------------
class _Swarm:
    def __init__(self, counts, offsets, variables):
        self.counts, self.offsets, self.variables = counts, offsets, variables
        self.x = variables.get("swarm.x")
        self.y = variables.get("swarm.y")
        self.z = variables.get("swarm.z")

    def Get(self, name):
        return self.variables.get(name)


class Dump:
    """What reference analysis/jhdf.py exposes of a dump."""

    def __init__(self, path: str):
        L = _load()
        f = _check(L.H5Fopen(path.encode(), _F_ACC_RDONLY, _P_DEFAULT), "H5Fopen " + path)
        self._L, self._f = L, f
        self.Time = float(self._attr("Info", "Time", np.float64)[0])
        self.NCycle = int(self._attr("Info", "NCycle", np.int32)[0])
        self.NumDims = int(self._attr("Info", "NumDims", np.int32)[0])
        self.NumBlocks = int(self._attr("Info", "NumMeshBlocks", np.int32)[0])
        self.MeshBlockSize = self._attr("Info", "MeshBlockSize", np.int32)
        self.Variables = self._attr_str("Info", "OutputDatasetNames").split(",")
        self.NX1, self.NX2, self.NX3 = (int(v) for v in self.MeshBlockSize)
        x, y, z = (self._dset("Locations/" + a) for a in "xyz")
        self.BlockBounds = [(x[b, 0], x[b, -1], y[b, 0], y[b, -1], z[b, 0], z[b, -1])
                            for b in range(self.NumBlocks)]
        self.Levels = self._dset("Levels")
        xc, yc, zc = (self._dset("VolumeLocations/" + a) for a in "xyz")
        shape = (self.NumBlocks, self.NX3, self.NX2, self.NX1)
        self.X1c = np.broadcast_to(xc[:, None, None, :], shape)
        self.X2c = np.broadcast_to(yc[:, None, :, None], shape)
        self.X3c = np.broadcast_to(zc[:, :, None, None], shape)

    def _attr(self, obj, name, dtype):
        L = self._L
        g = L.H5Gopen2(self._f, obj.encode(), _P_DEFAULT)
        a = _check(L.H5Aopen(g, name.encode(), _P_DEFAULT), name)
        sp = L.H5Aget_space(a)
        nd = L.H5Sget_simple_extent_ndims(sp)
        n = 1
        if nd > 0:
            dims = (C.c_uint64 * nd)()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            n = int(np.prod(list(dims)))
        out = np.zeros(n, dtype=dtype)
        L.H5Aread(a, _h5type(L, out), out.ctypes.data)
        L.H5Sclose(sp); L.H5Aclose(a); L.H5Gclose(g)
        return out

    def _attr_str(self, obj, name):
        L = self._L
        g = L.H5Gopen2(self._f, obj.encode(), _P_DEFAULT)
        a = _check(L.H5Aopen(g, name.encode(), _P_DEFAULT), name)
        t = L.H5Aget_type(a)
        n = L.H5Tget_size(t)
        buf = C.create_string_buffer(n + 1)
        L.H5Aread(a, t, buf)
        L.H5Tclose(t); L.H5Aclose(a); L.H5Gclose(g)
        return buf.raw[:n].rstrip(b"\0").decode()

    def _dset(self, name):
        L = self._L
        if L.H5Lexists(self._f, name.split("/")[0].encode(), _P_DEFAULT) <= 0:
            return None
        d = L.H5Dopen2(self._f, name.encode(), _P_DEFAULT)
        if d < 0:
            return None
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * nd)()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        t = L.H5Dget_type(d)
        size, cls = L.H5Tget_size(t), L.H5Tget_class(t)   # class 0 integer, 1 float
        dtype = np.float64 if cls == 1 else (np.int32 if size == 4 else np.int64)
        out = np.zeros(tuple(dims), dtype=dtype)
        if out.size:
            L.H5Dread(d, _h5type(L, out), _S_ALL, _S_ALL, _P_DEFAULT, out.ctypes.data)
        L.H5Tclose(t); L.H5Sclose(sp); L.H5Dclose(d)
        return out

    def Get(self, variable_name, flatten=False):
        if variable_name not in self.Variables:
            return None
        v = self._dset(variable_name)
        return v.reshape(-1) if flatten else v

    def GetSwarm(self, name):
        counts = self._dset(name + "/counts")
        if counts is None:
            return None
        offsets = self._dset(name + "/offsets")
        variables = {}
        for vname in ("swarm.x", "swarm.y", "swarm.z", "id", "weight", "time", "energy"):
            v = self._dset(f"{name}/SwarmVars/{vname}")
            if v is not None:
                variables[vname] = v
        return _Swarm(counts, offsets, variables)

    def close(self):
        if self._f is not None:
            self._L.H5Fclose(self._f)
            self._f = None


def read_dump(path: str) -> Dump:
    return Dump(path)
