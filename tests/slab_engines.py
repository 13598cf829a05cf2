"""CPU engine for the slab driver (pbf-sph_amd/slab.py): the oracle + numpy behind the same method
set as HipEngine, so that the driver's protocol (cuts, migration, ghost copies, per-phase refresh,
neighbour exchange) can be exercised with the "gloo" backend on CPU.  TEST INFRASTRUCTURE."""
import numpy as np
import torch

import oracle_lib as O

REC_MIGRANT, REC_GHOST, REC_FIELD = 0, 1, 2
GHOST = 2
FRAME_MARGIN = 6  # PBF_SLAB_FRAME_MARGIN (include/pbf_hip.h): keeps local x >= 0 for everything a rank can hold


def spread10(x):
    x = x & 0x3FF
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x


def shift_key_x(keys, shift):
    """Re-key into another rank-local x frame (mirrors shift_key_x in csrc/pbf_slab.hpp)."""
    k = keys.astype(np.int64)
    x = (compact10(k) + shift) & 1023
    return ((k & ~0x09249249) | spread10(x)).astype(np.uint64)


def compact10(v):
    v = v & 0x09249249
    v = (v | (v >> 2)) & 0x030C30C3
    v = (v | (v >> 4)) & 0x0300F00F
    v = (v | (v >> 8)) & 0x030000FF
    v = (v | (v >> 16)) & 0x3FF
    return v


class OracleEngine:
    torch = torch

    def __init__(self, fp64=False, device_pow=True):
        self.o = O.Oracle(fp64, device_pow=device_pow)
        self.fdt = np.float64 if fp64 else np.float32
        f = self.fdt
        self.mig_dt = np.dtype([("id", "<u8"), ("key", "<u8"), ("type", "u1"), ("pad", "u1", 7), ("mass", f),
                                ("pos", f, 3), ("vel", f, 3), ("colour", f, 4), ("pstar", f, 3), ("lam", f)])
        self.gho_dt = np.dtype([("key", "<u8"), ("type", "u1"), ("pad", "u1", 7), ("colour", f, 4), ("pstar", f, 3),
                                ("lam", f)])
        self.fld_dt = np.dtype([("pstar", f, 3), ("lam", f)])
        self.q = None
        self.n_owned_ = 0
        self.src = [np.zeros(0, np.int64), np.zeros(0, np.int64)]
        self.got = [0, 0]
        self.slot_of = None
        self.xoff, self.shift = 0, [0, 0]

    # -- plumbing ----------------------------------------------------------------------------------
    def alloc(self, nbytes):
        return torch.zeros(max(int(nbytes), 16), dtype=torch.uint8)

    def record_bytes(self, kind):
        return {REC_MIGRANT: self.mig_dt, REC_GHOST: self.gho_dt, REC_FIELD: self.fld_dt}[kind].itemsize

    def configure(self, cut, left_xlo, right_xlo):
        origin = lambda xlo, has_left: xlo - min(xlo, FRAME_MARGIN) if has_left and xlo > 0 else 0  # noqa: E731
        self.xoff = origin(cut[0], cut[2])
        self.shift = [origin(left_xlo, left_xlo > 0) - self.xoff, origin(right_xlo, True) - self.xoff]

    def _local(self, cut):
        return cut[0] - min(cut[0], self.xoff), cut[1] - min(cut[1], self.xoff), cut[2], cut[3]

    def upload(self, **sc):
        self.o.set_particles(**sc)
        self.n_owned_ = len(sc["id"])

    def sync(self):
        pass

    @property
    def n_owned(self):
        return self.n_owned_

    def _state(self):
        st = self.o.get_particles()
        st["key"], st["pstar"], st["lam"] = self.o.keys(), self.o.pstar(), self.o.lambdas()
        return st

    def _set(self, st):
        self.o.set_particles(st["id"], st["type"], st["mass"], st["pos"], st["vel"], st["colour"])
        self.o.set_scratch(st["key"], st["pstar"], st["lam"])

    @staticmethod
    def _take(st, idx):
        return {k: v[idx] for k, v in st.items()}

    @staticmethod
    def _cat(a, b):
        return {k: np.concatenate([a[k], b[k]]) for k in a}

    def _write(self, buf, arr):
        raw = np.frombuffer(arr.tobytes(), np.uint8)
        buf[:len(raw)] = torch.from_numpy(raw.copy())

    def _read(self, buf, dt, n):
        return np.frombuffer(buf[:n * dt.itemsize].numpy().tobytes(), dt)

    def _oparams(self, p):
        q = O.make_params(dt=p.dt, scale=p.scale, iteration=p.iteration, force=tuple(p.constant_force),
                          min_bound=tuple(p.min_bound), max_bound=tuple(p.max_bound), mode=O.JACOBI,
                          sort=O.SORT_STABLE, threads=1)
        return q

    # -- engine interface --------------------------------------------------------------------------
    def predict(self, p):
        self.q = self._oparams(p)
        self.o.predict(self.q)
        if self.xoff:  # keys in the rank-local x frame, like k_predict with StepConsts::xoff
            self.o.set_scratch(shift_key_x(self.o.keys(), -self.xoff), None, None)

    def migrate(self, cut, send_l, send_r, cap):
        xlo, xhi, has_l, has_r = self._local(cut)
        st = self._state()
        cx = compact10(st["key"].astype(np.int64))
        ghost = (st["type"] & GHOST) != 0
        to_l = (~ghost) & has_l & (cx < xlo)
        to_r = (~ghost) & has_r & (cx >= xhi) & ~to_l
        keep = (~ghost) & ~to_l & ~to_r
        for mask, buf in ((to_l, send_l), (to_r, send_r)):
            sub = self._take(st, np.flatnonzero(mask))
            rec = np.zeros(len(sub["id"]), self.mig_dt)
            for k in ("id", "key", "type", "mass", "pos", "vel", "colour", "pstar", "lam"):
                rec[k] = sub[k]
            assert len(rec) <= cap
            self._write(buf, rec)
        self._set(self._take(st, np.flatnonzero(keep)))
        self.n_owned_ = int(keep.sum())
        return int(to_l.sum()), int(to_r.sum())

    def add_migrants(self, recv_l, n_l, recv_r, n_r):
        st = self._state()
        for side, (buf, n) in enumerate(((recv_l, n_l), (recv_r, n_r))):
            if n:
                rec = self._read(buf, self.mig_dt, n)
                add = {k: rec[k].copy() for k in st}
                add["key"] = shift_key_x(add["key"], self.shift[side])
                st = self._cat(st, add)
        self._set(st)
        self.n_owned_ = len(st["id"])

    def ghosts(self, cut, send_l, send_r, cap):
        xlo, xhi, has_l, has_r = self._local(cut)
        st = self._state()
        cx = compact10(st["key"].astype(np.int64))
        sel = [np.flatnonzero((cx == xlo) & has_l), np.flatnonzero((cx + 1 == xhi) & has_r)]
        for idx, buf in zip(sel, (send_l, send_r)):
            rec = np.zeros(len(idx), self.gho_dt)
            for k in ("key", "colour", "pstar", "lam"):
                rec[k] = st[k][idx]
            rec["type"] = st["type"][idx] | GHOST
            assert len(rec) <= cap
            self._write(buf, rec)
        self.src = sel
        return len(sel[0]), len(sel[1])

    def add_ghosts(self, recv_l, n_l, recv_r, n_r):
        st = self._state()
        for side, (buf, n) in enumerate(((recv_l, n_l), (recv_r, n_r))):
            if n:
                rec = self._read(buf, self.gho_dt, n)
                add = {"id": np.full(n, 2 ** 64 - 1, np.uint64), "type": rec["type"].copy(),
                       "mass": np.zeros(n, self.fdt), "pos": np.zeros((n, 3), self.fdt),
                       "vel": np.zeros((n, 3), self.fdt), "colour": rec["colour"].copy(),
                       "key": shift_key_x(rec["key"], self.shift[side]),
                       "pstar": rec["pstar"].copy(), "lam": rec["lam"].copy()}
                st = self._cat(st, add)
        self.got = [n_l, n_r]
        self._set(st)

    def stage(self, name, p):
        if name == "sort":
            keys = self.o.keys()
            perm = np.argsort(keys, kind="stable")
            self.slot_of = np.empty(len(perm), np.int64)
            self.slot_of[perm] = np.arange(len(perm))
            self.o.sort(self.q).grid_table(self.q)
        else:
            getattr(self.o, {"lambda": "lambda_"}.get(name, name))(self.q)

    def pack(self, send_l, send_r):
        ps, la = self.o.pstar(), self.o.lambdas()
        for idx, buf in zip(self.src, (send_l, send_r)):
            slots = self.slot_of[idx]
            rec = np.zeros(len(idx), self.fld_dt)
            rec["pstar"], rec["lam"] = ps[slots], la[slots]
            self._write(buf, rec)

    def unpack(self, recv_l, recv_r):
        ps, la = self.o.pstar(), self.o.lambdas()
        at = self.n_owned_
        for buf, n in ((recv_l, self.got[0]), (recv_r, self.got[1])):
            if n:
                rec = self._read(buf, self.fld_dt, n)
                slots = self.slot_of[at:at + n]
                ps[slots], la[slots] = rec["pstar"], rec["lam"]
            at += n
        self.o.set_scratch(None, ps, la)

    # -- opt-in extras in slab mode: the owners refresh their copies' velocity / vorticity between the sub-stages --
    def extras_stage(self, name, p):
        getattr(self.o, name)(self.q)

    def pack_vec(self, which, send_l, send_r):
        v = self.o.get_vec(which)
        for idx, buf in zip(self.src, (send_l, send_r)):
            self._write(buf, np.ascontiguousarray(v[self.slot_of[idx]]))

    def unpack_vec(self, which, recv_l, recv_r):
        v = self.o.get_vec(which)
        at = self.n_owned_
        dt3 = np.dtype((self.fdt, 3))
        for buf, n in ((recv_l, self.got[0]), (recv_r, self.got[1])):
            if n:
                v[self.slot_of[at:at + n]] = self._read(buf, dt3, n)
            at += n
        self.o.set_vec(which, v)

    def finish(self):
        st = self._state()
        keep = np.flatnonzero((st["type"] & GHOST) == 0)
        self._set(self._take(st, keep))
        self.n_owned_ = len(keep)

    def column_histogram(self):
        """Owned particles per GLOBAL column as of the last predict (keys are kept in the rank-local frame)."""
        st = self._state()
        own = (st["type"] & GHOST) == 0
        cx = compact10(st["key"].astype(np.int64))[own] + self.xoff
        return np.bincount(cx, minlength=1024)[:1024].astype(np.int64)

    def download(self):
        return self.o.get_particles()
