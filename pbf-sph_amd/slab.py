"""Slab decomposition of the PBF-SPH step across GPUs: one rank per GPU.  CSlabSolver (production) attaches a
communicator and lets libpbf_hip.so run the whole step incl. the RCCL send/recv exchange (pbf_slab_step); SlabSolver
is the same protocol sequenced in Python for the CPU twin the tests compare against.

The reference is single-device — this layer has no counterpart there; its
correctness statement is "N ranks == 1 rank".  Layout: rank g owns the cell columns
[cuts[g], cuts[g+1]) of the GLOBAL grid (every rank passes the same bounds) plus a one-cell layer of
copies ("ghosts") of its x-neighbours' boundary columns.  Keys are kept in a rank-local x frame
(origin = PBF_SLAB_FRAME_MARGIN columns left of the slab; records are re-keyed on arrival), so each rank's grid table covers its
slab only — constant size under weak scaling instead of Morton(global extent).  Per step:

    predict ─ migrate ⇄ add_migrants ─ ghosts ⇄ add_ghosts ─ sort ─ diffuse ─
    K × { lambda ─ pack ⇄ unpack ─ delta ─ pack ⇄ unpack } ─ finalise ─ finish

⇄ = neighbour exchange (point-to-point send/recv with the left and right rank, one batch per
phase; messages are O(100 KB – few MB), so the cost is latency, not xGMI bandwidth).  All numerics
stay in the engine (libpbf_hip.so through the C ABI); this file only sequences and moves bytes.
"""
import ctypes as C

import numpy as np

REC_MIGRANT, REC_GHOST, REC_FIELD = 0, 1, 2


def column_of(x_world, scale=500.0, h=0.1, min_bound_x=0.0):
    """Grid column of a world-space x (ompsph.hpp:132-135,152: (x/scale - minExtent)/h, padding 2h)."""
    return int((x_world / scale - (min_bound_x / scale - 2 * h)) / h)


def columns_of(x_world, scale=500.0, h=0.1):
    """Vectorised column_of for min_bound_x = 0 (every bench / test scene)."""
    return ((np.asarray(x_world, np.float64) / scale + 2 * h) / h).astype(np.int64)


def even_cuts(nranks, box_x, scale=500.0, h=0.1):
    """Equal-width slabs over [0, box_x]; the first / last slab also take the padding columns."""
    cols = [column_of(box_x * g / nranks, scale, h) for g in range(nranks + 1)]
    cols[0], cols[-1] = 0, 1024
    return cols


def balanced_cuts(nranks, x_world_all, box_x, scale=500.0, h=0.1):
    """Cuts at particle-count quantiles of the column histogram (call with the gathered x of all ranks)."""
    order = np.sort(columns_of(x_world_all, scale, h))
    cuts = [0]
    for g in range(1, nranks):
        cuts.append(int(order[min(len(order) - 1, (len(order) * g) // nranks)]))
    cuts.append(1024)
    for g in range(1, nranks + 1):  # strictly increasing, at least one column each
        cuts[g] = max(cuts[g], cuts[g - 1] + 1)
    return cuts


MIN_SLAB_COLUMNS = 4   # a particle moves < 2 columns per step (v dt < 2 h): it then always lands in an ADJACENT slab
MAX_CUT_MOVE = 2       # columns a cut may move per re-cut: transferred columns go to the adjacent rank only


def recut(old_cuts, histogram, max_move=MAX_CUT_MOVE, min_width=MIN_SLAB_COLUMNS):
    """New cuts from the global per-column particle histogram (1024 bins): particle-count quantiles, each cut moved
    by at most `max_move` columns from where it is (so that every particle's new owner is its old owner or a direct
    neighbour: migration is point-to-point between adjacent slabs) and slabs kept >= `min_width` columns wide where the
    old ones were.  Deterministic: every rank computes the same cuts from the same histogram."""
    n = len(old_cuts) - 1
    h = np.asarray(histogram, np.int64)
    cum = np.concatenate([[0], np.cumsum(h)])
    total = int(cum[-1])
    new = list(old_cuts)
    for g in range(1, n):
        want = int(np.searchsorted(cum, total * g / n, side="left"))  # smallest c with cum[c] >= target
        new[g] = int(min(max(want, old_cuts[g] - max_move), old_cuts[g] + max_move))
    for g in range(1, n):      # keep the widths (left to right, then right to left): never below min_width
        new[g] = max(new[g], new[g - 1] + (min_width if g > 1 else 1))
    for g in range(n - 1, 0, -1):
        new[g] = min(new[g], new[g + 1] - (min_width if g < n - 1 else 1))
    for g in range(1, n):      # the clamps above never undo the max_move bound by more than they must
        new[g] = int(min(max(new[g], old_cuts[g] - max_move), old_cuts[g] + max_move))
    return new


class HipEngine:
    """The product engine: pbf_sph_amd.Solver (C ABI) + torch device buffers for the wire."""

    def __init__(self, solver, torch, device):
        self.s, self.torch, self.device = solver, torch, device
        self.L, self.ctx = solver.L, solver.ctx
        self.bytes = {k: self.L.pbf_slab_record_bytes(self.ctx, k) for k in (REC_MIGRANT, REC_GHOST, REC_FIELD)}

    def alloc(self, nbytes):
        return self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def record_bytes(self, kind):
        return self.bytes[kind]

    def _cut(self, cut):
        from . import capi
        return capi.SlabCut(cut[0], cut[1], int(cut[2]), int(cut[3]))

    def configure(self, cut, left_xlo, right_xlo):
        """Rank-local key frame: the grid table then covers the slab + ghost columns only."""
        c = self._cut(cut)
        self.s._chk(self.L.pbf_slab_configure(self.ctx, C.byref(c), left_xlo, right_xlo), "pbf_slab_configure")

    def predict(self, p):
        self.s.stage("predict", p)

    def migrate(self, cut, send_l, send_r, cap):
        c, out = self._cut(cut), (C.c_uint32 * 2)()
        self.s._chk(self.L.pbf_slab_migrate(self.ctx, C.byref(c), self._p(send_l), self._p(send_r), cap, out),
                    "pbf_slab_migrate")
        return int(out[0]), int(out[1])

    def add_migrants(self, recv_l, n_l, recv_r, n_r):
        self.s._chk(self.L.pbf_slab_add_migrants(self.ctx, self._p(recv_l), n_l, self._p(recv_r), n_r),
                    "pbf_slab_add_migrants")

    def ghosts(self, cut, send_l, send_r, cap):
        c, out = self._cut(cut), (C.c_uint32 * 2)()
        self.s._chk(self.L.pbf_slab_ghosts(self.ctx, C.byref(c), self._p(send_l), self._p(send_r), cap, out),
                    "pbf_slab_ghosts")
        return int(out[0]), int(out[1])

    def add_ghosts(self, recv_l, n_l, recv_r, n_r):
        self.s._chk(self.L.pbf_slab_add_ghosts(self.ctx, self._p(recv_l), n_l, self._p(recv_r), n_r),
                    "pbf_slab_add_ghosts")

    def stage(self, name, p):
        self.s.stage(name, p)

    def pack(self, send_l, send_r):
        self.s._chk(self.L.pbf_slab_pack(self.ctx, self._p(send_l), self._p(send_r)), "pbf_slab_pack")

    def unpack(self, recv_l, recv_r):
        self.s._chk(self.L.pbf_slab_unpack(self.ctx, self._p(recv_l), self._p(recv_r)), "pbf_slab_unpack")

    def finish(self):
        self.s._chk(self.L.pbf_slab_finish(self.ctx), "pbf_slab_finish")

    def column_histogram(self):
        """Owned particles per GLOBAL grid column (1024 bins) as of the last predict (device histogram of the keys)."""
        out = np.zeros(1024, np.uint32)
        self.s._chk(self.L.pbf_slab_column_histogram(self.ctx, out.ctypes.data_as(C.c_void_p)),
                    "pbf_slab_column_histogram")
        return out.astype(np.int64)

    def sync(self):
        self.s.sync()

    @property
    def n_owned(self):
        return self.L.pbf_owned_count(self.ctx)


class GlooHostTransport:
    """The library's host-callback transport over torch.distributed point-to-point (gloo): the library hands pinned
    HOST buffers, this moves them to / from rank - 1 and rank + 1 in one batched round.  Lets several ranks share
    ONE GPU for tests; production uses the RCCL transport (pbf_comm_create_rccl)."""

    def __init__(self, dist, torch, rank, nranks):
        from . import capi
        self.dist, self.torch, self.rank, self.nranks = dist, torch, rank, nranks
        self.rounds = 0
        self.fn = capi.EXCHANGE_FN(self._exchange)  # keep the ctypes thunk alive

    def _exchange(self, user, s_l, n_sl, s_r, n_sr, r_l, n_rl, r_r, n_rr):
        try:
            t = self.torch
            ops = []

            def view(ptr, n):
                return t.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)))

            for peer, ptr, n, recv in ((self.rank - 1, r_l, n_rl, True), (self.rank + 1, r_r, n_rr, True),
                                       (self.rank - 1, s_l, n_sl, False), (self.rank + 1, s_r, n_sr, False)):
                if n and 0 <= peer < self.nranks:
                    ops.append(self.dist.P2POp(self.dist.irecv if recv else self.dist.isend, view(ptr, n), peer))
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
                self.rounds += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("GlooHostTransport:", repr(e), flush=True)
            return 1


class CSlabSolver:
    """One rank of the slab decomposition with the WHOLE step — kernels and exchanges — inside libpbf_hip.so
    (pbf_slab_step): this class only creates the communicator, attaches it and drives the load balance.

    transport "rccl": ncclSend / ncclRecv over xGMI on the solver's stream (the unique id travels over `dist`);
    transport "gloo-host": host-staged callback (tests, several ranks on one GPU)."""

    def __init__(self, solver, dist, torch, rank, nranks, cuts, cap_migrants, cap_ghosts, transport="rccl",
                 rebalance_every=0, device=0):
        from . import capi
        self.s, self.L, self.dist, self.torch = solver, solver.L, dist, torch
        self.rank, self.nranks, self.transport = rank, nranks, transport
        self.comm = C.c_void_p()
        if transport == "rccl":
            ident = (C.c_uint8 * 128)()
            box = [None]
            if rank == 0:
                self._chk_comm(self.L.pbf_comm_unique_id(ident), "pbf_comm_unique_id")
                box = [bytes(ident)]
            if nranks > 1:
                dist.broadcast_object_list(box, src=0)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            self._chk_comm(self.L.pbf_comm_create_rccl(ident, nranks, rank, device, C.byref(self.comm)),
                           "pbf_comm_create_rccl")
            self.host = None
        else:
            self.host = GlooHostTransport(dist, torch, rank, nranks)
            self._chk_comm(self.L.pbf_comm_create_host_callback(self.host.fn, None, nranks, rank, C.byref(self.comm)),
                           "pbf_comm_create_host_callback")
        self.cuts = [int(c) for c in cuts]
        arr = (C.c_uint32 * (nranks + 1))(*self.cuts)
        solver._chk(self.L.pbf_slab_attach(solver.ctx, self.comm, arr, int(cap_migrants), int(cap_ghosts)),
                    "pbf_slab_attach")
        self.rebalance_every = int(rebalance_every)
        self.frame = 0
        self.stats = dict(recuts=0)

    def _chk_comm(self, rc, what):
        if rc != 0:
            msg = self.L.pbf_comm_last_error(None)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    @property
    def rounds(self):
        return int(self.L.pbf_comm_rounds(self.comm))

    @property
    def n_owned(self):
        return self.L.pbf_owned_count(self.s.ctx)

    def rebalance(self):
        """Re-cut from the global column histogram (device histogram -> all-reduce -> recut()).  Collective."""
        if self.nranks == 1:
            return False
        h = np.zeros(1024, np.uint32)
        self.s._chk(self.L.pbf_slab_column_histogram(self.s.ctx, h.ctypes.data_as(C.c_void_p)),
                    "pbf_slab_column_histogram")
        t = self.torch.from_numpy(h.astype(np.int64))
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        new = recut(self.cuts, t.cpu().numpy())
        if new != self.cuts:
            self.cuts = new
            arr = (C.c_uint32 * (self.nranks + 1))(*new)
            self.s._chk(self.L.pbf_slab_set_cuts(self.s.ctx, arr), "pbf_slab_set_cuts")
            self.stats["recuts"] += 1
            return True
        return False

    def step(self, p):
        if self.rebalance_every and self.frame and self.frame % self.rebalance_every == 0:
            self.rebalance()
        self.frame += 1
        self.s._chk(self.L.pbf_slab_step(self.s.ctx, C.byref(p)), "pbf_slab_step")

    def steps(self, p, count):
        if not self.rebalance_every:
            self.s._chk(self.L.pbf_slab_steps(self.s.ctx, C.byref(p), count), "pbf_slab_steps")
            self.frame += count
            return
        for _ in range(count):
            self.step(p)

    def close(self):
        if self.comm:
            self.L.pbf_comm_destroy(self.comm)
            self.comm = C.c_void_p()


class SlabSolver:
    """The slab protocol sequenced in Python, one engine call per phase: the readable twin of pbf_slab_step
    (csrc/pbf_hip.hip slab_step_impl).  The tests run it with the CPU oracle as the engine (tests/slab_engines.py,
    gloo, no GPU needed) and hold the library's step to its results bit for bit; with HipEngine it drives the
    library's low-level pbf_slab_* entry points (wire buffers bounced through host tensors).  Production runs
    CSlabSolver: the whole step and its RCCL exchanges inside libpbf_hip.so.

    engine : OracleEngine / HipEngine
    dist   : torch.distributed (initialised), or None for a single rank
    cuts   : nranks + 1 column boundaries, identical on every rank
    """

    def __init__(self, engine, dist, rank, nranks, cuts, cap_records, stage_via_host=True, rebalance_every=0):
        self.e, self.dist, self.rank, self.nranks = engine, dist, rank, nranks
        self.set_cuts(cuts)  # also switches the engine to rank-local keys
        self.cap = int(cap_records)
        big = max(engine.record_bytes(REC_MIGRANT), engine.record_bytes(REC_GHOST))
        self.send = [engine.alloc(self.cap * big) for _ in range(2)]
        self.recv = [engine.alloc(self.cap * big) for _ in range(2)]
        self.sent = [0, 0]
        self.got = [0, 0]
        self.stats = dict(migrated=0, ghosts=0, exchanges=0, recuts=0)
        self.rebalance_every = int(rebalance_every)
        self.frame = 0

    def set_cuts(self, cuts):
        assert len(cuts) == self.nranks + 1
        self.cuts = list(cuts)
        self.left = self.rank - 1 if self.rank > 0 else None
        self.right = self.rank + 1 if self.rank + 1 < self.nranks else None
        self.cut = (self.cuts[self.rank], self.cuts[self.rank + 1], self.left is not None, self.right is not None)
        if self.nranks > 1:
            self.e.configure(self.cut, self.cuts[self.rank - 1] if self.left is not None else 0,
                             self.cuts[self.rank + 1] if self.right is not None else 0)

    # -- neighbour exchange (host tensors over gloo) ------------------------------------------------
    def _exchange(self, send_bytes, recv_bytes):
        """One batched point-to-point round: send_bytes/recv_bytes = (left, right) byte counts."""
        if self.dist is None or self.nranks == 1:
            return
        t = self.e.torch
        ops, back = [], []
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is None:
                continue
            if recv_bytes[side]:
                host = t.empty(recv_bytes[side], dtype=t.uint8)
                back.append((self.recv[side][:recv_bytes[side]], host))
                ops.append(self.dist.P2POp(self.dist.irecv, host, peer))
            if send_bytes[side]:
                self.e.sync()
                ops.append(self.dist.P2POp(self.dist.isend, self.send[side][:send_bytes[side]].cpu(), peer))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
            for dst, host in back:
                dst.copy_(host)
            self.stats["exchanges"] += 1

    def _exchange_counts(self, n_l, n_r):
        if self.dist is None or self.nranks == 1:
            return 0, 0
        t = self.e.torch
        ops, got = [], {}
        for side, peer, n in ((0, self.left, n_l), (1, self.right, n_r)):
            if peer is None:
                continue
            got[side] = t.zeros(1, dtype=t.int64)
            ops.append(self.dist.P2POp(self.dist.irecv, got[side], peer))
            ops.append(self.dist.P2POp(self.dist.isend, t.tensor([n], dtype=t.int64), peer))
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()
        return int(got[0][0]) if 0 in got else 0, int(got[1][0]) if 1 in got else 0

    def _swap(self, kind, n_l, n_r):
        """Tell the neighbours how many records come, then move them. Returns (from_left, from_right).
        (pbf_slab_step carries the counts in the message header instead: no count round trip.)"""
        g_l, g_r = self._exchange_counts(n_l, n_r)
        if max(g_l, g_r) > self.cap:
            raise RuntimeError(f"slab wire buffer too small: {max(g_l, g_r)} records > cap {self.cap}")
        b = self.e.record_bytes(kind)
        self._exchange((n_l * b, n_r * b), (g_l * b, g_r * b))
        return g_l, g_r

    # -- load balance -----------------------------------------------------------------------------
    def rebalance(self):
        """Re-cut from the global column histogram (one small all-reduce + a host sync; every `rebalance_every`
        steps).  The particles of a transferred column migrate with the next step's ordinary migration round."""
        if self.dist is None or self.nranks == 1:
            return False
        t = self.e.torch
        h = t.from_numpy(self.e.column_histogram())
        self.dist.all_reduce(h)
        new = recut(self.cuts, h.cpu().numpy())
        if new != self.cuts:
            self.set_cuts(new)
            self.stats["recuts"] += 1
            return True
        return False

    # -- one step ---------------------------------------------------------------------------------
    def step(self, p):
        e = self.e
        if self.rebalance_every and self.frame and self.frame % self.rebalance_every == 0:
            self.rebalance()
        self.frame += 1
        e.predict(p)
        n_l, n_r = e.migrate(self.cut, self.send[0], self.send[1], self.cap)
        g_l, g_r = self._swap(REC_MIGRANT, n_l, n_r)
        e.add_migrants(self.recv[0], g_l, self.recv[1], g_r)
        self.stats["migrated"] += n_l + n_r
        n_l, n_r = e.ghosts(self.cut, self.send[0], self.send[1], self.cap)
        self.sent = [n_l, n_r]
        g_l, g_r = self._swap(REC_GHOST, n_l, n_r)
        self.got = [g_l, g_r]
        e.add_ghosts(self.recv[0], g_l, self.recv[1], g_r)
        self.stats["ghosts"] = g_l + g_r
        e.stage("sort", p)
        e.stage("diffuse", p)
        fb = e.record_bytes(REC_FIELD)
        for _ in range(int(p.iteration)):
            for name in ("lambda", "delta"):
                e.stage(name, p)
                e.pack(self.send[0], self.send[1])
                self._exchange((self.sent[0] * fb, self.sent[1] * fb), (self.got[0] * fb, self.got[1] * fb))
                e.unpack(self.recv[0], self.recv[1])
        e.stage("finalise", p)  # (engines run the core finalise here: the extras follow, with their refreshes)
        if p.xsph or p.vorticity:
            vb = 3 * (8 if getattr(e, "fdt", np.float32) == np.float64 else 4)

            def refresh(which):  # owners -> copies: velocity (0) / vorticity (1)
                e.pack_vec(which, self.send[0], self.send[1])
                self._exchange((self.sent[0] * vb, self.sent[1] * vb), (self.got[0] * vb, self.got[1] * vb))
                e.unpack_vec(which, self.recv[0], self.recv[1])

            refresh(0)
            if p.vorticity:
                e.extras_stage("vorticity", p)
                refresh(1)
                e.extras_stage("vorticity_force", p)
                if p.xsph:
                    refresh(0)
            if p.xsph:
                e.extras_stage("xsph", p)
        e.finish()

    def steps(self, p, count):
        for _ in range(count):
            self.step(p)
