"""Multi-GPU plumbing around the sharded C ABI (include/ekfslam.h, section "multi-GPU").

P is split over `world` shards: tile (I,J) of the landmark block lives on shard (I + J) mod world; x, s, the
robot block and the robot/landmark strip are replicated.  The only data-path exchange is ONE all-gather per
update-step of the 2 x 2N landmark row-panel (chunk k of T columns comes from shard (tile_row(j)+k) mod world).

  * ``attach_communicator``  one process per GPU (torchrun): gives the handle a native RCCL communicator
                             (ncclUniqueId from rank 0, broadcast through torch.distributed) -- or raises on every
                             rank; a host-run all-gather through torch.distributed only on explicit request.
  * ``ShardGroup``           every shard of one filter driven from ONE host thread in one process (what a
                             MATLAB host does; also how the sharded path is tested on a single GPU).
  * ``owner`` / ``panel_source`` host-only views of the shard plan.
"""
import ctypes

import numpy as np

from . import _lib as L
from .engine import Engine, _colmajor, _p, _vec


def owner(world, I, J):
    return int(L.lib().ekf_shard_owner(world, I, J))


def slot(world, I, J):
    return int(L.lib().ekf_shard_slot(world, I, J))


def panel_source(world, tile_row_j, chunk):
    o, k = ctypes.c_int32(), ctypes.c_int64()
    rc = L.lib().ekf_shard_panel_source(world, tile_row_j, chunk, ctypes.byref(o), ctypes.byref(k))
    if rc:
        raise L.EkfError(rc, "ekf_shard_panel_source")
    return int(o.value), int(k.value)


def attach_communicator(engine, dist, torch, transport="rccl"):
    """Give a sharded Engine its exchange.  Collective over all ranks.  Returns the transport in use.

    transport="rccl": the library's own RCCL communicator (ekf_comm_init; ncclUniqueId from rank 0, broadcast through
    torch.distributed).  If it cannot be attached on ANY rank this raises on EVERY rank -- there is no second transport
    behind it.  Everything that can fail on one rank alone (loading librccl, creating the id) is agreed on with an
    all_reduce BEFORE the collective ncclCommInitRank, so no rank is left waiting inside it.
    transport="torch": the all-gather is run by the host through torch.distributed between ekf_correct_begin / _finish
    (transport (b) of include/ekfslam.h): an explicit choice for rehearsals (gloo on one GPU), never a fallback."""
    world, rank = engine.cfg.world, engine.cfg.rank
    if transport not in ("rccl", "torch"):
        raise ValueError("transport must be 'rccl' or 'torch'")
    on_gpu = dist.get_backend() == "nccl"
    dev = "cuda" if on_gpu else "cpu"
    if transport == "rccl":
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        buf = torch.zeros(L.EKF_COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        why = ""
        if rank == 0:
            raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
            rc = L.lib().ekf_comm_unique_id(raw)
            if rc:
                ok.zero_()
                why = "ekf_comm_unique_id failed (status %d): librccl not loadable?" % rc
            else:
                buf.copy_(torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)                # agree BEFORE the collective init
        if int(ok.item()) != 1:
            raise L.EkfError(L.EKF_ERR_COMM, "native RCCL communicator unavailable on some rank" + (": " + why if why else ""))
        dist.broadcast(buf, src=0)
        err = ""
        try:
            engine.comm_init(bytes(buf.cpu().numpy().tobytes()))    # collective: ncclCommInitRank
        except L.EkfError as ex:
            ok.zero_()
            err = str(ex)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            raise L.EkfError(L.EKF_ERR_COMM, "ekf_comm_init failed on some rank" + (": " + err if err else ""))
        return "rccl-native"
    # host-run exchange over torch.distributed: the handle launches on a torch stream so that the collective is ordered
    # between extract and solve
    _, _, _, cap = engine.exchange_info()
    send = torch.zeros(cap, dtype=torch.float64, device="cuda")
    recv = torch.zeros(cap * world, dtype=torch.float64, device="cuda")
    engine.exchange_set_buffers(send.data_ptr(), recv.data_ptr())
    # The engine must launch on the stream the collective is ordered with.  torch's default stream is the NULL stream,
    # which ekf_set_stream reads as "use your own stream" -- so run everything on a dedicated torch stream.
    xstream = torch.cuda.Stream()
    engine.set_stream(xstream.cuda_stream)
    engine._xchg_tensors = (send, recv, xstream)
    staged = not on_gpu                        # e.g. gloo (rehearsals on one GPU): stage through host memory

    def host_exchange(e):
        _, _, cnt, _ = e.exchange_info()
        with torch.cuda.stream(xstream):
            if staged:
                xstream.synchronize()
                s_cpu = send[:cnt].cpu()
                parts = [torch.empty_like(s_cpu) for _ in range(world)]
                dist.all_gather(parts, s_cpu)
                recv[:cnt * world].copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(recv[:cnt * world], send[:cnt])

    engine._host_exchange = host_exchange
    engine._hints = engine.cfg.batch <= 1 and not engine.cfg.async_flush     # step_raw announces the next landmark (ekf_hint_next)
    return "torch.distributed" if not staged else "torch.distributed(%s, host-staged)" % dist.get_backend()


class ShardGroup:
    """All `world` shards of ONE filter, driven by one host thread (devices[r] is the HIP device of shard r)."""

    def __init__(self, world, devices=None, **engine_kw):
        devices = devices if devices is not None else [0] * world
        self.world = world
        self.shards = [Engine(rank=r, world=world, device=devices[r], **engine_kw) for r in range(world)]
        self._harr = (ctypes.c_void_p * world)(*[e.h for e in self.shards])
        self.lib = self.shards[0].lib

    def close(self):
        for e in self.shards:
            e.close()

    @property
    def N(self):
        return self.shards[0].N

    def predict(self, u):
        for e in self.shards:
            e.predict(u)

    def append(self, u, R, pos, signature):
        for e in self.shards:
            e.append(u, R, pos, signature)

    def correct(self, z, R, idx0):
        for e in self.shards:
            e.correct_begin(z, R, idx0)
        rc = self.lib.ekf_exchange_local(self._harr, self.world)
        if rc:
            raise L.EkfError(rc, self.lib.ekf_last_error(self.shards[0].h).decode())
        for e in self.shards:
            e.correct_finish()

    def hint_next(self, idx0):
        """The landmark the NEXT correct() names (ekf_hint_next): the current correction's pass extracts its row-panel."""
        for e in self.shards:
            e.hint_next(idx0)

    def prefetch_rows(self, idx0_list):
        for e in self.shards:
            e.prefetch_begin(idx0_list)
        rc = self.lib.ekf_exchange_local(self._harr, self.world)
        if rc:
            raise L.EkfError(rc, self.lib.ekf_last_error(self.shards[0].h).decode())
        for e in self.shards:
            e.prefetch_finish()

    def correct_local(self, z, R, idx0):
        """A correction on a prefetched landmark: no exchange, each shard proceeds on its own."""
        for e in self.shards:
            self._chk(e, self.lib.ekf_correct(e.h, _p(_vec(z[:2], 2)), _p(_colmajor(R).reshape(-1, order="F")), int(idx0)))

    @staticmethod
    def _chk(e, rc):
        if rc:
            raise L.EkfError(rc, e.lib.ekf_last_error(e.h).decode())

    def associate(self, z, R, want_costs=False):
        """estimateCorrespondence on the group.  Signature-only likelihood and no costs asked for: every shard decides alone
        from replicated data.  Otherwise (w_pos != 0, or the cost vectors): each shard scores the landmarks whose diagonal
        block it holds, one exchange of the candidates (+ position costs), the same arg-min on every shard."""
        if want_costs or self.shards[0].cfg.w_pos != 0.0:
            for e in self.shards:
                e.associate_begin(z, R, want_costs)
            rc = self.lib.ekf_exchange_local(self._harr, self.world)
            if rc:
                raise L.EkfError(rc, self.lib.ekf_last_error(self.shards[0].h).decode())
            res = [e.associate_finish(want_costs) for e in self.shards]
        else:
            res = [e.associate(z, R) for e in self.shards]
        for r in res[1:]:
            assert r[:2] == res[0][:2], "shards disagree on the association"
            for a, b in zip(r[2:], res[0][2:]):
                np.testing.assert_array_equal(a, b)
        return res[0]

    def run_threaded(self, fn):
        """fn(shard) on every shard, ONE HOST THREAD PER SHARD, with the exchange hook of transport (d) (include/ekfslam.h) set:
        wherever a library call needs an all-gather -- ekf_correct, ekf_prefetch_rows, the middle of ekf_measure's loop, a batch's
        pass with an announced prefetch (ekf_prefetch_next) -- every thread arrives at a barrier with its contribution queued, thread
        0 runs ekf_exchange_local over all handles, a second barrier releases them.  (The library calls release the GIL; the hook
        re-enters Python only for the two barrier waits.)"""
        import threading
        world = self.world
        bar = threading.Barrier(world)
        xrc = [0]
        errs = [None] * world

        def make_hook(r):
            def hook(_ctx):
                try:
                    bar.wait()
                    if r == 0:
                        xrc[0] = self.lib.ekf_exchange_local(self._harr, world)
                    bar.wait()
                    return int(xrc[0])
                except threading.BrokenBarrierError:
                    return 1
            return L.EXCHANGE_HOOK(hook)

        hooks = [make_hook(r) for r in range(world)]
        for e, hk in zip(self.shards, hooks):
            self._chk(e, self.lib.ekf_exchange_set_hook(e.h, ctypes.cast(hk, ctypes.c_void_p), None))

        def run(r):
            try:
                fn(self.shards[r])
            except BaseException as ex:  # noqa: BLE001 -- reported by the caller's thread below
                errs[r] = ex
                bar.abort()                                           # the other shards' hooks return an error instead of waiting forever

        try:
            ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        finally:
            for e in self.shards:
                self.lib.ekf_exchange_set_hook(e.h, None, None)
        for ex in errs:
            if ex is not None:
                raise ex

    def measure(self, observed_LL, u, lm_index, lm_loc):
        """EKF_SLAM.measure / EKF_SLAM_UC.measure on the group: ekf_measure on every shard (run_threaded)."""
        self.run_threaded(lambda e: e.measure(observed_LL, u, lm_index, lm_loc))

    def set_params(self, **kw):
        for e in self.shards:
            e.set_params(**kw)

    def set_state(self, x, P, s):
        for e in self.shards:
            e.set_state(x, P, s)

    def load_lowrank_state(self, x, s, d, U):
        for e in self.shards:
            e.load_lowrank_state(x, s, d, U)

    def get_x(self):
        xs = [e.get_x() for e in self.shards]
        for x in xs[1:]:
            np.testing.assert_array_equal(x, xs[0])      # replicated state must be bit-identical
        return xs[0]

    def get_P(self):
        """Merge the shards' views: each returns NaN for landmark-block entries it does not hold."""
        P = self.shards[0].get_P()
        for e in self.shards[1:]:
            Q = e.get_P()
            hole = np.isnan(P)
            P[hole] = Q[hole]
        return P

    def flush(self):
        for e in self.shards:
            e.flush()

    def digest(self):
        return sum(e.digest() for e in self.shards)
