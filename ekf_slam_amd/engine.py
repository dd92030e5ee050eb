"""Thin object wrapper over the C ABI handle: one Engine == one ekf_handle == one filter state in HBM."""
import ctypes

import numpy as np

from . import _lib as L

_dp = ctypes.POINTER(ctypes.c_double)


def _vec(v, n=None):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a


def _p(a):
    return a.ctypes.data_as(_dp)


def _colmajor(M):
    return np.asfortranarray(np.asarray(M, dtype=np.float64))


class Engine:
    def __init__(self, mode="known", capacity=1024, tile=0, storage="f64", device=0, rank=0, world=1, batch=1,
                 async_flush=False, device_assoc=None, **overrides):
        self.lib = L.lib()
        cfg = L.EkfConfig()
        m = L.EKF_MODE_KNOWN if mode in ("known", "EKF_SLAM") else L.EKF_MODE_UC
        self._check(self.lib.ekf_config_default(ctypes.byref(cfg), m), None)
        cfg.capacity_landmarks = int(capacity)
        cfg.tile = int(tile)
        if storage not in ("f64", "f32", "f32_mixed", "f32_split"):
            raise TypeError("Engine: storage is 'f64', 'f32' (float tiles, F64 arithmetic), 'f32_mixed' (float tiles, the pass over P "
                            "in F32 arithmetic on the matrix pipe: cfg.pass_arith = EKF_ARITH_F32) or 'f32_split' (the same with every "
                            "float operand cut into three bfloat16 pieces, on the bf16 matrix pipe: EKF_ARITH_SPLIT3)")
        cfg.storage = L.EKF_STORE_F64 if storage == "f64" else L.EKF_STORE_F32
        cfg.pass_arith = {"f32_mixed": L.EKF_ARITH_F32, "f32_split": L.EKF_ARITH_SPLIT3}.get(storage, L.EKF_ARITH_F64)
        cfg.device, cfg.rank, cfg.world, cfg.batch = int(device), int(rank), int(world), int(batch)
        cfg.async_flush = 1 if async_flush else 0
        if device_assoc is not None:              # None: the mode's default (uc: 3, the device-resident loop); include/ekfslam.h
            cfg.device_assoc = int(device_assoc)  # 0: host mirror, 1: device, waited for, 2: device, verified before measure() returns
        fields = {f[0] for f in L.EkfConfig._fields_}
        for k, v in overrides.items():
            if k not in fields:                   # (setattr on a ctypes struct would silently create a Python attribute)
                raise TypeError("Engine: unknown ekf_config field %r" % k)
            if k == "Rc":
                cfg.Rc[0], cfg.Rc[1] = float(v[0]), float(v[1])
            else:
                setattr(cfg, k, v)
        self.cfg = cfg
        self._host_exchange = None
        self._hints = False
        self._raw = None
        self.h = ctypes.c_void_p()
        rc = self.lib.ekf_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != L.EKF_OK:
            msg = self.lib.ekf_last_error(self.h).decode() if self.h else ""
            if self.h:
                self.lib.ekf_destroy(self.h)
                self.h = None
            raise L.EkfError(rc, self.lib.ekf_status_string(rc).decode() + (": " + msg if msg else ""))

    def close(self):
        if getattr(self, "h", None):
            self.lib.ekf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, h="self"):
        if rc != L.EKF_OK:
            msg = self.lib.ekf_last_error(self.h).decode() if (h == "self" and self.h) else ""
            raise L.EkfError(rc, self.lib.ekf_status_string(rc).decode() + (": " + msg if msg else ""))

    # ---- hot path ----
    def predict(self, u):
        self._check(self.lib.ekf_predict(self.h, _p(_vec(u, 2))))

    # ---- pre-marshalled inputs (hosts that stream many steps: bench.py) ----
    # The plain methods convert their arguments on every call (numpy + ctypes: ~10 us per predict + correct in CPython, more
    # than a shard of an 8-GPU run spends on the GPU per update-step).  `marshal_steps` lays a whole run out once in three
    # contiguous arrays; `step_raw(i)` then issues predict + correct of step i with integer addresses only.
    def marshal_steps(self, steps):
        """steps: iterable of (u[2], z[2], R[2x2], idx0).  Returns an opaque run object for step_raw / prefetch use."""
        steps = list(steps)
        m = len(steps)
        U = np.empty((m, 2)); Z = np.empty((m, 2)); Rm = np.empty((m, 4)); K = np.empty(m, dtype=np.int64)
        for i, (u, z, R, k) in enumerate(steps):
            U[i] = np.asarray(u, dtype=np.float64).reshape(-1)[:2]
            Z[i] = np.asarray(z, dtype=np.float64).reshape(-1)[:2]
            Rm[i] = np.asarray(R, dtype=np.float64).reshape(2, 2).reshape(-1, order="F")       # column-major, as the ABI takes it
            K[i] = int(k)
        if self._raw is None:
            proto_p = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)
            proto_c = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
            proto_h = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64)
            self._raw = (proto_p(("ekf_predict", self.lib)), proto_c(("ekf_correct", self.lib)),
                         proto_c(("ekf_correct_begin", self.lib)), proto_h(("ekf_hint_next", self.lib)))
        return {"U": U, "Z": Z, "R": Rm, "K": K, "k": K.tolist(), "u": U.ctypes.data, "z": Z.ctypes.data, "r": Rm.ctypes.data, "m": m}

    def step_raw(self, run, i):
        """predict(u_i) + correct(z_i, R_i, idx_i) of a marshalled run."""
        f_pred, f_corr, f_begin, f_hint = self._raw
        rc = f_pred(self.h, run["u"] + 16 * i)
        if rc:
            self._check(rc)
        if self._hints and i + 1 < run["m"]:     # sharded, library-owned communicator, batch 1: announce the next landmark
            f_hint(self.h, run["k"][i + 1])
        if self._host_exchange is not None:
            rc = f_begin(self.h, run["z"] + 16 * i, run["r"] + 32 * i, run["k"][i])
            if rc:
                self._check(rc)
            self._host_exchange(self)
            self.correct_finish()
            return
        rc = f_corr(self.h, run["z"] + 16 * i, run["r"] + 32 * i, run["k"][i])
        if rc:
            self._check(rc)

    def append(self, u, R, pos, signature):
        self._check(self.lib.ekf_append(self.h, _p(_vec(u, 2)), _p(_colmajor(R).reshape(-1, order="F")),
                                        _p(_vec(pos, 2)), float(signature)))

    def correct(self, z, R, idx0):
        if self._host_exchange is not None:      # sharded, all-gather done by the host (torch.distributed)
            self.correct_begin(z, R, idx0)
            self._host_exchange(self)
            self.correct_finish()
            return
        self._check(self.lib.ekf_correct(self.h, _p(_vec(z[:2], 2)), _p(_colmajor(R).reshape(-1, order="F")), int(idx0)))

    # ---- sharded correction, split so that the caller can run the all-gather (include/ekfslam.h, multi-GPU) ----
    def correct_begin(self, z, R, idx0):
        self._check(self.lib.ekf_correct_begin(self.h, _p(_vec(z[:2], 2)), _p(_colmajor(R).reshape(-1, order="F")),
                                               int(idx0)))

    def correct_finish(self):
        self._check(self.lib.ekf_correct_finish(self.h))

    def prefetch_rows(self, idx0_list):
        """One exchange for the base row-panels of the landmarks the next corrections touch (sharded handles)."""
        arr = (ctypes.c_int64 * len(idx0_list))(*[int(i) for i in idx0_list])
        if self._host_exchange is not None:
            self._check(self.lib.ekf_prefetch_begin(self.h, arr, len(idx0_list)))
            self._host_exchange(self)
            self._check(self.lib.ekf_prefetch_finish(self.h))
            return
        self._check(self.lib.ekf_prefetch_rows(self.h, arr, len(idx0_list)))

    def prefetch_next(self, idx0_list):
        """Announce the landmarks of the batch AFTER the current one (ekf_prefetch_next): extracted in front of the current batch's
        pass as that pass will leave them, exchanged beside it.  Library-owned communicator or exchange hook only."""
        if self._host_exchange is not None:
            raise L.EkfError(L.EKF_ERR_STATE, "prefetch_next: the host-run all-gather cannot run inside the library's flush")
        arr = (ctypes.c_int64 * len(idx0_list))(*[int(i) for i in idx0_list])
        self._check(self.lib.ekf_prefetch_next(self.h, arr, len(idx0_list)))

    def prefetch_begin(self, idx0_list):
        arr = (ctypes.c_int64 * len(idx0_list))(*[int(i) for i in idx0_list])
        self._check(self.lib.ekf_prefetch_begin(self.h, arr, len(idx0_list)))

    def prefetch_finish(self):
        self._check(self.lib.ekf_prefetch_finish(self.h))

    def exchange_info(self):
        send, recv = ctypes.c_void_p(), ctypes.c_void_p()
        cnt, cap = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.ekf_exchange_info(self.h, ctypes.byref(send), ctypes.byref(recv), ctypes.byref(cnt),
                                               ctypes.byref(cap)))
        return send.value, recv.value, int(cnt.value), int(cap.value)

    def exchange_set_buffers(self, send_ptr, recv_ptr):
        self._check(self.lib.ekf_exchange_set_buffers(self.h, ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr)))

    def comm_init(self, comm_id_bytes):
        assert len(comm_id_bytes) == L.EKF_COMM_ID_BYTES
        self._check(self.lib.ekf_comm_init(self.h, bytes(comm_id_bytes)))
        self._hints = self.cfg.batch <= 1 and not self.cfg.async_flush      # step_raw announces the next landmark (ekf_hint_next)

    def hint_next(self, idx0):
        self._check(self.lib.ekf_hint_next(self.h, int(idx0)))

    def associate(self, z, R, want_costs=False):
        if self._host_exchange is not None and (want_costs or self.cfg.w_pos != 0.0):
            # sharded, all-gather done by the host: the position cost needs the other shards' diagonal blocks
            self.associate_begin(z, R, want_costs)
            self._host_exchange(self)
            return self.associate_finish(want_costs)
        is_new, idx = ctypes.c_int32(), ctypes.c_int64()
        N = self.N
        pc = np.zeros(max(N, 1)) if want_costs else None
        sc = np.zeros(max(N, 1)) if want_costs else None
        self._check(self.lib.ekf_associate(self.h, _p(_vec(z, 3)), _p(_colmajor(R).reshape(-1, order="F")),
                                           ctypes.byref(is_new), ctypes.byref(idx),
                                           _p(pc) if want_costs else None, _p(sc) if want_costs else None))
        if want_costs:
            return bool(is_new.value), int(idx.value), pc[:N], sc[:N]
        return bool(is_new.value), int(idx.value)

    # ---- sharded association with position costs, split around the caller's all-gather (include/ekfslam.h) ----
    def associate_begin(self, z, R, want_costs=False):
        self._check(self.lib.ekf_associate_begin(self.h, _p(_vec(z, 3)), _p(_colmajor(R).reshape(-1, order="F")),
                                                 1 if want_costs else 0))

    def associate_finish(self, want_costs=False):
        is_new, idx = ctypes.c_int32(), ctypes.c_int64()
        N = self.N
        pc = np.zeros(max(N, 1)) if want_costs else None
        sc = np.zeros(max(N, 1)) if want_costs else None
        self._check(self.lib.ekf_associate_finish(self.h, ctypes.byref(is_new), ctypes.byref(idx),
                                                  _p(pc) if want_costs else None, _p(sc) if want_costs else None))
        if want_costs:
            return bool(is_new.value), int(idx.value), pc[:N], sc[:N]
        return bool(is_new.value), int(idx.value)

    def measure(self, observed_LL, u, lm_index, lm_loc):
        obs = np.asfortranarray(np.asarray(observed_LL, dtype=np.float64).reshape(-1, 3))
        idx = _vec(lm_index)
        loc = np.asfortranarray(np.asarray(lm_loc, dtype=np.float64).reshape(-1, 2))
        self._check(self.lib.ekf_measure(self.h, _p(obs.reshape(-1, order="F")), obs.shape[0], _p(_vec(u, 2)),
                                         _p(idx), _p(loc.reshape(-1, order="F")), idx.size))

    def set_params(self, C=None, Rc=None, s_cost=None, s_thresh=None, w_pos=None):
        c = self.cfg
        if C is not None: c.C = float(C)
        if Rc is not None: c.Rc[0], c.Rc[1] = float(Rc[0]), float(Rc[1])
        if s_cost is not None: c.s_cost = float(s_cost)
        if s_thresh is not None: c.s_thresh = float(s_thresh)
        if w_pos is not None: c.w_pos = float(w_pos)
        rc2 = (ctypes.c_double * 2)(c.Rc[0], c.Rc[1])
        self._check(self.lib.ekf_set_params(self.h, c.C, rc2, c.s_cost, c.s_thresh, c.w_pos))

    def flush(self):
        self._check(self.lib.ekf_flush(self.h))

    def pending(self):
        n = ctypes.c_int32()
        self._check(self.lib.ekf_pending(self.h, ctypes.byref(n)))
        return int(n.value)

    def sync(self):
        self._check(self.lib.ekf_sync(self.h))

    def set_stream(self, stream_ptr):
        self._check(self.lib.ekf_set_stream(self.h, ctypes.c_void_p(stream_ptr)))

    # ---- state ----
    @property
    def N(self):
        n = ctypes.c_int64()
        self._check(self.lib.ekf_num_landmarks(self.h, ctypes.byref(n)))
        return int(n.value)

    @property
    def n(self):
        return 3 + 2 * self.N

    def get_x(self):
        x = np.empty(self.n)
        self._check(self.lib.ekf_get_x(self.h, _p(x)))
        return x

    def get_s(self):
        s = np.empty(max(self.N, 1))
        self._check(self.lib.ekf_get_s(self.h, _p(s)))
        return s[:self.N]

    def get_P(self):
        n = self.n
        buf = np.empty(n * n)
        self._check(self.lib.ekf_get_P(self.h, _p(buf)))
        return buf.reshape(n, n, order="F")

    def get_P_block(self, r0, c0, nr, nc):
        buf = np.empty(nr * nc)
        self._check(self.lib.ekf_get_P_block(self.h, r0, c0, nr, nc, _p(buf)))
        return buf.reshape(nr, nc, order="F")

    def get_P_diag_blocks(self):
        """(N+1) x 2 x 2: P(1:2,1:2) and every landmark's diagonal block -- what plot() reads, one call."""
        nb = self.N + 1
        buf = np.empty(4 * nb)
        self._check(self.lib.ekf_get_P_diag_blocks(self.h, _p(buf)))
        return buf.reshape(nb, 2, 2).transpose(0, 2, 1).copy()      # each block arrives column-major

    def get_Q3(self):
        q = np.empty(9)
        self._check(self.lib.ekf_get_Q(self.h, _p(q)))
        return q.reshape(3, 3, order="F")

    def set_state(self, x, P, s):
        x = _vec(x)
        self._check(self.lib.ekf_set_x(self.h, _p(x), x.size))
        s = _vec(s)
        self._check(self.lib.ekf_set_s(self.h, _p(s) if s.size else None, s.size))
        Pf = np.asfortranarray(np.asarray(P, dtype=np.float64))
        self._check(self.lib.ekf_set_P(self.h, _p(Pf.reshape(-1, order="F")), Pf.shape[0]))

    def load_lowrank_state(self, x, s, d, U):
        x, s, d = _vec(x), _vec(s), _vec(d)
        U = np.asfortranarray(np.asarray(U, dtype=np.float64))
        N = (x.size - 3) // 2
        self._check(self.lib.ekf_load_lowrank_state(self.h, N, _p(x), _p(s) if s.size else _p(np.zeros(1)), _p(d),
                                                    _p(U.reshape(-1, order="F")), U.shape[1]))

    def checkpoint_save(self, path):
        self._check(self.lib.ekf_checkpoint_save(self.h, str(path).encode()))

    def checkpoint_load(self, path):
        self._check(self.lib.ekf_checkpoint_load(self.h, str(path).encode()))

    def digest(self):
        out = np.empty(3)
        self._check(self.lib.ekf_P_digest(self.h, _p(out)))
        return out

    def device_bytes(self):
        b = ctypes.c_int64()
        self._check(self.lib.ekf_device_bytes(self.h, ctypes.byref(b)))
        return int(b.value)

    # ---- measurement hooks ----
    def timing_enable(self, which, on=True, launches=0):
        """launches: event pairs to reserve up front (launches expected between two timing_read calls)."""
        self._check(self.lib.ekf_kernel_timing_enable(self.h, which, max(1, int(launches)) if on else 0))

    def downdate_kernel_name(self):
        """(name, pairs) of the kernel instance the last downdate / flush launch used, as the launcher chose it."""
        pairs = ctypes.c_int32()
        name = self.lib.ekf_downdate_kernel_name(self.h, ctypes.byref(pairs))
        return (name or b"").decode(), int(pairs.value)

    def timing_read(self, which):
        n, ms = ctypes.c_int64(), ctypes.c_double()
        self._check(self.lib.ekf_kernel_timing_read(self.h, which, ctypes.byref(n), ctypes.byref(ms)))
        return int(n.value), float(ms.value)

    def downdate_algorithmic_bytes(self):
        b = ctypes.c_int64()
        self._check(self.lib.ekf_downdate_algorithmic_bytes(self.h, ctypes.byref(b)))
        return int(b.value)
