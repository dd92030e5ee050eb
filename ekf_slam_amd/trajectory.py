"""On-disk trajectory log: everything the EKF hot path consumed, step by step.

The reference has no recorded data of any kind (its inputs are live ROS topics).  A log holds, per SLAM iteration,
the odometry `u` handed to predict() (SLAM.m:105-110) and what measure() saw after the landmark front-end ran
(EKF_SLAM.m:102,111,120): `observed_LL` (m x 3) and the landmark table (index, loc).  Replaying a log into an
engine reproduces the run bit for bit on the same hardware, and is how a run on one machine is compared with
another (or with the CPU oracle).  Format: one .npz with ragged arrays (`*_ptr` are CSR-style offsets).
"""
import numpy as np

FORMAT = "ekfslam-trajectory-1"


class TrajectoryLog:
    def __init__(self):
        self.u, self.obs, self.lm_index, self.lm_loc = [], [], [], []

    def __len__(self):
        return len(self.u)

    def record(self, u, observed_LL, lm_index, lm_loc):
        self.u.append(np.asarray(u, dtype=np.float64).reshape(2).copy())
        obs = np.zeros((0, 3)) if observed_LL is None or len(observed_LL) == 0 else np.asarray(observed_LL, dtype=np.float64)
        self.obs.append(obs.reshape(-1, 3).copy())
        self.lm_index.append(np.asarray(lm_index, dtype=np.float64).reshape(-1).copy())
        self.lm_loc.append(np.asarray(lm_loc, dtype=np.float64).reshape(-1, 2).copy())

    def save(self, path):
        def ragged(parts, width):
            ptr = np.cumsum([0] + [len(p) for p in parts])
            data = np.concatenate(parts) if parts and ptr[-1] else np.zeros((0, width) if width else (0,))
            return ptr, data
        obs_ptr, obs = ragged(self.obs, 3)
        lm_ptr, lmi = ragged(self.lm_index, 0)
        _, lml = ragged(self.lm_loc, 2)
        np.savez_compressed(path, format=np.array(FORMAT), u=np.array(self.u).reshape(-1, 2), obs_ptr=obs_ptr, obs=obs,
                            lm_ptr=lm_ptr, lm_index=lmi, lm_loc=lml)

    @staticmethod
    def load(path):
        g = np.load(path, allow_pickle=False)
        if str(g["format"]) != FORMAT:
            raise ValueError("not an %s file" % FORMAT)
        t = TrajectoryLog()
        for k in range(len(g["u"])):
            a, b = g["obs_ptr"][k], g["obs_ptr"][k + 1]
            c, d = g["lm_ptr"][k], g["lm_ptr"][k + 1]
            t.record(g["u"][k], g["obs"][a:b], g["lm_index"][c:d], g["lm_loc"][c:d])
        return t

    def replay(self, engine, start=0, stop=None):
        """predict + measure for steps [start, stop) on anything with predict(u) / measure(obs, u, idx, loc)."""
        stop = len(self) if stop is None else stop
        for k in range(start, stop):
            engine.predict(self.u[k])
            if len(self.obs[k]):
                engine.measure(self.obs[k], self.u[k], self.lm_index[k], self.lm_loc[k])
