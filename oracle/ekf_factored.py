"""Factored ("implicit") NumPy restatement of the reference's EKF-SLAM hot path -- the oracle that reaches 50 000 landmarks.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED by the reference (no MATLAB / Octave in the image, no golden vectors in
the reference); pinned to oracle/ekf_dense.py (the literal-dense restatement) at N <= 200 to 1e-11, appends and unknown correspondence
included (tests/test_oracle_factored.py).

Why: the dense restatements need the n x n covariance (80 GB at 50 000 landmarks), so BASELINE.json configs[4] (40 000 -> 50 000
landmarks) could only be checked engine against engine.  Here the landmark block of P is never formed:

    P = [ Prr  Prm ]      Prr 3 x 3, Prm 3 x m, Pmr m x 3   explicit (both strips: the reference's P is symmetric only to rounding)
        [ Pmr  Pmm ]      Pmm(i, j) = B(i, j) + sum_t L_t(i) R_t(j)          implicit
                          B   = diag(d) + U U'                               the bulk-loaded block (EKF_SLAM state loaded as D + U U'),
                                extended by every append's row panel / its transpose / its 2 x 2 block   (EKF_SLAM.m:91-97)
                          L_t = -K_m, R_t = (H P)_m of correction t          (EKF_SLAM.m:145, one rank-2 term per correction;
                                vectors are as long as the map was at time t: later landmarks see zeros)

and every read the reference makes of P -- the five rows and five columns S = {1, 2, 3, j, j+1} of a correction (EKF_SLAM.m:141-145), the
robot block and the column strip P(k, 1:3) of an append (:91-96), each landmark's 5 x 5 sub-block of an association
(Correspondence.m:66) -- is evaluated from that form in F64.  Cost per step O(n (k + 2 t)) instead of O(n^3).

One association differs from the dense evaluation: (eye(n) - K H) P is formed as P - K (H P) (as oracle/ekf_structured.c and the HIP
path do); entry by entry the two differ in the last bits of the update.

Follows (file:line in /root/reference): EKF_SLAM.m:26-34 (state), :40-51 + :56-65 (predict, f), :67-98 (append), :100-151 (measure),
EKF_SLAM_UC.m:13,16,102-152, Correspondence.m:28-88.  Angles are degrees everywhere.
"""
import numpy as np

from .ekf_dense import _lookup_loc
from .matlab_compat import atan2d, cosd, inv2, sind, wrapTo360


class FactoredEKF:
    """EKF_SLAM / EKF_SLAM_UC with an implicit landmark block.  x is the 1 x n row vector.  Indices of the public methods are 1-based
    where the reference's are (correct(idx), estimateCorrespondence -> index)."""

    def __init__(self, capacity, mode="known", max_terms=4096, max_appends=None):
        self.cap = int(capacity)
        self.mode = "uc" if mode in ("uc", "EKF_SLAM_UC") else "known"
        self.C = 0.2                                                    # EKF_SLAM.m:12
        self.Rc = [.1, 5] if self.mode == "uc" else [.01, 5]            # EKF_SLAM_UC.m:13 / EKF_SLAM.m:13
        self.s_cost, self.s_thresh, self.w_pos = .00000000001, 1000000000, 0.0       # EKF_SLAM.m:14,16; Correspondence.m:75 (w_pos = 0)
        self.x = np.zeros(3)                                            # EKF_SLAM.m:28
        self.Prr = np.eye(3) * 0.1                                      # :29-31
        M = 2 * self.cap
        self.Prm = np.zeros((3, M))
        self.Pmr = np.zeros((M, 3))
        self.s = []
        self.Q = None
        self.N = 0
        # the bulk-loaded block diag(d) + U U'
        self.m0 = 0
        self.d = np.zeros(0)
        self.U = np.zeros((0, 1))
        # appended landmarks: RP[2a:2a+2, :m_a] = row panel of appended landmark a at its append, Ca[a] its own 2 x 2 block, ma[a] = m_a
        self.amax = int(max_appends if max_appends is not None else self.cap)
        self.RP = None
        self.Ca = []
        self.ma = []
        # corrections: Lm[:, 2t:2t+2] = -K_m, Rm[2t:2t+2, :] = (H P)_m
        self.tmax = int(max_terms)
        self.Lm = None
        self.Rm = None
        self.nt = 0

    # ---- sizes ------------------------------------------------------------------------------------------------------------------
    @property
    def m(self):
        return 2 * self.N

    @property
    def n(self):
        return 3 + 2 * self.N

    def _terms(self):
        if self.Lm is None:
            self.Lm = np.zeros((2 * self.cap, 2 * self.tmax))
            self.Rm = np.zeros((2 * self.tmax, 2 * self.cap))
        return self.Lm, self.Rm

    # ---- bulk load: x, s, P = diag(d) + U U' (the configs[2..4] state) ----------------------------------------------
    def load_lowrank_state(self, x, s, d, U):
        x, d, U = np.asarray(x, float).reshape(-1), np.asarray(d, float).reshape(-1), np.asarray(U, float)
        N = (x.size - 3) // 2
        assert N <= self.cap and d.size == x.size and U.shape[0] == x.size
        self.x = x.copy()
        self.s = list(np.asarray(s, float).reshape(-1))
        self.N = N
        m = 2 * N
        P3 = U[:3] @ U.T                                                # rows 1:3 of U U'
        self.Prr = np.diag(d[:3]) + P3[:, :3]
        self.Prm[:, :m] = P3[:, 3:]
        self.Pmr[:m, :] = P3[:, 3:].T
        self.m0, self.d, self.U = m, d[3:].copy(), U[3:].copy()
        self.Ca, self.ma, self.nt = [], [], 0
        self.RP = None

    def set_state(self, x, P, s):
        """a dense state (small maps): taken as a bulk-loaded block with U = the dense landmark block's factor-free form"""
        x, P = np.asarray(x, float).reshape(-1), np.asarray(P, float)
        N = (x.size - 3) // 2
        self.x, self.s, self.N = x.copy(), list(np.asarray(s, float).reshape(-1)), N
        m = 2 * N
        self.Prr = P[:3, :3].copy()
        self.Prm[:, :m] = P[:3, 3:]
        self.Pmr[:m, :] = P[3:, :3]
        self.m0, self.d, self.U = m, np.zeros(m), np.zeros((m, 1))
        self._dense0 = P[3:, 3:].copy()                                 # B's bulk part given explicitly
        self.Ca, self.ma, self.nt = [], [], 0
        self.RP = None

    # ---- the implicit block --------------------------------------------------------------------------------------------------------
    def _base_rows(self, j):
        """B(j:j+2, :m)"""
        m = self.m
        out = np.zeros((2, m))
        if j < self.m0:
            if getattr(self, "_dense0", None) is not None:
                out[:, :self.m0] = self._dense0[j:j + 2, :]
            else:
                out[:, :self.m0] = self.U[j:j + 2] @ self.U.T
                out[0, j] += self.d[j]
                out[1, j + 1] += self.d[j + 1]
            for a, ma in enumerate(self.ma):                            # columns of the landmarks appended since: their row panels, transposed
                out[:, ma:ma + 2] = self.RP[2 * a:2 * a + 2, j:j + 2].T
        else:
            a = self.ma.index(j)
            out[:, :j] = self.RP[2 * a:2 * a + 2, :j]
            out[:, j:j + 2] = self.Ca[a]
            for b in range(a + 1, len(self.ma)):
                mb = self.ma[b]
                out[:, mb:mb + 2] = self.RP[2 * b:2 * b + 2, j:j + 2].T
        return out

    def _base_cols(self, j):
        """B(:m, j:j+2): B is symmetric by construction (the reference assigns the transposes: EKF_SLAM.m:93,96)"""
        return self._base_rows(j).T

    def Pmm_rows(self, j):
        """P(3 + j : 3 + j + 2, 3:)  (2 x m)"""
        out = self._base_rows(j)
        if self.nt:
            Lm, Rm = self._terms()
            out += Lm[j:j + 2, :2 * self.nt] @ Rm[:2 * self.nt, :self.m]
        return out

    def Pmm_cols(self, j):
        """P(3:, 3 + j : 3 + j + 2)  (m x 2)"""
        out = self._base_cols(j)
        if self.nt:
            Lm, Rm = self._terms()
            out += Lm[:self.m, :2 * self.nt] @ Rm[:2 * self.nt, j:j + 2]
        return out

    def diag_blocks(self):
        """every landmark's own 2 x 2 block, (N, 2, 2)"""
        N = self.N
        out = np.zeros((N, 2, 2))
        if self.m0:
            if getattr(self, "_dense0", None) is not None:
                D0 = self._dense0
                for k in range(self.m0 // 2):
                    out[k] = D0[2 * k:2 * k + 2, 2 * k:2 * k + 2]
            else:
                Ue, Uo = self.U[0::2], self.U[1::2]
                out[:self.m0 // 2, 0, 0] = np.einsum("ik,ik->i", Ue, Ue) + self.d[0::2]
                out[:self.m0 // 2, 1, 1] = np.einsum("ik,ik->i", Uo, Uo) + self.d[1::2]
                out[:self.m0 // 2, 0, 1] = out[:self.m0 // 2, 1, 0] = np.einsum("ik,ik->i", Ue, Uo)
        for a, ma in enumerate(self.ma):
            out[ma // 2] = self.Ca[a]
        if self.nt:
            Lm, Rm = self._terms()
            T = 2 * self.nt
            Le, Lo, Re, Ro = Lm[0:2 * N:2, :T], Lm[1:2 * N:2, :T], Rm[:T, 0:2 * N:2].T, Rm[:T, 1:2 * N:2].T
            out[:, 0, 0] += np.einsum("it,it->i", Le, Re)
            out[:, 0, 1] += np.einsum("it,it->i", Le, Ro)
            out[:, 1, 0] += np.einsum("it,it->i", Lo, Re)
            out[:, 1, 1] += np.einsum("it,it->i", Lo, Ro)
        return out

    def P_rows(self, r0, nr):
        """P(r0 : r0 + nr, :) for rows inside the robot block (r0 + nr <= 3) or one landmark's two rows (r0 = 3 + 2 k, nr = 2); 0-based"""
        m = self.m
        if r0 + nr <= 3:
            return np.hstack([self.Prr[r0:r0 + nr], self.Prm[r0:r0 + nr, :m]])
        j = r0 - 3
        assert nr == 2 and j % 2 == 0
        return np.hstack([self.Pmr[j:j + 2], self.Pmm_rows(j)])

    def dense_P(self):
        """the whole covariance (small maps only: tests)"""
        n, m = self.n, self.m
        P = np.zeros((n, n))
        P[:3, :3], P[:3, 3:], P[3:, :3] = self.Prr, self.Prm[:, :m], self.Pmr[:m]
        for j in range(0, m, 2):
            P[3 + j:5 + j, 3:] = self.Pmm_rows(j)
        return P

    @property
    def P(self):
        return self.dense_P()

    # ---- EKF_SLAM.m:40-51, :56-65 ---------------------------------------------------------------------------------------------------
    def predict(self, u):
        x, m = self.x, self.m
        W = np.array([[u[0] * cosd(x[2])], [u[0] * sind(x[2])], [u[1]]])
        Q3 = (W * self.C) @ W.T                                         # Q(1:3,1:3) = W*C*W'   (:43-44)
        f13, f23 = -1 * u[0] * sind(x[2]), u[0] * cosd(x[2])            # F(1,3), F(2,3) at the PRE-motion heading (:63-64)
        xn = x.copy()
        xn[0] = x[0] + u[0] * cosd(x[2] + u[1])                         # :58-60
        xn[1] = x[1] + u[0] * sind(x[2] + u[1])
        xn[2] = x[2] + u[1]
        # F P: rows 1, 2 += F(r,3) row 3 -- over the robot block and the strip
        Prr, Prm, Pmr = self.Prr.copy(), self.Prm, self.Pmr
        Prr[0] = Prr[0] + f13 * Prr[2]
        Prr[1] = Prr[1] + f23 * Prr[2]
        Prm[0, :m] = Prm[0, :m] + f13 * Prm[2, :m]
        Prm[1, :m] = Prm[1, :m] + f23 * Prm[2, :m]
        # (F P) F': columns 1, 2 += column 3 F(c,3) -- over the robot block and the column strip
        Prr[:, 0] = Prr[:, 0] + Prr[:, 2] * f13
        Prr[:, 1] = Prr[:, 1] + Prr[:, 2] * f23
        Pmr[:m, 0] = Pmr[:m, 0] + Pmr[:m, 2] * f13
        Pmr[:m, 1] = Pmr[:m, 1] + Pmr[:m, 2] * f23
        self.Prr = Prr + Q3                                             # :47
        self.Q = Q3
        xn[2] = wrapTo360(xn[2])                                        # :50
        self.x = xn

    # ---- EKF_SLAM.m:67-98 -------------------------------------------------------------------------------------------------------------
    def append(self, u, R, landmarkPos, signature):
        assert self.N < self.cap
        R = np.asarray(R, float).reshape(2, 2)
        self.s.append(signature)
        m = self.m
        self.x = np.concatenate([self.x, [landmarkPos[0], landmarkPos[1]]])
        x = self.x
        jxr = np.array([[1.0, 0.0, -u[0] * sind(x[2])], [0.0, 1.0, u[0] * cosd(x[2])]])
        jz = np.array([[cosd(u[1]), -u[0] * sind(u[1])], [sind(u[1]), u[0] * cosd(u[1])]])
        C = jxr @ self.Prr @ jxr.T + jz @ R @ jz.T                      # :91  C
        I = self.Prr @ jxr.T                                            # :92  I   P(1:3, new)
        self.Prm[:, m:m + 2] = I
        self.Pmr[m:m + 2, :] = I.T                                      # :93  H
        if self.RP is None:
            self.RP = np.zeros((2 * self.amax, 2 * self.cap))
        a = len(self.ma)
        assert a < self.amax
        self.RP[2 * a:2 * a + 2, :m] = jxr @ self.Pmr[:m].T             # :95  F   jxr * P(k, 1:3)' for every old landmark k (G = F': read transposed)
        self.Ca.append(C)
        self.ma.append(m)
        self.N += 1

    # ---- EKF_SLAM.m:124-145 -----------------------------------------------------------------------------------------------------------
    def _innovation(self, j):
        """z_k and the 2 x 5 block of H_k for the landmark at landmark-block row j (EKF_SLAM.m:125-138)"""
        x = self.x
        d0, d1 = x[3 + j] - x[0], x[4 + j] - x[1]
        q = d0 * d0 + d1 * d1
        sq = np.sqrt(q)
        z_k = np.array([sq, wrapTo360(atan2d(d1, d0) - x[2])])
        Hs = (1 / q) * np.array([[-sq * d0, -sq * d1, 0.0, sq * d0, sq * d1], [d1, -d0, -q, -d1, d0]])
        return z_k, Hs

    def correct(self, z, R, idx):
        """the correction body for the 1-based landmark idx"""
        assert 1 <= idx <= self.N
        R = np.asarray(R, float).reshape(2, 2)
        j, m = 2 * (idx - 1), self.m
        z_k, Hs = self._innovation(j)
        Hr, Hl = Hs[:, :3], Hs[:, 3:]
        rows, cols = self.Pmm_rows(j), self.Pmm_cols(j)                 # P(j:j+1, 4:end), P(4:end, j:j+1)
        # G = H_k P  (2 x n): rows S of P
        Gr = Hr @ self.Prr + Hl @ self.Pmr[j:j + 2]
        Gm = Hr @ self.Prm[:, :m] + Hl @ rows
        # P H_k'  (n x 2): columns S of P
        PHr = self.Prr @ Hr.T + self.Prm[:, j:j + 2] @ Hl.T
        PHm = self.Pmr[:m] @ Hr.T + cols @ Hl.T
        phi = Gr @ Hr.T + Gm[:, j:j + 2] @ Hl.T + R                     # :141
        ip = inv2(phi)
        Kr, Km = PHr @ ip, PHm @ ip                                     # :143
        nu = np.array([z[0], z[1]]) - z_k
        self.x = self.x + np.concatenate([Kr @ nu, Km @ nu])            # :144
        # (eye(n) - K H_k) P = P - K (H_k P)                              :145
        self.Prr = self.Prr - Kr @ Gr
        self.Prm[:, :m] = self.Prm[:, :m] - Kr @ Gm
        self.Pmr[:m] = self.Pmr[:m] - Km @ Gr
        Lm, Rm = self._terms()
        t = self.nt
        assert t < self.tmax, "FactoredEKF: max_terms corrections reached"
        Lm[:m, 2 * t:2 * t + 2] = -Km
        Lm[m:, 2 * t:2 * t + 2] = 0.0
        Rm[2 * t:2 * t + 2, :m] = Gm
        Rm[2 * t:2 * t + 2, m:] = 0.0
        self.nt = t + 1

    # ---- Correspondence.m:28-88 --------------------------------------------------------------------------------------------------------
    def estimateCorrespondence(self, z, R):
        """[newLL, index] (index 1-based); also keeps the position / signature costs of every landmark"""
        N, x = self.N, self.x
        R = np.asarray(R, float).reshape(2, 2)
        D = self.diag_blocks()
        pos, sig = np.zeros(N), np.zeros(N)
        newLL, index, best = True, N + 1, np.inf
        for kk in range(1, N + 1):
            j = 2 * (kk - 1)
            z_k, Hs = self._innovation(j)
            Hr, Hl = Hs[:, :3], Hs[:, 3:]
            Gr = Hr @ self.Prr + Hl @ self.Pmr[j:j + 2]
            Gl = Hr @ self.Prm[:, j:j + 2] + Hl @ D[kk - 1]
            phi = Gr @ Hr.T + Gl @ Hl.T + R                             # :66
            nu = np.array([z[0], z[1]]) - z_k
            pos[kk - 1] = float(nu @ inv2(phi) @ nu)                    # :69 (unused by the reference's decision)
            d = z[2] - self.s[kk - 1]
            sig[kk - 1] = d * (1.0 / self.s_cost) * d                   # :71
            ll = self.w_pos * pos[kk - 1] + sig[kk - 1] if self.w_pos != 0.0 else sig[kk - 1]      # :74-75
            if ll <= self.s_thresh and ll < best:                       # :78-85
                newLL, best, index = False, ll, kk
        self.last_position_cost, self.last_signature_cost = pos, sig
        return newLL, index

    # ---- EKF_SLAM.m:100-122 / EKF_SLAM_UC.m:102-124 ---------------------------------------------------------------------------------------
    def measure(self, laserData, u, landmark_list):
        observed_LL = landmark_list.getLandmark(laserData, self.x)
        if observed_LL is None or len(observed_LL) == 0:
            return
        observed_LL = np.asarray(observed_LL, float).reshape(-1, 3)
        for ii in range(1, observed_LL.shape[0] + 1):
            z = observed_LL[ii - 1]
            R = np.zeros((2, 2))
            R[0, 0], R[1, 1] = z[0] * self.Rc[0], z[1] * self.Rc[1]
            if len(self.x) < 4:
                self.append(u, R, _lookup_loc(landmark_list, None), 1)
            elif self.mode == "known":
                if z[2] > self.N:
                    self.append(u, R, _lookup_loc(landmark_list, z[2]), z[2])
                else:
                    self.correct(z, R, ii)                              # idx = ii (EKF_SLAM.m:123)
            else:
                new_LM, idx = self.estimateCorrespondence(z, R)
                if new_LM:
                    self.append(u, R, _lookup_loc(landmark_list, idx), idx)
                else:
                    self.correct(z, R, idx)
