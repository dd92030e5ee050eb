"""Hand-derived known-answer cases for the append (a4), correction (a6) and association (a8) rows of SURVEY.md section 8.

The reference holds no vectors and cannot run here (PARITY UNPINNED), so these are the only pins that do not come from
the builder's own restatements: each expected value below was worked out ON PAPER from the reference's source lines and is
asserted against oracle/ekf_dense.py, oracle/ekf_structured.c (tests/test_oracle_kat.py) AND the HIP path
(tests/test_kat_gpu.py).  All inputs are chosen so that sind / cosd / atan2d / wrapTo360 are exact and the answers are short
decimals or small rationals.

KAT-5  correction body, EKF_SLAM.m:124-145
    x = [0 0 0 | 2 0],  P = diag(.1 .1 .1 .5 .5),  z = [2.5, 10],  R = diag(.025, 50)  (= z .* Rc, Rc = [.01 5], :108)
    :125-127  delta = [2;0], q = 4, sqrt(q) = 2
    :130      z_k = [2; wrapTo360(atan2d(0,2) - 0)] = [2; 0]
    :137-138  H_s = (1/4)[-4 -0 0 4 0; 0 -2 -4 -0 2] = [-1 0 0 1 0; 0 -.5 -1 0 .5]   (bearing row is rad/m, NOT deg)
    :141      P H' = [-.1 0; 0 -.05; 0 -.1; .5 0; 0 .25]
              phi = [ .1+.5 , 0 ; 0 , .025+.1+.125 ] + R = diag(.625, 50.25)
    :143      K = P H' phi^-1 = [-.16 0; 0 -.2/201; 0 -.4/201; .8 0; 0 1/201]          (1/50.25 = 4/201)
    :144      nu = z - z_k = [.5; 10]  (the 10 deg bearing innovation meets a rad/m Jacobian row: reproduced, not fixed)
              x+ = [-.08, -2/201, -4/201, 2.4, 10/201]
    :145      P+ = P - K (H P);  rows/cols {1,4} couple through range, {2,3,5} through bearing:
              P11 = .1-.016 = .084   P14 = P41 = .08   P44 = .5-.4 = .1
              P22 = .1-.01/201   P23 = -.02/201   P25 = .05/201   P33 = .1-.04/201   P35 = .1/201   P55 = .5-.25/201

KAT-6  append, EKF_SLAM.m:67-98, twice from the KAT-1 state (ctor + predict([1 0]))
    x = [1 0 0], P = [.3 0 0; 0 .2 .1; 0 .1 .1]
    first:  u = [1 90], R = diag(.02, 10), pos = (3,4), signature 7
      :84-85  jxr = [1 0 -1*sind(0); 0 1 1*cosd(0)] = [1 0 0; 0 1 1]
      :87-88  jz  = [cosd(90) -sind(90); sind(90) cosd(90)] = [0 -1; 1 0]
      :91     C = jxr Prr jxr' + jz R jz' = [.3 0; 0 .2+.1+.1+.1] + [10 0; 0 .02] = [10.3 0; 0 .52]
      :92     P(1:3,new) = Prr jxr' = [.3 0; 0 .3; 0 .2]
    second: u = [2 180], R = diag(.5, 4), pos = (-1,6), signature 9
      jxr = [1 0 0; 0 1 2],  jz = [-1 -0; 0 -2]
      :91     C = [.3 0; 0 .2+2(.1)+2(.1)+4(.1)] + diag(.5, 16) = [.8 0; 0 17]
      :92     P(1:3,new) = [.3 0; 0 .4; 0 .3]
      :95     P(new, lm1) = jxr * P(lm1,1:3)' = [1 0 0; 0 1 2] * [.3 0; 0 .3; 0 .2] = [.3 0; 0 .7]

KAT-7  association, Correspondence.m:49-87, two landmarks, position cost made to count
    x = [0 0 0 | 2 0 | 0 4], P = diag(.1 .1 .1 .5 .5 .3 .3), s = [5 5], R = diag(.025, 50)
    landmark 1 (2,0):  z_k = [2;0],  phi = diag(.625, 50.25)                     (as KAT-5)
    landmark 2 (0,4):  delta = [0;4], q = 16, z_k = [4; 90]
                       H_s = (1/16)[-0 -16 0 0 16; 4 -0 -16 -4 0] = [0 -1 0 0 1; .25 0 -1 -.25 0]
                       phi = diag(.1+.3, .00625+.1+.01875) + R = diag(.425, 50.125)
    :69  position_cost = nu' phi^-1 nu;   :71  signiture_cost = (z3 - s_k)^2 / s_cost
    z = [2.5 10 5]:  pc = [.25/.625 + 100/50.25, 2.25/.425 + 6400/50.125] = [.4 + 400/201, 90/17 + 51200/401]
    z = [4.2 85 5]:  pc = [4.84/.625 + 7225/50.25, .04/.425 + 25/50.125]  = [7.744 + 28900/201, 8/85 + 200/401]
    live line :75 (w_pos = 0): both signature costs are 0 -> tie -> first index (strict '<', :81) -> (false, 1) for both z
    commented-out line :74 (w_pos = 1, s_cost = 1): z = [2.5 10 5] -> (false, 1);  z = [4.2 85 5] -> (false, 2)
    threshold :78 with w_pos = 1: s_thresh = 100 and z = [2.5 10 5] keeps only landmark 1; s_thresh = 2 keeps none ->
    (true, 3);  z3 = 6 adds 1 to both likelihoods.
"""
import numpy as np

# ---- KAT-5 ----
K5_X = np.array([0.0, 0, 0, 2, 0])
K5_P = np.diag([.1, .1, .1, .5, .5])
K5_Z = [2.5, 10.0]
K5_R = np.diag([.025, 50.0])
K5_X_OUT = np.array([-.08, -2 / 201, -4 / 201, 2.4, 10 / 201])
K5_P_OUT = np.array([
    [.084, 0, 0, .08, 0],
    [0, .1 - .01 / 201, -.02 / 201, 0, .05 / 201],
    [0, -.02 / 201, .1 - .04 / 201, 0, .1 / 201],
    [.08, 0, 0, .1, 0],
    [0, .05 / 201, .1 / 201, 0, .5 - .25 / 201]])

# ---- KAT-6 ----
K6_APPENDS = [dict(u=[1.0, 90.0], R=np.diag([.02, 10.0]), pos=[3.0, 4.0], sig=7.0),
              dict(u=[2.0, 180.0], R=np.diag([.5, 4.0]), pos=[-1.0, 6.0], sig=9.0)]
K6_X_OUT = np.array([1.0, 0, 0, 3, 4, -1, 6])
K6_S_OUT = np.array([7.0, 9.0])
K6_P_OUT = np.array([
    [.3, 0, 0, .3, 0, .3, 0],
    [0, .2, .1, 0, .3, 0, .4],
    [0, .1, .1, 0, .2, 0, .3],
    [.3, 0, 0, 10.3, 0, .3, 0],
    [0, .3, .2, 0, .52, 0, .7],
    [.3, 0, 0, .3, 0, .8, 0],
    [0, .4, .3, 0, .7, 0, 17.0]])

# ---- KAT-7 ----
K7_X = np.array([0.0, 0, 0, 2, 0, 0, 4])
K7_P = np.diag([.1, .1, .1, .5, .5, .3, .3])
K7_S = np.array([5.0, 5.0])
K7_R = np.diag([.025, 50.0])
K7_ZA, K7_ZB = [2.5, 10.0, 5.0], [4.2, 85.0, 5.0]
K7_PC_A = np.array([.4 + 400 / 201, 90 / 17 + 51200 / 401])
K7_PC_B = np.array([7.744 + 28900 / 201, 8 / 85 + 200 / 401])
