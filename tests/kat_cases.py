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


# =====================================================================================================================
# KAT-8 ... KAT-12 (round 3): heading != 0 everywhere, two corrections in a row, the UC new-landmark dispatch.
# KAT-1..7 all sit at x(3) = 0; a sign or argument-order slip in any heading-dependent expression that the oracle and the
# kernels shared would have passed them.  Worked on paper from the source lines cited; sind/cosd/atan2d/wrapTo360 only
# meet multiples of 90 deg, so every value below is an exact short decimal or a small rational.
#
# KAT-8  predict at heading 90, EKF_SLAM.m:40-51 + f :56-65       x = [1 2 90 | 3 4], u = [2 90], C = .2
#     P = [.5 .1 .2 .3  0 ;  .1 .4 .1  0 .2 ;  .2 .1 .3 .1 .1 ;  .3  0 .1  1  0 ;   0 .2 .1  0  2]
#     :42     W = [2 cosd(90); 2 sind(90); 90] = [0; 2; 90]                  (PRE-motion heading 90, not 180)
#     :44     Q(1:3,1:3) = W*.2*W' = .2*[0 0 0; 0 4 180; 0 180 8100] = [0 0 0; 0 .8 36; 0 36 1620]
#     :58-60  x_new = [1 + 2 cosd(180); 2 + 2 sind(180); 180] = [-1; 2; 180]
#     :63-64  F(1,3) = -2 sind(90) = -2,  F(2,3) = 2 cosd(90) = 0            (heading WITHOUT u2; with it: 0 and -2)
#     :47     F P: row1 <- row1 - 2 row3 = [.1 -.1 -.4 .1 -.2];  (F P) F': col1 <- col1 - 2 col3:
#             col1 = [.1+.8; .1-.2; .2-.6; .3-.2; 0-.2] = [.9; -.1; -.4; .1; -.2]   -> strip entries P(1,4), P(1,5) move
#             + Q -> P(2,2) = 1.2, P(2,3) = 36.1, P(3,3) = 1620.3
#     :50     x(3) = wrapTo360(180) = 180
K8_X = np.array([1.0, 2, 90, 3, 4])
K8_P = np.array([[.5, .1, .2, .3, 0], [.1, .4, .1, 0, .2], [.2, .1, .3, .1, .1], [.3, 0, .1, 1, 0], [0, .2, .1, 0, 2]])
K8_U = [2.0, 90.0]
K8_X_OUT = np.array([-1.0, 2, 180, 3, 4])
K8_Q_OUT = np.array([[0, 0, 0], [0, .8, 36], [0, 36, 1620.0]])
K8_P_OUT = np.array([
    [.9, -.1, -.4, .1, -.2],
    [-.1, 1.2, 36.1, 0, .2],
    [-.4, 36.1, 1620.3, .1, .1],
    [.1, 0, .1, 1, 0],
    [-.2, .2, .1, 0, 2]])

# KAT-9  correction body with a heading, EKF_SLAM.m:124-145       P = diag(.1 .1 .1 .5 .5), z = [2.5 10], R = diag(.025 50)
#   (a) x = [1 2 90 | 3 2]: the landmark is dead ahead in WORLD x (bearing 0 from the robot's position), heading 90
#     :125-127  delta = [2; 0], q = 4
#     :130      z_k = [2; wrapTo360(atan2d(0,2) - 90)] = [2; wrapTo360(-90)] = [2; 270]   (atan2d(y,x), MINUS x(3))
#     :137-138  H_s = [-1 0 0 1 0; 0 -.5 -1 0 .5],  phi = diag(.625, 50.25),  K as KAT-5 (H does not see the heading)
#     :144      nu = [.5; 10 - 270] = [.5; -260]   -- NOT wrapped to +100
#               x+ = [1 - .16(.5), 2 + (.2/201)260, 90 + (.4/201)260, 3 + .8(.5), 2 - 260/201]
#     :145      P+ = KAT-5's
#   (b) x = [1 2 270 | 1 0]: delta = [0; -2], q = 4
#     :130      z_k = [2; wrapTo360(atan2d(-2,0) - 270)] = [2; wrapTo360(-90 - 270)] = [2; wrapTo360(-360)] = [2; 0]
#     :137-138  H_s = (1/4)[-2*0, -2*(-2), 0, 2*0, 2*(-2); -2, -0, -4, 2, 0] = [0 1 0 0 -1; -.5 0 -1 .5 0]
#     :141      P H' = [0 -.05; .1 0; 0 -.1; 0 .25; -.5 0],  phi = diag(.1+.5, .025+.1+.125) + R = diag(.625, 50.25)
#     :143      K = [0 -.2/201; .16 0; 0 -.4/201; 0 1/201; -.8 0]
#     :144      nu = [.5; 10],  x+ = [1 - 2/201, 2.08, 270 - 4/201, 1 + 10/201, -.4]
#     :145      P+: {2,5} couple through range: P22 = .084, P25 = .08, P55 = .1;  {1,3,4} through bearing:
#               P11 = .1 - .01/201, P13 = -.02/201, P14 = .05/201, P33 = .1 - .04/201, P34 = .1/201, P44 = .5 - .25/201
K9_P = np.diag([.1, .1, .1, .5, .5])
K9_Z = [2.5, 10.0]
K9_R = np.diag([.025, 50.0])
K9A_X = np.array([1.0, 2, 90, 3, 2])
K9A_X_OUT = np.array([.92, 2 + 52 / 201, 90 + 104 / 201, 3.4, 2 - 260 / 201])
K9A_P_OUT = K5_P_OUT
K9B_X = np.array([1.0, 2, 270, 1, 0])
K9B_X_OUT = np.array([1 - 2 / 201, 2.08, 270 - 4 / 201, 1 + 10 / 201, -.4])
K9B_P_OUT = np.array([
    [.1 - .01 / 201, 0, -.02 / 201, .05 / 201, 0],
    [0, .084, 0, 0, .08],
    [-.02 / 201, 0, .1 - .04 / 201, .1 / 201, 0],
    [.05 / 201, 0, .1 / 201, .5 - .25 / 201, 0],
    [0, .08, 0, 0, .1]])

# KAT-10  append at heading 90 onto a one-landmark map, EKF_SLAM.m:67-98      state = KAT-8's INPUT (x = [1 2 90 | 3 4], P)
#     u = [2 90], R = diag(.5, 4), pos = (-1, 6), signature 9
#     :84-85  jxr = [1 0 -2 sind(90); 0 1 2 cosd(90)] = [1 0 -2; 0 1 0]              (heading 0 would give [1 0 0; 0 1 2])
#     :87-88  jz  = [cosd(90) -2 sind(90); sind(90) 2 cosd(90)] = [0 -2; 1 0]
#     :92     P(1:3,new) = Prr jxr' = [.5-.4 .1; .1-.2 .4; .2-.6 .1] = [.1 .1; -.1 .4; -.4 .1]
#     :91     jxr Prr jxr' = [.1+.8 .1-.2; -.1 .4] = [.9 -.1; -.1 .4];  jz R jz' = [0 -8; .5 0][0 1; -2 0] = diag(16, .5)
#             C = [16.9 -.1; -.1 .9]
#     :95     P(new, lm1) = jxr * P(lm1,1:3)' = [1 0 -2; 0 1 0] * [.3 0; 0 .2; .1 .1] = [.1 -.2; 0 .2]
K10_APPEND = dict(u=[2.0, 90.0], R=np.diag([.5, 4.0]), pos=[-1.0, 6.0], sig=9.0)
K10_X_OUT = np.array([1.0, 2, 90, 3, 4, -1, 6])
K10_P_OUT = np.array([
    [.5, .1, .2, .3, 0, .1, .1],
    [.1, .4, .1, 0, .2, -.1, .4],
    [.2, .1, .3, .1, .1, -.4, .1],
    [.3, 0, .1, 1, 0, .1, 0],
    [0, .2, .1, 0, 2, -.2, .2],
    [.1, -.1, -.4, .1, -.2, 16.9, -.1],
    [.1, .4, .1, 0, .2, -.1, .9]])

# KAT-11  two corrections in a row on different landmarks, EKF_SLAM.m:107,124-145: the second is linearised at the first's x+
#     and meets the cross-covariances the first one created.   x = [0 0 90 | 0 2 | 0 -2.55],  P = diag(.1 .1 .1 .5 .875 .3 .3)
#   1st: landmark 1, z = [2.5 0], R = diag(.025, 0)   (bearing 0 -> R22 = 0, :108)
#     delta = [0; 2], q = 4,  z_k = [2; wrapTo360(atan2d(2,0) - 90)] = [2; 0]
#     H_s = (1/4)[0 -4 0 0 4; 2 0 -4 -2 0] = [0 -1 0 0 1; .5 0 -1 -.5 0]
#     P H' = [0 .05; -.1 0; 0 -.1; 0 -.25; .875 0],  phi = diag(.1+.875+.025, .025+.1+.125+0) = diag(1, .25)
#     K = [0 .2; -.1 0; 0 -.4; 0 -1; .875 0],  nu = [.5; 0]  ->  x1 = [0 -.05 90 | 0 2.4375 | 0 -2.55]
#     P1: P22 = .1-.01 = .09, P25 = .0875, P55 = .875-.765625 = .109375;  P11 = .1-.01 = .09, P13 = .02, P14 = .05,
#         P33 = .1-.04 = .06, P34 = -.1, P44 = .5-.25 = .25;  landmark 2 untouched (.3, .3)
#   2nd: landmark 2, z = [1 .17232], R = diag(.01, .8616)
#     delta = [0 - 0; -2.55 - (-.05)] = [0; -2.5]  (2.55 without the re-linearisation), q = 6.25
#     z_k = [2.5; wrapTo360(atan2d(-2.5,0) - 90)] = [2.5; wrapTo360(-180)] = [2.5; 180]
#     H (2 x 7) = [0 1 0 0 0 0 -1; -.4 0 -1 0 0 .4 0]
#     c1 = P1 H'(:,1) = P1(:,2) - P1(:,7) = [0 .09 0 0 .0875 0 -.3]'
#     c2 = P1 H'(:,2) = -.4 P1(:,1) - P1(:,3) + .4 P1(:,6) = [-.056 0 -.068 .08 0 .12 0]'     (.08 = -.4(.05) + .1)
#     phi = diag(.09+.3+.01, .0224+.068+.048+.8616) = diag(.4, 1),  K = [c1/.4, c2]
#     nu = [1 - 2.5; .17232 - 180] = [-1.5; -179.82768]
#     x2 = x1 + K nu = [10.07035008, -.05-.3375, 90+12.22828224, -14.3862144, 2.4375-.328125, -21.5793216, -2.55+1.125]
#     P2 = P1 - c1 c1'/.4 - c2 c2'
K11_X = np.array([0.0, 0, 90, 0, 2, 0, -2.55])
K11_P = np.diag([.1, .1, .1, .5, .875, .3, .3])
K11_Z1, K11_R1 = [2.5, 0.0], np.diag([.025, 0.0])
K11_Z2, K11_R2 = [1.0, .17232], np.diag([.01, .8616])
K11_X1 = np.array([0.0, -.05, 90, 0, 2.4375, 0, -2.55])
K11_P1 = np.array([
    [.09, 0, .02, .05, 0, 0, 0],
    [0, .09, 0, 0, .0875, 0, 0],
    [.02, 0, .06, -.1, 0, 0, 0],
    [.05, 0, -.1, .25, 0, 0, 0],
    [0, .0875, 0, 0, .109375, 0, 0],
    [0, 0, 0, 0, 0, .3, 0],
    [0, 0, 0, 0, 0, 0, .3]])
K11_X2 = np.array([10.07035008, -.3875, 102.22828224, -14.3862144, 2.109375, -21.5793216, -1.425])
K11_P2 = np.array([
    [.086864, 0, .016192, .05448, 0, .00672, 0],
    [0, .06975, 0, 0, .0678125, 0, .0675],
    [.016192, 0, .055376, -.09456, 0, .00816, 0],
    [.05448, 0, -.09456, .2436, 0, -.0096, 0],
    [0, .0678125, 0, 0, .090234375, 0, .065625],
    [.00672, 0, .00816, -.0096, 0, .2856, 0],
    [0, .0675, 0, 0, .065625, 0, .075]])

# KAT-12  measure() with unknown correspondence, three rows on a two-landmark map, the middle one new
#     EKF_SLAM_UC.m:102-152 (Rc = [.1 5] :13, s_cost = 1e-11, s_thresh = 1e9 :16), Correspondence.m:71-85
#     x = [0 0 90 | 0 2 | 0 -2.55],  P = diag(.1 .1 .1 .5 .65 .3 .3),  s = [1; 2],  u = [2 90]
#     landmark table (index, loc): (1,(0,2)) (2,(0,-2.55)) (7,(9,9)) (3,(-1,6))
#     observed_LL = [2.5 0 1;  4 30 7;  1.1 .17232 2]
#   row 1: R = diag(.25, 0).  signature costs (1-1)^2/1e-11 = 0, (1-2)^2/1e-11 = 1e11 > 1e9 -> (false, 1) -> correction:
#     as KAT-11's first with P55 = .65: phi = diag(.1+.65+.25, .25) = diag(1, .25), K = [0 .2; -.1 0; 0 -.4; 0 -1; .65 0]
#     x = [0 -.05 90 | 0 2.325 | ...];  P22 = .09, P25 = .065, P55 = .65-.4225 = .2275; P11 = .09, P13 = .02, P14 = .05,
#     P33 = .06, P34 = -.1, P44 = .25
#   row 2: R = diag(.4, 150).  costs 36/1e-11, 25/1e-11 > 1e9 -> (true, N+1 = 3) -> :123 append(u, R, loc of the table entry
#     whose index == 3 -- (-1, 6), NOT the entry carrying the observed signature 7 and not a position derived from z --, 3):
#     the stored signature is 3, not 7.   Heading 90: jxr = [1 0 -2; 0 1 0], jz = [0 -2; 1 0]
#     Prr jxr' = [.09-.04 0; 0 .09; .02-.12 0] = [.05 0; 0 .09; -.1 0];  jxr Prr jxr' = diag(.05+.2, .09) = diag(.25, .09)
#     jz R jz' = [0 -300; .4 0][0 1; -2 0] = diag(600, .4);  C = diag(600.25, .49)
#     P(new, lm1) = jxr * [.05 0; 0 .065; -.1 0] = [.25 0; 0 .065];  P(new, lm2) = 0
#   row 3: R = diag(.11, .8616).  s = [1 2 3]: cost 0 at k = 2 -> (false, 2) -> correction on landmark 2 with n = 9:
#     delta = [0; -2.5], q = 6.25, z_k = [2.5; 180], H = [0 1 0 0 0 0 -1 0 0; -.4 0 -1 0 0 .4 0 0 0]
#     c1 = P(:,2) - P(:,7) = [0 .09 0 0 .065 0 -.3 0 .09]'        (row 9: the new landmark's P(9,2) = .09)
#     c2 = -.4 P(:,1) - P(:,3) + .4 P(:,6) = [-.056 0 -.068 .08 0 .12 0 .08 0]'     (row 8: -.4(.05) + .1)
#     phi = diag(.09+.3+.11, .0224+.068+.048+.8616) = diag(.5, 1),  K = [2 c1, c2],  nu = [1.1-2.5; .17232-180] = [-1.4; -179.82768]
#     x = [10.07035008, -.05-.252, 102.22828224, -14.3862144, 2.325-.182, -21.5793216, -2.55+.84, -1-14.3862144, 6-.252]
#     P = P - 2 c1 c1' - c2 c2'
K12_X = np.array([0.0, 0, 90, 0, 2, 0, -2.55])
K12_P = np.diag([.1, .1, .1, .5, .65, .3, .3])
K12_S = np.array([1.0, 2.0])
K12_U = [2.0, 90.0]
K12_TABLE = [(1, (0.0, 2.0)), (2, (0.0, -2.55)), (7, (9.0, 9.0)), (3, (-1.0, 6.0))]
K12_OBSERVED = np.array([[2.5, 0, 1], [4, 30, 7], [1.1, .17232, 2]])
K12_S_OUT = np.array([1.0, 2.0, 3.0])
K12_X_OUT = np.array([10.07035008, -.302, 102.22828224, -14.3862144, 2.143, -21.5793216, -1.71, -15.3862144, 5.748])
K12_P_OUT = np.array([
    [.086864, 0, .016192, .05448, 0, .00672, 0, .05448, 0],
    [0, .0738, 0, 0, .0533, 0, .054, 0, .0738],
    [.016192, 0, .055376, -.09456, 0, .00816, 0, -.09456, 0],
    [.05448, 0, -.09456, .2436, 0, -.0096, 0, .2436, 0],
    [0, .0533, 0, 0, .21905, 0, .039, 0, .0533],
    [.00672, 0, .00816, -.0096, 0, .2856, 0, -.0096, 0],
    [0, .054, 0, 0, .039, 0, .12, 0, .054],
    [.05448, 0, -.09456, .2436, 0, -.0096, 0, 600.2436, 0],
    [0, .0738, 0, 0, .0533, 0, .054, 0, .4738]])
# state after rows 1-2 only (before the third row's correction), for a stepwise check
K12_X_AFTER2 = np.array([0.0, -.05, 90, 0, 2.325, 0, -2.55, -1, 6])
K12_P_AFTER2 = np.array([
    [.09, 0, .02, .05, 0, 0, 0, .05, 0],
    [0, .09, 0, 0, .065, 0, 0, 0, .09],
    [.02, 0, .06, -.1, 0, 0, 0, -.1, 0],
    [.05, 0, -.1, .25, 0, 0, 0, .25, 0],
    [0, .065, 0, 0, .2275, 0, 0, 0, .065],
    [0, 0, 0, 0, 0, .3, 0, 0, 0],
    [0, 0, 0, 0, 0, 0, .3, 0, 0],
    [.05, 0, -.1, .25, 0, 0, 0, 600.25, 0],
    [0, .09, 0, 0, .065, 0, 0, 0, .49]])


# KAT-13  association costs at heading 90, Correspondence.m:49-87: KAT-7's scene turned by 90 deg about the robot.  Range and RELATIVE
#     bearing do not change under that rotation, so costs and decisions must be KAT-7's -- but only if z_k = wrapTo360(atan2d(dy,dx) -
#     x(3)) (:56) is evaluated as written: with + x(3), or atan2d's arguments swapped, landmark 1's predicted bearing is 180 or 270, not 0.
#     x = [0 0 90 | 0 2 | -4 0],  P = diag(.1 .1 .1 .5 .5 .3 .3), s = [5 5], R = diag(.025, 50)
#     landmark 1 (0,2):   delta = [0;2],  q = 4,  z_k = [2; wrapTo360(atan2d(2,0) - 90)] = [2; 0]
#                         H_s = [0 -1 0 0 1; .5 0 -1 -.5 0]  (KAT-11),  phi = diag(.1+.5, .025+.1+.125) + R = diag(.625, 50.25)
#     landmark 2 (-4,0):  delta = [-4;0], q = 16, z_k = [4; wrapTo360(atan2d(0,-4) - 90)] = [4; wrapTo360(180 - 90)] = [4; 90]
#                         H_s = (1/16)[16 0 0 -16 0; 0 4 -16 0 -4] = [1 0 0 -1 0; 0 .25 -1 0 -.25]
#                         phi = diag(.1+.3, .00625+.1+.01875) + R = diag(.425, 50.125)
#     z = [2.5 10 5]:  nu = [.5; 10], [-1.5; -80]  -> pc = [.4 + 400/201, 90/17 + 51200/401]  = K7_PC_A
#     z = [4.2 85 5]:  nu = [2.2; 85], [.2; -5]    -> pc = [7.744 + 28900/201, 8/85 + 200/401] = K7_PC_B
K13_X = np.array([0.0, 0, 90, 0, 2, -4, 0])


# KAT-14  measure() with KNOWN correspondence, the two dispatch quirks of EKF_SLAM.m:116-123
#   (a) :123  idx = ii -- the ROW NUMBER, not z(3).  x = [0 0 0 | 2 0 | 0 4], P = diag(.1 .1 .1 .5 .5 .3 .3), one observed row [2.5 10 2]:
#       z(3) = 2 <= N = 2 -> correction with idx = ii = 1: landmark ONE is corrected with an observation that names landmark two.
#       R = diag(2.5*.01, 10*5) = diag(.025, 50) (Rc = [.01 5], :13,:108) -> exactly KAT-5 on the first five states; landmark 2 (no
#       cross-covariance with anything) keeps (0, 4) and .3 I.
#   (b) :118-120  z(3) > N -> append(u, R, loc of the table entry whose index == z(3), z(3)): the stored signature is z(3) itself (5 here,
#       not N+1 = 3 as in the UC class).  State = KAT-12 after its first row (heading 90), u = [2 90], row [4 30 5], R = diag(.04, 150):
#       jxr = [1 0 -2; 0 1 0], jz = [0 -2; 1 0];  jz R jz' = [0 -300; .04 0][0 1; -2 0] = diag(600, .04);  C = diag(.25+600, .09+.04)
#       everything else as KAT-12's append (K12_P_AFTER2) -- only P(9,9) = .13 instead of .49.
K14A_X = K7_X
K14A_P = K7_P
K14A_OBSERVED = np.array([[2.5, 10.0, 2.0]])
K14A_X_OUT = np.concatenate([K5_X_OUT, [0.0, 4.0]])
K14A_P_OUT = np.zeros((7, 7)); K14A_P_OUT[:5, :5] = K5_P_OUT; K14A_P_OUT[5, 5] = K14A_P_OUT[6, 6] = .3
K14B_X = K12_X_AFTER2[:7]
K14B_P = K12_P_AFTER2[:7, :7]
K14B_TABLE = [(1, (0.0, 2.0)), (2, (0.0, -2.55)), (3, (9.0, 9.0)), (5, (-1.0, 6.0))]
K14B_OBSERVED = np.array([[4.0, 30.0, 5.0]])
K14B_X_OUT = K12_X_AFTER2
K14B_P_OUT = K12_P_AFTER2.copy(); K14B_P_OUT[8, 8] = .13
K14B_S_OUT = np.array([1.0, 2.0, 5.0])


class KatTable:
    """A landmark_list with a fixed struct array (RANSAC.m:238-241) and a scripted getLandmark: the duck type
    EKF_SLAM*.measure consumes (EKF_SLAM.m:102,111,120)."""

    class _Entry:
        def __init__(self, index, loc):
            self.index, self.loc, self.observe, self.fresh = index, np.array(loc, dtype=float), 1, 0

    class _Obj:
        def __init__(self, entries):
            self.landmark = entries

        def table(self):
            return (np.array([e.index for e in self.landmark], dtype=float),
                    np.array([e.loc for e in self.landmark], dtype=float).reshape(-1, 2))

    def __init__(self, table, observed):
        self.landmarkObj = KatTable._Obj([KatTable._Entry(i, l) for i, l in table])
        self._observed = np.asarray(observed, dtype=float)

    def getLandmark(self, laserdata, x):
        return self._observed

# KAT-15  append, then a correction of the landmark just appended -- the cross-covariances the append wrote are what the correction reads
#     EKF_SLAM.m:67-98, then :124-145.   x = [0 0 90], P = .1 I3, no landmark;  append(u = [2 90], R = diag(.4, .125), pos = (0, 2), sig = 5)
#     :84-88  heading 90: jxr = [1 0 -2; 0 1 0],  jz = [cosd90 -2 sind90; sind90 2 cosd90] = [0 -2; 1 0]
#     :91     jxr Prr jxr' = .1 [1+4 0; 0 1] = diag(.5, .1);  jz R jz' = diag(4(.125), .4) = diag(.5, .4);  C = diag(1, .5)
#     :92     Prr jxr' = [.1 0; 0 .1; -.2 0]
#     P_A = [.1 0 0 .1 0; 0 .1 0 0 .1; 0 0 .1 -.2 0; .1 0 -.2 1 0; 0 .1 0 0 .5],  x_A = [0 0 90 | 0 2],  s = [5]
#   correction of landmark 1 with z = [2.5 10], R = diag(.1, .125):
#     :125-130  delta = (0, 2), q = 4, z_k = [2; wrapTo360(atan2d(2,0) - 90)] = [2; 0]
#     :137-138  H_s = (1/4)[0 -4 0 0 4; 2 0 -4 -2 0] = [0 -1 0 0 1; .5 0 -1 -.5 0]
#     G = H P_A:  row 1 = -P(2,:) + P(5,:) = [0 0 0 0 .4];  row 2 = .5 P(1,:) - P(3,:) - .5 P(4,:) = [0 0 0 -.25 0]
#               (the robot columns cancel ONLY with P(1:3,new) = Prr jxr' as written: .5(.1) - .5(.1), -.1 - .5(-.2))
#     :141      phi = diag(.4 + .1, .125 + .125) = diag(.5, .25),  K = G' phi^-1 = [0 0; 0 0; 0 0; 0 -1; .8 0]
#     :144      nu = [.5; 10] (bearing innovation un-wrapped, H's bearing row in rad/m against degrees: -1 * 10 on the landmark's x)
#               x+ = [0 0 90 | -10 2.4]
#     :145      P+ = P_A - K G:  P44 = 1 - .25 = .75,  P55 = .5 - .32 = .18, nothing else moves
K15_X = np.array([0.0, 0, 90])
K15_P = np.diag([.1, .1, .1])
K15_APPEND = dict(u=[2.0, 90.0], R=np.diag([.4, .125]), pos=[0.0, 2.0], sig=5.0)
K15_X_A = np.array([0.0, 0, 90, 0, 2])
K15_P_A = np.array([
    [.1, 0, 0, .1, 0],
    [0, .1, 0, 0, .1],
    [0, 0, .1, -.2, 0],
    [.1, 0, -.2, 1, 0],
    [0, .1, 0, 0, .5]])
K15_Z, K15_R = np.array([2.5, 10.0]), np.diag([.1, .125])
K15_X_OUT = np.array([0.0, 0, 90, -10, 2.4])
K15_P_OUT = K15_P_A.copy()
K15_P_OUT[3, 3], K15_P_OUT[4, 4] = .75, .18

# KAT-16  correction body with a FULL phi (EKF_SLAM.m:124-145): the off-diagonals of phi_k^-1 in K = P H' phi^-1 (:141-143), and the
#         pivoting branch of the 2 x 2 inverse (|phi(2,1)| > |phi(1,1)|).  In KAT-5..15 phi is diagonal every time.
#     x = [0 0 0 | 2 0],  z = [2.5, .025],  R = diag(.025, .125)  (= z .* Rc, Rc = [.01 5]),  P symmetric positive definite with
#     P(1,2) = P(1,3) = P(1,4) = .1 and P(3,4) = -.1:
#         P = [.2 .1 .1 .1 0; .1 .2 0 0 0; .1 0 .3 -.1 0; .1 0 -.1 .2 0; 0 0 0 0 .1]
#     :125-138  delta = (2, 0), q = 4, z_k = [2; 0],  H_s = [-1 0 0 1 0; 0 -.5 -1 0 .5]                 (as KAT-5)
#     G = H P:  row 1 = -P(1,:) + P(4,:)              = [-.1  -.1  -.2  .1  0  ]
#               row 2 = -.5 P(2,:) - P(3,:) + .5 P(5,:) = [-.15 -.1  -.3  .1  .05]
#     :141      phi = G H' + R:  phi11 = .1 + .1 + .025 = .225 (9/40);  phi12 = phi21 = (-.1)(-.5) + (-.2)(-1) = .25;
#               phi22 = (-.1)(-.5) + (-.3)(-1) + (.05)(.5) + .125 = .5        -- |phi21| = .25 > phi11 = .225: the inverse pivots
#               det = 9/80 - 1/16 = 1/20,  phi^-1 = 20 [.5 -.25; -.25 .225] = [10 -5; -5 4.5]
#     :143      K = P H' phi^-1 = G' phi^-1 (P symmetric), row i = [G1(i) G2(i)] [10 -5; -5 4.5]:
#               K = [-.25 -.175; -.5 .05; -.5 -.35; .5 -.05; -.25 .225]
#     :144      nu = [.5; .025],  x+ = x + K nu = [-.129375  -.24875  -.25875  2.24875  -.119375]
#     :145      P+ = P - K G (every entry moves; exact values as 1/800ths below)
K16_X = np.array([0.0, 0, 0, 2, 0])
K16_P = np.array([
    [.2, .1, .1, .1, 0],
    [.1, .2, 0, 0, 0],
    [.1, 0, .3, -.1, 0],
    [.1, 0, -.1, .2, 0],
    [0, 0, 0, 0, .1]])
K16_Z, K16_R = np.array([2.5, .025]), np.diag([.025, .125])
K16_PHI = np.array([[.225, .25], [.25, .5]])
K16_K = np.array([[-.25, -.175], [-.5, .05], [-.5, -.35], [.5, -.05], [-.25, .225]])
K16_X_OUT = np.array([-.129375, -.24875, -.25875, 2.24875, -.119375])
K16_P_OUT = np.array([
    [119, 46, -2, 114, 7],
    [46, 124, -68, 36, -2],
    [-2, -68, 76, -12, 14],
    [114, 36, -12, 124, 2],
    [7, -2, 14, 2, 71]]) / 800.0

# KAT-17  association with robot-landmark cross-covariance (Correspondence.m:49-87): KAT-7's scene (x = [0 0 0 | 2 0 | 0 4], s = [5 5],
#         R = diag(.025, 50)) with a non-zero strip P(1:3, 4:7), so that the cross terms of H_k P H_k' (:66) enter the position cost (:69)
#         and FLIP the decision of the commented-out likelihood (:74, w_pos = 1, s_cost = 1) relative to KAT-7's diagonal P.
#     z = [3, 45, 5]:  landmark 1 (2,0): nu = [3 - 2; 45 - 0] = [1; 45];  landmark 2 (0,4): z_k = [4; 90], nu = [-1; -45]
#     KAT-7's P = diag(.1 .1 .1 .5 .5 .3 .3):  phi_1 = diag(.625, 50.25),  phi_2 = diag(.425, 50.125)
#         pc = [1/.625 + 2025/50.25,  1/.425 + 2025/50.125] = [8/5 + 2700/67,  40/17 + 16200/401] = [41.8985..., 42.7519...]  -> landmark 1
#     with the strip  P(1,4) = .2,  P(3,5) = .05,  P(2,7) = -.05  (and their mirrors; P stays positive definite):
#         landmark 1, H_s = [-1 0 0 1 0; 0 -.5 -1 0 .5] on rows {1,2,3,4,5}:
#             phi11 = P11 - 2 P14 + P44 + R11 = .1 - .4 + .5 + .025 = .225
#             phi22 = .25 P22 + P33 + .25 P55 - 2 (.5) P35 + R22 = .025 + .1 + .125 - .05 + 50 = 50.2;   phi12 = 0
#         landmark 2, H_s = [0 -1 0 0 1; .25 0 -1 -.25 0] on rows {1,2,3,6,7}:
#             phi11 = P22 - 2 P27 + P77 + R11 = .1 + .1 + .3 + .025 = .525;   phi22 = .00625 + .1 + .01875 + 50 = 50.125;   phi12 = 0
#         pc = [1/.225 + 2025/50.2,  1/.525 + 2025/50.125] = [40/9 + 10125/251,  40/21 + 16200/401] = [44.7830..., 42.3037...]  -> landmark 2
#     live line :75 (w_pos = 0): signature costs 0, 0 -> first index either way
K17_X = np.array([0.0, 0, 0, 2, 0, 0, 4])
K17_S = [5.0, 5.0]
K17_R = np.diag([.025, 50.0])
K17_Z = [3.0, 45.0, 5.0]
K17_P_DIAG = np.diag([.1, .1, .1, .5, .5, .3, .3])
K17_P = K17_P_DIAG.copy()
K17_P[0, 3] = K17_P[3, 0] = .2
K17_P[2, 4] = K17_P[4, 2] = .05
K17_P[1, 6] = K17_P[6, 1] = -.05
K17_PC_DIAG = np.array([8 / 5 + 2700 / 67, 40 / 17 + 16200 / 401])
K17_PC = np.array([40 / 9 + 10125 / 251, 40 / 21 + 16200 / 401])
