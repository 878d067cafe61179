/* orc_robot.c -- CPU ORACLE (test infrastructure): NAO data, utility maths and Robot state.
 * Follows reference src/robotParameters.cpp, src/generalizedFunctions.cpp, src/Robot.cpp. */
#include "lmh_oracle.h"
#include "orc_linalg.h"
#include "orc_internal.h"

/* Robot.cpp:3 -- the literal used by the reference, not M_PI */
static const double kPi = 3.14159265358979323846;

const int orc_parent[ORC_NF] = {-1, 0, 1, 2, 3, 4, 5, 6, 0, 8, 9, 10, 11, 12, 13,
                                0, 15, 16, 17, 18, 0, 20, 21, 22, 23, 0, 25, 26};      /* Robot.cpp:165 */
const int orc_act[ORC_NF] = {0, 1, 2, 3, 4, 5, 6, 0, 7, 8, 9, 10, 11, 12, 0,
                             13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 0};       /* Robot.cpp:172 */

/* ------------------------------------------------------------------------------------
 * NAO H25 inertial table (robotParameters.cpp:8-229).  One row per frame:
 * mass | com xyz | inertia row-major.  Values are restated VERBATIM, including the
 * reference's asymmetric entries: e.g. frame 13 I[1][2] =
 * 1.8740920495e-055 and I[0][2] != I[2][0]; frames 19/24 I[0][1] = 5.71599e-5 vs
 * I[1][0] = 5.71599e-6.  Do not "fix" them: M[0:6,0:6] inherits the asymmetry.
 * Frames 7, 14 (soles) and 27 (extra head frame) are massless virtual frames.
 * ------------------------------------------------------------------------------------ */
static const double kNao[ORC_NF][13] = {
/* 0 Trunk */          {1.0496, -0.00413, 0.0, 0.04342,
                        0.0050623407587, 1.4311580344e-05, 0.000155119082081,
                        1.4311580344e-05, 0.0048801358789, -2.7079340725e-05,
                        0.000155119082081, -2.7079340725e-05, 0.001610300038},
/* 1 RHipYawPitch */   {0.06981, -0.00781, 0.01114, 0.02661,
                        8.9971952548e-05, 5.0021899369e-06, 1.2735249584e-05,
                        5.0021899369e-06, 0.00010552610911, -2.770080027e-05,
                        1.2735249584e-05, -2.7700800274e-05, 6.6887238063e-05},
/* 2 RHipRoll */       {0.14053, -0.01549, -0.00029, -0.00515,
                        2.7586540455e-05, -1.9190000e-08, -4.108219855e-06,
                        -1.91900007e-08, 9.8269956652e-05, 2.5099999856e-05,
                        -4.108219855e-06, 2.5099999856e-05, 8.8103319285e-05},
/* 3 RHipPitch */      {0.38968, 0.00138, -0.00221, -0.05373,
                        0.0016374820843, -8.3954000729e-07, 8.5883009888e-05,
                        -8.3954000729e-07, 0.0015922139864, -3.9176258724e-05,
                        8.5883009888e-05, -3.9176258724e-05, 0.00030397824594},
/* 4 RKneePitch */     {0.30142, 0.00453, -0.00225, -0.04936,
                        0.0011828296119, -8.96500012e-07, 2.7996900826e-05,
                        -8.96500012e-07, 0.0011282785563, -3.8476038753e-05,
                        2.7996900826e-05, -3.8476038753e-05, 0.00019145276747},
/* 5 RAnklePitch */    {0.13416, 0.00045, -0.00029, 0.00685,
                        3.8508129364e-05, 6.4339999994e-08, 3.8746597966e-06,
                        6.4339999994e-08, 7.4310817581e-05, -4.5799999349e-09,
                        3.8746597966e-06, -4.5799999349e-09, 5.491311822e-05},
/* 6 RAnkleRoll */     {0.17184, 0.02542, -0.0033, -0.03239,
                        0.00026930202148, 5.87505001921e-06, 0.00013913327712,
                        5.8750501921e-06, 0.00064347387524, -1.8849170374e-05,
                        0.00013913327712, -1.884917037e-05, 0.000525034478946},
/* 7 R sole */         {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
/* 8 LHipYawPitch */   {0.06981, -0.00781, -0.01114, 0.02661,
                        8.1502330431e-05, -4.9944901548e-06, 1.27481698664e-05,
                        -4.9944901548e-06, 0.00010132555326, 2.3454740585e-05,
                        1.2748169866e-05, 2.3454740585e-05, 6.2623628764e-05},
/* 9 LHipRoll */       {0.14053, -0.01549, 0.00029, -0.00515,
                        2.7583539122e-05, -2.2329999183e-08, -4.0816398723e-06,
                        -2.2329999183e-08, 9.8270553281e-05, -4.1899999026e-09,
                        -4.0816398723e-06, -4.1899999026e-09, 8.809973223e-05},
/* 10 LHipPitch */     {0.38968, 0.00138, 0.00221, -0.05373,
                        0.001636719564, 9.2451000455e-07, 8.5306681285e-05,
                        9.2451000455e-07, 0.001591072767, 3.8361598854e-05,
                        8.5306681285e-05, 3.8361598854e-05, 0.00030374340713},
/* 11 LKneePitch */    {0.30142, 0.00453, 0.00225, -0.04936,
                        0.0011820796644, 6.3362000446e-07, 3.6496971006e-05,
                        6.3362000446e-07, 0.0011286522495, 3.949522943e-05,
                        3.6496971006e-05, 3.949522943e-05, 0.00019322744629},
/* 12 LAnklePitch */   {0.13416, 0.00045, 0.00029, 0.00685,
                        3.8508129364e-05, -2.6340000403e-08, 3.8619400584e-06,
                        -2.6340000403e-08, 7.4265262811e-05, 1.8339999741e-08,
                        3.8619400584e-06, 1.8339999741e-08, 5.4865398852e-05},
/* 13 LAnkleRoll */    {0.17184, 0.02542, 0.0033, -0.03239,
                        0.00026944180718, -5.6957201195e-06, 0.00013937948097,
                        -5.6957201195e-06, 0.0006443420817, 1.8740920495e-055,
                        0.000139379948097, 1.8740920495e-05, 0.00052575673908},
/* 14 L sole */        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
/* 15 RShoulderPitch */{0.09304, -0.00165, 0.02663, 0.00014,
                        8.4284300101e-05, 2.0280199351e-06, 2.3380000158e-08,
                        2.0280199351e-06, 1.4155610188e-05, 1.9719999e-08,
                        2.3380000158e-08, 1.9719999855e-08, 8.6419488071e-05},
/* 16 RShoulderRoll */ {0.15777, 0.02455, -0.00563, 0.0033,
                        0.00011012030882, 7.6691307186e-05, -2.6046069252e-05,
                        7.6691307186e-05, 0.00036757651833, 1.2098280422e-05,
                        -2.6046069252e-05, 1.2098280422e-05, 0.00035461771768},
/* 17 RElbowYaw */     {0.06483, -0.02744, 0.0, -0.00014,
                        5.5971499933e-06, 4.2099999042e-09, 4.3189999133e-08,
                        4.2099999042e-09, 7.5433119491e-05, -1.8400000412e-09,
                        4.3189999133e-08, -1.8400000412e-09, 7.6443393482e-05},
/* 18 RElbowRoll */    {0.07761, 0.02556, -0.00281, 0.00076,
                        2.5390700102e-05, 2.3324300855e-06, -6.0116997247e-07,
                        2.3324300855e-06, 8.9220360678e-05, 2.6940000453e-08,
                        -6.0116997247e-07, 2.6940000453e-08, 8.7248430646e-05},
/* 19 RWristYaw */     {0.18533, 0.03434, 0.00088, 0.00308,
                        7.0549329e-05, 5.71599e-5, -2.247437e-5,
                        5.71599e-6, 0.00035606, 3.17777e-6,
                        -2.247437e-5, 3.1777e-6, 0.000351},
/* 20 LShoulderPitch */{0.09304, -0.00165, -0.02663, 0.00014,
                        8.4284300101e-05, -2.0280199351e-06, 2.3380000158e-08,
                        -2.0280199351e-06, 1.4155610188e-05, -1.9719999e-08,
                        2.3380000158e-08, -1.9719999855e-08, 8.6419488071e-05},
/* 21 LShoulderRoll */ {0.15777, 0.02455, 0.00563, 0.0033,
                        9.3899929198e-05, -4.7144520067e-05, -2.6994710424e-05,
                        -4.7144520067e-05, 0.00037151877768, -2.4597700303e-06,
                        -2.6994710424e-05, -2.4597700303e-06, 0.00034190082806},
/* 22 LElbowYaw */     {0.06483, -0.02744, 0.0, -0.00014,
                        5.5971499933e-06, 4.2099999042e-09, 4.3189999133e-08,
                        4.2099999042e-09, 7.5433119491e-05, -1.8400000412e-09,
                        4.3189999133e-08, -1.8400000412e-09, 7.6443393482e-05},
/* 23 LElbowRoll */    {0.07761, 0.02556, 0.00281, 0.00076,
                        2.5332199584e-05, -2.3427101041e-06, 7.4589998178e-08,
                        -2.3427101041e-06, 8.91321979e-05, -2.6549999532e-08,
                        7.4589998178e-08, -2.6549999532e-08, 8.7287262431e-05},
/* 24 LWristYaw */     {0.18533, 0.03434, -0.00088, 0.00308,
                        7.0549329e-05, 5.71599e-5, -2.247437e-5,
                        5.71599e-6, 0.00035606, 3.17777e-6,
                        -2.247437e-5, 3.1777e-6, 0.000351},
/* 25 HeadYaw */       {0.07842, -1e-05, 0.0, -0.02742,
                        7.4992953159e-05, 1.5700000189e-09, -1.8339999741e-08,
                        1.5700000189e-09, 7.5999952969e-05, -5.294999994e-08,
                        -1.83399997e-08, -5.294999994e-08, 5.5337300182e-06},
/* 26 HeadPitch */     {0.60533, -0.00112, 0.0, 0.05258,
                        0.0026312952396, 8.788139894e-06, 4.0984661609e-05,
                        8.788139894e-06, 0.0024911249056, -2.995792056e-05,
                        4.0984661609e-05, -2.99579205e-05, 0.00098573567811},
/* 27 extra head */    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
};

void orc_nao_parameters(orc_link links[ORC_NF])
{
    for (int i = 0; i < ORC_NF; i++) {
        links[i].mass = kNao[i][0];
        for (int k = 0; k < 3; k++) links[i].com[k] = kNao[i][1 + k];
        for (int k = 0; k < 9; k++) links[i].inertia[k] = kNao[i][4 + k];
    }
}

/* ------------------------------- generalizedFunctions.cpp ------------------------------- */

void orc_cross_matrix(const double v[3], double A[9])            /* :3-9 */
{
    A[0] = 0;     A[1] = -v[2]; A[2] = v[1];
    A[3] = v[2];  A[4] = 0;     A[5] = -v[0];
    A[6] = -v[1]; A[7] = v[0];  A[8] = 0;
}

void orc_velocity_matrix(const double T[16], double X[36])       /* :11-19 */
{
    double R[9], p[3], cp[9], Rcp[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = T[j * 4 + i];  /* R = T.block.transpose() */
    p[0] = T[3]; p[1] = T[7]; p[2] = T[11];
    orc_cross_matrix(p, cp);
    orc_mm(3, 3, 3, R, cp, Rcp);
    memset(X, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X[i * 6 + j] = R[i * 3 + j];
            X[(3 + i) * 6 + j] = -Rcp[i * 3 + j];
            X[(3 + i) * 6 + 3 + j] = R[i * 3 + j];
        }
}

void orc_inverse_transform(const double T[16], double Ti[16])    /* :21-27 (R^T as inverse) */
{
    memset(Ti, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ti[i * 4 + j] = T[j * 4 + i];
    for (int i = 0; i < 3; i++) {
        double s = 0.0;
        for (int j = 0; j < 3; j++) s += (-T[j * 4 + i]) * T[j * 4 + 3];
        Ti[i * 4 + 3] = s;
    }
    Ti[15] = 1.0;
}

void orc_spatial_cross(const double v[6], double m[36])          /* :29-35 */
{
    double a[9], b[9];
    orc_cross_matrix(v, a);
    orc_cross_matrix(v + 3, b);
    memset(m, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            m[i * 6 + j] = a[i * 3 + j];
            m[(3 + i) * 6 + j] = b[i * 3 + j];
            m[(3 + i) * 6 + 3 + j] = a[i * 3 + j];
        }
}

void orc_spatial_cross_force(const double v[6], double f[36])    /* :37-41 */
{
    double m[36];
    orc_spatial_cross(v, m);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) f[i * 6 + j] = -m[j * 6 + i];
}

void orc_omega_to_euler_rate(const double eta[3], double Om[9])  /* :43-50 */
{
    Om[0] = cos(eta[2]) / cos(eta[1]); Om[1] = sin(eta[2]) / cos(eta[1]); Om[2] = 0;
    Om[3] = -sin(eta[2]);              Om[4] = cos(eta[2]);               Om[5] = 0;
    Om[6] = cos(eta[2]) * tan(eta[1]); Om[7] = sin(eta[2]) * tan(eta[1]); Om[8] = 1;
}

void orc_euler_to_so3(const double rpy[3], double R[9])          /* :52-72 */
{
    const double cr = cos(rpy[0]), sr = sin(rpy[0]);
    const double cp = cos(rpy[1]), sp = sin(rpy[1]);
    const double cy = cos(rpy[2]), sy = sin(rpy[2]);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

void orc_rot_to_axis_angle(const double R[9], double r[3])       /* :74-101 */
{
    double tr = R[0] + R[4] + R[8];
    double c = fmax(-1.0, fmin(1.0, (tr - 1.0) / 2.0));
    double phi = acos(c);
    double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};       /* vee(R - R^T) */
    double s = (phi < 1e-6) ? 0.5 : (phi / (2.0 * sin(phi)));
    for (int i = 0; i < 3; i++) r[i] = s * v[i];
}

double orc_polyval(const double *poly, int n, double x)          /* :165-176 */
{
    double value = 0, xPow = 1;
    for (int i = 0; i < n; i++) { value += poly[i] * xPow; xPow *= x; }
    return value;
}

int orc_polyder(const double *poly, int n, double *out)          /* :178-193 */
{
    if (n <= 1) { out[0] = 0.0; return 1; }
    for (int i = 0; i < n - 1; i++) out[i] = (i + 1) * poly[i + 1];
    return n - 1;
}

void orc_swap_base_velocity(const double X01[36], double v[ORC_NQ]) /* :195-206 */
{
    double t[6], o[6];
    for (int i = 0; i < 3; i++) { t[i] = v[3 + i]; t[3 + i] = v[i]; }
    orc_mv(6, 6, X01, t, o);
    for (int i = 0; i < 6; i++) v[i] = o[i];
}

int orc_find_poly_coeff(int nPos, const double *pos, int nVel, const double *vel,
                        int nAcc, const double *acc, double *coeff)   /* :103-163 */
{
    int n = nPos + nVel + nAcc;
    double A[64], b[8];
    int row = 0;
    for (int i = 0; i < nPos; i++) {
        double tPow = 1;
        for (int j = 0; j < n; j++) { A[row * n + j] = tPow; tPow *= pos[i * 2]; }
        b[row++] = pos[i * 2 + 1];
    }
    for (int i = 0; i < nVel; i++) {
        double tPow = 1;
        A[row * n] = 0;
        for (int j = 1; j < n; j++) { A[row * n + j] = j * tPow; tPow *= vel[i * 2]; }
        b[row++] = vel[i * 2 + 1];
    }
    for (int i = 0; i < nAcc; i++) {
        double tPow = 1;
        A[row * n] = 0; A[row * n + 1] = 0;
        for (int j = 2; j < n; j++) { A[row * n + j] = j * (j - 1) * tPow; tPow *= acc[i * 2]; }
        b[row++] = acc[i * 2 + 1];
    }
    return orc_solve_ge(n, A, b, coeff);  /* reference: colPivHouseholderQr, :161 */
}

/* ------------------------------------- Robot.cpp ------------------------------------- */

void orc_initial_configuration(double q[ORC_NQ])                 /* :242-251 */
{
    static const double q0[ORC_NQ] = {-0.0185, 0, 0.282, 0, 0, 0,
                                      0, 0, -0.5, 0.8, -0.3, 0,
                                      0, 0, -0.5, 0.8, -0.3, 0,
                                      1.6, 0, 0, 0, 0,
                                      -1.6, 0, 0, 0, 0,
                                      0, 0};
    memcpy(q, q0, sizeof(q0));
}

void orc_desired_posture(double q[ORC_NQ]) { orc_initial_configuration(q); } /* :253-262, same literals */

/* Khalil modified-DH link transforms, :176-223 */
static void mat_trans(const double theta[25], double T[25][16])
{
    const double r1 = -0.07071, r7 = 0.07071, r15 = 0.105, r17 = 0.05595, r20 = 0.105, r22 = 0.05595;
    const double d4 = -0.1, d5 = -0.1029, d10 = -0.1, d11 = -0.1029, d15 = -0.015, d20 = -0.015, d25 = 0.030;
    const double h = kPi / 2;
    const double r[25] = {r1, 0, 0, 0, 0, 0, r7, 0, 0, 0, 0, 0, 0, 0, r15, 0, r17, 0, 0, r20, 0, r22, 0, 0, 0};
    const double d[25] = {0, 0, 0, d4, d5, 0, 0, 0, 0, d10, d11, 0, 0, 0, d15, 0, 0, 0, 0, d20, 0, 0, 0, 0, d25};
    const double alpha[25] = {0, h, h, 0, 0, -h, -h, -h, h, 0, 0, -h, -h, h, h, -h, h, h, h, h, -h, h, 0, -h, 0};
    for (int i = 0; i < 25; i++) {
        double ct = cos(theta[i]), st = sin(theta[i]);
        double ca = cos(alpha[i]), sa = sin(alpha[i]);   /* cos(+-pi/2) ~ 6.1e-17, kept */
        double *t = T[i];
        t[0] = ct;      t[1] = -st;     t[2] = 0;   t[3] = d[i];
        t[4] = ca * st; t[5] = ca * ct; t[6] = -sa; t[7] = -r[i] * sa;
        t[8] = sa * st; t[9] = sa * ct; t[10] = ca; t[11] = r[i] * ca;
        t[12] = 0; t[13] = 0; t[14] = 0; t[15] = 1;
    }
}

static void mm4(const double *A, const double *B, double *C) { orc_mm(4, 4, 4, A, B, C); }

static void forward_kinematics(orc_robot *rb)                    /* :45-160 */
{
    const double *q = rb->q;
    double (*T)[16] = rb->T;
    double R0[9];
    orc_euler_to_so3(q + 3, R0);
    memset(T[0], 0, sizeof(T[0]));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[0][i * 4 + j] = R0[i * 3 + j];
        T[0][i * 4 + 3] = q[i];
    }
    T[0][15] = 1;

    double th[25];
    th[0] = q[6];
    th[1] = q[7] + (3.0 / 4) * kPi;
    for (int i = 2; i <= 5; i++) th[i] = q[6 + i];
    th[6] = q[12] - (1.0 / 2) * kPi;
    th[7] = q[13] + (1.0 / 4) * kPi;
    for (int i = 8; i <= 11; i++) th[i] = q[6 + i];
    th[12] = q[18];
    th[13] = q[19] + (1.0 / 2) * kPi;
    for (int i = 14; i <= 16; i++) th[i] = q[6 + i];
    th[17] = q[23];
    th[18] = q[24] + (1.0 / 2) * kPi;
    for (int i = 19; i <= 21; i++) th[i] = q[6 + i];
    th[22] = q[28];
    th[23] = q[29] - (1.0 / 2) * kPi;
    th[24] = -kPi / 2;

    double Temp[25][16];
    mat_trans(th, Temp);

    static const double auxT01[16] = {0, -1, 0, 0, 0.7071, 0, 0.7071, 0, -0.7071, 0, 0.7071, 0, 0, 0, 0, 1};
    static const double auxT09[16] = {1, 0, 0, 0, 0, 0.7071, 0.7071, 0, 0, -0.7071, 0.7071, 0, 0, 0, 0, 1};
    static const double auxFoot[16] = {1, 0, 0, -0.0452, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double tmp[16];

    mm4(T[0], auxT01, tmp); mm4(tmp, Temp[0], T[1]);                     /* :120 */
    for (int i = 1; i < 6; i++) mm4(T[i], Temp[i], T[i + 1]);            /* :122-124 */
    mm4(T[6], auxFoot, T[7]);                                            /* :125 */
    mm4(T[0], auxT09, tmp); mm4(tmp, Temp[6], T[8]);                     /* :127 */
    for (int i = 8; i < 13; i++) mm4(T[i], Temp[i - 1], T[i + 1]);       /* :128-130 */
    mm4(T[13], auxFoot, T[14]);                                          /* :131 */

    static const double shoulderR[3] = {0, -0.098, 0.13591};
    static const double shoulderL[3] = {0, 0.098, 0.13591};
    static const double headOff[3] = {0, 0, 0.1615};
    memcpy(tmp, Temp[12], sizeof(tmp));
    for (int k = 0; k < 3; k++) tmp[k * 4 + 3] = tmp[k * 4 + 3] + shoulderR[k];
    mm4(T[0], tmp, T[15]);                                               /* :135-137 */
    for (int i = 15; i < 19; i++) mm4(T[i], Temp[i - 2], T[i + 1]);
    memcpy(tmp, Temp[17], sizeof(tmp));
    for (int k = 0; k < 3; k++) tmp[k * 4 + 3] = tmp[k * 4 + 3] + shoulderL[k];
    mm4(T[0], tmp, T[20]);                                               /* :144-146 */
    for (int i = 20; i < 24; i++) mm4(T[i], Temp[i - 2], T[i + 1]);
    memcpy(tmp, Temp[22], sizeof(tmp));
    for (int k = 0; k < 3; k++) tmp[k * 4 + 3] = tmp[k * 4 + 3] + headOff[k];
    mm4(T[0], tmp, T[25]);                                               /* :153-155 */
    for (int i = 25; i < 27; i++) mm4(T[i], Temp[i - 2], T[i + 1]);
}

static void compute_com(orc_robot *rb)                           /* :225-238 */
{
    double com[3] = {0, 0, 0};
    for (int i = 0; i < ORC_NF; i++) {
        const double *T = rb->T[i];
        const double *c = rb->links[i].com;
        for (int k = 0; k < 3; k++) {
            double p = T[k * 4] * c[0] + T[k * 4 + 1] * c[1] + T[k * 4 + 2] * c[2] + T[k * 4 + 3] * 1.0;
            com[k] = com[k] + rb->links[i].mass * p;
        }
    }
    for (int k = 0; k < 3; k++) rb->CoM[k] = com[k] / rb->mass;
}

static void all_velocity_matrices(orc_robot *rb)                 /* :276-298 */
{
    double piTi[16], Tinv[16];
    orc_velocity_matrix(rb->T[0], rb->X[0]);
    for (int i = 1; i < ORC_NF; i++) {
        orc_inverse_transform(rb->T[orc_parent[i]], Tinv);
        mm4(Tinv, rb->T[i], piTi);
        orc_velocity_matrix(piTi, rb->X[i]);
    }
}

void orc_robot_update_state(orc_robot *rb, const double *q)      /* :264-269 */
{
    memcpy(rb->q, q, sizeof(rb->q));
    forward_kinematics(rb);
    compute_com(rb);
    all_velocity_matrices(rb);
}

void orc_robot_update_velocity(orc_robot *rb, const double *v, const double *AG) /* :271-274, 300-310 */
{
    double vh[ORC_NQ], h[6];
    memcpy(rb->v, v, sizeof(rb->v));
    memcpy(vh, v, sizeof(vh));
    orc_swap_base_velocity(rb->X[0], vh);
    orc_mv(6, ORC_NQ, AG, vh, h);
    for (int k = 0; k < 3; k++) {
        rb->comVel[k] = h[3 + k] / rb->mass;
        rb->comAngMom[k] = h[k];
    }
}

void orc_robot_init(orc_robot *rb, const orc_link *raw)          /* :5-43 */
{
    double q0[ORC_NQ];
    memset(rb, 0, sizeof(*rb));
    if (raw) memcpy(rb->links, raw, sizeof(rb->links));
    else orc_nao_parameters(rb->links);
    memset(q0, 0, sizeof(q0));
    rb->mass = 1.0;               /* reference reads an uninitialised mass_ here; CoM is recomputed below */
    orc_robot_update_state(rb, q0);
    /* re-express Aldebaran (world-aligned at q=0) inertial data in the joint frames, :14-22 */
    rb->mass = 0;
    for (int i = 0; i < ORC_NF; i++) {
        double Rj[9], Rt[9], c[3], t1[9], t2[9];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) { Rj[a * 3 + b] = rb->T[i][a * 4 + b]; Rt[b * 3 + a] = Rj[a * 3 + b]; }
        orc_mv(3, 3, Rt, rb->links[i].com, c);
        memcpy(rb->links[i].com, c, sizeof(c));
        orc_mm(3, 3, 3, Rt, rb->links[i].inertia, t1);
        orc_mm(3, 3, 3, t1, Rj, t2);
        memcpy(rb->links[i].inertia, t2, sizeof(t2));
        rb->mass += rb->links[i].mass;
    }
    orc_initial_configuration(q0);
    orc_robot_update_state(rb, q0);
    memset(rb->v, 0, sizeof(rb->v));
}
