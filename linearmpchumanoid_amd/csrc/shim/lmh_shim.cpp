// lmh_shim.cpp -- the reference's C++ class surface (Robot, Kinematics, ZMP, Mpc3dLip, Controller,
// footCoeffTrajectory, ...) as thin wrappers over the C ABI of include/lmh.h with B = 1, so that the
// reference's apps/offline/main.cpp builds and runs UNCHANGED against the MI355X kernels.
// Ownership / side effects follow the reference: Controller keeps references to the caller's Robot and
// Mpc3dLip and mutates them in standStep; ZMP and the coefficient vectors are copied (here: uploaded).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <utility>
#include <vector>
#include "linearMpcHumanoid/controller/controller.hpp"
#include "lmh.h"

static void die(const char *what)
{
    // the reference aborts on unusable data (controller.cpp:448-466); same policy for device errors
    std::cerr << what << ": " << lmh_last_error() << std::endl;
    std::abort();
}

// ------------------------------------------------------------------ utilities (host)
Eigen::Matrix3d crossMatrix(Eigen::Vector3d v)
{
    Eigen::Matrix3d A;
    A << 0, -v(2), v(1), v(2), 0, -v(0), -v(1), v(0), 0;
    return A;
}
Eigen::Matrix3d matrixAngularVelToEulerDot(Eigen::Vector3d eta)
{
    Eigen::Matrix3d O;
    O << std::cos(eta(2)) / std::cos(eta(1)), std::sin(eta(2)) / std::cos(eta(1)), 0,
        -std::sin(eta(2)), std::cos(eta(2)), 0,
        std::cos(eta(2)) * std::tan(eta(1)), std::sin(eta(2)) * std::tan(eta(1)), 1;
    return O;
}
double polyval(const Eigen::VectorXd &poly, double x)
{
    double value = 0, xPow = 1;
    for (int i = 0; i < poly.size(); i++) { value += poly(i) * xPow; xPow *= x; }
    return value;
}
Eigen::VectorXd findPolyCoeff(const std::vector<std::pair<double, double>> &pos, const std::vector<std::pair<double, double>> &vel,
                              const std::vector<std::pair<double, double>> &acc)
{
    const int n = static_cast<int>(pos.size() + vel.size() + acc.size());
    std::vector<double> A(static_cast<size_t>(n) * n, 0.0), b(static_cast<size_t>(n), 0.0);
    int row = 0;
    for (auto &p : pos) { double tp = 1; for (int j = 0; j < n; j++) { A[row * n + j] = tp; tp *= p.first; } b[row++] = p.second; }
    for (auto &p : vel) { double tp = 1; for (int j = 1; j < n; j++) { A[row * n + j] = j * tp; tp *= p.first; } b[row++] = p.second; }
    for (auto &p : acc) { double tp = 1; for (int j = 2; j < n; j++) { A[row * n + j] = j * (j - 1) * tp; tp *= p.first; } b[row++] = p.second; }
    for (int c = 0; c < n; c++) {                                   // Gaussian elimination, partial pivoting
        int pv = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(A[r * n + c]) > std::fabs(A[pv * n + c])) pv = r;
        for (int j = 0; j < n; j++) std::swap(A[c * n + j], A[pv * n + j]);
        std::swap(b[c], b[pv]);
        for (int r = c + 1; r < n; r++) {
            const double f = A[r * n + c] / A[c * n + c];
            for (int j = c; j < n; j++) A[r * n + j] -= f * A[c * n + j];
            b[r] -= f * b[c];
        }
    }
    Eigen::VectorXd x(n);
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int j = r + 1; j < n; j++) s -= A[r * n + j] * x(j);
        x(r) = s / A[r * n + r];
    }
    return x;
}
std::vector<Eigen::VectorXd> footCoeffTrajectory(const Eigen::Vector3d &cur, const Eigen::Vector3d &des, double stepHeight, double T)
{
    std::vector<Eigen::VectorXd> C(3);
    const std::vector<std::pair<double, double>> zero2 = {{0, 0}, {T, 0}};
    for (int ax = 0; ax < 2; ax++) C[ax] = findPolyCoeff({{0, cur(ax)}, {T, des(ax)}}, zero2, zero2);
    C[2] = findPolyCoeff({{0, cur(2)}, {T / 2, stepHeight}, {T, des(2)}}, {{0, 0}, {T / 2, 0}, {T, 0}}, zero2);
    return C;
}

// ------------------------------------------------------------------ ZMP
ZMP::ZMP(const Task task) : task_(task) { stanceZMP(); }
ZMP::ZMP(const Task, const double simulationTime, const double timeStep, const SupportFoot supportFoot)
    : simulationTime_(simulationTime), timeStep_(timeStep), supportFoot_(supportFoot) { stanceZMP(); }
ZMP::ZMP(const Task task, const int numSteps, const double timePerStep, const double simulationTime)
    : simulationTime_(simulationTime), task_(task), numSteps_(numSteps), timePerStep_(timePerStep) {}
void ZMP::stanceZMP()
{
    const int samples = static_cast<int>((simulationTime_ + 0.5) / timeStep_);
    zmpXRef_.resize(samples);
    zmpYRef_.resize(samples);
    const double y = (supportFoot_ == SupportFoot::Right) ? -0.05 : (supportFoot_ == SupportFoot::Left) ? 0.05 : 0.0;
    for (int i = 0; i < samples; i++) zmpYRef_(i) = y;
}

// ------------------------------------------------------------------ Robot
Eigen::VectorXd initialConfiguration()
{
    Eigen::VectorXd q = Eigen::VectorXd::Zero(30);
    const double v[30] = {-0.0185, 0, 0.282, 0, 0, 0, 0, 0, -0.5, 0.8, -0.3, 0, 0, 0, -0.5, 0.8, -0.3, 0,
                          1.6, 0, 0, 0, 0, -1.6, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 30; i++) q(i) = v[i];
    return q;
}
static lmh_handle *scratch_handle()
{
    static lmh_handle *h = nullptr;                                  // B = 1 handle for set-up computations (IK, CoM)
    if (!h) {
        lmh_config cfg;
        lmh_config_default(&cfg);
        if (lmh_create(&cfg, 1, 0, &h) != LMH_OK) die("lmh_create");
    }
    return h;
}
Robot::Robot()
{
    q_ = initialConfiguration();
    v_ = Eigen::VectorXd::Zero(30);
    if (lmh_get_mass(scratch_handle(), &mass_) != LMH_OK) die("lmh_get_mass");
    updateState(q_);
}
void Robot::updateState(const Eigen::VectorXd &q_new)
{
    q_ = q_new;
    double com[3];
    if (lmh_robot_com_host(scratch_handle(), q_.data(), com) != LMH_OK) die("Robot::updateState");
    for (int i = 0; i < 3; i++) CoM_(i) = com[i];
}
void Robot::setFromDevice(const double *q, const double *v, const double *com, const double *comVel)
{
    for (int i = 0; i < 30; i++) { if (q) q_(i) = q[i]; if (v) v_(i) = v[i]; }
    for (int i = 0; i < 3; i++) { if (com) CoM_(i) = com[i]; if (comVel) comVel_(i) = comVel[i]; }
}

// ------------------------------------------------------------------ Kinematics
Eigen::VectorXd Kinematics::desiredOperationalState(const Robot &robot, const Eigen::VectorXd &Rf, const Eigen::VectorXd &Lf, const Eigen::Vector3d &com)
{
    Eigen::VectorXd Qd = Eigen::VectorXd::Zero(robot.getNumJoints());
    const Eigen::VectorXd &q = robot.getJoints();
    for (int i = 0; i < 6; i++) { Qd(i) = Rf(i); Qd(6 + i) = Lf(i); }
    for (int i = 0; i < 12; i++) Qd(12 + i) = q(18 + i);
    for (int i = 0; i < 3; i++) { Qd(24 + i) = 0.0; Qd(27 + i) = com(i); }
    return Qd;
}
void Kinematics::compute(Robot &robot, const Eigen::VectorXd &desOp)
{
    Eigen::VectorXd q = robot.getJoints();
    double rf[6], lf[6], ct[3], com[3];
    for (int i = 0; i < 6; i++) { rf[i] = desOp(i); lf[i] = desOp(6 + i); }
    for (int i = 0; i < 3; i++) ct[i] = desOp(27 + i);
    int32_t iters = 0;
    if (lmh_ik_host(scratch_handle(), q.data(), ct, rf, lf, com, &iters) != LMH_OK) die("Kinematics::compute");
    iters_ = iters;
    if (iters >= 200) std::cout << "Inv Kinematics no solution founded" << std::endl;
    robot.setFromDevice(q.data(), nullptr, com, nullptr);
}

// ------------------------------------------------------------------ Controller
Controller::Controller(Robot &robot, Mpc3dLip &mpc, ZMP &zmp, std::vector<Eigen::VectorXd> &rFCoeff, std::vector<Eigen::VectorXd> &lFCoeff)
    : robot_(robot), mpc_(mpc)
{
    std::cout << "Controller Initiated" << std::endl;
    std::cout << "Initial conditions of the center of mass: " << "c = " << robot_.getCoM()(0);
    std::cout << ", " << robot_.getCoM()(1) << ", " << robot_.getCoM()(2) << std::endl << std::endl;
    lmh_config cfg;
    lmh_config_default(&cfg);
    // Mpc3dLip's dt is the MPC sample time; the Clock's step never reaches Controller (the caller's rk4Step integrates on the host), so
    // the control step of the handle is only a placeholder here and equals it
    cfg.dt = mpc.getDt(); cfg.mpc_dt = mpc.getDt(); cfg.time_horizon = mpc.getTimeHorizon(); cfg.z_com = mpc.getZCom();
    cfg.alpha = mpc.getAlpha(); cfg.beta = mpc.getBeta();
    cfg.warm_start = 0;                                             // reference: cold start every call (controller.cpp:467)
    if (lmh_create(&cfg, 1, 0, &h_) != LMH_OK) die("lmh_create");
    const Eigen::VectorXd zx = zmp.getZmpXRef(), zy = zmp.getZmpYRef();
    if (lmh_set_refs(h_, zx.data(), zy.data(), nullptr, zx.size()) != LMH_OK) die("lmh_set_refs");
    double r[24] = {0}, l[24] = {0};
    int32_t rn[3], ln[3];
    for (int a = 0; a < 3; a++) {
        rn[a] = rFCoeff[static_cast<size_t>(a)].size(); ln[a] = lFCoeff[static_cast<size_t>(a)].size();
        if (rn[a] > 8 || ln[a] > 8) die("foot polynomial with more than 8 coefficients");      // the staging rows below hold 8
        for (int k = 0; k < rn[a]; k++) r[8 * a + k] = rFCoeff[static_cast<size_t>(a)](k);
        for (int k = 0; k < ln[a]; k++) l[8 * a + k] = lFCoeff[static_cast<size_t>(a)](k);
    }
    if (lmh_set_foot_coeffs(h_, r, rn, l, ln) != LMH_OK) die("lmh_set_foot_coeffs");
    // Robot::v_ as the caller's Robot holds it (zeros after construction)
    if (lmh_set_prev_velocity_host(h_, robot_.getJointsVelocity().data()) != LMH_OK) die("lmh_set_prev_velocity_host");
    tau_ = Eigen::VectorXd::Zero(robot_.getNumJoints());
    last_.qpp = Eigen::VectorXd::Zero(30); last_.tau = Eigen::VectorXd::Zero(24); last_.f = Eigen::VectorXd::Zero(12);
}
Controller::~Controller() { lmh_destroy(h_); }

void Controller::standStep(const ControllerInput &in)
{
    int32_t status[LMH_STATUS_STRIDE];
    if (lmh_eval_host(h_, in.q.data(), in.dq.data(), in.time, last_.tau.data(), last_.f.data(), last_.qpp.data(), status) != LMH_OK) die("lmh_eval_host");
    double out[LMH_OUT_STRIDE];
    lmh_last_out_host(h_, out);
    k_ = status[0]; flags_ = status[2];
    if (flags_ & LMH_FLAG_NONFINITE) { std::cerr << "WBC output has NaN or Inf" << std::endl; std::abort(); }   // controller.cpp:448-466
    if (flags_ & LMH_FLAG_QP_MAXITER) std::cerr << "QP failed, status = " << flags_ << std::endl;               // controller.cpp:472-476
    robot_.setFromDevice(in.q.data(), in.dq.data(), out + 66, out + 69);    // the caller's Robot is mutated (controller.cpp:53,59)
    mpc_.setRefs(out + 72, out + 75);
    tau_ = last_.tau;
}
WBCOutput Controller::WBC(double) { return last_; }

