// apps/offline_stand.cpp -- this repository's own equivalent of the reference's apps/offline workload
// (build robot, IK to CoM (-0.02, 0, 0.26), stand for T seconds under RK4 on the controller's own
// acceleration, print CoM x after every tick), written against the reference's class surface as
// provided by linearmpchumanoid_amd/csrc/shim.  The reference's own apps/offline/main.cpp compiles
// against the same headers unchanged (tests/test_shim.py does that where /root/reference exists);
// this file exists because the reference source does not travel to the GPU box.
// usage: offline_stand [T=5] [dt=0.01] [horizon_s=0.5]
#include <cstdlib>
#include "linearMpcHumanoid/robotInfo/Robot.hpp"
#include "linearMpcHumanoid/controller/controller.hpp"
#include "linearMpcHumanoid/controller/invKinematics.hpp"
#include "linearMpcHumanoid/controller/mpcLinearPendulum.hpp"
#include "linearMpcHumanoid/general/Clock.hpp"
#include "linearMpcHumanoid/general/Task.hpp"

static Eigen::VectorXd plant(const Eigen::VectorXd &state, double t, Robot &robot, Controller &controller)
{
    const int n = robot.getNumJoints();
    Eigen::VectorXd q = state.segment(0, n), qD = state.segment(n, n);
    ControllerInput in;
    in.q = state.segment(0, n);
    in.dq = state.segment(n, n);
    in.time = t;
    controller.standStep(in);
    WBCOutput out = controller.WBC(t);
    qD.segment(0, 3) += crossMatrix(qD.segment(3, 3)) * q.segment(0, 3);
    qD.segment(3, 3) = matrixAngularVelToEulerDot(q.segment(3, 3)) * qD.segment(3, 3);
    Eigen::VectorXd xp(2 * n);
    xp.head(n) = qD;
    xp.tail(n) = out.qpp;
    return xp;
}

int main(int argc, char **argv)
{
    const double simulationTime = argc > 1 ? std::atof(argv[1]) : 5.0;
    const double timeStep = argc > 2 ? std::atof(argv[2]) : 0.01;
    const double timeHorizon = argc > 3 ? std::atof(argv[3]) : 0.5;
    Robot nao;
    Kinematics ik;
    Clock clock(timeStep, simulationTime);
    ZMP zmp(Task::Stand, simulationTime, timeStep, SupportFoot::Double);
    Eigen::VectorXd Rf = Eigen::VectorXd::Zero(6), Lf = Eigen::VectorXd::Zero(6);
    Rf(1) = -0.05; Lf(1) = 0.05;
    Eigen::Vector3d com;
    com << -0.02, 0.0, 0.26;
    ik.compute(nao, ik.desiredOperationalState(nao, Rf, Lf, com));
    Mpc3dLip mpc(clock.getTimeStep(), timeHorizon, nao.getCoM()(2));
    Eigen::Vector3d pr, pl;
    pr << 0, -0.05, 0; pl << 0, 0.05, 0;
    std::vector<Eigen::VectorXd> rF = footCoeffTrajectory(pr, pr, 0, simulationTime), lF = footCoeffTrajectory(pl, pl, 0, simulationTime);
    Controller controller(nao, mpc, zmp, rF, lF);
    const int n = nao.getNumJoints();
    Eigen::VectorXd state(2 * n);
    state.segment(0, n) = nao.getJoints();
    state.segment(n, n) = nao.getJointsVelocity();
    std::cout.precision(15);
    while (std::abs(clock.getTime() - clock.getSimulationTime()) > 0.01) {
        state = rk4Step([&](const Eigen::VectorXd &x, double t) { return plant(x, t, nao, controller); }, state, clock.getTime(), clock.getTimeStep());
        std::cout << nao.getCoM()(0) << std::endl;
        clock.step();
    }
    return 0;
}
