#pragma once
#include <Eigen/Dense>
#include <vector>
#include "linearMpcHumanoid/controller/mpcLinearPendulum.hpp"
#include "linearMpcHumanoid/controller/invKinematics.hpp"
#include "linearMpcHumanoid/robotInfo/Robot.hpp"
#include "linearMpcHumanoid/trajectories/zmpGeneration.hpp"
#include "linearMpcHumanoid/trajectories/footRefTrajectory.hpp"
#include "linearMpcHumanoid/general/Task.hpp"
#include "linearMpcHumanoid/general/Clock.hpp"
#include "linearMpcHumanoid/general/generalizedFunctions.hpp"
#include "linearMpcHumanoid/general/rk4.hpp"
// Same call surface as the reference's controller/controller.hpp:33-63 (which re-exports the headers
// above transitively and, unlike this one, pulls in <qpOASES.hpp>).
struct ControllerInput { Eigen::VectorXd q; Eigen::VectorXd dq; double time; };
struct ControllerOutput { Eigen::VectorXd tau; };
struct WBCOutput { Eigen::VectorXd qpp; Eigen::VectorXd tau; Eigen::VectorXd f; };
struct lmh_handle;
class Controller {
public:
    Controller(Robot &robot, Mpc3dLip &mpc, ZMP &zmp, std::vector<Eigen::VectorXd> &rFCoeff, std::vector<Eigen::VectorXd> &lFCoeff);
    ~Controller();
    Controller(const Controller &) = delete;
    Controller &operator=(const Controller &) = delete;
    void standStep(const ControllerInput &in);      // controller.cpp:48-79 -> one lmh_eval on the GPU
    WBCOutput WBC(double t);                        // controller.cpp:81-154: pure function of the state standStep left
    const Eigen::VectorXd &getTorques() { return tau_; }
    int lastStatusFlags() const { return flags_; }
    int lastPreviewIndex() const { return k_; }
private:
    Eigen::VectorXd tau_;
    Robot &robot_;
    Mpc3dLip &mpc_;
    lmh_handle *h_ = nullptr;
    WBCOutput last_;
    int flags_ = 0, k_ = 0;
};
