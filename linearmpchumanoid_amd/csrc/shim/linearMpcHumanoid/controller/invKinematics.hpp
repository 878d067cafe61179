#pragma once
#include <Eigen/Dense>
#include "linearMpcHumanoid/robotInfo/Robot.hpp"
// Newton IK of the initial posture (reference src/invKinematics.cpp:11-52), solved on the GPU.
class Kinematics {
public:
    Kinematics() = default;
    Eigen::VectorXd desiredOperationalState(const Robot &robot, const Eigen::VectorXd &Rf, const Eigen::VectorXd &Lf,
                                            const Eigen::Vector3d &com);
    void compute(Robot &robot, const Eigen::VectorXd &desOp);
    int lastIterations() const { return iters_; }
private:
    int iters_ = 0;
};
