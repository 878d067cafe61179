#pragma once
#include <Eigen/Dense>
#include <cmath>
#include "linearMpcHumanoid/general/generalizedFunctions.hpp"
#define NUM_JOINTS 30
#define NUM_ACTUAL_JOINTS 24
#define NUM_FRAMES 28
#define NUM_BODIES 25
// Host-side mirror of the reference's Robot (robotInfo/Robot.hpp:16-73): it only HOLDS the state
// the caller reads (q, v, CoM); every kinematic / dynamic quantity is evaluated on the GPU behind
// include/lmh.h.  As in the reference, Controller::standStep mutates the caller's Robot.
class Robot {
public:
    Robot();
    int getNumFrames() const { return NUM_FRAMES; }
    int getNumJoints() const { return NUM_JOINTS; }
    int getNumActualJoints() const { return NUM_ACTUAL_JOINTS; }
    int getNumBodies() const { return NUM_BODIES; }
    const Eigen::VectorXd &getJoints() const { return q_; }
    const Eigen::VectorXd &getJointsVelocity() const { return v_; }
    const Eigen::Vector3d &getCoM() const { return CoM_; }
    const Eigen::Vector3d &getComVel() const { return comVel_; }
    double getMass() const { return mass_; }
    void updateState(const Eigen::VectorXd &q_new);                 // Robot.cpp:264-269 (FK + CoM on the GPU)
    // used by the shim's Controller / Kinematics
    void setFromDevice(const double *q, const double *v, const double *com, const double *comVel);
private:
    Eigen::VectorXd q_, v_;
    Eigen::Vector3d CoM_, comVel_;
    double mass_ = 0;
};
Eigen::VectorXd initialConfiguration();                             // Robot.cpp:242-251
