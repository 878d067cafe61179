#pragma once
#include <Eigen/Dense>
#include "linearMpcHumanoid/general/Task.hpp"
// ZMP reference arrays (reference trajectories/zmpGeneration.hpp, src/zmpGeneration.cpp:4-60)
class ZMP {
public:
    ZMP(const Task task);
    ZMP(const Task task, const double simulationTime, const double timeStep, const SupportFoot supportFoot);
    ZMP(const Task task, const int numSteps, const double timePerStep, const double simulationTime);
    void stanceZMP();
    const Eigen::VectorXd getZmpXRef() const { return zmpXRef_; }
    const Eigen::VectorXd getZmpYRef() const { return zmpYRef_; }
private:
    Eigen::VectorXd zmpXRef_, zmpYRef_;
    double simulationTime_ = 1;
    double timeStep_ = 0.01;
    SupportFoot supportFoot_ = SupportFoot::Double;
    Task task_ = Task::Stand;
    int numSteps_ = 1;
    double timePerStep_ = 0.5;
};
