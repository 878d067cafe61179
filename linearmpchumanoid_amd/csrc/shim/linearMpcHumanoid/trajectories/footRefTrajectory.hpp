#pragma once
#include <vector>
#include <Eigen/Dense>
// x, y: 6 coefficients; z: 8 coefficients, ascending powers (reference src/footRefTrajectory.cpp:4-47)
std::vector<Eigen::VectorXd> footCoeffTrajectory(const Eigen::Vector3d &currentPos, const Eigen::Vector3d &desPos,
                                                 double stepHeight, double T);
