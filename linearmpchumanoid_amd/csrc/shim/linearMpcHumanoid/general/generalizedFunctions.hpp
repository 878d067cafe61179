#pragma once
#include <Eigen/Dense>
#include <cmath>
#include <iostream>
// host-side utilities the caller (apps/offline/main.cpp:111,116) uses directly
Eigen::Matrix3d crossMatrix(Eigen::Vector3d v);                       // generalizedFunctions.cpp:3-9
Eigen::Matrix3d matrixAngularVelToEulerDot(Eigen::Vector3d eta);      // generalizedFunctions.cpp:43-50
Eigen::VectorXd findPolyCoeff(const std::vector<std::pair<double, double>> &pos,
                              const std::vector<std::pair<double, double>> &vel,
                              const std::vector<std::pair<double, double>> &acc);   // :103-163 (rows = (t, value))
double polyval(const Eigen::VectorXd &poly, double x);                // :165-176
