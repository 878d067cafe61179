#pragma once
#include <Eigen/Dense>
// LIPM preview controller parameters + the last references (reference controller/mpcLinearPendulum.hpp:5-31).
// The preview QP itself is evaluated inside the GPU controller kernel.
class Mpc3dLip {
public:
    Mpc3dLip() {}
    Mpc3dLip(const double dt, const double timeHorizon, const double zCoM) : dt_(dt), timeHorizon_(timeHorizon), zCom_(zCoM) {}
    Mpc3dLip(const double dt, const double timeHorizon, const double zCoM, const double alpha, const double beta)
        : dt_(dt), timeHorizon_(timeHorizon), zCom_(zCoM), alpha_(alpha), beta_(beta) {}
    Eigen::Vector3d getXRef() const { return xRef_; }
    Eigen::Vector3d getYRef() const { return yRef_; }
    double getZCom() const { return zCom_; }
    double getDt() const { return dt_; }
    double getTimeHorizon() const { return timeHorizon_; }
    double getAlpha() const { return alpha_; }
    double getBeta() const { return beta_; }
    void setRefs(const double *x3, const double *y3) { for (int i = 0; i < 3; i++) { xRef_(i) = x3[i]; yRef_(i) = y3[i]; } }
private:
    Eigen::Vector3d xRef_, yRef_;
    double dt_ = 0.01, timeHorizon_ = 0.5, zCom_ = 0.26, alpha_ = 1e-3, beta_ = 1;
};
