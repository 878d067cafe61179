#pragma once
// Fixed-step clock; t accumulates by repeated addition (reference Clock.hpp:3-18) -- the float
// accumulation is what k = int(t/dt) sees, so it must not be replaced by tick*dt.
class Clock {
public:
    Clock(double dt, double simulationTime) : dt_(dt), T_(simulationTime) {}
    double getTime() const { return t_; }
    double getTimeStep() const { return dt_; }
    double getSimulationTime() const { return T_; }
    void step() { t_ += dt_; }
private:
    double t_ = 0;
    double dt_;
    double T_;
};
