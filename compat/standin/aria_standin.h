// aria_standin.h -- TEST DOUBLE, not MobileRobots ARIA.  Only what odometry/kalmanfilter.cpp:17-20
// touches: a lockable robot that reports translational (mm/s) and rotational (deg/s) velocity.
// Where "Aria.h" exists, compat/kalmanfilter.h uses the real ArRobot instead.
#pragma once
#include <mutex>

class ArRobot {
public:
    void lock() { m_.lock(); }
    void unlock() { m_.unlock(); }
    double getVel() const { return vel_mm_s_; }
    double getRotVel() const { return rotvel_deg_s_; }
    void setVelocities(double vel_mm_s, double rotvel_deg_s) {  // what the replay driver feeds
        vel_mm_s_ = vel_mm_s;
        rotvel_deg_s_ = rotvel_deg_s;
    }

private:
    std::mutex m_;
    double vel_mm_s_ = 0.0, rotvel_deg_s_ = 0.0;
};
