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

// ---- the laser side, what features/featuredetector.cpp:18-22 and slam.cpp:186-196 touch --------------------------------
#include <vector>

class ArSensorReading {
public:
    ArSensorReading(unsigned int range_mm = 0, double local_x = 0, double local_y = 0) : range_(range_mm), lx_(local_x), ly_(local_y) {}
    unsigned int getRange() const { return range_; }
    double getLocalX() const { return lx_; }
    double getLocalY() const { return ly_; }

private:
    unsigned int range_;
    double lx_, ly_;
};

class ArTime {
public:
    explicit ArTime(long long stamp = -1) : stamp_(stamp) {}
    bool isAt(ArTime other) const { return stamp_ == other.stamp_; }

private:
    long long stamp_;
};

class ArSick {
public:
    void lockDevice() { m_.lock(); }
    void unlockDevice() { m_.unlock(); }
    std::vector<ArSensorReading> *getRawReadingsAsVector() { return &readings_; }
    ArTime getLastReadingTime() const { return ArTime(stamp_); }
    void setScan(const std::vector<ArSensorReading> &r) {  // what the replay driver feeds: a new sweep has arrived
        readings_ = r;
        stamp_++;
    }

private:
    std::mutex m_;
    std::vector<ArSensorReading> readings_;
    long long stamp_ = 0;
};
