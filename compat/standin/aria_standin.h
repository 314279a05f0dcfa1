// aria_standin.h -- TEST DOUBLE, not MobileRobots ARIA.  What odometry/kalmanfilter.cpp:17-20
// touches: a lockable robot that reports translational (mm/s) and rotational (deg/s) velocity.
// Where "Aria.h" exists, compat/kalmanfilter.h uses the real ArRobot instead.
// Below the two classes the replay driver feeds, the NAMES slam.cpp:54-106,208,216 needs to be type-checked against compat/ (connection,
// key handler, range devices, motors): declarations that do nothing, so that tests/test_compat.py can run the compiler's front end
// over the reference's unchanged slam.cpp with compat/kalmanfilter.h and compat/featuredetector.h in the reference headers' places.
// Nothing of this is linked into anything that runs.
#pragma once
#include <mutex>

class ArKeyHandler {};
class ArRangeDevice {};
class ArSonarDevice : public ArRangeDevice {};
namespace ArCommands { enum Commands { SOUNDTOG = 92 }; }

class ArRobot {
public:
    // slam.cpp:71,76-77,86,97,105-106,120,208 (no-ops)
    void attachKeyHandler(ArKeyHandler *) {}
    void addRangeDevice(ArRangeDevice *) {}
    void runAsync(bool) {}
    bool disconnect() { return true; }
    void enableMotors() {}
    bool comInt(unsigned char, short int) { return true; }
    void requestEncoderPackets() {}
    void waitForRunExit() {}
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

class ArSick : public ArRangeDevice {
public:
    // slam.cpp:90,92,95 (no-ops)
    enum BaudRate { BAUD9600, BAUD19200, BAUD38400 };
    enum Degrees { DEGREES180, DEGREES100 };
    enum Increment { INCREMENT_ONE, INCREMENT_HALF };
    void configureShort(bool, BaudRate = BAUD38400, Degrees = DEGREES180, Increment = INCREMENT_ONE) {}
    void runAsync() {}
    bool blockingConnect() { return true; }
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

// ---- slam.cpp:54-92: start-up names (no-ops) ---------------------------------------------------------------------------
class Aria {
public:
    static void init() {}
    static void shutdown() {}
    static void exit(int) {}  // slam.cpp:216
    static void setKeyHandler(ArKeyHandler *) {}
};

class ArArgumentParser {
public:
    ArArgumentParser(int *, char **) {}
    void loadDefaultArguments() {}
};

class ArSimpleConnector {
public:
    ArSimpleConnector(int *, char **) {}
    bool parseArgs() { return true; }
    void logOptions() const {}
    bool connectRobot(ArRobot *) { return true; }
    bool setupLaser(ArSick *) { return true; }
};
