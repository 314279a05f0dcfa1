// compat/kalmanfilter.h -- header-compatible replacement for the reference's
// odometry/kalmanfilter.h:21-43.  Same class name, same public members (X, Y, Phi, Num_Landmarks),
// same three methods with the same signatures; the state lives in HBM behind the C ABI of
// include/ekfslam_c.h (libekfslam_hip.so).  slam.cpp:127,136,146,170 compile against it unchanged.
//
// What stays on the host, exactly where the reference has it:
//   - the ARIA velocity reads under robot->lock() and their unit conversions (kalmanfilter.cpp:17-26)
//   - the two per-step file writes (kalmanfilter.cpp:51,56-59), same formats, same index quirk
//   - the public mirrors refreshed after every call (kalmanfilter.cpp:46-48,85-89)
// What the reference does not have: an error channel.  A failing C-ABI call throws std::runtime_error (a full map is not a failure:
// doUpdate grows the capacity, as the reference's state grows).
#ifndef KALMANFILTER_H
#define KALMANFILTER_H

#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#else
#include "standin/eigen_standin.h"
#endif
#if __has_include("Aria.h")
#include "Aria.h"
#else
#include "standin/aria_standin.h"
#endif

#include "../include/ekfslam_c.h"

#define INF 999999999999 /* kalmanfilter.h:17 */
#define PI 3.141592653589793238462643383279502884197169399375105820974944592307816406286 /* kalmanfilter.h:18 (unused by the filter itself, which writes 3.141592654) */

class KalmanFilter {
public:
    double X = 0.0;
    double Y = 0.0;
    double Phi = 0.0;
    int Num_Landmarks = 0;
    // not in the reference: the reference's only diagnostics -- "New " / "Old " / "Ignore " on stdout, one token per measurement
    // (Update.cpp:154,183,191; slam.cpp:169-171 prints "Update: " in front and Num_Landmarks behind) -- can be switched off
    bool Print_Decisions = true;

    // capacity_landmarks / device_id are additions with defaults: `new KalmanFilter(&robot)` still works
    explicit KalmanFilter(ArRobot *robot, int capacity_landmarks = 4096, int device_id = 0) : robot(robot) {
        // This class synchronises after every call (the public mirrors must be current, kalmanfilter.cpp:46-48,85-89).  Its calls travel as
        // commands to a resident "streaming" launch (round 6, ekfslam_c.h "Tunables": EKF_STREAM), so what a call costs is one trip to the
        // device and back plus the operation itself; what is left to choose is where the window's dense pass runs.  In place (overlap = 0)
        // the call that follows a full window waits for the pass: nothing at N = 1024 (15 us), 100 us at N = 4096.  From 2048 landmarks on
        // the pass therefore runs beside the next window's calls (overlap = 1, a second P_LL buffer): per 5-call step at N = 4096 median
        // 75 us / p90 88 us against 70 / 163 in place; at N = 1024 in place is the faster one (67 / 77 against 65 / 85); N = 50: 50 / 70.
        // (Before round 6, one launch per call: 89 / 103 / 110 us median at N = 50 / 1024 / 4096, p90 185 us at N = 4096.)
        ekf_params params;
        ekf_default_params(&params);
        params.overlap = capacity_landmarks >= 2048 ? 1 : 0;
        check(ekf_create(&h, capacity_landmarks, device_id, &params));  // x = 0_3, P = 0_3x3: kalmanfilter.cpp:10-11
    }
    ~KalmanFilter() { ekf_destroy(h); }
    KalmanFilter(const KalmanFilter &) = delete;
    KalmanFilter &operator=(const KalmanFilter &) = delete;

    void doPropagation(double dt, std::ofstream &covFile, std::ofstream &knownfeaturesFile) {
        robot->lock();  // kalmanfilter.cpp:17-20
        double V = robot->getVel();
        double RTV = robot->getRotVel() * 3.141592654 / 180.0;
        robot->unlock();
        double v = V / 1000.0;  // :26
        double w = RTV;
        check(ekf_propagate(h, v, w, dt));  // Q = (v*v) diag(0.01, 0.04)^2 and Propagate, :28-44
        mirror();
        double Prr[9];
        check(ekf_get_robot_cov(h, Prr));
        covFile << Prr[0] << " " << Prr[1] << " " << Prr[3] << " " << Prr[4] << std::endl;  // :51
        if (Num_Landmarks > 0 && knownfeaturesFile.good()) {  // :53-60, including its stride-1 indexing (a stream that cannot take the lines is not worth the O(N) copy of x)
            xbuf.resize(3 + 2 * (size_t)Num_Landmarks);
            check(ekf_get_x(h, 0, xbuf.data(), (int)xbuf.size()));
            // (same lines as the reference; one flush per call instead of its std::endl per line)
            for (int i = 1; i < Num_Landmarks; i++) knownfeaturesFile << xbuf[3 + i] << " " << xbuf[4 + i] << "\n";
            knownfeaturesFile.flush();
        }
    }

    void doUpdate(Eigen::MatrixXd z_chunk, Eigen::MatrixXd R_chunk) {
        int n_z = (int)(z_chunk.size() / 2);  // Update.cpp:27
        // The reference's state grows with every New landmark and never runs out (Update.cpp:158-177, kalmanfilter.cpp:78-84); the
        // device buffers are sized by a capacity, so make room BEFORE a chunk that could exceed it (each measurement adds at most
        // one landmark): the capacity doubles, the state moves over on the device (ekf_reserve), nothing is dropped, nothing throws
        const int cap = ekf_capacity(h);
        if (Num_Landmarks + n_z > cap) check(ekf_reserve(h, grown_capacity(cap, Num_Landmarks + n_z)));
        decisions.resize(n_z);
        // z_chunk.data() / R_chunk.data() are column-major, which is what the C ABI takes
        check(ekf_update(h, z_chunk.data(), R_chunk.data(), n_z, decisions.data()));  // Gamma 50 / 10: kalmanfilter.cpp:67-68
        if (Print_Decisions)  // Update.cpp:154,183,191: the same tokens, in measurement order, no newline, no flush
            for (const ekf_decision &d : decisions)
                std::cout << (d.decision == EKF_DECISION_NEW ? "New " : d.decision == EKF_DECISION_OLD ? "Old " : "Ignore ");
        mirror();
    }

    void doUpdateCompass(double z, double R) {
        check(ekf_update_compass(h, z, R));
        mirror();
    }

    // The capacity asked for when `need` landmarks no longer fit `cap`: double, but never beyond what the library can hold
    // (EKF_MAX_CAPACITY, ekfslam_c.h) -- only a map that really needs more than that fails (ekf_reserve then says so).
    static int grown_capacity(int cap, int need) {
        int want = 2 * cap > need ? 2 * cap : need;
        if (want > EKF_MAX_CAPACITY && need <= EKF_MAX_CAPACITY) want = EKF_MAX_CAPACITY;
        return want;
    }

    // not in the reference: the gate decisions of the last doUpdate ("New " / "Old " / "Ignore ", Update.cpp:154,183,191)
    const std::vector<ekf_decision> &lastDecisions() const { return decisions; }
    ekf_handle handle() const { return h; }

private:
    ArRobot *robot;
    ekf_handle h = nullptr;
    std::vector<ekf_decision> decisions;
    std::vector<double> xbuf;

    void mirror() {  // kalmanfilter.cpp:46-48, 85-89
        double pose[3];
        check(ekf_get_pose(h, pose));
        X = pose[0], Y = pose[1], Phi = pose[2];
        int n = ekf_num_landmarks(h);
        check(n);
        Num_Landmarks = n;
    }
    static void check(int rc) {
        if (rc < 0) throw std::runtime_error(std::string("libekfslam_hip: ") + ekf_last_error());
    }
};

#endif  // KALMANFILTER_H
