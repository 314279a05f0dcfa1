// compat/replay.cpp -- headless stand-in for the reference's main loop (slam.cpp:130-204) so that the
// whole path can run without a Pioneer robot: it reads one record per loop iteration
//     dt  vel_mm_s  rotvel_deg_s  compass|nan  n  fx_mm fy_mm ... (n corner features, robot frame, mm)
// and drives the header-compatible KalmanFilter exactly as slam.cpp does: doPropagation, optional
// doUpdateCompass(compass, 0.0005) (:144-147), then one doUpdate per feature with
// z = (fx, fy)/1000 and R = G diag(0.0025, 0.0001) G^T (:152-170).  It writes the reference's output
// files in the reference's formats (slam.cpp:177,181; kalmanfilter.cpp:51,56-59).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "kalmanfilter.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <records.txt> <output-dir> [capacity_landmarks]\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[1]);
    if (!in) return std::fprintf(stderr, "cannot open %s\n", argv[1]), 2;
    std::string dir = argv[2];
    int cap = argc > 3 ? std::atoi(argv[3]) : 1024;
    std::ofstream odomFile(dir + "/odomRun.txt"), featuresFile(dir + "/featuresRun.txt"), covFile(dir + "/covRun.txt"),
        knownfeaturesFile(dir + "/knownfeaturesRun.txt"), decisionFile(dir + "/decisionsRun.txt");
    odomFile.precision(17), featuresFile.precision(17), covFile.precision(17), knownfeaturesFile.precision(17), decisionFile.precision(17);

    ArRobot robot;
    try {
        KalmanFilter *ekf = new KalmanFilter(&robot, cap);  // slam.cpp:127
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            double dt, vel, rot;
            std::string comp;
            int n;
            if (!(ls >> dt >> vel >> rot >> comp >> n)) continue;
            robot.setVelocities(vel, rot);
            ekf->doPropagation(dt, covFile, knownfeaturesFile);  // slam.cpp:136
            if (comp != "nan") ekf->doUpdateCompass(std::stod(comp), 0.0005);  // :144-147
            for (int i = 0; i < n; i++) {
                double fxmm, fymm;
                ls >> fxmm >> fymm;
                Eigen::MatrixXd z_chunk(2, 1), R(2, 2), R_chunk(2, 2), G(2, 2);
                z_chunk(0, 0) = fxmm / 1000.0, z_chunk(1, 0) = fymm / 1000.0;  // :157
                double fx = fxmm / 1000.0, fy = fymm / 1000.0;
                double dist = std::sqrt(fx * fx + fy * fy);
                double bearing = std::atan2(fy, fx);
                R(0, 0) = 0.0025, R(0, 1) = 0, R(1, 0) = 0, R(1, 1) = 0.0001;  // :165
                G(0, 0) = std::cos(bearing), G(0, 1) = -dist * std::sin(bearing), G(1, 0) = std::sin(bearing), G(1, 1) = dist * std::cos(bearing);
                for (int r = 0; r < 2; r++)  // R_chunk = G * R * G^T, :167
                    for (int c = 0; c < 2; c++) {
                        double s = 0;
                        for (int a = 0; a < 2; a++) {
                            double gr = G(r, 0) * R(0, a) + G(r, 1) * R(1, a);
                            s += gr * G(c, a);
                        }
                        R_chunk(r, c) = s;
                    }
                ekf->doUpdate(z_chunk, R_chunk);  // :170
                const ekf_decision &d = ekf->lastDecisions()[0];
                decisionFile << d.decision << " " << d.matched << " " << d.mahal << std::endl;
                double newX = fx * std::cos(ekf->Phi) - fy * std::sin(ekf->Phi);  // :173-177
                double newY = fx * std::sin(ekf->Phi) + fy * std::cos(ekf->Phi);
                featuresFile << newX + ekf->X << " " << newY + ekf->Y << std::endl;
            }
            odomFile << ekf->X << " " << ekf->Y << std::endl;  // :181
        }
        std::printf("final %.17g %.17g %.17g %d\n", ekf->X, ekf->Y, ekf->Phi, ekf->Num_Landmarks);
        delete ekf;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
