// compat/replay.cpp -- headless stand-in for the reference's main loop (slam.cpp:130-204) so that the
// whole path can run without a Pioneer robot.  Input: a text file with one record per loop iteration
//     dt  vel_mm_s  rotvel_deg_s  compass|nan  n  fx_mm fy_mm ... (n corner features, robot frame, mm)
// optionally interleaved with lines that stand for what the SICK's own thread delivers meanwhile:
//     scan  k  range_mm lx_mm ly_mm ...     (the k raw readings sick.getRawReadingsAsVector() would return
//                                            from now on: range, local x, local y; slam.cpp:186-196)
// It drives the header-compatible KalmanFilter exactly as slam.cpp does: doPropagation (:136), optional
// doUpdateCompass(compass, 0.0005) (:144-147), one doUpdate per feature with z = (fx, fy)/1000 and
// R = G diag(0.0025, 0.0001) G^T (:152-170), the feature / odometry lines (:173-181), and, whenever more
// than a second of dt has accumulated, the scan dump of :184-203 (readings beyond 7000 mm skipped).
// Output: the reference's files in the reference's places and formats under <output-dir>:
//     data/odom/odomRun.txt  data/features/featuresRun.txt  data/features/knownfeaturesRun.txt
//     data/scan/scanRun.txt  data/cov/covRun.txt                  (slam.cpp:21-50; kalmanfilter.cpp:51,56-59)
// so that plot.py (run from <output-dir>) and mapping/RealTimePlotting.m (run from <output-dir>/mapping) read
// them as they read the reference's; plus data/decisionsRun.txt (the "New "/"Old "/"Ignore " stdout tokens of
// Update.cpp:154,183,191 as numbers).
// Options (not in the reference): --state FILE starts from an injected state instead of x = 0_3, P = 0
// (binary doubles: n, x[n], P[n*n]); --dump-state FILE writes the final state in the same format (through
// ekf_get_state); --timing prints per-iteration wall-clock statistics of the filter calls (and leaves
// knownfeaturesRun.txt empty: its O(N) text lines per iteration would otherwise be what is timed); --detect takes the
// corner features and the structural compass from the header-compatible FeatureDetector (compat/featuredetector.h, Hough
// and corner extraction on the GPU) run on the `scan` lines, exactly as slam.cpp:138-147 does, instead of from the record
// (every `scan` line is then a NEW sweep: featuredetector.cpp:31-34 skips iterations without one).
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "featuredetector.h"
#include "kalmanfilter.h"

struct Reading {
    double range, lx, ly;  // ArSensorReading::getRange / getLocalX / getLocalY, mm
};

static bool read_state(const char *path, std::vector<double> &x, std::vector<double> &P) {
    std::ifstream f(path, std::ios::binary);
    double nd = 0;
    if (!f.read((char *)&nd, sizeof nd)) return false;
    size_t n = (size_t)nd;
    x.resize(n), P.resize(n * n);
    return (bool)f.read((char *)x.data(), n * sizeof(double)) && (bool)f.read((char *)P.data(), n * n * sizeof(double));
}

int main(int argc, char **argv) {
    std::vector<std::string> pos;
    const char *state_in = nullptr, *state_out = nullptr;
    bool timing = false, detect = false, quiet = false;  // --quiet: without slam.cpp's stdout chatter ("Compass: ...", "Update: New 12")
    // digits of the data files.  The reference's streams are never given a precision (slam.cpp:177,181,200; kalmanfilter.cpp:51,58), so it
    // writes the default 6 significant digits: that is the default here too -- a replay's files compare byte for byte with a reference run's on
    // the same values.  --precision 17 writes round-trip digits (the parity tests compare the files value for value).
    int precision = 6;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--state") && i + 1 < argc) state_in = argv[++i];
        else if (!std::strcmp(argv[i], "--dump-state") && i + 1 < argc) state_out = argv[++i];
        else if (!std::strcmp(argv[i], "--timing")) timing = true;
        else if (!std::strcmp(argv[i], "--detect")) detect = true;
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else if (!std::strcmp(argv[i], "--precision") && i + 1 < argc) precision = std::atoi(argv[++i]);
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 2) {
        std::fprintf(stderr, "usage: %s <records.txt> <output-dir> [capacity_landmarks] [--state f] [--dump-state f] [--timing] [--detect] [--quiet] [--precision digits (default 6, as the reference's streams)]\n", argv[0]);
        return 2;
    }
    std::ifstream in(pos[0]);
    if (!in) return std::fprintf(stderr, "cannot open %s\n", pos[0].c_str()), 2;
    const std::string dir = pos[1];
    const int cap = pos.size() > 2 ? std::atoi(pos[2].c_str()) : 1024;
    // the reference's Makefile:5 creates ./data, ./data/features, ./data/odom, ./data/scan (and forgets ./data/cov)
    for (const char *d : {"/data", "/data/odom", "/data/features", "/data/scan", "/data/cov", "/maps"}) mkdir((dir + d).c_str(), 0777);
    std::ofstream odomFile(dir + "/data/odom/odomRun.txt"), scanFile(dir + "/data/scan/scanRun.txt"),
        featuresFile(dir + "/data/features/featuresRun.txt"), knownfeaturesFile(dir + "/data/features/knownfeaturesRun.txt"),
        covFile(dir + "/data/cov/covRun.txt"), decisionFile(dir + "/data/decisionsRun.txt"), compassFile(dir + "/data/compassRun.txt");  // slam.cpp:21-50
    if (!odomFile || !scanFile || !featuresFile || !knownfeaturesFile || !covFile) return std::fprintf(stderr, "cannot create the data files under %s\n", dir.c_str()), 2;
    for (std::ofstream *f : {&odomFile, &scanFile, &featuresFile, &knownfeaturesFile, &covFile, &decisionFile, &compassFile}) f->precision(precision > 0 ? precision : 6);
    // --timing measures the filter calls: the O(N) text lines of knownfeaturesRun.txt (kalmanfilter.cpp:56-59) would be what
    // is timed at large N, so that file stays empty in a timing run (a failed stream ignores its insertions)
    if (timing) knownfeaturesFile.setstate(std::ios::badbit);

    ArRobot robot;
    ArSick sick;
    try {
        FeatureDetector *f = detect ? new FeatureDetector(&sick) : nullptr;  // slam.cpp:110
        KalmanFilter *ekf = new KalmanFilter(&robot, cap);  // slam.cpp:127
        if (quiet || timing) ekf->Print_Decisions = false;
        const bool chatter = !(quiet || timing);
        if (state_in) {
            std::vector<double> x, P;
            if (!read_state(state_in, x, P)) return std::fprintf(stderr, "cannot read %s\n", state_in), 2;
            if (ekf_set_state(ekf->handle(), 0, x.data(), P.data(), (int)x.size(), (int)x.size()) < 0) throw std::runtime_error(ekf_last_error());
        }
        std::vector<Reading> readings;  // what the laser's thread holds at the moment
        std::vector<double> step_us;
        double loopTime = 0.0;          // slam.cpp:115,184
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            if (line.compare(0, 4, "scan") == 0) {
                std::string tag;
                int k = 0;
                ls >> tag >> k;
                readings.assign((size_t)std::max(k, 0), Reading{0, 0, 0});
                for (auto &r : readings) ls >> r.range >> r.lx >> r.ly;
                if (detect) {
                    std::vector<ArSensorReading> sr;
                    for (const Reading &r : readings) sr.push_back(ArSensorReading((unsigned int)r.range, r.lx, r.ly));
                    sick.setScan(sr);
                }
                continue;
            }
            double dt, vel, rot;
            std::string comp;
            int n;
            if (!(ls >> dt >> vel >> rot >> comp >> n)) continue;
            robot.setVelocities(vel, rot);
            auto t0 = std::chrono::steady_clock::now();
            ekf->doPropagation(dt, covFile, knownfeaturesFile);  // slam.cpp:136
            std::vector<Feature> fvec;  // :139-141
            if (detect) {
                double compass;
                f->getFeatures(&fvec, &compass, ekf->Phi);
                if (compass != f->NO_COMPASS) {  // :144-147
                    if (chatter) std::cout << "Compass: " << compass << std::endl;  // :145
                    compassFile << compass << "\n";
                    ekf->doUpdateCompass(compass, 0.0005);
                }
                n = (int)fvec.size();
            } else if (comp != "nan") {
                if (chatter) std::cout << "Compass: " << std::stod(comp) << std::endl;  // :145
                ekf->doUpdateCompass(std::stod(comp), 0.0005);
            }
            for (int i = 0; i < n; i++) {
                double fxmm, fymm;
                if (detect) fxmm = fvec[i].x, fymm = fvec[i].y;
                else ls >> fxmm >> fymm;
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
                if (chatter) std::cout << "Update: ";  // :169
                ekf->doUpdate(z_chunk, R_chunk);  // :170 (prints "New " / "Old " / "Ignore " as Update.cpp:154,183,191 do)
                if (chatter) std::cout << ekf->Num_Landmarks << std::endl;  // :171
                const ekf_decision &d = ekf->lastDecisions()[0];
                decisionFile << d.decision << " " << d.matched << " " << d.mahal << "\n";
                double newX = fx * std::cos(ekf->Phi) - fy * std::sin(ekf->Phi);  // :173-177
                double newY = fx * std::sin(ekf->Phi) + fy * std::cos(ekf->Phi);
                featuresFile << newX + ekf->X << " " << newY + ekf->Y << std::endl;
            }
            step_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            odomFile << ekf->X << " " << ekf->Y << std::endl;  // :181
            loopTime += dt;                                    // :184
            if (loopTime > 1.0) {                              // :185-203
                for (const Reading &r : readings) {
                    if (r.range > 7000) continue;              // :191
                    double fx = r.lx / 1000.0, fy = r.ly / 1000.0;
                    double newX = fx * std::cos(ekf->Phi) - fy * std::sin(ekf->Phi);
                    double newY = fx * std::sin(ekf->Phi) + fy * std::cos(ekf->Phi);
                    scanFile << newX + ekf->X << " " << newY + ekf->Y << std::endl;  // :200
                }
                loopTime = 0.0;
            }
        }
        std::cout.flush();
        std::printf("final %.17g %.17g %.17g %d\n", ekf->X, ekf->Y, ekf->Phi, ekf->Num_Landmarks);
        if (timing && !step_us.empty()) {
            std::vector<double> s(step_us.begin() + std::min<size_t>(step_us.size() / 5, 10), step_us.end());  // skip the warm-up iterations
            std::sort(s.begin(), s.end());
            std::printf("timing iterations %zu median_us %.1f p90_us %.1f max_us %.1f\n", s.size(), s[s.size() / 2], s[(s.size() * 9) / 10], s.back());
            long long starts = 0, ops = 0;  // (how the calls travelled: streaming launches started, operations posted to them; 0 0 = one launch per call)
            const int streams = ekf_debug_stream(ekf->handle(), &starts, &ops);
            std::printf("streaming %d launches %lld operations %lld\n", streams, starts, ops);
        }
        if (state_out) {
            int n = ekf_get_state(ekf->handle(), 0, nullptr, nullptr, 0);
            if (n < 0) throw std::runtime_error(ekf_last_error());
            std::vector<double> x((size_t)n), P((size_t)n * n);
            if (ekf_get_state(ekf->handle(), 0, x.data(), P.data(), n) < 0) throw std::runtime_error(ekf_last_error());
            std::ofstream f(state_out, std::ios::binary);
            double nd = n;
            f.write((const char *)&nd, sizeof nd);
            f.write((const char *)x.data(), x.size() * sizeof(double));
            f.write((const char *)P.data(), P.size() * sizeof(double));
        }
        delete ekf;
        delete f;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
