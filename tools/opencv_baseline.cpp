// opencv_baseline — the SECONDARY CPU baseline of BASELINE.md §3.2 / SURVEY.md §8d: the per-frame pipeline of the reference
// (src/vo.cpp:41-137) driven through the SAME seven cv:: functions with the reference's parameters — cv::FAST
// (feature_set.cpp:60-61), cv::buildOpticalFlowPyramid (vo.cpp:50,52,200,201), cv::calcOpticalFlowPyrLK x4 (vo.cpp:203-215),
// cv::triangulatePoints + cv::convertPointsFromHomogeneous (vo.cpp:89-94), cv::solvePnPRansac + cv::Rodrigues
// (vo.cpp:287-308) — as own code (nothing of the reference is compiled or copied).  Built ONLY when bench.py finds OpenCV 4 on
// the box that runs it (it never is in this image; nothing is ever installed); then it gives (a) a CPU frame-pairs/s number
// that IS OpenCV, with default threads and with one, and (b) per-frame poses of the same frames the oracle and the HIP path
// process — the first bit-level check against the real dependency.
//
//   opencv_baseline frames.raw W H N fx cx cy bf win max_level ransac_iters max_translation threads poses_out.txt
//     frames.raw: N stereo pairs, left then right, W*H bytes each (gray)
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/video/tracking.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

namespace {

struct Tracks { std::vector<cv::Point2f> pts; std::vector<int> age, strength; };

// one feature per grid cell: highest score wins, the earliest of equals stays (feature_set.cpp:20-53 with capacity 1);
// score = age + (strength - 20) / 20 in integers; tracks of age >= 20 are dropped; the top 4 grid rows hold nothing
void keep_best_per_cell(Tracks& t, int W, int H) {
    const int rows = 92, cols = 160, first_row = 4, max_age = 20, fast_thr = 20;
    const int ch = (H + rows - 1) / rows, cw = (W + cols - 1) / cols;
    std::vector<int> winner(rows * cols, -1), best(rows * cols, 0);
    for (size_t i = 0; i < t.pts.size(); i++) {
        if (t.age[i] >= max_age) continue;
        const int r = (int)(t.pts[i].y / ch), c = (int)(t.pts[i].x / cw);
        if (r < first_row || r >= rows || c < 0 || c >= cols) continue;
        const int score = t.age[i] + (t.strength[i] - fast_thr) / 20, cell = r * cols + c;
        if (winner[cell] < 0 || score > best[cell]) { winner[cell] = (int)i; best[cell] = score; }
    }
    Tracks out;
    for (int cell = 0; cell < rows * cols; cell++)
        if (winner[cell] >= 0) { out.pts.push_back(t.pts[winner[cell]]); out.age.push_back(t.age[winner[cell]]); out.strength.push_back(t.strength[winner[cell]]); }
    t = out;
}

void detect_into(Tracks& t, const cv::Mat& img, int threshold) {
    std::vector<cv::KeyPoint> kps;
    cv::FAST(img, kps, threshold, true);
    for (const auto& k : kps) { t.pts.push_back(k.pt); t.age.push_back(0); t.strength.push_back((int)k.response); }
    keep_best_per_cell(t, img.cols, img.rows);
}

template <class T> void keep_flagged(std::vector<T>& v, const std::vector<uchar>& ok) {
    size_t m = 0;
    for (size_t i = 0; i < v.size(); i++) if (ok[i]) v[m++] = v[i];
    v.resize(m);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 15) { std::fprintf(stderr, "usage: see the header of tools/opencv_baseline.cpp\n"); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
    const float fx = (float)std::atof(argv[5]), cx = (float)std::atof(argv[6]), cy = (float)std::atof(argv[7]), bf = (float)std::atof(argv[8]);
    const int win = std::atoi(argv[9]), max_level = std::atoi(argv[10]), ransac_iters = std::atoi(argv[11]);
    const double max_translation = std::atof(argv[12]);
    const int threads = std::atoi(argv[13]);
    if (threads > 0) cv::setNumThreads(threads);
    std::vector<uchar> raw((size_t)W * H * 2 * N);
    { std::ifstream f(argv[1], std::ios::binary); f.read((char*)raw.data(), (std::streamsize)raw.size()); if (!f) { std::fprintf(stderr, "short read\n"); return 2; } }
    auto frame = [&](int k, int cam) { return cv::Mat(H, W, CV_8UC1, raw.data() + ((size_t)k * 2 + cam) * W * H); };

    cv::Mat_<float> Pl = (cv::Mat_<float>(3, 4) << fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0);
    cv::Mat_<float> Pr = Pl.clone(); Pr(0, 3) = bf;
    cv::Mat_<float> K = Pl(cv::Rect(0, 0, 3, 3)).clone();
    const cv::Size lk_win(win, win);
    const cv::TermCriteria crit(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30, 0.0001);

    Tracks feats;
    std::vector<cv::Mat> pyrL0, pyrR0;
    cv::Mat R = cv::Mat::eye(3, 3, CV_64F), t = cv::Mat::zeros(3, 1, CV_64F);
    cv::Mat_<double> last = cv::Mat_<double>::eye(4, 4);
    std::ofstream poses(argv[14]);
    poses.precision(17);
    int n_ok = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < N; k++) {
        const cv::Mat L1 = frame(k, 0), R1 = frame(k, 1);
        bool ok = false;
        if (k == 0) {
            cv::buildOpticalFlowPyramid(L1, pyrL0, lk_win, max_level);
            cv::buildOpticalFlowPyramid(R1, pyrR0, lk_win, max_level);
        } else {
            const cv::Mat L0 = frame(k - 1, 0);
            detect_into(feats, L0, 20);
            if (feats.pts.size() < 100) detect_into(feats, L0, 5);
            std::vector<cv::Point2f> pL0 = feats.pts, pL1, pR1, pR0, back;
            if (!pL0.empty()) {
                std::vector<cv::Mat> pyrL1, pyrR1;
                cv::buildOpticalFlowPyramid(L1, pyrL1, lk_win, max_level);
                cv::buildOpticalFlowPyramid(R1, pyrR1, lk_win, max_level);
                std::vector<uchar> s0, s1, s2, s3; std::vector<float> err;
                cv::calcOpticalFlowPyrLK(pyrL0, pyrL1, pL0, pL1, s0, err, lk_win, max_level, crit, 0, 0.001);
                cv::calcOpticalFlowPyrLK(pyrL1, pyrR1, pL1, pR1, s1, err, lk_win, max_level, crit, 0, 0.001);
                cv::calcOpticalFlowPyrLK(pyrR1, pyrR0, pR1, pR0, s2, err, lk_win, max_level, crit, 0, 0.001);
                cv::calcOpticalFlowPyrLK(pyrR0, pyrL0, pR0, back, s3, err, lk_win, max_level, crit, 0, 0.001);
                std::vector<uchar> keep(pL0.size());
                for (size_t i = 0; i < pL0.size(); i++) {
                    const float off = std::max(std::fabs(pL0[i].x - back[i].x), std::fabs(pL0[i].y - back[i].y));
                    bool in = true;
                    for (const cv::Point2f* p : {&pL0[i], &pL1[i], &pR1[i], &pR0[i]}) in = in && !(p->x < 0 || p->y < 0 || p->y >= H || p->x >= W);
                    keep[i] = s0[i] && s1[i] && s2[i] && s3[i] && !(off > 0.15f) && in;
                }
                keep_flagged(pL0, keep); keep_flagged(pL1, keep); keep_flagged(pR1, keep); keep_flagged(pR0, keep);
                keep_flagged(feats.pts, keep); keep_flagged(feats.age, keep); keep_flagged(feats.strength, keep);
                pyrL0 = pyrL1; pyrR0 = pyrR1;
            }
            for (auto& a : feats.age) a++;
            if (pL0.size() > 15) {
                cv::Mat X4, X3;
                cv::triangulatePoints(Pl, Pr, pL0, pR0, X4);
                cv::convertPointsFromHomogeneous(X4.t(), X3);
                cv::Mat rvec, inliers, tvec = t.clone();
                cv::Rodrigues(R, rvec);
                const bool found = cv::solvePnPRansac(X3, pL1, K, cv::Mat::zeros(4, 1, CV_64F), rvec, tvec, true, ransac_iters, 8.0f, 0.98, inliers, cv::SOLVEPNP_ITERATIVE);
                if (found && inliers.rows >= 15) {
                    cv::Rodrigues(rvec, R); t = tvec;
                    std::vector<uchar> is_in(pL1.size(), 0);
                    for (int i = 0; i < inliers.rows; i++) is_in[inliers.at<int>(i)] = 1;
                    feats.pts = pL1;
                    keep_flagged(feats.pts, is_in); keep_flagged(feats.age, is_in); keep_flagged(feats.strength, is_in);
                    cv::Mat ang; cv::Rodrigues(R, ang);
                    if (cv::norm(t) <= max_translation && cv::norm(ang) <= 0.5) {
                        cv::Mat_<double> T = cv::Mat_<double>::eye(4, 4);
                        cv::Mat Rt = R.t(); cv::Mat c = -Rt * t;
                        Rt.copyTo(T(cv::Rect(0, 0, 3, 3))); c.copyTo(T(cv::Rect(3, 0, 1, 3)));
                        last = T; ok = true;
                    }
                }
            }
        }
        n_ok += ok;
        poses << k << ' ' << (int)ok;
        for (int i = 0; i < 16; i++) poses << ' ' << last(i / 4, i % 4);
        poses << '\n';
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%.2f frame-pairs/s over %d pairs (%d poses ok), cv::getNumThreads() = %d, OpenCV %s\n", (N - 1) / secs, N - 1, n_ok, cv::getNumThreads(), CV_VERSION);
    return 0;
}
