// visual_odometry.hpp — header-only C++ facade over the C-ABI (include/svo.h) that mirrors the
// reference's C++ surface, namespace visual_odometry (reference include/vo.h:46-472): same class and
// function names, same argument order and meaning, same failure behaviour.  Only the OpenCV types
// are substituted: cv::Mat (8-bit image) -> Image, cv::Point2f -> Point2f, cv::Mat_<float> 3x4 ->
// Mat34f, cv::Mat_<double> 4x4 -> Mat44, cv::Mat inliers (Nx1 int32) -> std::vector<int>.
// Link with -lsvo_hip.  All arithmetic runs on the GPU; errors surface as std::runtime_error with
// svo_last_error() (the reference would surface cv::Exception).
#pragma once
#include <array>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../svo.h"

namespace visual_odometry {

// the reference's constants (include/vo.h:53-127)
const int BUCKET_START_ROW = 4;
const int BUCKETS_ALONG_HEIGHT = 92;
const int BUCKETS_ALONG_WIDTH = 160;
const int FEATURES_PER_BUCKET = 1;
const int FEATURES_THRESHOLD = 15;
const int PRE_MATCHING_FEATURE_THRESHOLD = 100;
const int AGE_THRESHOLD = 20;
const int FAST_THRESHOLD = 20;
const float RANSAC_REPROJECTION_ERROR = 8;
const int RANSAC_ITERATIONS = 100;
const double OPTICAL_FLOW_MIN_EIG_THRESHOLD = 0.001;
const double CIRCULAR_MATCHING_SUCCESS_THRESHOLD = .15;
const double MAX_TRANSLATION_NORM = .1;
const double MAX_ROTATION_NORM = .5;

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
using Mat34f = std::array<float, 12>;     // 3x4 row-major
using Mat33f = std::array<float, 9>;
using Mat33d = std::array<double, 9>;
using Vec3d = std::array<double, 3>;
using Mat44 = std::array<double, 16>;     // 4x4 row-major

// 8-bit single-channel image view (what cv_bridge MONO8 delivers, reference src/stereo_vo.cpp:9)
struct Image {                                                        // the part of cv::Mat the path uses: 8-bit, 1 or 3 channels
    const uint8_t* data = nullptr;
    int rows = 0, cols = 0, step = 0;                                 // step in bytes
    int channels = 1;                                                 // 3 = interleaved BGR, as cv::imread returns it (main.cpp:38-39)
    Image() {}
    Image(const uint8_t* d, int r, int c, int s = 0, int cn = 1) : data(d), rows(r), cols(c), step(s ? s : c * cn), channels(cn) {}
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
};

inline void svo_throw(int rc) { if (rc < 0) throw std::runtime_error(std::string("libsvo_hip: ") + svo_last_error()); }
inline int& default_device() { static int d = 0; return d; }

class FeatureSet {                                                   // include/vo.h:132-188
   public:
    std::vector<Point2f> points;
    std::vector<int> ages;
    std::vector<int> strengths;
    int size() { return (int)points.size(); }
    void clear() { points.clear(); ages.clear(); strengths.clear(); }
    void filterByBucketLocationInternal(const Image& image, const int buckets_along_height, const int buckets_along_width,
                                        const int bucket_start_row, const int features_per_bucket) {
        int n = size();
        svo_throw(svo_bucket_filter(default_device(), image.cols, image.rows, &n, n ? &points[0].x : nullptr,
                                    n ? ages.data() : nullptr, n ? strengths.data() : nullptr, buckets_along_height,
                                    buckets_along_width, bucket_start_row, features_per_bucket, AGE_THRESHOLD, FAST_THRESHOLD));
        points.resize(n); ages.resize(n); strengths.resize(n);
    }
    void filterByBucketLocation(const Image& image) {
        filterByBucketLocationInternal(image, BUCKETS_ALONG_HEIGHT, BUCKETS_ALONG_WIDTH, BUCKET_START_ROW, FEATURES_PER_BUCKET);
    }
    void appendFeaturesFromImage(const Image& image, const int fast_threshold) {
        int n = size();
        int cap = (BUCKETS_ALONG_HEIGHT - BUCKET_START_ROW) * BUCKETS_ALONG_WIDTH;
        if (cap < n) cap = n;
        points.resize(cap); ages.resize(cap); strengths.resize(cap);
        svo_throw(svo_append_features_from_image(default_device(), nullptr, image.data, image.cols, image.rows, image.step,
                                                 fast_threshold, cap, &n, &points[0].x, ages.data(), strengths.data()));
        points.resize(n); ages.resize(n); strengths.resize(n);
    }
};

class Bucket {                                                       // include/vo.h:195-229
   public:
    int max_size;
    FeatureSet features;
    Bucket(int max_size_) : max_size(max_size_) {}
    int compute_score(const int age, const int strength) { return age + (strength - FAST_THRESHOLD) / 20; }
    void add_feature(const Point2f point, const int age, const int strength) {
        // the insertion rule (feature_set.cpp:20-53) runs on the GPU: the bucket's current content, in slot order, followed by
        // the new feature goes through a 1x1 grid of this capacity — the stored features refill their slots in the same order
        // (they passed the age test when they entered), then the new one meets exactly the state the reference's bucket is in.
        // One call with at most max_size + 1 inputs per insertion (not a replay of everything ever offered).
        if (!max_size) return;
        features.points.push_back(point); features.ages.push_back(age); features.strengths.push_back(strength);
        float mx = 1;
        for (auto& p : features.points) { if (p.x > mx) mx = p.x; if (p.y > mx) mx = p.y; }
        Image dims(reinterpret_cast<const uint8_t*>(this), (int)mx + 2, (int)mx + 2);   // only rows/cols are read
        features.filterByBucketLocationInternal(dims, 1, 1, 0, max_size);
    }
    int size() { return features.size(); }
};

inline std::vector<Point2f> featureDetectionFast(const Image image, const int fast_threshold,
                                                 std::vector<float>& response_strengths) {          // vo.h:393-395
    int cap = 8192, n = 0;
    std::vector<Point2f> pts;
    for (;;) {
        pts.resize(cap); response_strengths.resize(cap);
        svo_throw(svo_fast_detect(default_device(), image.data, image.cols, image.rows, image.step, fast_threshold, cap,
                                  &pts[0].x, response_strengths.data(), &n));
        if (n <= cap) break;
        cap = n;
    }
    pts.resize(n); response_strengths.resize(n);
    return pts;
}

inline void deletePointsWithFailureStatus(std::vector<Point2f>& point_vector, const std::vector<bool>& isok) {   // vo.h:406-407
    size_t m = 0;
    for (size_t i = 0; i < point_vector.size(); i++)
        if (i >= isok.size() || isok[i]) point_vector[m++] = point_vector[i];
    point_vector.resize(m);
}

inline void deleteFeaturesWithFailureStatus(FeatureSet& f, const std::vector<bool>& isok) {                    // vo.h:416-417
    size_t m = 0;
    for (size_t i = 0; i < f.points.size(); i++)
        if (i >= isok.size() || isok[i]) { f.points[m] = f.points[i]; f.ages[m] = f.ages[i]; f.strengths[m] = f.strengths[i]; m++; }
    f.points.resize(m); f.ages.resize(m); f.strengths.resize(m);
}

inline std::vector<bool> findClosePoints(const std::vector<Point2f>& points_1, const std::vector<Point2f>& points_2,
                                         float threshold) {                                                     // vo.h:430-432
    std::vector<uint8_t> ok(points_1.size());
    if (!points_1.empty())
        svo_throw(svo_find_close_points(default_device(), (int)points_1.size(), &points_1[0].x, &points_2[0].x, threshold, ok.data()));
    return std::vector<bool>(ok.begin(), ok.end());
}

// vo.h:452-456.  rotation (3x3) and translation (3) are in/out exactly as in the reference.
inline std::pair<std::vector<int>, bool> cameraToWorld(const Mat33f& cameraProjection, const std::vector<Point2f>& cameraPoints,
                                                       const std::vector<Point3f>& worldPoints, Mat33d& rotation, Vec3d& translation) {
    int n = (int)cameraPoints.size(), nin = 0, ok = 0;
    std::vector<int> inl(n > 0 ? n : 1);
    svo_throw(svo_camera_to_world(default_device(), cameraProjection.data(), n, n ? &cameraPoints[0].x : nullptr,
                                  n ? &worldPoints[0].x : nullptr, rotation.data(), translation.data(), inl.data(), &nin, &ok,
                                  RANSAC_ITERATIONS, RANSAC_REPROJECTION_ERROR, 0.98f, nullptr));
    inl.resize(nin);
    return std::make_pair(inl, ok != 0);
}

inline Mat44 getInverseTransform(const Mat33d& rotation, const Vec3d& translation_stereo) {                     // vo.h:469-470
    Mat44 T;
    svo_throw(svo_inverse_transform(default_device(), rotation.data(), translation_stereo.data(), T.data()));
    return T;
}

class VisualOdometry {                                               // include/vo.h:231-380
   public:
    Mat34f leftCameraProjection_, rightCameraProjection_;           // vo.h:273
    VisualOdometry() { svo_config_default(&cfg_); }
    explicit VisualOdometry(const svo_config& cfg) : cfg_(cfg) {}
    // the 2-argument form src/stereo_vo.cpp:50 calls (missing from the reference's own header)
    VisualOdometry(const Mat34f& Pl, const Mat34f& Pr) { svo_config_default(&cfg_); initalize_projection_matricies(Pl, Pr); }
    ~VisualOdometry() { svo_destroy(ctx_); }
    VisualOdometry(const VisualOdometry&) = delete;
    VisualOdometry& operator=(const VisualOdometry&) = delete;

    void initalize_projection_matricies(const Mat34f leftCameraProjection, const Mat34f rightCameraProjection) {   // vo.h:307-309
        leftCameraProjection_ = leftCameraProjection; rightCameraProjection_ = rightCameraProjection;
        have_p_ = true;
        if (ctx_) svo_throw(svo_set_projection(ctx_, -1, leftCameraProjection_.data(), rightCameraProjection_.data()));
    }

    // vo.h:333-334: (success, transform).  On failure the transform is the last successful one (identity at first).
    std::pair<bool, Mat44> stereo_callback(const Image& image_left, const Image& image_right) {
        if (image_left.empty() || image_right.empty()) throw std::runtime_error("stereo_callback: empty image");
        if (image_left.channels != image_right.channels) throw std::runtime_error("stereo_callback: channel mismatch");
        if (!ctx_) {                                                  // the reference learns size and type from the first frame
            cfg_.channels = image_left.channels;
            svo_throw(svo_create(&cfg_, default_device(), 1, image_left.cols, image_left.rows, &ctx_));
            width_ = image_left.cols; height_ = image_left.rows;
            // like the reference, a first frame needs no projection matrices (vo.cpp:47-56 only caches); until
            // initalize_projection_matricies is called they are all-zero, as the reference's empty Mats effectively are
            if (!have_p_) { leftCameraProjection_.fill(0.f); rightCameraProjection_.fill(0.f); }
            svo_throw(svo_set_projection(ctx_, -1, leftCameraProjection_.data(), rightCameraProjection_.data()));
        }
        check_frame(image_left, "left"); check_frame(image_right, "right");
        Mat44 T;
        int rc = svo_process(ctx_, image_left.data, image_right.data, image_left.step, T.data(), &stats);
        svo_throw(rc);
        return std::make_pair(rc == 1, T);
    }
    // vo.h:374-379, vo.cpp:169-240.  Tracks pointsLeftT0 around the loop T0-left -> T1-left -> T1-right -> T0-right -> T0-left,
    // removes every point (and its feature) that lost an LK status or does not close the loop within 0.15 px, and makes the
    // T1 pyramids the cached pair (vo.cpp:231-232; on empty input it returns before doing so, :179-181).  As in the
    // reference this is a member call on the SAME state stereo_callback works on: the T0 side is the pyramid pair cached on
    // the device by the previous stereo_callback / circularMatching (prime it with stereo_callback, as main.cpp:191 does), and
    // the next stereo_callback tracks against the pair cached here.
    void circularMatching(const Image& imgLeftT1, const Image& imgRightT1, std::vector<Point2f>& pointsLeftT0,
                          std::vector<Point2f>& pointsRightT0, std::vector<Point2f>& pointsLeftT1,
                          std::vector<Point2f>& pointsRightT1, FeatureSet& current_features) {
        if (pointsLeftT0.empty()) return;                                                          // vo.cpp:179-181
        if (!ctx_) throw std::runtime_error("circularMatching: no cached pyramids (call stereo_callback first)");
        check_frame(imgLeftT1, "left"); check_frame(imgRightT1, "right");
        const int n = (int)pointsLeftT0.size();
        std::vector<Point2f> loop(n);
        std::vector<uint8_t> ok(n);
        pointsLeftT1.resize(n); pointsRightT1.resize(n); pointsRightT0.resize(n);
        svo_throw(svo_circular_matching(ctx_, imgLeftT1.data, imgRightT1.data, imgLeftT1.step, n, &pointsLeftT0[0].x, &pointsLeftT1[0].x,
                                        &pointsRightT1[0].x, &pointsRightT0[0].x, &loop[0].x, ok.data()));
        const std::vector<bool> keep(ok.begin(), ok.end());
        deleteFeaturesWithFailureStatus(current_features, keep);                                    // vo.cpp:233
        deletePointsWithFailureStatus(pointsLeftT0, keep); deletePointsWithFailureStatus(pointsLeftT1, keep);   // :234-238
        deletePointsWithFailureStatus(pointsRightT0, keep); deletePointsWithFailureStatus(pointsRightT1, keep);
    }

    // vo.h:354-362, vo.cpp:315-366: detect on imageLeftT0 (second pass at threshold / 4 if fewer than 100 features), track the
    // feature set through circularMatching, then drop everything that left the image in any of the four views.
    // As in the reference, imageLeftT0 is only what FAST runs on (vo.cpp:325); the loop's T0 side is the cached pyramid pair
    // (imageRightT0 is not read at all, vo.cpp:315-366), so the caller primes the object with stereo_callback(imageLeftT0, imageRightT0).
    void matchingFeatures(const Image& imageLeftT0, const Image& imageRightT0, const Image& imageLeftT1, const Image& imageRightT1,
                          FeatureSet& currentVOFeatures, std::vector<Point2f>& pointsLeftT0, std::vector<Point2f>& pointsRightT0,
                          std::vector<Point2f>& pointsLeftT1, std::vector<Point2f>& pointsRightT1) {
        currentVOFeatures.appendFeaturesFromImage(imageLeftT0, FAST_THRESHOLD);                     // vo.cpp:325
        if (currentVOFeatures.size() < PRE_MATCHING_FEATURE_THRESHOLD)                              // :327-332
            currentVOFeatures.appendFeaturesFromImage(imageLeftT0, FAST_THRESHOLD / 4);
        pointsLeftT0 = currentVOFeatures.points;                                                    // :338
        (void)imageRightT0;
        circularMatching(imageLeftT1, imageRightT1, pointsLeftT0, pointsRightT0, pointsLeftT1, pointsRightT1, currentVOFeatures);
        std::vector<bool> inside(pointsLeftT0.size());                                              // :341-359
        for (size_t i = 0; i < inside.size(); i++) {
            const Point2f* q[4] = {&pointsLeftT0[i], &pointsLeftT1[i], &pointsRightT0[i], &pointsRightT1[i]};
            bool in = true;
            for (const Point2f* p : q) if (p->x < 0 || p->y < 0 || p->y >= imageLeftT1.rows || p->x >= imageLeftT1.cols) in = false;
            inside[i] = in;
        }
        deleteFeaturesWithFailureStatus(currentVOFeatures, inside);                                 // :360-364
        deletePointsWithFailureStatus(pointsLeftT0, inside); deletePointsWithFailureStatus(pointsLeftT1, inside);
        deletePointsWithFailureStatus(pointsRightT0, inside); deletePointsWithFailureStatus(pointsRightT1, inside);
    }

    // functor form for boost::bind / message_filters style registration (src/stereo_vo.cpp:61-62)
    void operator()(const Image& l, const Image& r) { stereo_callback(l, r); }

    svo_frame_stats stats{};                                          // the counters the reference printf's (vo.cpp:226..365)
    svo_context* handle() { return ctx_; }

   private:
    // cv::Mat carries its own size and type and OpenCV asserts on mismatches; raw pointers do not, so the facade checks every
    // frame against the context (a shorter or narrower buffer would be read past its end)
    void check_frame(const Image& im, const char* which) const {
        if (im.empty()) throw std::runtime_error(std::string(which) + " image is empty");
        if (im.cols != width_ || im.rows != height_ || im.channels != cfg_.channels || im.step < im.cols * im.channels)
            throw std::runtime_error(std::string(which) + " image does not match the size / channels of the first frame");
    }
    svo_config cfg_;
    svo_context* ctx_ = nullptr;
    bool have_p_ = false;
    int width_ = 0, height_ = 0;                                      // learnt from the first frame
};

}   // namespace visual_odometry
