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
        // the insertion rule runs on the GPU: replay the recorded inputs through a 1x1 grid of this capacity
        in_.points.push_back(point); in_.ages.push_back(age); in_.strengths.push_back(strength);
        features = in_;
        float mx = 1;
        for (auto& p : in_.points) { if (p.x > mx) mx = p.x; if (p.y > mx) mx = p.y; }
        Image dims(reinterpret_cast<const uint8_t*>(this), (int)mx + 2, (int)mx + 2);   // only rows/cols are read
        if (max_size) features.filterByBucketLocationInternal(dims, 1, 1, 0, max_size); else features.clear();
    }
    int size() { return features.size(); }
   private:
    FeatureSet in_;
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
            if (have_p_) svo_throw(svo_set_projection(ctx_, -1, leftCameraProjection_.data(), rightCameraProjection_.data()));
        }
        Mat44 T;
        int rc = svo_process(ctx_, image_left.data, image_right.data, image_left.step, T.data(), &stats);
        svo_throw(rc);
        return std::make_pair(rc == 1, T);
    }
    // functor form for boost::bind / message_filters style registration (src/stereo_vo.cpp:61-62)
    void operator()(const Image& l, const Image& r) { stereo_callback(l, r); }

    svo_frame_stats stats{};                                          // the counters the reference printf's (vo.cpp:226..365)
    svo_context* handle() { return ctx_; }

   private:
    svo_config cfg_;
    svo_context* ctx_ = nullptr;
    bool have_p_ = false;
};

}   // namespace visual_odometry
