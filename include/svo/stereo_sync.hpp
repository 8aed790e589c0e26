// stereo_sync.hpp — ROS-callback-shaped front door, ROS-free.
//
// The reference's node (src/stereo_vo.cpp:53-62) subscribes to left/image_rect and right/image_rect, pairs them with a
// message_filters ApproximateTime synchroniser of queue size 10 and binds the pair to VisualOdometry::stereo_callback
// after a MONO8 conversion.  This header provides the same shape without ROS: feed timestamped mono8 frames per side,
// get the callback invoked once per matched pair, oldest first.  A real ROS1 node only has to forward its two
// subscriber callbacks to push_left / push_right.  (ROS is not present in the build image; nothing here links to it.)
//
// Pairing rule (a deliberately small restatement of ApproximateTime for two topics): whenever both queues are
// non-empty, take the older of the two heads as pivot; its partner is the message of the other queue closest in time,
// provided a later message of that queue can no longer be closer (i.e. one with a stamp >= the pivot's has been seen)
// or the queue is full; a pivot whose best partner is closer to the pivot queue's next message is dropped (its own
// partner was lost); messages older than the chosen partner are dropped.  Queues are bounded (default 10): on
// overflow the oldest message is dropped, as message_filters does.
#pragma once
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <utility>
#include <vector>
#include "visual_odometry.hpp"

namespace visual_odometry {

struct StampedImage {
    double stamp = 0;                 // seconds
    int rows = 0, cols = 0;
    std::vector<uint8_t> mono8;       // owned copy (cv_bridge::toCvCopy semantics, stereo_vo.cpp:6-14)
    Image view() const { return Image(mono8.data(), rows, cols); }
};

class StereoSynchronizer {
   public:
    using Callback = std::function<void(const StampedImage&, const StampedImage&)>;
    explicit StereoSynchronizer(Callback cb, size_t queue_size = 10, double max_interval = 1e9)
        : cb_(std::move(cb)), cap_(queue_size ? queue_size : 1), slop_(max_interval) {}

    void push_left(StampedImage m) { push(left_, std::move(m)); match(); }
    void push_right(StampedImage m) { push(right_, std::move(m)); match(); }
    size_t pairs_emitted() const { return emitted_; }
    size_t dropped() const { return dropped_; }

   private:
    void push(std::deque<StampedImage>& q, StampedImage m) {
        q.push_back(std::move(m));
        if (q.size() > cap_) { q.pop_front(); dropped_++; }
    }
    void match() {
        for (;;) {
            if (left_.empty() || right_.empty()) return;
            const bool pivot_left = left_.front().stamp <= right_.front().stamp;
            std::deque<StampedImage>& P = pivot_left ? left_ : right_;
            std::deque<StampedImage>& O = pivot_left ? right_ : left_;
            const double t = P.front().stamp;
            // closest candidate in O, and whether a later arrival could still beat it
            size_t best = 0;
            for (size_t i = 1; i < O.size(); i++)
                if (std::abs(O[i].stamp - t) < std::abs(O[best].stamp - t)) best = i;
            const bool settled = O.back().stamp >= t || O.size() >= cap_;
            if (!settled) return;
            if (std::abs(O[best].stamp - t) > slop_) { P.pop_front(); dropped_++; continue; }
            // a later message of the pivot's own queue fits that partner better: the pivot has lost its partner, drop it
            if (P.size() > 1 && std::abs(P[1].stamp - O[best].stamp) < std::abs(t - O[best].stamp)) { P.pop_front(); dropped_++; continue; }
            for (size_t i = 0; i < best; i++) { O.pop_front(); dropped_++; }
            const StampedImage& l = pivot_left ? P.front() : O.front();
            const StampedImage& r = pivot_left ? O.front() : P.front();
            cb_(l, r);
            emitted_++;
            P.pop_front(); O.pop_front();
        }
    }
    Callback cb_;
    size_t cap_;
    double slop_;
    std::deque<StampedImage> left_, right_;
    size_t emitted_ = 0, dropped_ = 0;
};

// The binding of stereo_vo.cpp:61-62: synchroniser -> VisualOdometry::stereo_callback.
inline StereoSynchronizer make_stereo_vo_callback(VisualOdometry& vo, std::function<void(bool, const Mat44&)> on_pose = nullptr,
                                                  size_t queue_size = 10) {
    return StereoSynchronizer(
        [&vo, on_pose](const StampedImage& l, const StampedImage& r) {
            auto out = vo.stereo_callback(l.view(), r.view());
            if (on_pose) on_pose(out.first, out.second);
        },
        queue_size);
}

}   // namespace visual_odometry
