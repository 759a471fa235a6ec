// erasor_shim_queue.h -- which sequence goes to which device when there are more sequences than devices
#ifndef ERASOR_SHIM_QUEUE_H
#define ERASOR_SHIM_QUEUE_H
#include <atomic>
#include <cstddef>
#include <vector>
namespace erasor {
// Independent sequences over fewer devices than sequences (BASELINE config 3: "the fifth queued onto the first free GPU"): jobs are
// handed out in order to whichever worker asks next -- an idle device takes the next sequence, nothing is dealt in advance.
// (One updater per sequence like in the reference, main_kitti.cpp:4-11; one worker thread per device.)
class WorkQueue {
public:
    explicit WorkQueue(size_t n_jobs) : n_(n_jobs), taken_by_(n_jobs, -1) {}
    // the next job for `worker`, or -1 when none is left
    long next(int worker) {
        const size_t j = head_.fetch_add(1);
        if (j >= n_) return -1;
        taken_by_[j] = worker;
        return (long)j;
    }
    size_t size() const { return n_; }
    int taken_by(size_t job) const { return taken_by_[job]; }  // (read after the workers have joined)
private:
    size_t n_;
    std::atomic<size_t> head_{0};
    std::vector<int> taken_by_;
};
}  // namespace erasor
#endif
