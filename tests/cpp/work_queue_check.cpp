// erasor::WorkQueue (erasor_shim.h): jobs go, in order, to whichever worker asks next -- the "fifth sequence onto the first free GPU" rule
// of BASELINE config 3.  Workers with very different job durations: every job is taken exactly once, and the fast worker takes more.
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../erasor_amd/csrc/shim/erasor_shim_queue.h"
int main() {
    const size_t n_jobs = 12;
    erasor::WorkQueue q(n_jobs);
    std::vector<int> count(3, 0);
    std::vector<int> seen(n_jobs, 0);
    std::vector<std::thread> th;
    for (int w = 0; w < 3; ++w)
        th.emplace_back([&, w] {
            for (long j; (j = q.next(w)) >= 0;) {
                ++seen[j];
                ++count[w];
                std::this_thread::sleep_for(std::chrono::milliseconds(w == 0 ? 2 : 30));  // worker 0 is fifteen times faster
            }
        });
    for (auto &t : th) t.join();
    bool ok = q.next(0) == -1;
    for (size_t j = 0; j < n_jobs; ++j) ok = ok && seen[j] == 1 && q.taken_by(j) >= 0 && q.taken_by(j) < 3;
    ok = ok && count[0] + count[1] + count[2] == (int)n_jobs && count[0] > count[1] && count[0] > count[2];
    printf("jobs per worker: %d %d %d -> %s\n", count[0], count[1], count[2], ok ? "WORK-QUEUE-OK" : "FAILED");
    return ok ? 0 : 1;
}
