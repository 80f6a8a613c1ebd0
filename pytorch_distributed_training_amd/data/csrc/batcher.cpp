// Batch assembly for packed fixed-size records: gather records of a memory-mapped file straight into a caller-owned
// (pinned) batch buffer with a pool of native threads.
//
// A torch DataLoader moves every record through four copies on its way to the GPU (page cache -> tensor in a worker,
// -> collated batch in shared memory, -> pinned staging in the parent, -> device) and pays Python per record; at 196 KB
// per record that caps one MI355X's input at ~24 k images/s on 16 cores.  Here there is ONE host copy (page cache ->
// pinned batch), done by threads that never touch the interpreter, and the calling thread only submits index lists.
//
// Jobs are asynchronous: submit() returns a ticket at once, wait() blocks until that job's records are in place, so the
// caller keeps a few batches in flight under the training step.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Job {
    std::vector<long long> idx;
    unsigned char* dst = nullptr;
    std::atomic<int> next{0};          // next record to claim
    std::atomic<int> done{0};          // records finished
    std::atomic<int> bad{0};
};

struct Batcher {
    int fd = -1;
    const unsigned char* base = nullptr;
    size_t bytes = 0, rec = 0;
    long long n = 0;
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable work_cv, done_cv;
    std::deque<std::shared_ptr<Job>> queue;      // jobs with unclaimed records, oldest first
    std::map<long long, std::shared_ptr<Job>> jobs;
    long long next_ticket = 1;
    bool stop = false;

    void worker() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                work_cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop) return;
                job = queue.front();
            }
            const int total = (int)job->idx.size();
            // claim records in small runs: neighbours in the batch land in neighbouring destination rows
            for (;;) {
                const int first = job->next.fetch_add(4);
                if (first >= total) break;
                const int last = first + 4 < total ? first + 4 : total;
                for (int i = first; i < last; ++i) {
                    const long long r = job->idx[i];
                    if (r < 0 || r >= n) {
                        job->bad.store(1);
                        std::memset(job->dst + (size_t)i * rec, 0, rec);
                    } else {
                        std::memcpy(job->dst + (size_t)i * rec, base + (size_t)r * rec, rec);
                    }
                }
                if (job->done.fetch_add(last - first) + (last - first) == total) {
                    std::lock_guard<std::mutex> lk(mu);
                    done_cv.notify_all();
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!queue.empty() && queue.front() == job) queue.pop_front();     // fully claimed: next job's turn
            }
        }
    }
};

}  // namespace

extern "C" {

// Maps ``path`` (n records of rec_bytes each) read-only and starts ``threads`` gather threads.  NULL on failure.
void* pdt_batcher_open(const char* path, long long n, long long rec_bytes, int threads) {
    if (!path || n <= 0 || rec_bytes <= 0 || threads < 1 || threads > 256) return nullptr;
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || (long long)st.st_size < n * rec_bytes) {
        ::close(fd);
        return nullptr;
    }
    void* m = mmap(nullptr, (size_t)(n * rec_bytes), PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) {
        ::close(fd);
        return nullptr;
    }
    madvise(m, (size_t)(n * rec_bytes), MADV_RANDOM);      // shuffled access: no read-ahead beyond the record asked for
    Batcher* b = new Batcher();
    b->fd = fd;
    b->base = static_cast<const unsigned char*>(m);
    b->bytes = (size_t)(n * rec_bytes);
    b->rec = (size_t)rec_bytes;
    b->n = n;
    for (int i = 0; i < threads; ++i) b->pool.emplace_back([b] { b->worker(); });
    return b;
}

// Queue the gather of ``count`` records (indices are copied) into dst[count][rec_bytes]; returns a ticket (> 0).
long long pdt_batcher_submit(void* handle, const long long* idx, int count, void* dst) {
    Batcher* b = static_cast<Batcher*>(handle);
    if (!b || !idx || count <= 0 || !dst) return -1;
    auto job = std::make_shared<Job>();
    job->idx.assign(idx, idx + count);
    job->dst = static_cast<unsigned char*>(dst);
    std::lock_guard<std::mutex> lk(b->mu);
    const long long t = b->next_ticket++;
    b->jobs[t] = job;
    b->queue.push_back(job);
    b->work_cv.notify_all();
    return t;
}

// Block until the job is complete.  0: ok; 1: an index was out of range (its record is zero-filled); -1: unknown ticket.
int pdt_batcher_wait(void* handle, long long ticket) {
    Batcher* b = static_cast<Batcher*>(handle);
    if (!b) return -1;
    std::unique_lock<std::mutex> lk(b->mu);
    auto it = b->jobs.find(ticket);
    if (it == b->jobs.end()) return -1;
    std::shared_ptr<Job> job = it->second;
    b->done_cv.wait(lk, [&] { return job->done.load() == (int)job->idx.size(); });
    b->jobs.erase(ticket);
    return job->bad.load() ? 1 : 0;
}

void pdt_batcher_close(void* handle) {
    Batcher* b = static_cast<Batcher*>(handle);
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;
        b->work_cv.notify_all();
    }
    for (auto& t : b->pool) t.join();
    munmap(const_cast<unsigned char*>(b->base), b->bytes);
    ::close(b->fd);
    delete b;
}

}  // extern "C"
