// dtw_main.cpp -- `sigfish-amd dtw`: the reference's `sigfish dtw` command line (src/dtw_main.c:125-352) over the
// MI355X alignment stage.  Same positional arguments, options and PAF output; the batch loop keeps the reference's
// order (load -> process -> output, src/dtw_main.c:299-326) so reads come out in file order.
//
// Host stages run on a thread fan-out per batch (parse, events, normalise: src/sigfish.c:317-505); the DTW stage
// is one call into the C-ABI (sfa_align_events, the align_db hook).  There is no CPU DTW path in this binary.
#include <getopt.h>
#include <sys/prctl.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/sigfish_amd.h"
#include "blow5.hpp"
#include "events.hpp"
#include "refio.hpp"

namespace {

// option bits beyond the ones the library reads: same values as src/sigfish.h:30-39
enum : uint32_t { F_RNA = 0x001, F_DTW = 0x002, F_INV = 0x004, F_SEC = 0x008, F_REF = 0x010, F_END = 0x020, F_PRF = 0x040, F_SAM = 0x100, F_R10 = 0x200 };

struct Opt {
    uint32_t flag = 0;
    int32_t batch_size = 4096;            // -K  (reference default 512; a GPU batch wants thousands of reads)
    int64_t batch_bytes = 200 * 1000 * 1000;  // -B  (reference default 20M)
    int32_t threads = 8;
    int32_t prefix = 50, query = 250;
    int32_t debug_break = -1;
    int verbosity = 4;
    std::vector<int> devices{0};  // --device 0,1,...: batches go to the devices in turn
    bool host_events = false;  // --host-events: event detection on host threads instead of the GPU
    int gpu_parse = -1;        // --gpu-parse / --host-parse: records decompressed and parsed on the GPU / on host threads (-1: by device count)
    int streams = 0;           // --streams: device contexts that take batches in turn (0 = 2)
    // read sharding over processes (one per GPU): what part of the file THIS process maps
    int ranks = 0;                       // --ranks G: start G processes on disjoint parts of the file, print their output in rank order (0: one per distinct device)
    int shard_r = 0, shard_n = 1;        // --shard r/G: the records starting in the r-th of G equal byte slices of the file
    int64_t range_a = 0, range_b = -1;   // --read-range A:B: records [A, B) by position in the file (B omitted: to the end)
    bool no_header = false;              // --no-header: no SAM header (every rank but the first of a sharded run)
    int64_t rank_buffer = 256 * 1000 * 1000;  // --rank-buffer: gathered output kept in memory per rank; what exceeds it goes to an unnamed temporary file
    int device_share = 1;                // ranks of a sharded run that were given the same device as this one (a one-GPU rehearsal of --ranks)
    const char *model_file = nullptr;
    const char *pore = nullptr;
    int pore_flag = 0;  // 0 r9, 1 r10, 2 rna004
};

double realtime() {
    timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}
double cputime() {
    rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

// Errors travel as exceptions: die() may be called on a helper thread (the GPU stage or the output stage of a batch,
// both std::async) while the main thread, the worker pool and another GPU stage are still running.  exit() from there
// would run static destructors and the HIP runtime's teardown under live kernels and threads; instead the exception
// crosses the future, the stack unwinds (futures of std::async wait for their thread, the pool joins its workers) and
// dtw_main() prints the message and returns the failure status from the main thread.
struct Fatal : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void die(const std::string &msg) { throw Fatal(msg); }

bool yes_or_no(const char *arg, const char *what) {  // yes_or_no(), src/dtw_main.c:92-113
    if (!strcmp(arg, "yes") || !strcmp(arg, "y")) return true;
    if (!strcmp(arg, "no") || !strcmp(arg, "n")) return false;
    die(std::string("option '--") + what + "' only accepts 'yes' or 'no'.");
}

int64_t parse_num(const char *s) {  // K/M/G suffixes as src/dtw_main.c:46-58
    char *e;
    double x = strtod(s, &e);
    if (*e == 'G' || *e == 'g')
        x *= 1e9;
    else if (*e == 'M' || *e == 'm')
        x *= 1e6;
    else if (*e == 'K' || *e == 'k')
        x *= 1e3;
    return static_cast<int64_t>(x + .499);
}

void help(FILE *fp, const Opt &o) {
    fprintf(fp, "Usage: sigfish-amd dtw [OPTIONS] genome.fa reads.blow5|reads.slow5\n\nbasic options:\n");
    fprintf(fp, "   -t INT                     number of host threads for parsing and event detection [%d]\n", o.threads);
    fprintf(fp, "   -K INT                     batch size (max number of reads loaded at once) [%d]\n", o.batch_size);
    fprintf(fp, "   -B FLOAT[K/M/G]            max number of bytes loaded at once [%.1fM]\n", o.batch_bytes / 1e6);
    fprintf(fp, "   -h                         help\n   -o FILE                    output to file [stdout]\n");
    fprintf(fp, "   --verbose INT              verbosity level [%d]\n   --version                  print version\n", o.verbosity);
    fprintf(fp, "   --pore STR                 set the pore chemistry (r9, r10 or rna004) [auto]\n");
    fprintf(fp, "   --device INT[,INT...]      GPU(s) to use; batches are dealt to them in turn [0]\n   --host-events              detect events on host threads instead of the GPU\n   --gpu-parse | --host-parse decompress and parse the records on the GPU | on host threads [host threads up to 2 GPUs per process, GPU beyond]\n   --streams INT              device contexts taking batches in turn [2]\n");
    fprintf(fp, "   --ranks INT                read-shard the run over INT processes (rank r: device r of the list, -t/INT threads, the r-th\n"
                "                              byte slice of the file, which must be a regular file); output is printed in rank order = file\n"
                "                              order [one per distinct device]\n"
                "   --shard r/G                map only the records starting in the r-th of G equal byte slices of the file\n"
                "   --read-range A:B           map only records A..B-1 of the file (B omitted: to the end)\n"
                "   --rank-buffer FLOAT[K/M/G] gathered output kept in memory per rank; the rest waits in an unnamed temporary file [%.0fM]\n"
                "   --no-header                do not print the SAM header (ranks after the first)\n\nadvanced options:\n", o.rank_buffer / 1e6);
    fprintf(fp, "   --kmer-model FILE          nucleotide k-mer model file (required: builtin models are not bundled)\n");
    fprintf(fp, "   --rna                      the dataset is direct RNA\n");
    fprintf(fp, "   -q INT                     the number of events in query signal to align [%d]\n", o.query);
    fprintf(fp, "   -p INT                     the number of events to trim at query signal start [%d]\n", o.prefix);
    fprintf(fp, "   --debug-break INT          break after processing the specified no. of batches\n");
    fprintf(fp, "   --dtw-std                  use DTW standard instead of DTW subsequence\n");
    fprintf(fp, "   --invert                   reverse the reference events instead of query\n");
    fprintf(fp, "   --full-ref                 map to the full reference\n");
    fprintf(fp, "   --from-end                 map the end portion of the query instead of the beginning\n");
    fprintf(fp, "   --sam                      output in SAM format\n");
    fprintf(fp, "   --profile-cpu=yes|no       run the stages one after the other and report Parse/Events/Normalise/DTW time [no]\n");
    fprintf(fp, "   --accel=yes|no             run the alignment on the accelerator [yes]; 'no' is an error: this build has no CPU path\n");
}

struct Read {
    std::vector<uint8_t> mem;          // record bytes when the file cannot be mapped
    const uint8_t *view = nullptr;     // record bytes inside the mapped file otherwise
    size_t view_size = 0;
    sfa::Blow5Record rec;
    std::vector<sfa_event_t> ev;
    std::vector<float> pa;             // --profile-cpu=yes: picoamps kept between the events and the normalise stage
    int64_t qstart = 0, qend = 0;
    bool keep = false;
    int status = 0;
};

// work_db() of the reference forks and joins `-t` threads for every stage of every batch (src/thread.c:119-132); here
// the workers are created once and woken per stage.  Items are handed out through an atomic counter,
// the calling thread works too.  Several threads may call run() (the loader and the output stage do): calls queue up.
class WorkerPool {
  public:
    explicit WorkerPool(int nthreads) {
        for (int t = 1; t < nthreads; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    template <typename F>
    void run(int64_t n, F fn) {
        if (workers_.empty() || n <= 1) {
            for (int64_t i = 0; i < n; ++i) fn(i);
            return;
        }
        std::lock_guard<std::mutex> one_job(submit_mu_);
        std::function<void(int64_t)> f = fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f;
            n_ = n;
            next_.store(0);
            busy_ = static_cast<int>(workers_.size());
            ++generation_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return busy_ == 0; });
        fn_ = nullptr;
    }

  private:
    void drain() {
        for (;;) {
            const int64_t i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--busy_ == 0) done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, submit_mu_;
    std::condition_variable cv_, done_cv_;
    std::atomic<int64_t> next_{0};
    const std::function<void(int64_t)> *fn_ = nullptr;
    int64_t n_ = 0;
    int busy_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// load_db() of the reference reads the records of a batch on the main thread, between two batches' worth of work (src/sigfish.c:
// 262-315).  Framing a mapped file is cheap -- a size prefix per record -- but it is a cache miss per record, 0.25 us x 1.6 M records
// = 0.4 s of a 2.3 s run during which the -t workers have nothing to do.  Here a helper frames batch i + 1 (and i + 2) while the main
// thread and the workers are in the host stages of batch i: the main thread only collects finished lists.
struct Frames {
    std::vector<const uint8_t *> view;
    std::vector<size_t> size;
    int32_t n = 0;
    int64_t bytes = 0;
    bool more = true;  // false: the file (or this process's part of it) ends with this batch
    bool failed = false;
};
class FrameLoader {
  public:
    FrameLoader(sfa::Blow5Reader &reader, int32_t batch_size, int64_t batch_bytes) : reader_(reader), batch_size_(batch_size), batch_bytes_(batch_bytes) {
        for (Frames &f : ring_) {
            f.view.resize(batch_size);
            f.size.resize(batch_size);
        }
        th_ = std::thread([this] { loop(); });
    }
    ~FrameLoader() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    // the next batch's list; valid until the call after the next one (three buffers: one with the caller, two ahead)
    const Frames &next() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return produced_ > consumed_; });
        const Frames &f = ring_[consumed_ % kDepth];
        ++consumed_;
        lk.unlock();
        cv_.notify_all();
        return f;
    }

  private:
    static constexpr uint64_t kDepth = 3;
    void loop() {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                // the caller still reads the list it took last: keep one buffer behind `consumed_` untouched
                cv_.wait(lk, [this] { return quit_ || produced_ + 1 < consumed_ + kDepth; });
                if (quit_) return;
            }
            Frames &f = ring_[produced_ % kDepth];
            f.n = 0;
            f.bytes = 0;
            f.more = true;
            f.failed = false;
            while (f.n < batch_size_ && f.bytes < batch_bytes_) {
                const int rc = reader_.next_view(&f.view[f.n], &f.size[f.n]);
                if (rc < 0) f.failed = true;
                if (rc <= 0) {
                    f.more = false;
                    break;
                }
                f.bytes += static_cast<int64_t>(f.size[f.n]);
                ++f.n;
            }
            const bool last = !f.more;
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++produced_;
            }
            cv_.notify_all();
            if (last) return;
        }
    }
    sfa::Blow5Reader &reader_;
    const int32_t batch_size_;
    const int64_t batch_bytes_;
    Frames ring_[kDepth];
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t produced_ = 0, consumed_ = 0;
    bool quit_ = false;
    std::thread th_;
};

// ---- read sharding over processes (the reference's analogue is the serial loop src/dtw_main.c:299-326) ----
// `--ranks G`: this process becomes a supervisor BEFORE anything has touched the GPU (no HIP call is made ahead of sfa_init,
// and the supervisor never makes one).  It forks G ranks; rank r continues into the ordinary run with --shard r/G, device r of
// the --device list and its share of the host threads.  Rank 0 writes to the supervisor's own stdout; the later ranks write
// into pipes that the supervisor drains into memory as they run (a rank never waits for the printer) and prints in rank order
// once the ranks in front have finished -- the "gather of PAF rows".  Shards are contiguous in the file and disjoint, every
// rank prints its reads in file order (src/sigfish.c:1051-1086), so the concatenation is the single-process output byte for byte.
// Returns -1 in a rank (which carries on with `o` rewritten), the exit status in the supervisor.
int supervise_ranks(Opt &o, double t0) {
    const int G = o.ranks;
    struct Rank {
        pid_t pid = -1;
        int fd = -1;  // read end of the rank's stdout pipe (rank 0: none)
        std::string out;
        FILE *spill = nullptr;  // output beyond --rank-buffer (tmpfile(): unnamed, gone with the process)
        int64_t spilled = 0;
        bool spill_failed = false;
        int status = -1;
        double wall = 0;
        std::thread th;
    };
    std::vector<Rank> rk(G);
    fflush(stdout);
    fflush(stderr);
    const int threads_each = std::max(1, o.threads / G);
    if (threads_each < 4 && o.verbosity >= 1)  // (a rank's host stages -- inflating and decoding its records -- want a GPU's worth of cores)
        fprintf(stderr, "[sigfish-amd] WARNING: -t %d over %d ranks leaves %d host thread(s) per rank; the host stages of a rank scale to about 16 (-t %d)\n",
                o.threads, G, threads_each, 16 * G);
    const pid_t supervisor = getpid();
    for (int r = 0; r < G; ++r) {
        int pfd[2] = {-1, -1};
        if (r > 0 && pipe(pfd) != 0) die("--ranks: cannot create a pipe");
        const pid_t pid = fork();
        if (pid < 0) die("--ranks: cannot start a rank (fork failed)");
        if (pid == 0) {  // the rank
            prctl(PR_SET_PDEATHSIG, SIGTERM);  // a supervisor that dies takes its ranks with it
            if (getppid() != supervisor) _exit(EXIT_FAILURE);  // (... and one that died before the line above, too)
            for (int q = 1; q < r; ++q) close(rk[q].fd);
            if (r > 0) {
                close(pfd[0]);
                if (dup2(pfd[1], STDOUT_FILENO) < 0) _exit(EXIT_FAILURE);
                close(pfd[1]);
            }
            o.shard_r = r;
            o.shard_n = G;
            o.ranks = 1;
            o.threads = threads_each;
            int share = 0;
            for (int q = 0; q < G; ++q) share += o.devices[q % o.devices.size()] == o.devices[r % o.devices.size()];
            o.device_share = share;
            o.devices = {o.devices[r % o.devices.size()]};
            o.no_header = o.no_header || r > 0;
            return -1;
        }
        rk[r].pid = pid;
        if (r > 0) {
            close(pfd[1]);
            rk[r].fd = pfd[0];
        }
    }
    for (int r = 0; r < G; ++r)
        rk[r].th = std::thread([&rk, r, t0, cap = o.rank_buffer] {
            Rank &k = rk[r];
            if (k.fd >= 0) {
                char buf[1 << 16];
                for (;;) {
                    const ssize_t n = read(k.fd, buf, sizeof buf);
                    if (n > 0) {
                        // (keep reading whatever happens to the spill file: a rank must never block on its pipe)
                        if (!k.spill && !k.spill_failed && static_cast<int64_t>(k.out.size()) + n > cap) {
                            k.spill = tmpfile();
                            k.spill_failed = !k.spill;
                        }
                        if (k.spill) {
                            if (fwrite(buf, 1, static_cast<size_t>(n), k.spill) != static_cast<size_t>(n)) k.spill_failed = true;
                            k.spilled += n;
                        } else if (!k.spill_failed) {
                            k.out.append(buf, static_cast<size_t>(n));
                        }
                    } else if (n == 0 || errno != EINTR) {
                        break;
                    }
                }
                close(k.fd);
            }
            int st = 0;
            while (waitpid(k.pid, &st, 0) < 0 && errno == EINTR) {}
            k.status = st;
            k.wall = realtime() - t0;
        });
    int rc = EXIT_SUCCESS;
    for (int r = 0; r < G; ++r) {
        rk[r].th.join();
        const bool ok = WIFEXITED(rk[r].status) && WEXITSTATUS(rk[r].status) == 0 && !rk[r].spill_failed;
        if (o.verbosity >= 3)
            fprintf(stderr, "[dtw_main] rank %d/%d (device %d, %d host threads): %s after %.3f sec, %zu bytes of output gathered\n", r, G,
                    o.devices[r % o.devices.size()], threads_each, ok ? "done" : "FAILED", rk[r].wall, rk[r].out.size() + static_cast<size_t>(rk[r].spilled));
        if (rk[r].spill_failed) fprintf(stderr, "[sigfish-amd] ERROR: rank %d: the temporary file for output beyond --rank-buffer could not be written\n", r);
        if (!ok) rc = EXIT_FAILURE;
        if (rc == EXIT_SUCCESS && !rk[r].out.empty() && fwrite(rk[r].out.data(), 1, rk[r].out.size(), stdout) != rk[r].out.size()) rc = EXIT_FAILURE;
        std::string().swap(rk[r].out);
        if (rk[r].spill) {
            if (rc == EXIT_SUCCESS) {
                rewind(rk[r].spill);
                std::vector<char> buf(1 << 20);
                for (size_t n; (n = fread(buf.data(), 1, buf.size(), rk[r].spill)) > 0;)
                    if (fwrite(buf.data(), 1, n, stdout) != n) {
                        rc = EXIT_FAILURE;
                        break;
                    }
                if (ferror(rk[r].spill)) rc = EXIT_FAILURE;
            }
            fclose(rk[r].spill);
        }
    }
    fflush(stdout);
    if (rc != EXIT_SUCCESS) fprintf(stderr, "[sigfish-amd] ERROR: a rank of the sharded run failed; output is incomplete\n");
    return rc;
}

}  // namespace

static int dtw_run(int argc, char **argv);

static double g_t0 = 0;
static int g_verbosity = 0;

int dtw_main(int argc, char **argv) {
    try {
        const int rc = dtw_run(argc, argv);
        // (--verbose 4: where a short run's tail goes -- contexts, page-locked buffers, the mapped file and the slots have been
        // released by now; what follows is the runtime's own exit.  Measured on a compressed 400 000-read file: release 0.06-0.10 s,
        // exit 0.12 s; leaving everything to the kernel with _exit() right after the last line takes the same 0.18-0.2 s, and
        // bringing the contexts up beside the first batches (parsing ahead into spare slots) gains what the spare slots then cost
        // at exit: profiles/r03_logs/rejected_cli_async_init_run_ahead_and_fast_exit.log)
        if (g_verbosity >= 4) fprintf(stderr, "[dtw_main::%.3f] device contexts, staging buffers and the file mapping released\n", realtime() - g_t0);
        return rc;
    } catch (const Fatal &e) {  // every helper thread has been joined by the unwinding (see Fatal)
        fflush(stdout);
        fprintf(stderr, "[sigfish-amd] ERROR: %s\n", e.what());
        return EXIT_FAILURE;
    }
}

static int dtw_run(int argc, char **argv) {
    const double t0 = realtime();
    g_t0 = t0;
    static option lo[] = {{"threads", required_argument, 0, 't'},   {"batchsize", required_argument, 0, 'K'},
                          {"max-bytes", required_argument, 0, 'B'}, {"verbose", required_argument, 0, 'v'},
                          {"help", no_argument, 0, 'h'},            {"version", no_argument, 0, 'V'},
                          {"kmer-model", required_argument, 0, 1},  {"output", required_argument, 0, 'o'},
                          {"rna", no_argument, 0, 2},               {"prefix", required_argument, 0, 'p'},
                          {"query-size", required_argument, 0, 'q'}, {"debug-break", required_argument, 0, 3},
                          {"dtw-std", no_argument, 0, 4},           {"invert", no_argument, 0, 5},
                          {"full-ref", no_argument, 0, 6},          {"from-end", no_argument, 0, 7},
                          {"profile-cpu", required_argument, 0, 8}, {"accel", required_argument, 0, 9},
                          {"sam", no_argument, 0, 'a'},             {"pore", required_argument, 0, 10},
                          {"device", required_argument, 0, 11},     {"secondary", required_argument, 0, 12},
                          {"window", required_argument, 0, 'w'},    {"meth-model", required_argument, 0, 13},   {"host-events", no_argument, 0, 14},   {"streams", required_argument, 0, 15},   {"host-parse", no_argument, 0, 16},   {"gpu-parse", no_argument, 0, 17},   
                          {"ranks", required_argument, 0, 20},      {"shard", required_argument, 0, 21},
                          {"read-range", required_argument, 0, 22}, {"no-header", no_argument, 0, 23},
                          {"rank-buffer", required_argument, 0, 25},
                          {0, 0, 0, 0}};
    Opt o;
    FILE *fp_help = stderr;
    int c, li = 0;
    while ((c = getopt_long(argc, argv, "p:q:t:B:K:v:o:w:ahV", lo, &li)) >= 0) {
        switch (c) {
            case 'B': o.batch_bytes = parse_num(optarg); if (o.batch_bytes <= 0) die("Maximum number of bytes should be larger than 0."); break;
            case 'K': o.batch_size = atoi(optarg); if (o.batch_size < 1) die("Batch size should larger than 0."); break;
            case 't': o.threads = atoi(optarg); if (o.threads < 1) die("Number of threads should larger than 0."); break;
            case 'v': o.verbosity = atoi(optarg); break;
            case 'V': fprintf(stdout, "sigfish-amd %s\n", sfa_version()); exit(EXIT_SUCCESS);
            case 'h': fp_help = stdout; break;
            case 'p': o.prefix = atoi(optarg); break;
            case 'q': o.query = atoi(optarg); if (o.query < 1) die("Query size should larger than 0."); break;
            case 'o': if (strcmp(optarg, "-") != 0 && !freopen(optarg, "wb", stdout)) die(std::string("failed to write the output to file ") + optarg); break;
            case 'a': o.flag |= F_SAM; break;
            case 'w': break;  // parsed and unused by the reference as well
            case 1: o.model_file = optarg; break;
            case 2: o.flag |= F_RNA; break;
            case 3: o.debug_break = atoi(optarg); break;
            case 4: o.flag |= F_DTW; break;
            case 5: o.flag |= F_INV; break;
            case 6: o.flag |= F_REF; break;
            case 7: o.flag |= F_END; break;
            case 8: if (yes_or_no(optarg, "profile-cpu")) o.flag |= F_PRF; else o.flag &= ~F_PRF; break;  // src/dtw_main.c:213-214
            case 9:  // src/dtw_main.c:215-220: the reference falls back to work_db(dtw_single) on its CPU; there is none here
                if (!yes_or_no(optarg, "accel"))
                    die("--accel=no: this build has no CPU alignment path (the stage only exists as gfx950 kernels); "
                        "run the reference binary for a CPU run, or drop the option");
                break;
            case 12: case 13: break;  // dead options of the reference (--secondary, --meth-model): parsed, no effect
            case 10:
                o.pore = optarg;
                if (strcmp(optarg, "r9") && strcmp(optarg, "r10") && strcmp(optarg, "rna004")) die("Pore model should be r9, r10 or rna004");
                if (!strcmp(optarg, "r10")) { o.flag |= F_R10; o.pore_flag = 1; }
                if (!strcmp(optarg, "rna004")) { o.flag |= F_RNA | F_R10; o.pore_flag = 2; }
                break;
            case 11: {
                o.devices.clear();
                for (const char *p = optarg; *p;) {
                    char *e = nullptr;
                    const long d = strtol(p, &e, 10);
                    if (e == p || d < 0) die("--device takes a comma separated list of GPU indices");
                    o.devices.push_back(static_cast<int>(d));
                    p = (*e == ',') ? e + 1 : e;
                    if (*e && *e != ',') die("--device takes a comma separated list of GPU indices");
                }
                if (o.devices.empty()) die("--device takes a comma separated list of GPU indices");
                break;
            }
            case 14: o.host_events = true; break;
            case 16: o.gpu_parse = 0; break;
            case 17: o.gpu_parse = 1; break;
            case 20: o.ranks = atoi(optarg); if (o.ranks < 1 || o.ranks > 64) die("--ranks should be 1..64"); break;
            case 21:
                if (sscanf(optarg, "%d/%d", &o.shard_r, &o.shard_n) != 2 || o.shard_n < 1 || o.shard_r < 0 || o.shard_r >= o.shard_n)
                    die("--shard takes r/G with 0 <= r < G");
                break;
            case 22: {
                char *e = nullptr;
                o.range_a = strtoll(optarg, &e, 10);
                if (e == optarg || *e != ':' || o.range_a < 0) die("--read-range takes A:B (records A up to, not including, B; B may be omitted)");
                ++e;  // past the colon
                o.range_b = *e ? strtoll(e, &e, 10) : -1;
                if (*e || (o.range_b >= 0 && o.range_b < o.range_a)) die("--read-range takes A:B (records A up to, not including, B; B may be omitted)");
                break;
            }
            case 23: o.no_header = true; break;
            case 25: o.rank_buffer = parse_num(optarg); if (o.rank_buffer < 0) die("--rank-buffer should not be negative"); break;
            case 15: o.streams = atoi(optarg); if (o.streams < 1 || o.streams > 8) die("--streams should be 1..8"); break;
            default: help(stderr, o); exit(EXIT_FAILURE);
        }
    }
    if (argc - optind != 2 || fp_help == stdout) {
        help(fp_help, o);
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    const char *fasta = argv[optind], *blow5 = argv[optind + 1];
    // same order of checks as src/dtw_main.c:248-277 (before RNA auto-detection)
    if (!(o.flag & F_RNA)) {
        if (o.flag & F_DTW) die("DTW is only available for RNA.");
        if (o.flag & F_INV) die("Inversion is only available for RNA.");
        if (o.flag & F_REF) die("--full-ref is only available for RNA.");
    }
    if (o.prefix < 0) {
        if (!(o.flag & F_RNA)) die("DNA does not support auto query start detection.");
        if (o.flag & F_INV) die("Inversion is not compatible with auto query start detection.");
        if (o.flag & F_END) die("Mapping from query end is not compatible with auto query start detection.");
    }

    if (o.shard_n > 1 && (o.range_a > 0 || o.range_b >= 0)) die("--shard and --read-range exclude each other");
    if (o.ranks == 0) {  // one process per DISTINCT device of the list; a device listed twice is two contexts of one process
        std::vector<int> d = o.devices;
        std::sort(d.begin(), d.end());
        o.ranks = static_cast<int>(std::unique(d.begin(), d.end()) - d.begin());
    }
    if (o.ranks > 1) {
        if (o.shard_n > 1 || o.range_a > 0 || o.range_b >= 0) die("--ranks shards the whole file: it cannot be combined with --shard or --read-range");
        if (o.debug_break >= 0) die("--debug-break counts the batches of one process: use --ranks 1 with it");
        const int rc = supervise_ranks(o, t0);
        if (rc >= 0) return rc;
    }

    // ---- init_core(), src/sigfish.c:81-207 ----
    double t_init[4] = {0, 0, 0, 0};  // reader, model + reference events, device contexts, (teardown)
    double ti = realtime();
    sfa::Blow5Reader reader;
    if (!reader.open(blow5)) die(reader.error());
    // This process's part of the file.  Finding it means walking the size prefixes of every record in front of it (0.36 us each: 0.3 s
    // for the second half of a 1.6 M-read file), which needs nothing but the mapping: a helper thread does it while this one reads
    // the model, builds the reference events and brings up the device contexts (0.3-0.4 s).  Joined in front of the batch loop.
    std::future<bool> selected = std::async(std::launch::async, [&reader, &o] {
        if (o.shard_n > 1 && !reader.select_shard(static_cast<uint32_t>(o.shard_r), static_cast<uint32_t>(o.shard_n))) return false;
        if ((o.range_a > 0 || o.range_b >= 0) &&
            !reader.select_records(static_cast<uint64_t>(o.range_a), o.range_b < 0 ? UINT64_MAX : static_cast<uint64_t>(o.range_b - o.range_a)))
            return false;
        reader.start_prefault();  // a helper thread takes the page faults of the mapped file ahead of the batch loop (blow5.hpp)
        return true;
    });
    t_init[0] = realtime() - ti;
    ti = realtime();
    if (const char *exp = reader.attr("experiment_type")) {
        if (!strcmp(exp, "rna")) o.flag |= F_RNA;
    }
    if (!o.pore) {
        if (const char *kit = reader.attr("sequencing_kit")) {
            if (strstr(kit, "114")) { o.flag |= F_R10; o.pore_flag = 1; }
            else if (strstr(kit, "rna004")) { o.flag |= F_R10; o.pore_flag = 2; }
            if (o.pore_flag == 1 && (o.flag & F_RNA)) die("R10 RNA data does not exist! But the header indicates that the data is R10 RNA.");
        }
    }
    const bool rna = (o.flag & F_RNA) != 0;
    if (!o.model_file)
        die("builtin pore models are not bundled with this build (the reference's src/model.h tables are not part of "
            "this tree); pass --kmer-model FILE");
    std::vector<float> levels;
    uint32_t k = 0;
    std::string err;
    std::string model_warnings;
    if (!sfa::read_kmer_model(o.model_file, &levels, &k, &err, &model_warnings)) die(err);
    if (!model_warnings.empty() && o.verbosity >= 1) fprintf(stderr, "[sigfish-amd] ERROR: %s", model_warnings.c_str());  // logged, not fatal: src/model.c:98-100
    std::vector<sfa::FastaRecord> contigs;
    if (!sfa::read_fasta(fasta, &contigs, &err)) die(err);
    if (contigs.empty()) die(std::string("no sequences in ") + fasta);
    const int32_t nref = static_cast<int32_t>(contigs.size());
    std::vector<std::vector<float>> fwd(nref), rev(nref);
    std::vector<int32_t> ref_len(nref), ref_off(nref), seq_len(nref);
    for (int32_t i = 0; i < nref; ++i) {
        const int32_t l = static_cast<int32_t>(contigs[i].seq.size());
        if (l < static_cast<int32_t>(k)) die("contig " + contigs[i].name + " is shorter than the k-mer size");
        fwd[i].resize(l + 1 - k);
        if (!rna) rev[i].resize(l + 1 - k);
        const int32_t n = sfa_gen_ref_record(contigs[i].seq.c_str(), l, levels.data(), k, o.flag, o.query, fwd[i].data(),
                                             rna ? nullptr : rev[i].data(), &ref_off[i]);
        if (n <= 0) die("cannot build reference events for " + contigs[i].name);
        ref_len[i] = n;
        seq_len[i] = l;
    }
    std::vector<const float *> fp(nref), rp(nref);
    for (int32_t i = 0; i < nref; ++i) {
        fp[i] = fwd[i].data();
        rp[i] = rna ? nullptr : rev[i].data();
    }
    sfa_ref_t sref{nref, ref_len.data(), ref_off.data(), fp.data(), rna ? nullptr : rp.data()};
    t_init[1] = realtime() - ti;
    ti = realtime();
    // two contexts (streams + scratch) on the same device: consecutive batches alternate between them, so the uploads
    // and the event detection of batch i+1 overlap the DTW of batch i
    // (--streams: more than two were measured to add nothing, the stages of one batch already serialise on syncs)
    // Several devices (--device 0,1,...): reads shard by batch, every device holds its own copy of the reference
    // arrays (uploaded by sfa_init: a single process needs no collective), rows come back in batch order.
    const int n_ctx = (o.streams > 0 ? o.streams : 2) * static_cast<int>(o.devices.size());
    struct Contexts : std::vector<sfa_ctx_t *> {  // destroyed on every way out, after the helper threads that use them
        using std::vector<sfa_ctx_t *>::vector;
        ~Contexts() {
            for (sfa_ctx_t *c : *this) sfa_destroy(c);
        }
    } ctxs(n_ctx, nullptr);
    for (int j = 0; j < n_ctx; ++j)
    {
        if (sfa_init(&ctxs[j], &sref, o.flag, o.devices[j % o.devices.size()]) != SFA_OK)
            die(std::string("accelerator init failed: ") + sfa_last_error());
        // the small-batch shapes pay off while ONE batch leaves the chip idle; with s batches in flight per device the
        // threshold (waves per SIMD of a single batch) shrinks accordingly
        const int per_dev = n_ctx / static_cast<int>(o.devices.size()) * o.device_share;  // batches in flight on this device, all processes
        if (sfa_set_option(ctxs[j], "widen_below", std::max(1, 5 / per_dev)) != SFA_OK) die(sfa_last_error());
        // SFA_OPTS="name=value,name=value": planner / launch options of the library (sfa_set_option) for experiments from the
        // command line; rows do not depend on them (the library's test hooks are not options: refused here whatever the environment)
        if (const char *e = getenv("SFA_OPTS")) {
            std::string all(e);
            for (size_t p = 0; p < all.size();) {
                const size_t q = std::min(all.find(',', p), all.size());
                const std::string kv = all.substr(p, q - p);
                const size_t eq = kv.find('=');
                if (eq == std::string::npos || eq == 0) die("SFA_OPTS takes name=value[,name=value...]");
                if (kv.compare(0, 6, "debug_") == 0) die("SFA_OPTS: '" + kv.substr(0, eq) + "' is a test hook of the library, not an option");
                if (sfa_set_option(ctxs[j], kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1)) != SFA_OK) die(std::string("SFA_OPTS: ") + sfa_last_error());
                p = q + 1;
            }
        }
    }

    t_init[2] = realtime() - ti;
    ti = realtime();
    if (!selected.get()) die(reader.error());
    const double t_select = realtime() - ti;  // what of the walk the initialisation did not hide
    if (o.verbosity >= 4 && (o.shard_n > 1 || o.range_a > 0)) fprintf(stderr, "[dtw_main::%.3f] waited %.3f s more for this process's part of the file\n", realtime() - t0, t_select);
    if (o.verbosity >= 4)
        fprintf(stderr, "[dtw_main::%.3f] initialised: input %.3f s, model + reference events %.3f s, %d device context(s) %.3f s\n", realtime() - t0,
                t_init[0], t_init[1], n_ctx, t_init[2]);

    if ((o.flag & F_SAM) && !o.no_header) {  // sam_hdr_wr(), src/dtw_main.c:118-123 (LN is the k-mer count, as the reference prints it)
        for (int32_t i = 0; i < nref; ++i) fprintf(stdout, "@SQ\tSN:%s\tLN:%ld\n", contigs[i].name.c_str(), static_cast<long>(ref_len[i]));
        fprintf(stdout, "@PG\tID:sigfish\tPN:sigfish\tVN:0.2.0\n");
    }

    // ---- batch loop, src/dtw_main.c:299-326, as a pipeline over four slots: while the GPU stages of batches i and i-1
    // (one per context) and the output of batch i-2 run on helper threads, the main thread loads and pre-processes
    // batch i+1.  Batches are printed strictly in order, so the output is the same as the serial loop's. ----
    double t_load = 0, t_proc = 0, t_dtw = 0, t_out = 0;
    double t_wait_gpu = 0, t_wait_out = 0, t_pin = 0;  // main thread: waiting for a context / for the printer; page-locked staging (re)allocation
    // --profile-cpu=yes (src/dtw_main.c:213-214, src/sigfish.c:1021-1040): the stages of a batch run one after the other,
    // each under its own timer, and the batches are not overlapped.  Host stages are wall time of their fan-out over -t
    // threads, as in the reference; stages that run on the device are the device's own time (HIP events, sfa_get_profile).
    const bool prf = (o.flag & F_PRF) != 0;
    double t_parse = 0, t_events = 0, t_norm = 0, t_dtw_stage = 0;
    WorkerPool pool(o.threads);  // -t host threads, alive for the whole run
    int64_t total = 0, prefix_fail = 0, ignored = 0, too_short = 0, sum_bytes = 0;
    struct Slot {
        std::vector<Read> reads;
        std::vector<const sfa_event_t *> evp;
        std::vector<int64_t> nev, qs, qe;
        std::vector<sfa_result_t> rows;
        // device-side event detection: concatenated raw samples + scaling instead of event tables
        int16_t *raw = nullptr;  // page-locked (sfa_pinned_alloc), grown on demand
        size_t raw_cap = 0;
        std::vector<int64_t> raw_off;
        std::vector<double> scaling;
        std::vector<sfa_query_info_t> info;
        std::vector<sfa_event_t> qev;  // [n][query] event tables of the query windows (SAM with device-side events)
        // device-side record decoding: the records' bytes as they are in the file, back to back, page-locked
        uint8_t *rec_bytes = nullptr;
        size_t rec_cap = 0;
        std::vector<int64_t> rec_off;
        std::vector<sfa_read_head_t> heads;
        int32_t n = 0;
        int64_t bytes = 0;
        bool via_device = false;  // this batch's records go to the device as they are in the file (sfa_align_blow5)
    };
    // events on the GPU unless the RNA auto prefix is asked for (adaptor/poly-A detection stays on the host); for SAM the
    // event tables of the query windows come back from the device with the rows
    const bool gpu_events = !o.host_events && o.prefix >= 0;
    // ... and so can the records themselves: inflate, field parsing and signal decoding on the device (sfa_align_blow5), the host
    // only framing the records and copying their bytes into page-locked staging.  Measured on one GPU with 16 host threads
    // (profiles/r02_logs/e2e_compressed_streams_x_batch.log): the two routes are level, 0.49-0.52 M reads/s from a compressed
    // file -- 16 cores inflate 0.75 M records/s, the device 1.0 M/s but in competition with the alignment kernels for the same
    // LDS -- and the host route is the better one at the default -K 4096.  What the device route buys is independence from the
    // host: a node's cores do not grow with its GPUs, so it is the default from three devices on.
    // (the device route decodes BLOW5 records; the lines of a SLOW5 ASCII file are parsed by the host threads)
    if (reader.ascii() && o.gpu_parse == 1) die("--gpu-parse decodes BLOW5 records: a SLOW5 ASCII file is parsed on the host threads");
    const bool gpu_parse = gpu_events && !reader.ascii() && (o.gpu_parse < 0 ? o.devices.size() > 2 : o.gpu_parse == 1);
    const bool sam = (o.flag & F_SAM) != 0;
    const int n_slots = n_ctx + 2;  // one being filled, one per GPU stage in flight, one being printed
    std::vector<Slot> slots(n_slots);
    for (Slot &sl : slots) {
        sl.reads.resize(o.batch_size);
        sl.evp.resize(o.batch_size);
        sl.nev.resize(o.batch_size);
        sl.qs.resize(o.batch_size);
        sl.qe.resize(o.batch_size);
        sl.rows.resize(o.batch_size);
    }
    std::mutex stat_mu;  // two GPU stages may finish together
    auto align = [&](Slot &sl, sfa_ctx_t *ctx) {
        const int32_t n = sl.n;
        std::vector<sfa_result_t> &rows = sl.rows;
        const double a = realtime();
        if (sl.via_device) {
            sl.info.resize(n);
            sl.heads.resize(n);
            if (sam) sl.qev.resize(static_cast<size_t>(n) * o.query);
            if (n > 0 && sfa_align_blow5(ctx, sl.rec_bytes, sl.rec_off.data(), n, reader.records_zlib(), reader.signal_svb(), o.prefix, o.query,
                                         rows.data(), sl.info.data(), sl.heads.data(), sam ? sl.qev.data() : nullptr) != SFA_OK)
                die(std::string("alignment failed: ") + sfa_last_error());
        } else if (gpu_events) {
            sl.info.resize(n);
            if (sam) sl.qev.resize(static_cast<size_t>(n) * o.query);
            if (n > 0 && sfa_align_raw_ex(ctx, sl.raw, sl.raw_off.data(), sl.scaling.data(), n, o.prefix, o.query, rows.data(),
                                          sl.info.data(), sam ? sl.qev.data() : nullptr) != SFA_OK)
                die(std::string("alignment failed: ") + sfa_last_error());
        } else if (n > 0 && sfa_align_events(ctx, sl.evp.data(), sl.nev.data(), sl.qs.data(), sl.qe.data(), n, rows.data()) != SFA_OK) {
            die(std::string("alignment failed: ") + sfa_last_error());
        }
        sfa_profile_t pr{};
        if (prf && n > 0 && sfa_get_profile(ctx, &pr) != SFA_OK) die(std::string("sfa_get_profile failed: ") + sfa_last_error());
        std::lock_guard<std::mutex> lock(stat_mu);
        t_dtw += realtime() - a;
        if (prf) {
            if (gpu_events) {  // events and normalisation ran on the device, inside the same call
                t_parse += pr.decode_ms * 1e-3;  // ... and so did parse_single's work, when the records went up as they are
                t_events += pr.events_ms * 1e-3;
                t_norm += pr.normalise_ms * 1e-3;
                t_dtw_stage += pr.total_ms * 1e-3;
            } else {
                t_dtw_stage += realtime() - a;  // the reference's timer around align_db (src/sigfish.c:1037-1040)
            }
        }
        if (gpu_events)
            for (int32_t i = 0; i < n; ++i) {
                ignored += (sl.info[i].status & 2) != 0;
                too_short += (sl.info[i].status & 1) != 0;
            }
        if (o.verbosity >= 4)
            fprintf(stderr, "[dtw_main::%.3f*%.2f] %d Entries (%.1fM bytes) processed\n", realtime() - t0, cputime() / (realtime() - t0), n, sl.bytes / 1e6);
    };
    auto output = [&](Slot &sl) {
        const int32_t n = sl.n;
        std::vector<Read> &batch = sl.reads;
        std::vector<sfa_result_t> &rows = sl.rows;
        const double a = realtime();
        if (o.flag & F_SAM) {
            // the warp path of every winner is rebuilt on the host from its band (sam.hpp), one read per task
            std::vector<std::string> sam_rows(n);
            pool.run(n, [&](int64_t i) {
                const Read &r = batch[i];
                const sfa_result_t &row = rows[i];
                if (!row.valid || row.rid < 0 || (!gpu_events && !r.keep)) return;
                // the event table and the window inside it: the read's own (host events) or the window alone (device events)
                const sfa_event_t *ev = gpu_events ? sl.qev.data() + static_cast<size_t>(i) * o.query : r.ev.data();
                const int64_t qs = gpu_events ? 0 : r.qstart, qe = gpu_events ? sl.info[i].qend - sl.info[i].qstart : r.qend;
                const float *y = row.strand == '+' ? fwd[row.rid].data() : rev[row.rid].data();
                std::string buf(1 << 16, '\0');
                const char *rid = sl.via_device ? sl.heads[i].read_id : r.rec.read_id.c_str();
                int len = sfa_sam_row(&buf[0], buf.size(), &row, rid, contigs[row.rid].name.c_str(), ev, qs, qe, y,
                                      ref_len[row.rid], ref_off[row.rid], o.flag);
                if (len == SFA_ERANGE) {  // very long ss strings (full-reference alignments)
                    buf.assign(1 << 22, '\0');
                    len = sfa_sam_row(&buf[0], buf.size(), &row, rid, contigs[row.rid].name.c_str(), ev, qs, qe, y,
                                      ref_len[row.rid], ref_off[row.rid], o.flag);
                }
                if (len > 0) sam_rows[i].assign(buf.data(), len);
            });
            for (int32_t i = 0; i < n; ++i) fwrite(sam_rows[i].data(), 1, sam_rows[i].size(), stdout);
        } else {
            std::string line(4096, '\0');
            for (int32_t i = 0; i < n; ++i) {  // output_db + aln_to_str, src/sigfish.c:796-826,1051-1086
                const Read &r = batch[i];
                if (!rows[i].valid || rows[i].rid < 0) continue;
                uint64_t start_raw, end_raw, qsize;
                if (gpu_events) {
                    start_raw = sl.info[i].start_raw_idx;
                    end_raw = sl.info[i].end_raw_idx;
                    qsize = static_cast<uint64_t>((sl.info[i].qend - 1) - sl.info[i].qstart);
                } else {
                    if (!r.keep) continue;
                    const sfa_event_t &e0 = r.ev[r.qstart], &e1 = r.ev[r.qend - 1];
                    start_raw = e0.start;
                    end_raw = static_cast<uint64_t>(static_cast<float>(e1.start) + e1.length);  // u64 + float, as in C
                    qsize = static_cast<uint64_t>((r.qend - 1) - r.qstart);
                }
                const char *rid = sl.via_device ? sl.heads[i].read_id : r.rec.read_id.c_str();
                const uint64_t n_raw = sl.via_device ? static_cast<uint64_t>(sl.heads[i].n_samples) : r.rec.raw.size();
                const int len = sfa_paf_row(&line[0], line.size(), &rows[i], rid, contigs[rows[i].rid].name.c_str(),
                                            start_raw, end_raw, qsize, n_raw, static_cast<uint64_t>(seq_len[rows[i].rid]));
                if (len < 0) die("PAF line too long");
                fwrite(line.data(), 1, len, stdout);
            }
        }
        fflush(stdout);
        t_out += realtime() - a;
    };

    std::vector<std::future<void>> gpu_pending(n_ctx);  // GPU stage per context
    std::future<void> out_pending;                      // output stage
    int64_t bi = 0;  // batch index; batch bi lives in slot bi % n_slots and runs on context bi % n_ctx
    int32_t counter = 0;
    bool more = true;
    std::unique_ptr<FrameLoader> loader;  // (mapped files; anything else is read record by record below)
    if (reader.mapped()) loader.reset(new FrameLoader(reader, o.batch_size, o.batch_bytes));
    while (more) {
        Slot &sl = slots[bi % n_slots];
        std::vector<Read> &batch = sl.reads;
        double a = realtime();
        int32_t n = 0;
        int64_t bytes = 0;
        if (loader) {
            const Frames &f = loader->next();
            if (f.failed) die(reader.error());
            n = f.n;
            bytes = f.bytes;
            more = f.more;
            for (int32_t i = 0; i < n; ++i) {
                batch[i].view = f.view[i];
                batch[i].view_size = f.size[i];
            }
        }
        while (!loader && n < o.batch_size && bytes < o.batch_bytes) {
            int rc = reader.next_view(&batch[n].view, &batch[n].view_size);
            if (rc == -2) {  // not mappable: copy the record
                batch[n].view = nullptr;
                rc = reader.next_mem(&batch[n].mem);
                batch[n].view_size = batch[n].mem.size();
            }
            if (rc < 0) die(reader.error());
            if (rc == 0) {
                more = false;
                break;
            }
            bytes += static_cast<int64_t>(batch[n].view_size);
            ++n;
        }
        sl.n = n;
        sl.bytes = bytes;
        sl.via_device = gpu_parse;
        t_load += realtime() - a;
        if (o.verbosity >= 4)
            fprintf(stderr, "[dtw_main::%.3f*%.2f] %d Entries (%.1fM bytes) loaded\n", realtime() - t0, cputime() / (realtime() - t0), n, bytes / 1e6);
        a = realtime();
        std::atomic<int> bad(0);
        auto parse_one = [&](int64_t i) {  // parse_single, src/sigfish.c:317-328
            Read &r = batch[i];
            std::string perr;
            if (!(r.view ? reader.parse(r.view, r.view_size, &r.rec, &perr) : reader.parse(r.mem, &r.rec, &perr))) bad = 1;
            r.keep = false;
            r.ev.clear();
            r.status = 0;
        };
        auto events_one = [&](int64_t i, std::vector<float> &pa) {  // event_single, src/sigfish.c:330-378
            Read &r = batch[i];
            const int64_t ns = static_cast<int64_t>(r.rec.raw.size());
            pa.resize(ns);
            sfa::raw_to_picoamps(r.rec.raw.data(), ns, r.rec.digitisation, r.rec.offset, r.rec.range, pa.data());
            r.ev = sfa::detect_events(pa.data(), ns, rna);
        };
        auto normalise_one = [&](int64_t i, const std::vector<float> &pa) {  // normalise_single, src/sigfish.c:424-505
            Read &r = batch[i];
            if (!r.ev.empty())
                r.keep = sfa::select_and_normalise(r.ev, r.rec.raw.data(), static_cast<int64_t>(r.rec.raw.size()), pa.data(), o.prefix, o.query,
                                                   o.flag, o.pore_flag, &r.qstart, &r.qend, &r.status);
        };
        if (sl.via_device) {  // nothing to parse here: the records go to the device as they are
            double b = realtime();
            sl.rec_off.resize(n + 1);
            sl.rec_off[0] = 0;
            for (int32_t i = 0; i < n; ++i) sl.rec_off[i + 1] = sl.rec_off[i] + static_cast<int64_t>(batch[i].view_size);
            const size_t need = static_cast<size_t>(sl.rec_off[n]) + 64;
            if (need > sl.rec_cap) {
                sfa_pinned_free(sl.rec_bytes);
                sl.rec_cap = need + need / 4;
                sl.rec_bytes = static_cast<uint8_t *>(sfa_pinned_alloc(sl.rec_cap));
                if (!sl.rec_bytes) die(std::string("cannot allocate the record staging buffer: ") + sfa_last_error());
            }
            pool.run(n, [&](int64_t i) {
                const uint8_t *src = batch[i].view ? batch[i].view : batch[i].mem.data();
                memcpy(sl.rec_bytes + sl.rec_off[i], src, static_cast<size_t>(sl.rec_off[i + 1] - sl.rec_off[i]));
            });
            t_parse += realtime() - b;
        } else if (!prf) {  // one fan-out per batch, every read through all its host stages (work_per_single_read, src/sigfish.c:995-1001)
            auto rest_of = [&](int64_t i) {
                if (bad || gpu_events || batch[i].rec.raw.empty()) return;
                std::vector<float> pa;
                events_one(i, pa);
                normalise_one(i, pa);
            };
            // reads go two at a time: their zlib streams are inflated side by side by one thread (blow5.hpp: parse_pair)
            pool.run((n + 1) / 2, [&](int64_t j) {
                const int64_t i0 = 2 * j, i1 = 2 * j + 1;
                if (i1 >= n) {
                    parse_one(i0);
                } else {
                    Read &a0 = batch[i0], &a1 = batch[i1];
                    const uint8_t *const mem[2] = {a0.view ? a0.view : a0.mem.data(), a1.view ? a1.view : a1.mem.data()};
                    const size_t size[2] = {a0.view_size, a1.view_size};
                    sfa::Blow5Record *const rec[2] = {&a0.rec, &a1.rec};
                    std::string e0, e1;
                    std::string *const perr[2] = {&e0, &e1};
                    bool ok[2];
                    reader.parse_pair(mem, size, rec, perr, ok);
                    if (!ok[0] || !ok[1]) bad = 1;
                    for (Read *r : {&a0, &a1}) {
                        r->keep = false;
                        r->ev.clear();
                        r->status = 0;
                    }
                    rest_of(i1);
                }
                rest_of(i0);
            });
        } else {  // stage by stage, each under its timer
            double b = realtime();
            pool.run(n, parse_one);
            t_parse += realtime() - b;
            if (!gpu_events && !bad) {
                b = realtime();
                pool.run(n, [&](int64_t i) {
                    if (!batch[i].rec.raw.empty()) events_one(i, batch[i].pa);
                });
                t_events += realtime() - b;
                b = realtime();
                pool.run(n, [&](int64_t i) {
                    if (!batch[i].rec.raw.empty()) normalise_one(i, batch[i].pa);
                    std::vector<float>().swap(batch[i].pa);
                });
                t_norm += realtime() - b;
            }
        }
        if (bad) die("error parsing a BLOW5 record");
        if (gpu_events && !sl.via_device) {  // pack the samples of the batch for one upload
            sl.raw_off.resize(n + 1);
            sl.scaling.resize(3 * static_cast<size_t>(n));
            sl.raw_off[0] = 0;
            for (int32_t i = 0; i < n; ++i) {
                sl.raw_off[i + 1] = sl.raw_off[i] + static_cast<int64_t>(batch[i].rec.raw.size());
                sl.scaling[3 * i] = batch[i].rec.digitisation;
                sl.scaling[3 * i + 1] = batch[i].rec.offset;
                sl.scaling[3 * i + 2] = batch[i].rec.range;
            }
            const size_t need = static_cast<size_t>(sl.raw_off[n]) + 1;
            if (need > sl.raw_cap) {
                const double pa = realtime();
                sfa_pinned_free(sl.raw);
                sl.raw_cap = need + need / 4;
                sl.raw = static_cast<int16_t *>(sfa_pinned_alloc(sl.raw_cap * sizeof(int16_t)));
                t_pin += realtime() - pa;
                if (!sl.raw) die(std::string("cannot allocate the sample staging buffer: ") + sfa_last_error());
            }
            pool.run(n, [&](int64_t i) {
                memcpy(sl.raw + sl.raw_off[i], batch[i].rec.raw.data(), sizeof(int16_t) * batch[i].rec.raw.size());
            });
        }
        for (int32_t i = 0; i < n && !gpu_events; ++i) {
            const Read &r = batch[i];
            sl.evp[i] = r.keep ? r.ev.data() : nullptr;
            sl.nev[i] = r.keep ? static_cast<int64_t>(r.ev.size()) : 0;
            sl.qs[i] = r.qstart;
            sl.qe[i] = r.qend;
            prefix_fail += (r.status & 4) != 0;
            ignored += (r.status & 2) != 0;
            too_short += (r.status & 1) != 0;
        }
        t_proc += realtime() - a;
        a = realtime();
        if (gpu_pending[bi % n_ctx].valid()) gpu_pending[bi % n_ctx].get();  // batch bi-n_ctx has its rows, its context is free
        t_wait_gpu += realtime() - a;
        a = realtime();
        if (out_pending.valid()) out_pending.get();  // batch bi-n_ctx-1 is printed (its slot is filled next)
        t_wait_out += realtime() - a;
        if (bi >= n_ctx) {
            Slot *done = &slots[(bi - n_ctx) % n_slots];
            out_pending = std::async(std::launch::async, [&output, done] { output(*done); });
        }
        {
            Slot *mine = &sl;
            sfa_ctx_t *c = ctxs[bi % n_ctx];
            gpu_pending[bi % n_ctx] = std::async(std::launch::async, [&align, mine, c] { align(*mine, c); });
            if (prf) gpu_pending[bi % n_ctx].get();  // sectional mode: nothing of the next batch starts before this one is through
        }
        ++bi;
        total += n;
        sum_bytes += bytes;
        if (o.debug_break == counter) break;
        ++counter;
    }
    if (out_pending.valid()) out_pending.get();
    for (int64_t b = std::max<int64_t>(bi - n_ctx, 0); b < bi; ++b) {  // the last batches, in order
        if (gpu_pending[b % n_ctx].valid()) gpu_pending[b % n_ctx].get();
        output(slots[b % n_slots]);
    }
    for (Slot &sl : slots) {
        sfa_pinned_free(sl.raw);
        sfa_pinned_free(sl.rec_bytes);
    }
    if (o.verbosity >= 3 && prf) {  // the reference's lines, src/dtw_main.c:331-343
        fprintf(stderr, "[dtw_main] total entries: %ld\tprefix fail: %ld\tignored: %ld\ttoo short: %ld", (long)total, (long)prefix_fail, (long)ignored, (long)too_short);
        fprintf(stderr, "\n[dtw_main] total bytes: %.1f M", sum_bytes / 1e6);
        fprintf(stderr, "\n[dtw_main] Data loading time: %.3f sec", t_load);
        fprintf(stderr, "\n[dtw_main] Data processing time: %.3f sec", t_proc + t_dtw);
        fprintf(stderr, "\n[dtw_main]     - Parse time: %.3f sec", t_parse);
        fprintf(stderr, "\n[dtw_main]     - Events time: %.3f sec", t_events);
        fprintf(stderr, "\n[dtw_main]     - Normalise time: %.3f sec", t_norm);
        fprintf(stderr, "\n[dtw_main]     - DTW time: %.3f sec", t_dtw_stage);
        fprintf(stderr, "\n[dtw_main] Data output time: %.3f sec\n", t_out);
    } else if (o.verbosity >= 3) {
        fprintf(stderr, "[dtw_main] total entries: %ld\tprefix fail: %ld\tignored: %ld\ttoo short: %ld\n", (long)total, (long)prefix_fail, (long)ignored, (long)too_short);
        fprintf(stderr, "[dtw_main] total bytes: %.1f M\n[dtw_main] Data loading time: %.3f sec\n", sum_bytes / 1e6, t_load);
        fprintf(stderr, "[dtw_main] Data processing time: %.3f sec (host stages) + %.3f sec (DTW stage, overlapped with the next batch)\n",
                t_proc, t_dtw);
        fprintf(stderr, "[dtw_main] Data output time: %.3f sec\n", t_out);
        if (o.verbosity >= 4)
            fprintf(stderr, "[dtw_main] main thread waited %.3f sec for a free device context and %.3f sec for the printer; page-locked staging (re)allocated in %.3f sec (inside the host stages)\n",
                    t_wait_gpu, t_wait_out, t_pin);
    }
    if (o.verbosity >= 4) fprintf(stderr, "[dtw_main::%.3f] all output written; releasing the device\n", realtime() - t0);
    g_verbosity = o.verbosity;
    return 0;
}

int eval_main(int argc, char **argv);  // eval_main.cpp

int main(int argc, char **argv) {
    if (argc >= 2 && (!strcmp(argv[1], "--version") || !strcmp(argv[1], "-V"))) {
        fprintf(stdout, "sigfish-amd %s\n", sfa_version());
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "dtw")) return dtw_main(argc - 1, argv + 1);
    if (argc >= 2 && !strcmp(argv[1], "eval")) return eval_main(argc - 1, argv + 1);
    fprintf(argc >= 2 && (!strcmp(argv[1], "--help") || !strcmp(argv[1], "-h")) ? stdout : stderr,
            "Usage: sigfish-amd <command> [options]\n\ncommand:\n         dtw           map raw signal reads to a reference with subsequence DTW on an MI355X\n         eval          compare a test PAF with a truth PAF (mapping accuracy)\n\n");
    return (argc >= 2 && (!strcmp(argv[1], "--help") || !strcmp(argv[1], "-h"))) ? 0 : 1;
}
