// gpsx_steps.cpp -- the reference's step-level entry points on top of the batched engine ("Tier 2"):
//   acquisition_process / acquisition_start_* / acquisition_get_hist      PM/GPS/acquisition.c, acquisition.h:7-12
//   gps_tracking_process                                                   PM/GPS/tracking.c,    tracking.h:6
// The data-parallel part of every step (replica, wipe-off, correlation, search) runs on the GPU through
// gpsx_acq_jobs / gpsx_track_epl_batch / gpsx_rewind; what remains here is the reference's serial decision logic
// (best-phase voting, histograms, DLL/PLL/FLL float loops), restated with its exact integer widths and float32
// expression order so that channel state evolves identically (tests/test_gpu_steps.py replays golden traces).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <initializer_list>
#include <sched.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <climits>
#include <dlfcn.h>
#include <thread>
#include <vector>

#include "../../include/gpsx_compat.h"
#include "gpsx_compat_internal.hpp"
#include "gpsx_libm.hpp"   // the reference build's float arctangents (glibc <= 2.40 = fdlibm), whatever libm this host has

namespace {

// gps_misc.h:16 defines M_PI as the float 3.1415926535f, but tracking.c includes <math.h> AFTER gps_misc.h, whose
// double M_PI replaces it (the compiler's "M_PI redefined" warning): every pi in the tracking loops is a double and
// the float operands around it are promoted, then the result is rounded back to float on assignment.
constexpr double kPi = 3.14159265358979323846;
constexpr int kBinLength = 10;                       // ACQ_SINGLE_FREQ_LENGTH, acquisition.c:18
constexpr int kBinCapacity = 25;                     // FREQ_SEARCH_POINTS_MAX_CNT, acquisition.c:12
constexpr uint32_t kCodeSearchTimeoutMs = 120000;    // acquisition.c:13
constexpr int kStage2Width = 500, kStage3Width = 60; // acquisition.c:15-16
constexpr int kPreTrackZone = 30;                    // tracking.c:17
constexpr int kPreTrackStep = kPreTrackZone / TRACKING_CH_LENGTH;   // 7 correlations per ms
constexpr int kFineRatio = 8;                        // tracking.c:23
constexpr int kSnrLength = 200;                      // tracking.c:26
constexpr int kPllBadThreshold = 80;                 // tracking.c:14
constexpr int kFullRange = 2 * PRN_LENGTH;

uint32_t g_ticks = 0;

// frequency-search bookkeeping shared by all channels, exactly as the reference shares it (acquisition.c:28-33)
uint32_t g_freq_votes[ACQ_COUNT];
uint16_t g_bin_phases[kBinCapacity];
uint8_t g_bin_count = 0;

// State the reference keeps in file/function statics because only ONE channel is served per 4 ms time slot:
// pre-tracking's running best of the slot (tracking.c:33-34) and the nav-bit hook's slot buffers (nav_data.c:27,48-51).
// The reference-named entry points share one instance, exactly like the reference; the batched step
// (gps_tracking_process_batch, every channel served every millisecond) gives each channel its own.
struct SlotState {
  uint16_t best_value = 0;
  uint16_t best_phase = 0;
  uint32_t start_ticks = 0;
  int16_t ip[TRACKING_CH_LENGTH] = {0, 0, 0, 0};
  uint8_t bits[TRACKING_CH_LENGTH] = {0, 0, 0, 0};
  // the FLL's arctangent of the previous millisecond's prompt pair: this millisecond's "before" is last millisecond's "now"
  // (same inputs, same function: the value is reused, not recomputed; the inputs are kept to prove it)
  float fll_now = 0.0f;
  int16_t fll_now_i = 0, fll_now_q = 0;
  bool fll_now_valid = false;
};
SlotState g_shared_slot;

// The millisecond tick as the tracking-side logic reads it.  Inside gps_tracking_process_batch the host's time source
// (signal_capture_get_packet_cnt, a weak hook the host may override with anything -- a ctypes callback, a non-atomic
// counter) is read ONCE, on the calling thread, and every channel of the step sees that value: worker threads never call
// into the host.  Outside a batched step it is the hook itself, call by call, as the reference reads it.
std::atomic<bool> g_batch_tick_valid{false};
uint32_t g_batch_tick = 0;
thread_local bool t_tick_valid = false;   // gps_tracking_words_batch: a worker replays the milliseconds of a launch one by one
thread_local uint32_t t_tick = 0;
inline uint32_t tick()
{
  if (t_tick_valid)
    return t_tick;
  return g_batch_tick_valid.load(std::memory_order_acquire) ? g_batch_tick : signal_capture_get_packet_cnt();
}
struct BatchTick {
  explicit BatchTick(uint32_t now)
  {
    g_batch_tick = now;
    g_batch_tick_valid.store(true, std::memory_order_release);
  }
  ~BatchTick() { g_batch_tick_valid.store(false, std::memory_order_release); }
};

// true when `fn` (the address of one of this library's weak hooks, taken through the GOT) resolves outside this library:
// the host overrode it
bool resolves_outside_this_library(const void *fn)
{
  static const int anchor = 0;
  Dl_info theirs, ours;
  if (!dladdr(fn, &theirs) || !dladdr(&anchor, &ours))
    return false;
  return theirs.dli_fbase != ours.dli_fbase;
}

// ---- the batched step's host workers (gps_tracking_process_batch) ---------------------------------------------------------
constexpr int kStepThreadsFrom = 2048;    // channels from which the per-channel host loops are spread over worker threads
                                          // ($GPSX_STEP_THREADS_FROM lowers it: tests put the 64-channel reference trace on workers)
constexpr int kStepOverlapFrom = 65536;   // tracked channels from which those loops overlap the correlators, piece by piece

// the cache lines of a channel record the batched step touches: tracking_data (offset 60, 152 bytes) and, after the
// correlators, the head of nav_data behind it
inline void prefetch_channel(const gps_ch_t &ch, bool with_nav)
{
  const char *p = reinterpret_cast<const char *>(&ch.tracking_data);
  __builtin_prefetch(p, 1);
  __builtin_prefetch(p + 64, 1);
  __builtin_prefetch(p + 128, 1);
  if (with_nav) {
    __builtin_prefetch(p + 192, 1);
    __builtin_prefetch(p + 256, 1);
  }
}

struct WorkerLists {   // what one worker's contiguous channel range contributes to the step's work lists
  std::vector<gpsx_acq_job_t> jobs;
  std::vector<gpsx_trk_state_t> st;
  std::vector<uint8_t> skipped;
  int job_base = 0, st_base = 0;
  bool any_skipped = false;
  // overlapped step: this worker's tracked channels go to the GPU as kRuns runs, run q inside piece q of the launch
  // (entries [run_start[q], run_start[q + 1]) of `st` at run_base[q] of the step's arrays); `cursor` = the next channel
  // whose loops have not run yet, `cursor_run` = the run its entry is in
  static constexpr int kMaxRuns = 16;
  int run_start[kMaxRuns + 1] = {0}, run_base[kMaxRuns] = {0};
  int cursor = 0, cursor_run = 0;
  // channels of this worker's range that stopped in front of a false-lock reseed (channel, entry of the step's arrays)
  std::vector<std::pair<int, size_t>> deferred;
};

// A fixed set of threads that run `fn(job)` for job = 0..n-1 and meet again.  Every thread takes the job of its own number
// first (the caller: 0) -- a job is a contiguous channel range, and a range's records then live in one core's caches from
// millisecond to millisecond -- and then whatever job nobody has started: on a shared host a thread is now and then not
// scheduled for milliseconds after its wake-up, and its range must not wait for it.  Sized once, from
// the CPUs the creating thread may run on ($GPSX_STEP_THREADS overrides; at most 64), so a process that pinned itself next to
// its GPU (gpsx_bind_thread_to_device) gets workers on those cores.  Between steps the workers spin briefly -- the next
// millisecond is never far -- then sleep on the generation counter (a futex).
class StepPool {
 public:
  static StepPool &instance()
  {
    static StepPool pool;
    return pool;
  }
  int size() const { return n_; }
  // while set, waiting workers keep spinning instead of going to sleep: for a caller whose runs follow each other within a
  // few tens of microseconds (the pieces of an overlapped step) and who clears it before the long wait for the next millisecond
  void keep_hot(bool on) { hot_.store(on, std::memory_order_relaxed); }
  template <typename F>
  void run(int n_workers, F &&fn)
  {
    if (n_workers <= 1) {
      fn(0);
      return;
    }
    start_threads();
    std::function<void(int)> f = std::ref(fn);
    // (what a generation runs sits in the slot of its parity: a thread that looks late, while the NEXT run is being posted,
    //  still reads its own generation's function and job count, or sees the counter move and looks again)
    const unsigned gen = generation_.load(std::memory_order_relaxed) + 1;   // (only run() moves the counter)
    slot_[gen & 1].job.store(&f, std::memory_order_release);
    slot_[gen & 1].active.store(n_workers, std::memory_order_release);
    pending_.store(n_workers, std::memory_order_relaxed);
    generation_.fetch_add(1);
    // one system call wakes every sleeper, and none of them takes a lock on its way out (a condition variable hands its mutex
    // from thread to thread: thirteen wake-ups in a row at the head of every phase of every millisecond)
    if (sleepers_.load() != 0)   // (sequentially consistent with the sleeper's own "count myself, look again")
      futex(FUTEX_WAKE_PRIVATE, INT_MAX);
    take_jobs(f, 0, n_workers, gen);
    for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; spin++)
      if (spin > 64)
        std::this_thread::yield();
    slot_[gen & 1].job.store(nullptr, std::memory_order_relaxed);
  }

 private:
  StepPool()
  {
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0)
      n = CPU_COUNT(&set);
    if (n <= 0)
      n = (int)std::thread::hardware_concurrency();
    // A container's CPU quota (cgroup cpu.max / cfs_quota_us): workers that spin between steps burn it, and a group that
    // exhausts its quota is frozen until the next 100 ms period -- measured on the GPU box (16 CPUs of quota): 64 workers = a 75 ms stall
    // every 100 ms.
    const double quota = cgroup_cpu_quota();
    if (quota > 0.0 && n > (int)quota - 2)
      n = (int)quota - 2;   // (two CPUs' worth left for the HIP runtime's own threads and the caller's other work)
    if (const char *e = std::getenv("GPSX_STEP_THREADS"))
      n = std::atoi(e);
    n_ = n < 1 ? 1 : (n > 64 ? 64 : n);
  }
  static double cgroup_cpu_quota()   // CPUs' worth of quota, 0 = unlimited / unknown
  {
    if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota|max> <period>"
      char q[32] = {0};
      long period = 0;
      const int got = std::fscanf(f, "%31s %ld", q, &period);
      std::fclose(f);
      if (got == 2 && period > 0 && std::strcmp(q, "max") != 0)
        return std::atof(q) / (double)period;
      return 0.0;
    }
    long quota = -1, period = 0;
    if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
      if (std::fscanf(f, "%ld", &quota) != 1) quota = -1;
      std::fclose(f);
    }
    if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(f, "%ld", &period) != 1) period = 0;
      std::fclose(f);
    }
    return quota > 0 && period > 0 ? (double)quota / (double)period : 0.0;
  }
  ~StepPool()
  {
    quit_.store(true);
    generation_.fetch_add(1);
    futex(FUTEX_WAKE_PRIVATE, INT_MAX);
    for (std::thread &t : threads_)
      t.join();
  }
  // job `own` if nobody has it yet, then every other job nobody has started (claimed[j] == gen: taken in this generation)
  void take_jobs(std::function<void(int)> &f, int own, int n_jobs, unsigned gen)
  {
    for (int i = 0; i < n_jobs; i++) {
      const int j = own + i < n_jobs ? own + i : own + i - n_jobs;
      unsigned c = claimed_[j].load(std::memory_order_relaxed);
      // (only forward: a thread that was held up with an older generation in hand must find nothing to take)
      if ((int)(gen - c) <= 0 || !claimed_[j].compare_exchange_strong(c, gen, std::memory_order_acq_rel))
        continue;
      f(j);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  long futex(int op, unsigned val)   // on the generation counter
  {
    static_assert(sizeof(std::atomic<unsigned>) == sizeof(unsigned), "futex word");
    return syscall(SYS_futex, reinterpret_cast<unsigned *>(&generation_), op, val, nullptr, nullptr, 0);
  }
  void start_threads()
  {
    if (!threads_.empty() || n_ <= 1)
      return;
    for (int w = 1; w < n_; w++)
      threads_.emplace_back([this, w] { worker(w); });
  }
  void worker(int w)
  {
    unsigned seen = 0;
    for (;;) {
      // wait for the next generation: spin briefly (the phases of a step follow each other within microseconds), then sleep --
      // a worker that spins through the GPU's part of the millisecond burns the CPU quota the step needs (measured on a
      // 16-CPU quota: 2000 pauses -> deadline misses at 65536 channels, 200 -> none)
      static const int kSpin = [] { const char *e = std::getenv("GPSX_STEP_SPIN"); return e ? std::atoi(e) : 200; }();
      for (int spin = 0; generation_.load(std::memory_order_acquire) == seen; spin++) {
        if (spin < kSpin || hot_.load(std::memory_order_relaxed)) {
          __builtin_ia32_pause();   // (the pool is x86-64 host code: the GPU boxes are EPYC)
          continue;
        }
        sleepers_.fetch_add(1);
        if (generation_.load() == seen)
          futex(FUTEX_WAIT_PRIVATE, seen);   // (returns at once if the counter has moved on)
        sleepers_.fetch_sub(1);
        spin = 0;
      }
      // (job and worker count of THE generation taken: a worker outside a run's count may look late, when the next is posted)
      std::function<void(int)> *job;
      int active;
      do {
        seen = generation_.load();
        // (acquire loads between two sequentially consistent loads of the counter: the re-check below cannot be satisfied
        //  by a counter value older than the slot contents just read -- the C++ memory model's guarantee, not only x86's)
        job = slot_[seen & 1].job.load(std::memory_order_acquire);
        active = slot_[seen & 1].active.load(std::memory_order_acquire);
      } while (generation_.load() != seen);
      if (quit_.load(std::memory_order_relaxed))
        return;
      // (a job is only entered through a claim of THIS generation, and none is left once the run that owns *job has returned)
      if (job && w < active)
        take_jobs(*job, w, active, seen);
    }
  }
  int n_ = 1;
  std::vector<std::thread> threads_;
  std::atomic<unsigned> generation_{0};
  std::atomic<int> pending_{0}, sleepers_{0};
  std::atomic<bool> hot_{false}, quit_{false};
  struct Posted {
    std::atomic<std::function<void(int)> *> job{nullptr};
    std::atomic<int> active{0};
  } slot_[2];
  std::atomic<unsigned> claimed_[64] = {};
};

struct StepBuffers {
  gpsx_trk_state_t *st = nullptr;
  int16_t *iq = nullptr;
  size_t cap = 0;
  static StepBuffers &instance()
  {
    static StepBuffers b;
    return b;
  }
  void reserve(gpsx_ctx *gx, size_t n)
  {
    if (n <= cap)
      return;
    if (st)
      (void)gpsx_host_free(gx, st);
    if (iq)
      (void)gpsx_host_free(gx, iq);
    st = nullptr;
    iq = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 64;
    void *a = nullptr, *b = nullptr;
    if (gpsx_host_alloc(gx, &a, want * sizeof(gpsx_trk_state_t)) != GPSX_OK || gpsx_host_alloc(gx, &b, want * 12) != GPSX_OK)
      gpsx_compat_die("gps_tracking_process_batch(page-locked staging)", GPSX_ENOMEM);
    st = static_cast<gpsx_trk_state_t *>(a);
    iq = static_cast<int16_t *>(b);
    cap = want;
  }
};

void reset_search_buffers()
{
  std::memset(g_freq_votes, 0, sizeof g_freq_votes);
  std::memset(g_bin_phases, 0, sizeof g_bin_phases);
  g_bin_count = 0;
}

void clear_phase_histogram(gps_acq_t &a) { std::memset(a.code_phase_histogram, 0, ACQ_PHASE1_HIST_SIZE); }

// narrowed code-phase window around the current estimate (acquisition.c:112-124,156-167)
void narrow_window(gps_acq_t &a, int width)
{
  a.code_search_start = (uint16_t)(a.found_code_phase - width / 2);
  a.code_search_stop = (uint16_t)(a.found_code_phase + width / 2);
  if (a.code_search_start > kFullRange)   // unsigned wrap below zero
    a.code_search_start = 0;
  if (a.code_search_stop > kFullRange)
    a.code_search_stop = kFullRange;
  a.code_hist_step = (uint16_t)(width / ACQ_PHASE1_HIST_SIZE + 1);
}

// ---- frequency search: votes of one Doppler bin (acquisition.c:322-416) ----------------------------------------
void vote_on_bin(gps_ch_t &ch, uint8_t count)
{
  std::sort(g_bin_phases, g_bin_phases + count);
  uint8_t run = 0;
  uint16_t longest = 0;
  uint8_t tight = 0;   // some neighbours in the run were closer than 3 half-chips
  for (uint8_t i = 1; i < count; i++) {
    const int16_t gap = (int16_t)((int16_t)g_bin_phases[i] - (int16_t)g_bin_phases[i - 1]);
    if (std::abs((int)gap) < 3)
      tight = 1;
    if (std::abs((int)gap) < 15) {
      run++;
    } else {
      if (run > longest && tight)
        longest = run;
      run = 0;
      tight = 0;
    }
  }
  if (run > longest)
    longest = run;
  if (longest >= 2)
    g_freq_votes[ch.acq_data.freq_index] += longest;
}

void judge_frequency_votes(gps_ch_t &ch)
{
  gps_acq_t &a = ch.acq_data;
  uint8_t nonzero = 0, best_at = 0, best = 0;
  for (uint8_t i = 0; i < ACQ_COUNT; i++) {
    if (g_freq_votes[i] > 0)
      nonzero++;
    if (g_freq_votes[i] > best) {
      best = (uint8_t)g_freq_votes[i];
      best_at = i;
    }
  }
  if (nonzero == 1 && best >= 3) {
    a.state = GPS_ACQ_FREQ_SEARCH_DONE;
    a.found_freq_offset_hz = (int16_t)(-ACQ_SEARCH_FREQ_HZ + best_at * ACQ_SEARCH_STEP_HZ);
    a.hist_ratio = 10.0f;
  } else if (nonzero > 1) {
    float worst = 10.0;
    for (uint8_t i = 0; i < ACQ_COUNT; i++) {
      if (g_freq_votes[i] > 0 && i != best_at) {
        const float r = (float)best / (float)g_freq_votes[i];
        if (r < worst)
          worst = r;
      }
    }
    if (worst > 1.7f) {
      a.hist_ratio = worst;
      a.state = GPS_ACQ_FREQ_SEARCH_DONE;
      a.found_freq_offset_hz = (int16_t)(-ACQ_SEARCH_FREQ_HZ + best_at * ACQ_SEARCH_STEP_HZ);
    }
  }
  if (a.state == GPS_ACQ_FREQ_SEARCH_DONE)
    std::printf("PRN=%d FINAL FREQ=%dHz\n", ch.prn, a.found_freq_offset_hz);
}

void after_frequency_search(gps_ch_t &ch, const gpsx_peak_t &pk)
{
  g_bin_phases[g_bin_count] = (uint16_t)pk.phase;
  g_bin_count++;
  if (g_bin_count >= kBinLength) {
    vote_on_bin(ch, g_bin_count);
    judge_frequency_votes(ch);
    reset_search_buffers();
    ch.acq_data.freq_index++;
    if (ch.acq_data.freq_index >= ACQ_COUNT)
      ch.acq_data.freq_index = 0;
  }
}

// ---- code-phase search: histogram of best phases (acquisition.c:211-274) -----------------------------------------
void after_code_phase_search(gps_ch_t &ch, const gpsx_peak_t &pk)
{
  gps_acq_t &a = ch.acq_data;
  const uint16_t best_phase = (uint16_t)pk.phase;
  if (best_phase < a.code_search_start || best_phase >= a.code_search_stop)
    return;
  const uint32_t waited = signal_capture_get_packet_cnt() - a.start_timestamp;
  if (waited > kCodeSearchTimeoutMs) {
    clear_phase_histogram(a);
    a.start_timestamp = signal_capture_get_packet_cnt();
  }
  const uint8_t bin = (uint8_t)((best_phase - a.code_search_start) / a.code_hist_step);
  if (bin < ACQ_PHASE1_HIST_SIZE)
    a.code_phase_histogram[bin]++;

  uint8_t top = 0, top_at = 0, populated = 0;
  const uint16_t bins = (uint16_t)((a.code_search_stop + 2 - a.code_search_start) / a.code_hist_step);
  for (uint8_t i = 0; i < bins; i++) {
    if (a.code_phase_histogram[i] > top) {
      top = a.code_phase_histogram[i];
      top_at = i;
    }
    if (a.code_phase_histogram[i] > 0)
      populated++;
  }
  if (top < 2)
    return;
  uint32_t sum = 0;
  uint8_t cnt = 0;
  for (uint8_t i = 0; i < ACQ_PHASE1_HIST_SIZE; i++) {
    if (a.code_phase_histogram[i] > 0) {
      sum += a.code_phase_histogram[i];
      cnt++;
    }
  }
  const float mean = (float)sum / (float)cnt;
  if (mean < 0.01f)
    return;
  float ratio = (float)top / mean;
  if (populated == 1 && top > 3)
    ratio = 10.0f;
  if (ratio > 3.2f) {
    a.found_code_phase = (uint16_t)(a.code_search_start + top_at * a.code_hist_step);
    if (a.state == GPS_ACQ_CODE_PHASE_SEARCH1)
      a.state = GPS_ACQ_CODE_PHASE_SEARCH1_DONE;
    if (a.state == GPS_ACQ_CODE_PHASE_SEARCH2)
      a.state = GPS_ACQ_CODE_PHASE_SEARCH2_DONE;
    if (a.state == GPS_ACQ_CODE_PHASE_SEARCH3)
      a.state = GPS_ACQ_CODE_PHASE_SEARCH3_DONE;
  }
}

bool in_code_phase_search(gps_acq_state_t s)
{
  return s == GPS_ACQ_CODE_PHASE_SEARCH1 || s == GPS_ACQ_CODE_PHASE_SEARCH2 || s == GPS_ACQ_CODE_PHASE_SEARCH3;
}

// ---- tracking loops (tracking.c:175-393) ------------------------------------------------------------------------
void dll_update(gps_ch_t &ch, int16_t IE, int16_t QE, int16_t IL, int16_t QL)
{
  gps_tracking_t &t = ch.tracking_data;
  const int32_t e2 = (int32_t)IE * (int32_t)IE + (int32_t)QE * (int32_t)QE;
  const int32_t l2 = (int32_t)IL * (int32_t)IL + (int32_t)QL * (int32_t)QL;
  const int32_t diff = e2 - l2;
  const int32_t total = e2 + l2;
  float err = (float)diff / (float)total;
  err = -err;
  const float dt = 0.001f;
  t.code_phase_fine += (TRACKING_DLL1_C1 * (err - t.dll_code_err) + TRACKING_DLL1_C2 * dt * err);

  const float span = (float)(PRN_LENGTH * 2 * kFineRatio);
  uint8_t wrapped = 0;
  if (t.code_phase_fine < 0.0f) {
    t.code_phase_fine = span - t.code_phase_fine;   // sic (tracking.c:356-361)
    wrapped = 1;
  } else if (t.code_phase_fine > span) {
    t.code_phase_fine = t.code_phase_fine - span;
    wrapped = 1;
  }
  if (wrapped) {
    t.code_phase_fine_filt = -1.0f;
  } else if (t.code_phase_fine_filt >= 0.0f) {
    t.code_phase_fine_filt += t.code_phase_fine;
    t.code_filt_cnt++;
  }
  t.dll_code_err = err;
}

void pll_update(gps_ch_t &ch, uint8_t index, int16_t IP, int16_t QP)
{
  gps_tracking_t &t = ch.tracking_data;
  if (index != 0)    // (the reference evaluates the arctangent first and returns here, tracking.c:178-190: it has no effect then)
    return;
  float phase_err;   // in units of pi
  if (IP > 0)
    phase_err = (float)((double)gpsx_libm::atan2f_fdlibm((float)QP, (float)IP) / kPi);
  else  // the reference calls the double-precision atan2 here (tracking.c:184)
    phase_err = (float)(atan2((double)(float)-QP, (double)(float)-IP) / kPi);
  float step = phase_err - t.pll_code_err;
  if ((double)step > kPi / 2)
    step = (float)(kPi - (double)step);
  if ((double)step < -kPi / 2)
    step = (float)(-kPi - (double)step);
  const float dt = 0.001f;
  if (ch.nav_data.period_sync_ok_flag)
    t.if_freq_offset_hz -= TRACKING_PLL2_C1 * step + (TRACKING_PLL2_C2 * dt * phase_err);
  else
    t.if_freq_offset_hz -= TRACKING_PLL1_C1 * step + (TRACKING_PLL1_C2 * dt * phase_err);
  t.pll_code_err = phase_err;
}

// The PLL's stability check in two parts (tracking.c:261-327).  false_lock_detect is the bookkeeping: it returns true when
// the channel has been flipping signs for long enough that its carrier must jump to a random offset (counters already
// cleared, as the reference clears them before it draws); false_lock_reseed is that jump -- the ONE place of the tracking
// path that touches process-global state (libc's rand()).  The batched step's worker threads only detect; the jumps, and
// the rest of such a channel's millisecond, are done afterwards on the calling thread in channel order, so the draws come
// out of rand() in the order the single-threaded step (and the reference's own loop over its channels) makes them.
bool false_lock_detect(gps_ch_t &ch, uint8_t index, int16_t ip)
{
  gps_tracking_t &t = ch.tracking_data;
  if (index >= TRACKING_CH_LENGTH)
    return false;
  t.pll_check_buf[index] = ip;
  if (index < TRACKING_CH_LENGTH - 1)
    return false;
  uint8_t flips = 0;
  uint8_t prev = t.pll_check_buf[0] > 0 ? 1 : 0;
  for (uint8_t i = 1; i < TRACKING_CH_LENGTH; i++) {
    const uint8_t cur = t.pll_check_buf[i] > 0 ? 1 : 0;
    if (cur != prev)
      flips++;
    prev = cur;
  }
  if (flips > 1) {
    t.pll_bad_state_cnt++;
    if (t.pll_bad_state_cnt > 10)
      t.pll_bad_state_cnt = 10;
  } else if (t.pll_bad_state_cnt > 0) {
    t.pll_bad_state_cnt--;
  }
  if (t.pll_bad_state_cnt > 9)
    t.pll_bad_state_master_cnt++;
  else if (t.pll_bad_state_cnt == 0)
    t.pll_bad_state_master_cnt = 0;

  if (t.pll_bad_state_master_cnt <= kPllBadThreshold)
    return false;
  t.pll_bad_state_master_cnt = 0;
  t.pll_bad_state_cnt = 0;
  return true;
}

void false_lock_reseed(gps_ch_t &ch)
{
  // false lock: jump to a random carrier offset around the acquired one (tracking.c:309-326)
  gps_tracking_t &t = ch.tracking_data;
  int16_t delta = 0, candidate;
  do {
    const uint16_t r = (uint16_t)(std::rand() % ACQ_SEARCH_STEP_HZ);
    candidate = (int16_t)(ch.acq_data.found_freq_offset_hz - r + (ACQ_SEARCH_STEP_HZ / 2));
    delta = (int16_t)((int16_t)t.if_freq_offset_hz - candidate);
  } while (std::abs((int)delta) < 200);
  t.if_freq_offset_hz = (float)candidate;
}

// the frequency-locked loop proper: what gps_tracking_fll does after its call of gps_tracking_pll_check (tracking.c:208-256)
void fll_update(gps_ch_t &ch, uint8_t index, int16_t IP, int16_t QP, SlotState *slot)
{
  gps_tracking_t &t = ch.tracking_data;
  if (index == 0) {   // first ms after a channel swap: only remember
    t.fll_old_i = IP;
    t.fll_old_q = QP;
    if (slot)
      slot->fll_now_valid = false;
    return;
  }
  const int16_t oldI = t.fll_old_i, oldQ = t.fll_old_q;
  const float now = (IP == 0) ? (float)(kPi / 2) : gpsx_libm::atanf_fdlibm((float)QP / (float)IP);
  float before;
  if (slot && slot->fll_now_valid && slot->fll_now_i == oldI && slot->fll_now_q == oldQ)
    before = slot->fll_now;
  else
    before = (oldI == 0) ? (float)(kPi / 2) : gpsx_libm::atanf_fdlibm((float)oldQ / (float)oldI);
  if (slot) {
    slot->fll_now = now;
    slot->fll_now_i = IP;
    slot->fll_now_q = QP;
    slot->fll_now_valid = true;
  }
  float rot = now - before;
  if ((double)rot > kPi / 2)
    rot = (float)(kPi - (double)rot);
  if ((double)rot < -kPi / 2)
    rot = (float)(-kPi - (double)rot);
  float change = rot - t.fll_err;
  if ((double)change > kPi / 2)
    change = (float)(kPi - (double)change);
  if ((double)change < -kPi / 2)
    change = (float)(-kPi - (double)change);
  const float dt = 0.001f;
  const float hz = TRACKING_FLL1_C1 * dt * change + (TRACKING_FLL1_C2 * dt * rot);
  t.if_freq_offset_hz -= hz;
  t.fll_old_i = IP;
  t.fll_old_q = QP;
  t.fll_err = rot;
}

// ---- pre-tracking (tracking.c:398-499) ----------------------------------------------------------------------------
void settle_pre_track(gps_ch_t &ch, uint8_t count)
{
  gps_tracking_t &t = ch.tracking_data;
  std::sort(t.pre_track_phases, t.pre_track_phases + count);
  uint8_t run = 0;
  uint16_t longest = 0, mode = 0;
  for (uint8_t i = 1; i < count; i++) {
    const uint16_t gap = (uint16_t)(t.pre_track_phases[i] - t.pre_track_phases[i - 1]);
    if (std::abs((int)gap) < 1) {
      run++;
    } else {
      if (run > longest) {
        longest = run;
        mode = t.pre_track_phases[i - 1];
      }
      run = 0;
    }
  }
  if (run > longest) {
    longest = run;
    mode = t.pre_track_phases[count - 1];
  }
  if (mode) {
    t.code_phase_fine = (float)(mode * kFineRatio);
    t.state = GPS_PRE_TRACK_DONE;
  }
}

// window of this millisecond's 7 correlations; false if it is empty
bool pre_track_window(const gps_ch_t &ch, uint8_t index, uint16_t &first, uint16_t &last)
{
  const gps_tracking_t &t = ch.tracking_data;
  first = (uint16_t)(t.code_search_start + index * kPreTrackStep);
  last = (uint16_t)(first + kPreTrackStep);
  if (last > kFullRange)
    last = kFullRange;
  return first < last;
}

gpsx_acq_job_t pre_track_job(const gps_ch_t &ch, uint16_t first, uint16_t last)
{
  // K2 + K3 + 7 x gps_correlation8 in one search; the window's first maximum competes with the slot's running best
  gpsx_acq_job_t job;
  job.block = 0;
  job.n_ms = 1;
  job.prn = ch.prn;
  job.freq_hz = (float)IF_FREQ_HZ + ch.tracking_data.if_freq_offset_hz;
  job.offset_bits = 0;
  job.win_start = first;
  job.win_stop = last;
  return job;
}

void pre_track_apply(gps_ch_t &ch, uint8_t index, const gpsx_peak_t *pk, SlotState &slot)
{
  gps_tracking_t &t = ch.tracking_data;
  if (pk && (int16_t)pk->max_val > (int)slot.best_value) {
    slot.best_value = (uint16_t)pk->max_val;
    slot.best_phase = (uint16_t)pk->phase;
  }
  if (index == TRACKING_CH_LENGTH - 1) {   // end of this channel's time slot
    t.pre_track_phases[t.pre_track_count] = slot.best_phase;
    t.pre_track_count++;
    if (t.pre_track_count > PRE_TRACK_POINTS_MAX_CNT - 10)
      settle_pre_track(ch, t.pre_track_count);
    if (t.pre_track_count >= PRE_TRACK_POINTS_MAX_CNT) {
      t.pre_track_count = 0;
      std::memset(t.pre_track_phases, 0, PRE_TRACK_POINTS_MAX_CNT * 2);
    }
    slot.best_value = 0;
  }
}

void pre_track_step(gps_ch_t &ch, uint8_t *data, uint8_t index)
{
  if (index >= TRACKING_CH_LENGTH)
    return;
  uint16_t first, last;
  gpsx_peak_t pk;
  const bool has_job = pre_track_window(ch, index, first, last);
  if (has_job) {
    const gpsx_acq_job_t job = pre_track_job(ch, first, last);
    const int rc = gpsx_acq_jobs(gpsx_compat_ctx(), &job, 1, data, 1, &pk, nullptr);
    if (rc != GPSX_OK)
      gpsx_compat_die("gps_tracking_process(pre-track)", rc);
  }
  pre_track_apply(ch, index, has_job ? &pk : nullptr, g_shared_slot);
}

// ---- tracking step (tracking.c:92-170) ------------------------------------------------------------------------------
void nav_bit_sync(gps_ch_t *channel, uint8_t index, int16_t new_i, SlotState &slot);

// Bookkeeping before the correlators: elapsed milliseconds since this channel was last served -> how many skipped
// milliseconds the carrier NCO must be advanced by (0 when served every millisecond).
uint8_t tracking_skipped_ms(gps_ch_t &ch)
{
  gps_tracking_t &t = ch.tracking_data;
  const uint32_t now = tick();
  uint32_t elapsed = now - t.prev_track_timestamp;
  t.prev_track_timestamp = now;
  if (elapsed > 50)
    elapsed = 1;
  return elapsed != 1 ? (uint8_t)(elapsed - 1) : (uint8_t)0;
}

// Everything after the correlators: DLL / PLL / FLL, nav-bit hook, SNR.  slot == nullptr: call the overridable
// gps_nav_data_analyse_new_code (reference linkage); otherwise the built-in bit synchroniser on the given slot state.
// tracking_tail is the part behind the false-lock check.
void tracking_tail(gps_ch_t &ch, uint8_t index, const int16_t iq[6], SlotState *slot)
{
  gps_tracking_t &t = ch.tracking_data;
  const int16_t IP = iq[2], QP = iq[3];
  fll_update(ch, index, IP, QP, slot);
  if (slot)
    nav_bit_sync(&ch, index, IP, *slot);
  else
    gps_nav_data_analyse_new_code(&ch, index, IP);

  t.i_part_summ += (uint32_t)std::abs((int)IP);
  t.q_part_summ += (uint32_t)std::abs((int)QP);
  t.snr_summ_cnt++;
  if (t.snr_summ_cnt > kSnrLength) {
    if (t.q_part_summ == 0) {
      t.snr_value = 1.0f;
      return;   // sic: the sums are not cleared on this path (tracking.c:152-156)
    }
    const float ratio = (float)t.i_part_summ / (float)t.q_part_summ;
    t.snr_value = 10.0f * log10f(ratio);
    t.snr_summ_cnt = 0;
    t.i_part_summ = 0;
    t.q_part_summ = 0;
  }
}

// Returns true when the channel stopped in front of a false-lock reseed it may not draw itself (defer_reseed: a worker
// thread of the batched step): the caller owes it false_lock_reseed + tracking_tail, in channel order.
bool tracking_apply(gps_ch_t &ch, uint8_t index, const int16_t iq[6], SlotState *slot, bool defer_reseed = false)
{
  const int16_t IE = iq[0], QE = iq[1], IP = iq[2], QP = iq[3], IL = iq[4], QL = iq[5];
  dll_update(ch, IE, QE, IL, QL);
  pll_update(ch, index, IP, QP);
  if (false_lock_detect(ch, index, IP)) {
    if (defer_reseed)
      return true;
    false_lock_reseed(ch);
  }
  tracking_tail(ch, index, iq, slot);
  return false;
}

// ---- tracking step (tracking.c:92-170) ------------------------------------------------------------------------------
void tracking_step(gps_ch_t &ch, uint8_t *data, uint8_t index)
{
  gps_tracking_t &t = ch.tracking_data;
  (void)signal_capture_get_packet_cnt();   // the reference reads the tick before the index test (tracking.c:94-99)
  if (index >= TRACKING_CH_LENGTH)
    return;
  const uint8_t skipped = tracking_skipped_ms(ch);
  if (skipped)
    gps_rewind_if_phase(&t, skipped);   // the carrier NCO kept running while other channels were served

  gpsx_trk_state_t st;
  st.prn = ch.prn;
  st.code_phase_fine = t.code_phase_fine;
  st.if_freq_offset_hz = t.if_freq_offset_hz;
  st.if_freq_accum = t.if_freq_accum;
  int16_t iq[6];
  const int rc = gpsx_track_epl_batch(gpsx_compat_ctx(), data, &st, 1, iq);   // K2 + K3 + K5
  if (rc != GPSX_OK)
    gpsx_compat_die("gps_tracking_process", rc);
  t.if_freq_accum = st.if_freq_accum;
  tracking_apply(ch, index, iq, nullptr);
}

// shared by gps_tracking_process and the batched step: state transitions around pre-tracking (tracking.c:52-87)
void enter_pre_track_if_needed(gps_ch_t &ch)
{
  gps_tracking_t &t = ch.tracking_data;
  if (t.state != GPS_NEED_PRE_TRACK)
    return;
  t.code_search_start = (uint16_t)(ch.acq_data.found_code_phase - kPreTrackZone / 2);
  t.code_search_stop = (uint16_t)(ch.acq_data.found_code_phase + kPreTrackZone / 2);
  if (t.code_search_start > kFullRange)
    t.code_search_start = 0;
  if (t.code_search_stop > kFullRange)
    t.code_search_stop = kFullRange;
  t.if_freq_offset_hz = (float)ch.acq_data.found_freq_offset_hz;
  t.pre_track_count = 0;
  std::memset(t.pre_track_phases, 0, PRE_TRACK_POINTS_MAX_CNT * 2);
  t.state = GPS_PRE_TRACK_RUN;
}

}  // namespace

extern "C" {

// ---- link-time hooks with weak defaults ---------------------------------------------------------------------------
void gpsx_compat_set_packet_cnt(uint32_t ticks_ms) { g_ticks = ticks_ms; }

__attribute__((weak)) uint32_t signal_capture_get_packet_cnt(void) { return g_ticks; }

// ---- capture interface (PM/signal_capture.h) on two capture rings of the default context ---------------------------
// g_ring: the two-slot circular buffer the DMA fills (PM/signal_capture.c:14-19); g_copy: the "long processing" copy
// (spi_rx_copy_buffer, :22).  gpsx_compat_capture_push is the half / full transfer interrupt (:57-82).  Blocks go to
// the device when they arrive, so the step calls that get these pointers back find them in HBM already.
namespace {

gpsx_capture *g_ring = nullptr;
gpsx_capture *g_copy = nullptr;
uint8_t g_need_copy = 0;        // signal_capture_need_copy_flag
uint8_t g_irq_unprocessed = 0;  // signal_capture_irq_unprocessed_flag

void capture_open()
{
  if (g_ring)
    return;
  gpsx_ctx *gx = gpsx_compat_ctx();
  int rc = gpsx_capture_create(gx, 2, &g_ring);
  if (rc == GPSX_OK)
    rc = gpsx_capture_create(gx, 1, &g_copy);
  if (rc != GPSX_OK)
    gpsx_compat_die("gpsx_capture_create", rc);
}

}  // namespace

}  // extern "C"

void gpsx_compat_capture_forget()
{
  g_ring = g_copy = nullptr;
  g_need_copy = g_irq_unprocessed = 0;
}

extern "C" {

__attribute__((weak)) void signal_capture_init(void) { capture_open(); }

void gpsx_compat_capture_push(const uint8_t *block)
{
  capture_open();
  const int rc = gpsx_capture_push(g_ring, block);
  if (rc != GPSX_OK)
    gpsx_compat_die("gpsx_capture_push", rc);
  g_ticks++;
  g_irq_unprocessed = 1;
}

__attribute__((weak)) uint8_t *signal_capture_get_ready_buf(void)
{
  capture_open();
  g_irq_unprocessed = 0;
  const uint8_t *p = gpsx_capture_ready_buf(g_ring);
  return const_cast<uint8_t *>(p ? p : gpsx_capture_write_slot(g_ring));   // nothing received yet: a zeroed slot
}

__attribute__((weak)) uint8_t *signal_capture_get_copy_buf(void)
{
  capture_open();
  const uint8_t *p = gpsx_capture_ready_buf(g_copy);
  return const_cast<uint8_t *>(p ? p : gpsx_capture_write_slot(g_copy));   // one slot: a fixed address either way
}

__attribute__((weak)) void signal_capture_handling(void)   // PM/signal_capture.c:108-135 without the 900 us deadline
{
  if (!g_irq_unprocessed || !g_ring)
    return;
  if (g_need_copy) {
    const uint8_t *ready = gpsx_capture_ready_buf(g_ring);
    const int rc = ready ? gpsx_capture_push(g_copy, ready) : GPSX_OK;
    if (rc != GPSX_OK)
      gpsx_compat_die("gpsx_capture_push", rc);
    g_need_copy = 0;
    g_irq_unprocessed = 0;
  }
}

__attribute__((weak)) uint8_t signal_capture_have_irq(void) { return g_irq_unprocessed; }
__attribute__((weak)) void signal_capture_need_data_copy(void) { g_need_copy = 1; }
__attribute__((weak)) uint8_t signal_capture_check_copied(void) { return g_need_copy == 0; }

// ---- word layer (PM/GPS/nav_data.c:257-451): 50 bit/s stream -> 30-bit words -> 10-word subframes --------------------
namespace {

constexpr uint32_t kBadPolarityTimeoutMs = 12000;   // two subframes without a good word: forget the polarity (nav_data.c:22)
const uint8_t kPreamble[8] = {1, 0, 0, 0, 1, 0, 1, 1};

// IS-GPS-200 table 20-XIV: which of the source bits d1..d24 (bit i - 1 of the mask) enter parity bits D25..D30, and
// which of the previous word's last two bits (D29*, D30*) each starts from
constexpr uint32_t parity_mask(std::initializer_list<int> bits)
{
  uint32_t m = 0;
  for (int b : bits)
    m |= 1u << (b - 1);
  return m;
}
constexpr uint32_t kParityMask[6] = {
    parity_mask({1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23}),
    parity_mask({2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24}),
    parity_mask({1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22}),
    parity_mask({2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23}),
    parity_mask({1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24}),
    parity_mask({3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24}),
};
constexpr bool kParityFromD30[6] = {false, true, false, true, true, false};

bool starts_with_preamble(const gps_nav_data_t &n, uint8_t invert)
{
  for (int i = 0; i < 8; i++)
    if (n.word_buf[i] != (kPreamble[i] ^ invert))
      return false;
  return true;
}

// the collected word becomes word `word_cnt` of the subframe image; its last two bits seed the next word's parity
void store_word(gps_nav_data_t &n)   // nav_data.c:409-429
{
  const int first = n.word_cnt * GPS_NAV_WORD_LENGTH;
  for (int i = 0; i < GPS_NAV_WORD_LENGTH; i++) {
    const int bit = first + i;
    if (n.word_buf[i] == 1)
      n.subframe_data[bit >> 3] |= (uint8_t)(1u << (bit & 7));
    else
      n.subframe_data[bit >> 3] &= (uint8_t)~(1u << (bit & 7));
  }
  n.old_D29 = n.word_buf[28];
  n.old_D30 = n.word_buf[29];
}

// Removes the D30* inversion from the 24 data bits IN PLACE (as the reference does: what is stored afterwards are source
// bits), then checks the six parity bits.
bool word_parity_ok(gps_nav_data_t &n)   // nav_data.c:433-452
{
  uint32_t d = 0;
  for (int i = 0; i < 24; i++) {
    n.word_buf[i] ^= n.old_D30;
    d |= (uint32_t)(n.word_buf[i] & 1u) << i;
  }
  for (int k = 0; k < 6; k++) {
    const uint8_t p = (uint8_t)((__builtin_popcount(d & kParityMask[k]) & 1) ^ (kParityFromD30[k] ? n.old_D30 : n.old_D29));
    if (n.word_buf[24 + k] != p)
      return false;
  }
  return true;
}

// The subframe that just completed began at the last accurately located bit edge at or before now (nav_data.c:356-378).
void stamp_subframe(gps_nav_data_t &n)
{
  if (n.accurate_swap_ok == 0)
    return;
  const uint32_t now = tick();
  uint32_t edge = now / 20 * 20 + n.accurate_swap_time;
  if ((int32_t)(now - edge) < 0)
    edge -= 20;
  n.subframe_cnt++;
  n.last_subframe_time = edge;
}

}  // namespace

__attribute__((weak)) void gps_nav_data_words_detection(gps_ch_t *channel, uint8_t new_bit)
{
  gps_nav_data_t &n = channel->nav_data;
  if (n.word_cnt == 0) {
    // hunting: slide the 30-bit window by one bit and look for the preamble at its head
    std::memmove(n.word_buf, n.word_buf + 1, GPS_NAV_WORD_LENGTH - 1);
    n.word_buf[GPS_NAV_WORD_LENGTH - 1] = new_bit;
    if (starts_with_preamble(n, 0)) {
      store_word(n);
      n.word_cnt = 1;
      n.word_bit_cnt = 0;
      n.inv_preabmle_cnt = 0;
    }
    if (n.polarity_found == 0 && n.word_cnt == 0) {
      if (starts_with_preamble(n, 1))
        n.inv_preabmle_cnt++;
      if (n.inv_preabmle_cnt >= 2)
        n.inv_polarity_flag = 1;
    }
    if (n.polarity_found) {
      if (tick() - n.word_detection_timestamp > kBadPolarityTimeoutMs) {
        n.word_detection_timestamp = tick();
        n.polarity_found = 0;
        n.inv_polarity_flag = 0;
      }
    }
    return;
  }
  // synchronised: collect the next word bit by bit
  n.word_buf[n.word_bit_cnt++] = new_bit;
  if (n.word_bit_cnt < GPS_NAV_WORD_LENGTH)
    return;
  if (!word_parity_ok(n)) {
    n.word_cnt = 0;
    std::memset(n.word_buf, 0, GPS_NAV_WORD_LENGTH);
    return;
  }
  n.word_cnt_test++;
  store_word(n);
  n.word_cnt++;
  n.word_bit_cnt = 0;
  n.word_detection_timestamp = tick();
  n.polarity_found = 1;
  if (n.word_cnt == 10) {
    (void)gps_nav_data_decode_subframe(channel);
    stamp_subframe(n);
    n.word_cnt = 0;
    n.new_subframe_flag = 1;
    std::memset(n.word_buf, 0, GPS_NAV_WORD_LENGTH);   // no false preamble out of stale bits
  }
}

// Default prompt-I hook: 20 ms bit-period synchronisation and bit integration (PM/GPS/nav_data.c:46-250), feeding the
// word layer above.  A host that links the reference's nav_data.c overrides it.
namespace {

void refine_bit_edge(gps_ch_t *ch, const int16_t *ip, uint32_t slot_start_ticks)   // nav_data.c:145-218
{
  uint8_t edge = 0;
  if (std::abs((int)ip[1]) > std::abs((int)ip[0]))
    return;
  if (ip[3] == 0)
    return;
  const float whole = (float)std::abs((int)ip[0]) / (float)std::abs((int)ip[3]);
  if (whole > 1.5f || whole < 0.7f)
    return;
  const int16_t chip = (int16_t)((int16_t)ch->tracking_data.code_phase_fine / 16);
  if (chip < 0 || chip > PRN_LENGTH)
    return;
  if (chip < PRN_LENGTH / 4 || chip > PRN_LENGTH * 3 / 4) {
    if (ip[1] == 0)
      return;
    const float jump = (float)std::abs((int)ip[0]) / (float)std::abs((int)ip[1]);
    if (jump > 1.5f || jump < 0.7f)
      return;
    edge = chip < PRN_LENGTH / 4 ? 2 : 1;
  } else {
    const uint16_t d1 = (uint16_t)std::abs(ip[0] - ip[1]);
    const uint16_t d2 = (uint16_t)std::abs(ip[2] - ip[3]);
    if (d1 > d2) {
      if (d2 == 0)
        return;
      if ((float)d1 / (float)d2 < 2.5f)
        return;
      edge = 1;
    } else {
      if (d1 == 0)
        return;
      if ((float)d2 / (float)d1 < 2.5f)
        return;
      edge = 2;
    }
  }
  if (edge == 0)
    return;
  ch->nav_data.accurate_swap_time = (uint8_t)((slot_start_ticks + edge) % 20);
  ch->nav_data.accurate_swap_ok = 1;
}

void integrate_bit(gps_ch_t *ch, uint8_t ms_bit, uint32_t now)   // nav_data.c:223-253
{
  gps_nav_data_t &n = ch->nav_data;
  const uint32_t since = now - n.old_swap_time;
  const uint8_t rem = (uint8_t)(since % 20);
  if (rem < n.old_reminder) {
    const uint8_t bit = n.last_bit_pos_cnt > n.last_bit_neg_cnt ? 1 : 0;
    gps_nav_data_words_detection(ch, bit);
    n.last_bit_pos_cnt = 0;
    n.last_bit_neg_cnt = 0;
  }
  if (ms_bit)
    n.last_bit_pos_cnt++;
  else
    n.last_bit_neg_cnt++;
  n.old_reminder = rem;
}
}  // namespace

namespace {
void nav_bit_sync(gps_ch_t *channel, uint8_t index, int16_t new_i, SlotState &slot)
{
  if (index >= TRACKING_CH_LENGTH)
    return;
  gps_nav_data_t &n = channel->nav_data;
  uint8_t bit = new_i > 0 ? 1 : 0;
  if (n.inv_polarity_flag)
    bit ^= 1;
  slot.bits[index] = bit;
  slot.ip[index] = new_i;
  const uint32_t now = tick();
  if (index == 0)
    slot.start_ticks = now;
  if (n.period_sync_ok_flag == 1)
    integrate_bit(channel, bit, now);
  if (index < TRACKING_CH_LENGTH - 1)
    return;

  uint8_t flips = 0, flip_at = 0, prev = slot.bits[0];
  for (uint8_t i = 1; i < TRACKING_CH_LENGTH; i++) {
    if (slot.bits[i] != prev) {
      flips++;
      flip_at = i;
    }
    prev = slot.bits[i];
  }
  if (flips != 1)
    return;
  const uint32_t edge_time = slot.start_ticks + flip_at;
  const uint8_t rem = (uint8_t)((edge_time - n.old_swap_time) % 20);
  if (rem < 2 || rem == 19) {
    if (n.right_period_cnt < 10)
      n.right_period_cnt++;
    if (n.right_period_cnt > 8)
      n.period_sync_ok_flag = 1;
  } else {
    if (n.right_period_cnt > 0)
      n.right_period_cnt--;
    if (n.right_period_cnt < 3)
      n.period_sync_ok_flag = 0;
  }
  n.old_swap_time = edge_time;
  if (n.period_sync_ok_flag && flip_at == 2)
    refine_bit_edge(channel, slot.ip, slot.start_ticks);
}
}  // namespace

__attribute__((weak)) void gps_nav_data_analyse_new_code(gps_ch_t *channel, uint8_t index, int16_t new_i)
{
  nav_bit_sync(channel, index, new_i, g_shared_slot);
}

// ---- acquisition.h ----------------------------------------------------------------------------------------------------
uint32_t *acquisition_get_hist(void) { return g_freq_votes; }

void acquisition_start_channel(gps_ch_t *channel)
{
  gps_acq_t &a = channel->acq_data;
  if (a.state != GPS_ACQ_NEED_FREQ_SEARCH)
    return;
  if (a.given_freq_offset_hz != 0) {   // Doppler hint: no frequency search (acquisition.c:72-79)
    a.found_freq_offset_hz = a.given_freq_offset_hz;
    a.state = GPS_ACQ_FREQ_SEARCH_DONE;
    return;
  }
  reset_search_buffers();
  a.freq_index = 0;
  a.state = GPS_ACQ_FREQ_SEARCH_RUN;
}

void acquisition_start_code_search_channel(gps_ch_t *channel)
{
  gps_acq_t &a = channel->acq_data;
  if (a.state != GPS_ACQ_FREQ_SEARCH_DONE)
    return;
  clear_phase_histogram(a);
  a.code_search_start = 0;
  a.code_search_stop = kFullRange;
  a.code_hist_step = 64;   // ACQ_PHASE1_HIST_STEP
  a.start_timestamp = signal_capture_get_packet_cnt();
  a.state = GPS_ACQ_CODE_PHASE_SEARCH1;
}

void acquisition_start_code_search3_channel(gps_ch_t *channel)
{
  gps_acq_t &a = channel->acq_data;
  if (a.state != GPS_ACQ_CODE_PHASE_SEARCH2_DONE)
    return;
  clear_phase_histogram(a);
  narrow_window(a, kStage3Width);
  reset_search_buffers();
  a.start_timestamp = signal_capture_get_packet_cnt();
  a.state = GPS_ACQ_CODE_PHASE_SEARCH3;
}

void acquisition_process(gps_ch_t *channel, uint8_t *data)
{
  // Every channel's search parameters depend only on its own state on entry, so all searches of this millisecond go
  // out as one job list; the serial logic then consumes the triplets in channel order, as the reference's loop does.
  gpsx_acq_job_t jobs[GPS_SAT_CNT];
  int job_of[GPS_SAT_CNT];
  int n_jobs = 0;
  for (int i = 0; i < GPS_SAT_CNT; i++) {
    job_of[i] = -1;
    const gps_acq_t &a = channel[i].acq_data;
    if (channel[i].prn < 1 || a.state == GPS_ACQ_DONE)
      continue;
    gpsx_acq_job_t j;
    j.block = 0;
    j.n_ms = 1;
    j.prn = channel[i].prn;
    j.offset_bits = 0;
    if (a.state == GPS_ACQ_FREQ_SEARCH_RUN) {
      const int16_t hz = (int16_t)(-ACQ_SEARCH_FREQ_HZ + a.freq_index * ACQ_SEARCH_STEP_HZ);
      j.freq_hz = (float)(IF_FREQ_HZ + hz);
      j.win_start = 0;
      j.win_stop = kFullRange;
    } else if (in_code_phase_search(a.state)) {
      j.freq_hz = (float)(IF_FREQ_HZ + a.found_freq_offset_hz);
      j.win_start = a.code_search_start;
      j.win_stop = a.code_search_stop > kFullRange ? kFullRange : a.code_search_stop;
      if (j.win_start > j.win_stop)
        j.win_start = j.win_stop;
    } else {
      continue;
    }
    job_of[i] = n_jobs;
    jobs[n_jobs++] = j;
  }
  gpsx_peak_t peaks[GPS_SAT_CNT];
  if (n_jobs > 0) {
    const int rc = gpsx_acq_jobs(gpsx_compat_ctx(), jobs, n_jobs, data, 1, peaks, nullptr);
    if (rc != GPSX_OK)
      gpsx_compat_die("acquisition_process", rc);
  }

  for (int i = 0; i < GPS_SAT_CNT; i++) {
    gps_ch_t &ch = channel[i];
    gps_acq_t &a = ch.acq_data;
    if (ch.prn < 1 || a.state == GPS_ACQ_DONE)
      continue;
    if (a.state == GPS_ACQ_FREQ_SEARCH_RUN) {
      after_frequency_search(ch, peaks[job_of[i]]);
      continue;
    }
    if (a.state == GPS_ACQ_CODE_PHASE_SEARCH1_DONE) {   // open stage 2 around the stage-1 estimate
      clear_phase_histogram(a);
      narrow_window(a, kStage2Width);
      reset_search_buffers();
      a.start_timestamp = signal_capture_get_packet_cnt();
      a.state = GPS_ACQ_CODE_PHASE_SEARCH2;
      continue;
    }
    if (a.state == GPS_ACQ_CODE_PHASE_SEARCH3_DONE) {
      a.state = GPS_ACQ_DONE;
      continue;
    }
    if (in_code_phase_search(a.state) && job_of[i] >= 0)
      after_code_phase_search(ch, peaks[job_of[i]]);
  }
}

// ---- gps_master.h: channel sequencing (PM/GPS/gps_master.c:68-129, 458-510) -------------------------------------------------
// The part of the reference's "GPS master" that drives the step calls above: start acquisition channel by channel,
// open the code-phase searches together, hand finished channels to tracking.  Its pseudorange / PVT duty
// (gps_master_nav_handling) is gpsx_nav_master.cpp; the UI is out of scope, key_up_presed a weak variable.
namespace {
uint8_t g_need_acq = 1;     // gps_common_need_acq
uint8_t g_start_flag = 1;   // gps_start_flag
}  // namespace

__attribute__((weak)) uint8_t key_up_presed = 0;

uint8_t gps_master_need_acq(void) { return g_need_acq; }

// What a reboot does to the firmware: every file-scope variable of the receiver's step logic back at its initial value
// (gps_master.c's start flag and need-acquisition flag, acquisition.c's search buffers, tracking.c / nav_data.c's slot
// statics, the pseudorange step's and the solver's memories).  The channel table is the caller's.
void gpsx_compat_receiver_reset(void)
{
  g_need_acq = 1;
  g_start_flag = 1;
  g_ticks = 0;
  reset_search_buffers();
  g_shared_slot = SlotState{};
  gpsx_nav_master_reset();
  gpsx_pvt_reset();
}

uint8_t gps_master_need_freq_search(gps_ch_t *channels)
{
  uint8_t need = 0;
  for (int i = 0; i < GPS_SAT_CNT; i++)
    if (channels[i].acq_data.state < GPS_ACQ_FREQ_SEARCH_DONE)
      need = 1;
  return need;
}

uint8_t gps_master_is_code_search3(gps_ch_t *channels)
{
  int n = 0;
  for (int i = 0; i < GPS_SAT_CNT; i++)
    if (channels[i].acq_data.state > GPS_ACQ_CODE_PHASE_SEARCH2)
      n++;
  return n == GPS_SAT_CNT;
}

void gps_master_reset_to_aqc_start(gps_ch_t *channels)
{
  for (int i = 0; i < GPS_SAT_CNT; i++)
    if (channels[i].acq_data.state < GPS_ACQ_FREQ_SEARCH_DONE)
      return;
  for (int i = 0; i < GPS_SAT_CNT; i++) {
    uint32_t good_words;   // gps_nav_data_t.word_cnt_test, byte 56 of the reference's struct (inside the opaque part)
    std::memcpy(&good_words, reinterpret_cast<const uint8_t *>(&channels[i].nav_data) + 56, 4);
    if (good_words > 1)
      channels[i].acq_data.found_freq_offset_hz = (int16_t)channels[i].tracking_data.if_freq_offset_hz;
    channels[i].acq_data.state = GPS_ACQ_FREQ_SEARCH_DONE;
    std::memset(&channels[i].tracking_data, 0, sizeof(gps_tracking_t));
    std::memset(&channels[i].nav_data, 0, sizeof(gps_nav_data_t));
  }
}

void gps_master_handling(gps_ch_t *channels, uint8_t index)
{
  if (g_start_flag) {
    g_start_flag = 0;
    acquisition_start_channel(&channels[0]);
  }
  g_need_acq = 0;
  uint8_t need_freq = 0, stage3_ready = 0;
  for (int i = 0; i < GPS_SAT_CNT; i++) {
    if (channels[i].acq_data.state != GPS_ACQ_DONE)
      g_need_acq = 1;
    if (channels[i].acq_data.state < GPS_ACQ_FREQ_SEARCH_DONE)
      need_freq = 1;
    if (channels[i].acq_data.state == GPS_ACQ_CODE_PHASE_SEARCH2_DONE)
      stage3_ready++;
  }
  if (g_need_acq == 1) {   // frequency searches (or hints) one channel at a time
    for (int i = 0; i < GPS_SAT_CNT - 1; i++) {
      if (channels[i].acq_data.state == GPS_ACQ_FREQ_SEARCH_DONE &&
          channels[i + 1].acq_data.state == GPS_ACQ_NEED_FREQ_SEARCH) {
        acquisition_start_channel(&channels[i + 1]);
        return;
      }
    }
  }
  if (need_freq == 0 && g_need_acq == 1) {   // code-phase searches for all channels together
    for (int i = 0; i < GPS_SAT_CNT; i++) {
      if (channels[i].acq_data.state == GPS_ACQ_FREQ_SEARCH_DONE)
        acquisition_start_code_search_channel(&channels[i]);
      if (stage3_ready == GPS_SAT_CNT)
        acquisition_start_code_search3_channel(&channels[i]);
    }
  }
  if (g_need_acq == 0) {   // everything acquired: start tracking
    for (int i = 0; i < GPS_SAT_CNT; i++)
      if (channels[i].tracking_data.state == GPS_TRACKNG_IDLE)
        channels[i].tracking_data.state = GPS_NEED_PRE_TRACK;
  }
  if (key_up_presed) {
    key_up_presed = 0;
    gps_master_reset_to_aqc_start(channels);
  }
  if (index == 0xFF)   // the idle slot of the 17 ms cycle: navigation / pseudoranges / PVT in the reference
    gps_master_nav_handling(channels);
}

// ---- tracking.h ---------------------------------------------------------------------------------------------------------
void gps_tracking_process(gps_ch_t *channel, uint8_t *data, uint8_t index)
{
  gps_tracking_t &t = channel->tracking_data;
  enter_pre_track_if_needed(*channel);
  if (t.state == GPS_PRE_TRACK_RUN)
    pre_track_step(*channel, data, index);
  else if (t.state == GPS_PRE_TRACK_DONE)
    t.state = GPS_TRACKING_RUN;
  if (t.state == GPS_TRACKING_RUN)
    tracking_step(*channel, data, index);
}

// ---- batched step (not in the reference) --------------------------------------------------------------------------------
// gps_tracking_process for n_ch channels on the SAME millisecond, each channel treated as the single channel of its own
// receiver (the schedule of project_single_sat/main.c:96-109: every channel is served every millisecond, `index`
// cycles 0..3 or is 0xFF for an idle slot).  All pre-tracking searches go out as ONE job list and all E/P/L correlators
// as ONE launch; the per-channel loop logic is the same code gps_tracking_process runs.
int gps_tracking_batch_workers(void) { return StepPool::instance().size(); }
namespace {
int g_last_batch_workers = 0;
}
// how many of them the last gps_tracking_process_batch call actually used (1 below the threshold, and whenever the host
// overrides one of the weak hooks the per-channel logic reaches)
int gps_tracking_batch_last_workers(void) { return g_last_batch_workers; }

void gps_tracking_process_batch(gps_ch_t *channel, int n_ch, uint8_t *data, uint8_t index)
{
  static std::vector<SlotState> slots;
  static std::vector<gpsx_acq_job_t> jobs;
  static std::vector<gpsx_peak_t> peaks;
  static std::vector<int> job_of, trk_of;
  static std::vector<gpsx_trk_state_t> st;
  static std::vector<uint8_t> skipped;
  static std::vector<int16_t> iq;
  if (n_ch <= 0)
    return;
  if ((int)slots.size() < n_ch)
    slots.resize(n_ch);
  if ((int)job_of.size() < n_ch) {
    job_of.resize(n_ch);
    trk_of.resize(n_ch);
  }
  // The host side of the step -- the work lists before the correlators, the reference's float loops after them -- is
  // independent per channel (each owns its gps_ch_t and its slot state): from kStepThreadsFrom channels on it is spread over
  // the calling thread's CPUs (after gpsx_bind_thread_to_device: the cores next to the GPU), contiguous channel ranges per
  // worker.  Below that -- the reference's four channels, the golden traces -- everything runs on the calling thread in
  // channel order, as before.  Shared state this path can reach: rand() in the PLL's false-lock reseed (tracking.c:309-326)
  // -- never drawn on a worker: false_lock_detect / finish_deferred below -- and the host's weak hooks (kForeignHooks).
  StepPool &pool = StepPool::instance();
  static const int kThreadsFrom = [] { const char *e = std::getenv("GPSX_STEP_THREADS_FROM"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : kStepThreadsFrom; }();
  // A host that overrides one of the weak hooks this path reaches (its own tick source, the reference's nav_data.c with its
  // function-static scratch, a ctypes callback) gets every call from the thread that called the step, as the
  // reference's single-threaded loop makes them: no workers then.
  static const bool kForeignHooks = resolves_outside_this_library((const void *)&signal_capture_get_packet_cnt) ||
                                    resolves_outside_this_library((const void *)&gps_nav_data_words_detection) ||
                                    resolves_outside_this_library((const void *)&gps_nav_data_decode_subframe);
  const int n_workers = (n_ch >= kThreadsFrom && !kForeignHooks) ? pool.size() : 1;
  g_last_batch_workers = n_workers;
  static const int kAhead = [] { const char *e = std::getenv("GPSX_STEP_PREFETCH"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 2; }();
  static std::vector<WorkerLists> lists;
  if ((int)lists.size() < n_workers)
    lists.resize(n_workers);
  const BatchTick one_tick(signal_capture_get_packet_cnt());   // ONE read, here, for every channel and worker of the step: tick()
  for (int w = 0; w < n_workers; w++)
    lists[w].deferred.clear();

  // pass 1: state transitions that precede the correlators, and the work lists (per worker, in channel order)
  auto pass1 = [&](int w) {
    WorkerLists &L = lists[w];
    L.jobs.clear();
    L.st.clear();
    L.skipped.clear();
    L.any_skipped = false;
    const int lo = (int)((long)n_ch * w / n_workers), hi = (int)((long)n_ch * (w + 1) / n_workers);
    for (int c = lo; c < hi; c++) {
      gps_ch_t &ch = channel[c];
      gps_tracking_t &t = ch.tracking_data;
      if (c + kAhead < hi)   // (a gps_ch_t is 1688 bytes: every channel's state is a fresh set of cache lines)
        prefetch_channel(channel[c + kAhead], false);
      job_of[c] = trk_of[c] = -1;
      enter_pre_track_if_needed(ch);
      if (t.state == GPS_PRE_TRACK_RUN) {
        uint16_t first, last;
        if (index < TRACKING_CH_LENGTH && pre_track_window(ch, index, first, last)) {
          job_of[c] = (int)L.jobs.size();
          L.jobs.push_back(pre_track_job(ch, first, last));
        }
      } else {
        if (t.state == GPS_PRE_TRACK_DONE)
          t.state = GPS_TRACKING_RUN;
        if (t.state == GPS_TRACKING_RUN && index < TRACKING_CH_LENGTH) {
          gpsx_trk_state_t s1;
          s1.prn = ch.prn;
          s1.code_phase_fine = t.code_phase_fine;
          s1.if_freq_offset_hz = t.if_freq_offset_hz;
          s1.if_freq_accum = t.if_freq_accum;
          trk_of[c] = (int)L.st.size();
          L.st.push_back(s1);
          L.skipped.push_back(tracking_skipped_ms(ch));
          L.any_skipped |= L.skipped.back() != 0;
        }
      }
    }
  };
  pool.run(n_workers, pass1);
  // the workers' lists, one behind the other (their order is the channel order)
  size_t n_jobs = 0, n_st = 0;
  bool any_skipped = false;
  for (int w = 0; w < n_workers; w++) {
    lists[w].job_base = (int)n_jobs;
    lists[w].st_base = (int)n_st;
    n_jobs += lists[w].jobs.size();
    n_st += lists[w].st.size();
    any_skipped |= lists[w].any_skipped;
  }
  gpsx_ctx *gx = gpsx_compat_ctx();
  StepBuffers &pin = StepBuffers::instance();
  gpsx_trk_state_t *st_all = nullptr;
  int16_t *iq_all = nullptr;
  if (n_st) {   // page-locked (gpsx_host_alloc): the step's copies then run at the link's rate and can be chunked
    pin.reserve(gx, n_st);
    st_all = pin.st;
    iq_all = pin.iq;
  }
  jobs.resize(n_jobs);
  skipped.resize(n_st);
  // From kStepOverlapFrom tracked channels on (and with workers to spread the loops over) the correlators go through in
  // pieces (gpsx_track_epl_batch_chunked) and the loops of a piece run while the GPU works on the next ones.  Every worker
  // keeps ITS channels (their records live in its core's caches from one millisecond to the next): piece q of the launch
  // is made of the q-th part of every worker's list, so the order of the step's arrays is no longer the channel order.
  // (Closed loop, 32 signals, one box: 131072 channels 821-870 us per step without, 712 us with 4 pieces, 753 with 2, 892
  // with 8; 65536 channels 445-546 without, 486 with.  Pieces re-divided over all workers instead: 1175-1421 us -- a
  // channel's record then moves between cores twice per millisecond.  $GPSX_STEP_CHUNKS = 0 / 1 turns it off.)
  static const int kStepChunks = [] { const char *e = std::getenv("GPSX_STEP_CHUNKS"); const int v = e ? std::atoi(e) : -1; return v >= 0 && v <= WorkerLists::kMaxRuns ? v : 4; }();
  static const int kOverlapFrom = [] { const char *e = std::getenv("GPSX_STEP_OVERLAP_FROM"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : kStepOverlapFrom; }();
  const bool overlapped = n_workers > 1 && kStepChunks > 1 && n_st >= (size_t)kOverlapFrom;
  if (overlapped) {
    int base = 0;
    for (int q = 0; q < kStepChunks; q++)
      for (int w = 0; w < n_workers; w++) {
        WorkerLists &L = lists[w];
        const long sz = (long)L.st.size();
        L.run_start[q] = (int)(sz * q / kStepChunks);
        L.run_start[q + 1] = (int)(sz * (q + 1) / kStepChunks);
        L.run_base[q] = base;
        base += L.run_start[q + 1] - L.run_start[q];
      }
  }
  auto gather = [&](int w) {
    WorkerLists &L = lists[w];
    if (!L.jobs.empty())
      std::memcpy(&jobs[L.job_base], L.jobs.data(), L.jobs.size() * sizeof(gpsx_acq_job_t));
    if (L.st.empty())
      return;
    if (!overlapped) {
      std::memcpy(st_all + L.st_base, L.st.data(), L.st.size() * sizeof(gpsx_trk_state_t));
      std::memcpy(&skipped[L.st_base], L.skipped.data(), L.skipped.size());
      return;
    }
    for (int q = 0; q < kStepChunks; q++) {
      const int from = L.run_start[q], cnt = L.run_start[q + 1] - from;
      std::memcpy(st_all + L.run_base[q], L.st.data() + from, cnt * sizeof(gpsx_trk_state_t));
      std::memcpy(&skipped[L.run_base[q]], L.skipped.data() + from, cnt);
    }
  };
  pool.run(n_workers, gather);

  // pass 2: the GPU work of this millisecond
  if (n_jobs) {
    peaks.resize(n_jobs);
    const int rc = gpsx_acq_jobs(gx, jobs.data(), (int)n_jobs, data, 1, peaks.data(), nullptr);
    if (rc != GPSX_OK)
      gpsx_compat_die("gps_tracking_process_batch(pre-track)", rc);
  }
  if (n_st) {
    if (any_skipped) {
      const int rc = gpsx_rewind(gx, st_all, (int)n_st, skipped.data());
      if (rc != GPSX_OK)
        gpsx_compat_die("gps_tracking_process_batch(rewind)", rc);
    }
    if (!overlapped) {
      const int rc = gpsx_track_epl_batch(gx, data, st_all, (int)n_st, iq_all);
      if (rc != GPSX_OK)
        gpsx_compat_die("gps_tracking_process_batch", rc);
    }
  }
  // pass 3: per-channel serial logic, in channel order within a worker
  auto loops = [&](WorkerLists &L, int c, int hi, size_t k) {
    gps_ch_t &ch = channel[c];
    gps_tracking_t &t = ch.tracking_data;
    if (c + kAhead < hi)
      prefetch_channel(channel[c + kAhead], true);
    if (t.state == GPS_PRE_TRACK_RUN && trk_of[c] < 0) {
      if (index < TRACKING_CH_LENGTH)
        pre_track_apply(ch, index, job_of[c] >= 0 ? &peaks[L.job_base + job_of[c]] : nullptr, slots[c]);
      // a channel whose pre-tracking just settled starts tracking on the NEXT millisecond (its correlators were not
      // part of this launch); the reference's single-channel call does the same one call later
    } else if (trk_of[c] >= 0) {
      t.if_freq_accum = st_all[k].if_freq_accum;
      if (tracking_apply(ch, index, &iq_all[k * 6], &slots[c], n_workers > 1))
        L.deferred.emplace_back(c, k);
    }
  };
  // The reseeds the workers left undone, in channel order (a worker's range is contiguous, it walks it upwards, and the
  // ranges follow each other), each followed by the rest of that channel's millisecond: rand() is drawn exactly as the
  // single-threaded step draws it.
  auto finish_deferred = [&]() {
    for (int w = 0; w < n_workers; w++)
      for (const auto &d : lists[w].deferred) {
        gps_ch_t &ch = channel[d.first];
        false_lock_reseed(ch);
        tracking_tail(ch, index, &iq_all[d.second * 6], &slots[d.first]);
      }
  };
  if (!overlapped) {
    auto pass3 = [&](int w) {
      WorkerLists &L = lists[w];
      const int lo = (int)((long)n_ch * w / n_workers), hi = (int)((long)n_ch * (w + 1) / n_workers);
      for (int c = lo; c < hi; c++)
        loops(L, c, hi, (size_t)L.st_base + (trk_of[c] >= 0 ? trk_of[c] : 0));
    };
    pool.run(n_workers, pass3);
    finish_deferred();
    return;
  }
  // overlapped: when entries [0, ready) of the step's arrays are back, every worker moves its cursor over the channels whose
  // entry is among them (and the untracked channels on the way); the last piece takes each worker to the end of its range
  size_t ready = 0;
  for (int w = 0; w < n_workers; w++) {
    lists[w].cursor = (int)((long)n_ch * w / n_workers);
    lists[w].cursor_run = 0;
  }
  auto pass3_piece = [&](int w) {
    WorkerLists &L = lists[w];
    const int hi = (int)((long)n_ch * (w + 1) / n_workers);
    int c = L.cursor, q = L.cursor_run;
    for (; c < hi; c++) {
      size_t k = 0;
      if (trk_of[c] >= 0) {
        while (trk_of[c] >= L.run_start[q + 1])
          q++;
        k = (size_t)L.run_base[q] + (trk_of[c] - L.run_start[q]);
        if (k >= ready)
          break;
      }
      loops(L, c, hi, k);
    }
    L.cursor = c;
    L.cursor_run = q;
  };
  struct Piece {
    decltype(pass3_piece) *fn;
    StepPool *pool;
    size_t *ready;
    int n_workers;
  } piece = {&pass3_piece, &pool, &ready, n_workers};
  auto on_piece = [](void *user, int first, int n) {
    Piece &p = *static_cast<Piece *>(user);
    *p.ready = (size_t)first + n;
    p.pool->run(p.n_workers, *p.fn);
  };
  pool.keep_hot(true);   // (workers that slept between the pieces: 827 us per step at 131072 channels instead of 712)
  const int rc = gpsx_track_epl_batch_chunked(gx, data, st_all, (int)n_st, iq_all, kStepChunks, on_piece, &piece);
  pool.keep_hot(false);
  if (rc != GPSX_OK)
    gpsx_compat_die("gps_tracking_process_batch", rc);
  finish_deferred();
}


// ---- the word layer behind the DEVICE tracking loops (include/gpsx.h gpsx_track_loop) ------------------------------------------
// flags: the [n_blocks][n_ch] bytes of one gpsx_track_loop launch whose first block had tick first_tick.  Per channel and
// millisecond, in order: a completed navigation bit goes to gps_nav_data_words_detection with the tick of its
// millisecond (preamble hunt, parity, polarity, subframe assembly, ephemeris decode, subframe time stamp -- all as in
// the host mode), then a located bit edge is written into the channel's nav_data (what the NEXT stamp is made from, as
// nav_data.c orders them).  The bit synchroniser's own state lives on the device; of it the host record keeps
// period_sync_ok_flag, accurate_swap_time / _ok and inv_polarity_flag current.  Channels whose inv_polarity_flag is different
// after the launch from what it was before are listed in changed_opt, each once, in channel order (at most max_changed are
// written; the return value is how many there are, at most n_ch).  Under GPSX_WORDSYNC_DEVICE the device's own word sync has
// taken the same decisions on the same milliseconds and the list is information; under GPSX_WORDSYNC_HOST the caller hands
// it to gpsx_loop_set_polarity, and the change reaches the device's votes (nav_data.c:60-66) with the next launch.
// From kStepThreadsFrom channels on: worker threads.
int gps_tracking_words_batch(gps_ch_t *channel, int n_ch, const uint8_t *flags, int n_blocks, uint32_t first_tick,
                             int *changed_opt, int max_changed)
{
  if (!channel || !flags || n_ch <= 0 || n_blocks <= 0)
    return 0;
  StepPool &pool = StepPool::instance();
  static const int kThreadsFrom = [] { const char *e = std::getenv("GPSX_STEP_THREADS_FROM"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : kStepThreadsFrom; }();
  static const bool kForeignHooks = resolves_outside_this_library((const void *)&gps_nav_data_words_detection) ||
                                    resolves_outside_this_library((const void *)&gps_nav_data_decode_subframe);
  const int n_workers = (n_ch >= kThreadsFrom && !kForeignHooks) ? pool.size() : 1;
  static std::vector<std::vector<int>> changed;
  if ((int)changed.size() < n_workers)
    changed.resize(n_workers);
  // (channel, its flag before the launch) of the channels whose flag a word of this launch touched: a handful per launch --
  // the records themselves (1.7 KB each, gigabytes at a million channels) are only visited where a flag byte says so
  static std::vector<std::vector<std::pair<int, uint8_t>>> touched;
  if ((int)touched.size() < n_workers)
    touched.resize(n_workers);
  auto work = [&](int w) {
    std::vector<int> &mine = changed[w];
    std::vector<std::pair<int, uint8_t>> &seen = touched[w];
    mine.clear();
    seen.clear();
    const int lo = (int)((long)n_ch * w / n_workers), hi = (int)((long)n_ch * (w + 1) / n_workers);
    t_tick_valid = true;
    for (int ms = 0; ms < n_blocks; ms++) {
      const uint8_t *f = flags + (size_t)ms * n_ch;
      t_tick = first_tick + (uint32_t)ms;
      for (int c = lo; c < hi; c++) {
        const uint8_t v = f[c];
        if ((v & 128) == 0)   // not served this millisecond (GPSX_SCHED_MUX17)
          continue;
        // The record (1.7 KB, a cache miss at a million channels) is visited only where the flag byte has something for it: a
        // completed bit, a located edge, or the period flag changing between two milliseconds of this launch.
        const bool sync_changed = ms > 0 && (f[c - n_ch] & 128) && ((f[c - n_ch] ^ v) & 8);
        if ((v & (2 | 32)) == 0 && !sync_changed)
          continue;
        gps_nav_data_t &n = channel[c].nav_data;
        n.period_sync_ok_flag = (v & 8) ? 1 : 0;
        if (v & 2) {
          const uint8_t was = n.inv_polarity_flag;
          gps_nav_data_words_detection(&channel[c], (uint8_t)((v >> 2) & 1));
          if (n.inv_polarity_flag != was)
            seen.emplace_back(c, was);
        }
        if (v & 32) {
          n.accurate_swap_time = (uint8_t)((t_tick - 3u + ((v & 64) ? 2u : 1u)) % 20u);
          n.accurate_swap_ok = 1;
        }
      }
    }
    t_tick_valid = false;
    // listed once, in channel order, and only if the launch left the flag different from how it found it (a channel's FIRST
    // entry holds the value from before the launch: the sort is stable)
    std::stable_sort(seen.begin(), seen.end(), [](const std::pair<int, uint8_t> &a, const std::pair<int, uint8_t> &b) { return a.first < b.first; });
    for (size_t i = 0; i < seen.size(); i++)
      if ((i == 0 || seen[i].first != seen[i - 1].first) && channel[seen[i].first].nav_data.inv_polarity_flag != seen[i].second)
        mine.push_back(seen[i].first);
  };
  pool.run(n_workers, work);
  int total = 0;
  for (int w = 0; w < n_workers; w++)
    for (int c : changed[w]) {
      if (changed_opt && total < max_changed)
        changed_opt[total] = c;
      total++;
    }
  return total;
}

}  // extern "C"
