// rvc_abi.cpp -- the C ABI of include/reevr_amd/rvc.h and rvc_debug.h: every extern "C" entry point of the convolver sets. Entry
// points of a set with child sets forward to the children; everything else calls into rvc_state.cpp (init / lifetime) and
// rvc_schedule.cpp (process).
//
// What it replaces in the reference (paths relative to the reference tree):
//   TwoStageFFTConvolver::{init,process,clear,reset}   libs/FFTConvolver/TwoStageFFTConvolver.h:54-83
//   FFTConvolver::{init,process,clear,reset}           libs/FFTConvolver/FFTConvolver.h:52-80
//   Convolver::isFinished                              src/dsp/Convolver.cpp:79
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "rvc_set.h"

// Staging of many channels: the caller's per-channel buffers are copied into / out of the pinned staging rows by a few host
// threads (one thread moves ~15 GB/s: 0.55 ms each way for 4096 channels of 512 frames, more than the kernels of the call). One
// process-wide crew, started on first use; a call that finds it busy (another handle on another thread) copies inline -- nobody
// ever waits for it. Small sets (the plug-in's 2-4 channels) never touch it. (Callers that can write their audio into the staging
// rows themselves -- rvc_set_host_buffers -- skip the copy altogether.)
namespace {
// [copy-crew begin] (tests/test_host_logic.py compiles this block alone and stresses it on the CPU)
constexpr int kCrewWorkers = 15;
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}
struct CopyCrew {
  std::mutex busy;                       // one job at a time (try_lock: never waited for)
  std::mutex m;                          // sleepers only
  std::condition_variable wake;
  std::vector<std::thread> workers;
  std::atomic<const std::function<void(int)> *> job{nullptr};
  std::atomic<int> count{0}, finished{0};
  std::atomic<unsigned long long> next{~0ull};     // (epoch << 32) | next chunk index of the job of that epoch; ~0: no job
  std::atomic<unsigned long long> epoch{0};
  std::atomic<bool> stop{false};
  // claim chunks of job `ep` until none is left. A claim is a compare-exchange on (epoch, index): a straggler of an earlier job
  // finds another epoch (or ~0) there and leaves without having consumed anything.
  void drain(const unsigned long long ep) {
    for (;;) {
      unsigned long long v = next.load(std::memory_order_acquire);
      if ((v >> 32) != (ep & 0xffffffffull)) return;
      const int i = (int)(v & 0xffffffffull);
      if (i >= count.load(std::memory_order_relaxed)) return;
      if (!next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
      (*job.load(std::memory_order_relaxed))(i);
      finished.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      // a caller in a block loop comes back within tens of microseconds: spin that long before going to sleep (a condition
      // variable costs the NEXT job ~50 us per sleeper to wake)
      for (int i = 0; i < 20000 && epoch.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed); ++i) cpu_relax();
      if (epoch.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(m);
        wake.wait(lk, [&] { return stop.load() || epoch.load(std::memory_order_acquire) != seen; });
      }
      if (stop.load()) return;
      seen = epoch.load(std::memory_order_acquire);
      drain(seen);
    }
  }
  void start(int n) {
    for (int i = 0; i < n; ++i) workers.emplace_back([this] { worker(); });
  }
  ~CopyCrew() {
    { std::lock_guard<std::mutex> lk(m); stop.store(true); }
    wake.notify_all();
    for (std::thread &t : workers) t.join();
  }
  // f(i) for i < n, the calling thread included
  void run(int n, const std::function<void(int)> &f) {
    if (n <= 1 || !busy.try_lock()) { for (int i = 0; i < n; ++i) f(i); return; }
    if (workers.empty()) {
      // crew size: the caller + up to kCrewWorkers helpers, never more than half the host's hardware threads; RVC_COPY_THREADS (read
      // once, at the first large call) overrides it for measurements
      int w = std::min(kCrewWorkers, (int)std::thread::hardware_concurrency() / 2 - 1);
      if (const char *e = std::getenv("RVC_COPY_THREADS")) w = std::atoi(e) - 1;
      start(std::max(1, std::min(w, 63)));
    }
    const unsigned long long ep = (epoch.load(std::memory_order_relaxed) + 1) & 0xffffffffull;
    job.store(&f, std::memory_order_relaxed);
    count.store(n, std::memory_order_relaxed);
    finished.store(0, std::memory_order_relaxed);
    next.store(ep << 32, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(m); epoch.store(ep, std::memory_order_release); }
    wake.notify_all();
    drain(ep);
    while (finished.load(std::memory_order_acquire) < n) cpu_relax();     // (chunks are tens of microseconds long)
    next.store(~0ull, std::memory_order_release);                          // no straggler claims anything of this job from here on
    busy.unlock();
  }
};
// [copy-crew end]
CopyCrew &copy_crew() { static CopyCrew c; return c; }
constexpr size_t kCrewMinBytes = (size_t)1 << 20;   // below this one thread is faster than waking the crew

// rows of `len` floats between the caller's per-channel buffers and the staging rows (max_len apart) of the `n` sets in `sets` --
// a set's child sets in ONE job --, set k serving the caller's channels c0[k] ..; a caller's pointer that IS the staging row
// (rvc_set_host_buffers) is skipped. Exactly one of in / out is given.
void stage_rows(rvc_set *const *sets, const int *c0, int n, const float *const *in, float *const *out, size_t len) {
  const size_t bytes = len * sizeof(float);
  int total = 0;
  for (int k = 0; k < n; ++k) total += sets[k]->nch;
  auto one = [&](int k, int c) {
    rvc_set *s = sets[k];
    float *row = (in ? s->h_in : s->h_out) + (size_t)c * s->max_len;
    if (in) { if (in[c0[k] + c] != row) std::memcpy(row, in[c0[k] + c], bytes); }
    else if (out[c0[k] + c] && out[c0[k] + c] != row) std::memcpy(out[c0[k] + c], row, bytes);
  };
  auto range = [&](int lo, int hi) {               // global channel indices [lo, hi) over the concatenated sets
    int base = 0;
    for (int k = 0; k < n && lo < hi; ++k) {
      const int e = base + sets[k]->nch;
      for (; lo < hi && lo < e; ++lo) one(k, lo - base);
      base = e;
    }
  };
  if ((size_t)total * bytes < kCrewMinBytes) { range(0, total); return; }
  const int chunks = std::min(total, 64);
  copy_crew().run(chunks, [&](int i) { range((int)((long long)total * i / chunks), (int)((long long)total * (i + 1) / chunks)); });
}
}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

rvc_set *rvc_set_create(int n_channels, int device, unsigned flags) {
  if (n_channels < 1) return nullptr;
  rvc_set *s = new (std::nothrow) rvc_set();
  if (!s) return nullptr;
  s->nch = n_channels;
  s->device = device;
  s->flags = flags;
  s->tune = tune_defaults_now();
  s->timing = (flags & RVC_FLAG_TIMING) != 0;
  if ((flags & RVC_FLAG_PERSISTENT) != 0) { s->err = RVC_ERR_UNSUPPORTED; s->errstr = "RVC_FLAG_PERSISTENT was removed (round 4)"; }
  s->in_ptrs.assign((size_t)n_channels, nullptr);
  s->out_ptrs.assign((size_t)n_channels, nullptr);
  return s;
}

rvc_set *rvc_set_create_tuned(int n_channels, int device, unsigned flags, const char *knobs) {
  Tuning t = tune_defaults_now();
  if (!apply_knobs(t, knobs)) return nullptr;
  rvc_set *s = rvc_set_create(n_channels, device, flags);
  if (s) s->tune = t;
  return s;
}

void rvc_set_destroy(rvc_set *s) {
  if (!s) return;
  drop_kids(s);
  free_device_state(s);
  if (s->timed_base) hipEventDestroy(s->timed_base);
  drop_streams(s);
  delete s;
}

int rvc_set_init(rvc_set *s, size_t head_block, size_t tail_block, const float *const *irs,
                 const size_t *ir_lens, size_t max_len) {
  if (!s) return 0;
  if (irs && ir_lens && make_kids(s, subset_count(s, head_block, max_len))) {
    bool ok = true;
    size_t longest = 0;                  // (trimmed like do_init does: TwoStageFFTConvolver.cpp:107-110)
    for (int c = 0; c < s->nch; ++c) {
      size_t l = irs[c] ? ir_lens[c] : 0;
      while (l > longest && std::fabs(irs[c][l - 1]) < 0.000001f) --l;
      longest = std::max(longest, l);
    }
    for (size_t k = 0; k < s->kids.size(); ++k) {
      s->kids[k]->longest_hint = longest;
      ok = rvc_set_init(s->kids[k], head_block, tail_block, irs + s->kid_c0[k], ir_lens + s->kid_c0[k], max_len) != 0 && ok;
    }
    adopt_kid_geometry(s, ok);
    return ok ? 1 : 0;
  }
  const bool ok = do_init(s, head_block, tail_block, true, irs, ir_lens, max_len);
  if (!ok) release_after_failed_init(s);
  return ok ? 1 : 0;
}

int rvc_set_init_uniform(rvc_set *s, size_t block, const float *const *irs, const size_t *ir_lens,
                         size_t max_len) {
  if (!s) return 0;
  if (irs && ir_lens && make_kids(s, subset_count(s, block, max_len))) {
    bool ok = true;
    for (size_t k = 0; k < s->kids.size(); ++k)
      ok = rvc_set_init_uniform(s->kids[k], block, irs + s->kid_c0[k], ir_lens + s->kid_c0[k], max_len) != 0 && ok;
    adopt_kid_geometry(s, ok);
    return ok ? 1 : 0;
  }
  const bool ok = do_init(s, block, 0, false, irs, ir_lens, max_len);
  if (!ok) release_after_failed_init(s);
  return ok ? 1 : 0;
}

int rvc_set_init_impulse(rvc_set *s, size_t head_block, size_t tail_block, rvc_impulse *m, const int *channels,
                         size_t max_len) {
  if (!s) return 0;
  if (channels && make_kids(s, subset_count(s, head_block, max_len))) {
    bool ok = true;
    size_t longest = 0;
    rvc::ImpulseView pv{};
    if (m && rvc::impulse_view(m, &pv))
      for (int c = 0; c < s->nch; ++c)
        if (channels[c] >= 0 && channels[c] < pv.channels) longest = std::max(longest, (size_t)pv.trimmed[channels[c]]);
    for (size_t k = 0; k < s->kids.size(); ++k) {
      s->kids[k]->longest_hint = longest;
      ok = rvc_set_init_impulse(s->kids[k], head_block, tail_block, m, channels + s->kid_c0[k], max_len) != 0 && ok;
    }
    adopt_kid_geometry(s, ok);
    return ok ? 1 : 0;
  }
  rvc::ImpulseView v{};
  if (!m || !channels || !rvc::impulse_view(m, &v)) {
    s->err = RVC_OK;
    fail(s, RVC_ERR_BAD_ARG, hipSuccess, "impulse");
    return 0;
  }
  std::vector<const float *> irs(s->nch, nullptr);
  std::vector<size_t> lens(s->nch, 0);
  for (int c = 0; c < s->nch; ++c) {
    const int k = channels[c];
    if (k < 0 || k >= v.channels || v.device != s->device) {
      s->err = RVC_OK;
      fail(s, RVC_ERR_BAD_ARG, hipSuccess, "impulse channel / device");
      return 0;
    }
    irs[c] = v.size ? v.ch[k] : nullptr;
    lens[c] = v.trimmed[k];
  }
  const bool ok = do_init(s, head_block, tail_block, true, irs.data(), lens.data(), max_len, /*on_device=*/true);
  if (!ok) release_after_failed_init(s);
  return ok ? 1 : 0;
}

void rvc_set_process_device(rvc_set *s, const float *d_in, size_t in_stride, float *d_out,
                            size_t out_stride, size_t len) {
  if (!s || len == 0) return;
  if (!s->kids.empty()) {
    (void)fence_children_in(s);
    forward_device_call(s, d_in, in_stride, d_out, out_stride, len);
    (void)fence_children_out(s);
    return;
  }
  if (!s->live || s->err != RVC_OK) {   // not initialised / empty IR / failed: zeros
    zero_device_out(s, d_out, out_stride, len);
    return;
  }
  if (!use_device(s)) return;
  const TuneScope tune_scope(s);
  size_t done = 0;
  while (done < len) {   // calls longer than max_len are split; results are call-pattern independent
    const size_t chunk = std::min(len - done, s->max_len);
    if (!step_device(s, d_in + done, in_stride, d_out + done, out_stride, chunk)) {
      zero_device_out(s, d_out, out_stride, len);
      return;
    }
    done += chunk;
  }
}

void rvc_set_process_device_blocks(rvc_set *s, const float *d_in, size_t in_stride, float *d_out,
                                   size_t out_stride, size_t len, size_t block) {
  if (!s || block == 0) return;
  if (!s->kids.empty()) {        // one fence around the whole loop: the buffers are complete before and read after it
    (void)fence_children_in(s);
    for (size_t done = 0; done < len; done += block)
      forward_device_call(s, d_in + done, in_stride, d_out + done, out_stride, std::min(block, len - done));
    (void)fence_children_out(s);
    return;
  }
  for (size_t done = 0; done < len; done += block)
    rvc_set_process_device(s, d_in + done, in_stride, d_out + done, out_stride, std::min(block, len - done));
}

long rvc_set_process_device_blocks_stamped(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride,
                                           size_t len, size_t block, double *done_ms) {
  if (!s || block == 0) return -1;
  std::vector<rvc_set *> units = s->kids.empty() ? std::vector<rvc_set *>{s} : s->kids;
  for (rvc_set *u : units)
    if (!u->streams_ok) return -1;
  if (hipSetDevice(s->device) != hipSuccess) return -1;
  const size_t calls = (len + block - 1) / block, nu = units.size();
  std::vector<hipEvent_t> ev((calls + 1) * nu, nullptr);
  bool ok = true;
  for (hipEvent_t &e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
  if (ok) {
    rvc_set_sync(s);
    if (!s->kids.empty()) (void)fence_children_in(s);
    for (size_t u = 0; u < nu; ++u) ok = ok && hipEventRecord(ev[u], units[u]->st_main) == hipSuccess;     // the loop starts
    size_t call = 0;
    for (size_t done = 0; done < len; done += block, ++call) {
      const size_t n = std::min(block, len - done);
      if (!s->kids.empty()) forward_device_call(s, d_in + done, in_stride, d_out + done, out_stride, n);
      else rvc_set_process_device(s, d_in + done, in_stride, d_out + done, out_stride, n);
      for (size_t u = 0; u < nu; ++u) ok = ok && hipEventRecord(ev[(call + 1) * nu + u], units[u]->st_main) == hipSuccess;
    }
    if (!s->kids.empty()) (void)fence_children_out(s);
    rvc_set_sync(s);
    // one clock: the children's first stamps were recorded back to back on idle streams; later stamps relative to the earliest
    for (size_t i = 0; ok && i < calls; ++i) {
      double t = 0.0;
      for (size_t u = 0; u < nu; ++u) {
        float ms = 0.f;
        ok = ok && hipEventElapsedTime(&ms, ev[0], ev[(i + 1) * nu + u]) == hipSuccess;
        t = std::max(t, (double)ms);
      }
      if (done_ms) done_ms[i] = t;
    }
  }
  for (hipEvent_t e : ev)
    if (e) hipEventDestroy(e);
  return ok ? (long)calls : -1;
}

// The host's per-block loop over HOST buffers with a stopwatch around every call: what the plug-in's audio thread sees
// per process() (pinned staging + hand-off + kernel + copy back), measured without any host-language overhead.
void rvc_set_process_host_blocks_timed(rvc_set *s, const float *const *in, float *const *out, size_t len, size_t block,
                                       double *us_per_call) {
  if (!s || !in || !out || block == 0) return;
  std::vector<const float *> ins((size_t)s->nch);
  std::vector<float *> outs((size_t)s->nch);
  size_t call = 0;
  for (size_t done = 0; done < len; done += block, ++call) {
    const size_t n = std::min(block, len - done);
    for (int c = 0; c < s->nch; ++c) { ins[c] = in[c] + done; outs[c] = out[c] + done; }
    const auto a = std::chrono::steady_clock::now();
    rvc_set_process_begin(s, ins.data(), n);
    rvc_set_process_end(s, outs.data());
    if (us_per_call) us_per_call[call] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
  }
}

// process_begin behind the staging (h_in holds the call's input): copy-in, kernels, copy-out enqueued; false = nothing is pending
static bool begin_core(rvc_set *s, size_t len) {
  const TuneScope tune_scope(s);
  // Per-block calls of SMALL sets (the latency path: one fused launch) skip both DMA copies: the pinned staging
  // buffers are device-visible, the kernel reads its 2 KB per channel over PCIe and writes the
  // result straight back; the host polls the flags its workgroups publish. Longer calls and many channels use DMA (a kernel
  // that fetches megabytes over PCIe with its own loads holds its CUs for the whole transfer; the copy engines do not).
  const long long hb = (long long)s->A.B;
  // (a call across one block boundary is two such launches: step_device)
  const bool small = s->tune.host_zero_copy >= 0 ? s->tune.host_zero_copy != 0 : (size_t)s->nch * len * sizeof(float) <= kZeroCopyMaxBytes;
  s->zero_copy = small && !s->block_general && rvc::fused_supported(s->A.logB, s->A.f64()) && ((s->n + (long long)len - 1) / hb - s->n / hb) <= 1;
  bool ok = true;
  const size_t ml = s->max_len;
  if (!s->zero_copy) {
    if (len == ml) ok = hipMemcpyAsync(s->d_in, s->h_in, sizeof(float) * len * s->nch, hipMemcpyHostToDevice, s->st_main) == hipSuccess;
    else ok = hipMemcpy2DAsync(s->d_in, sizeof(float) * ml, s->h_in, sizeof(float) * ml, sizeof(float) * len, (size_t)s->nch,
                               hipMemcpyHostToDevice, s->st_main) == hipSuccess;
  }
  s->out_copy_len = len;                       // step_device emits the copy-back / event right behind the output kernel
  ok = ok && (s->zero_copy ? step_device(s, s->h_in, ml, s->h_out, ml, len)
                           : step_device(s, s->d_in, ml, s->d_out, ml, len));
  ok = ok && emit_output_copy(s);              // (paths whose last kernel is the output kernel)
  s->out_copy_len = 0;
  if (!ok) fail(s, RVC_ERR_HIP, hipGetLastError(), "process_begin");
  s->pending_ok = ok;
  return ok;
}
// whether this call of `len` frames will run on the set at all (else process_end delivers zeros)
static bool begin_accepts(rvc_set *s, const float *const *in, size_t len) {
  s->pending_len = len;
  s->pending_ok = false;
  s->flag_count = 0;
  if (len == 0 || !s->live || s->err != RVC_OK || !in) return false;
  if (len > s->max_len) return false;   // refused: process_end writes zeros for this call; the handle stays usable
                                        // (rvc_set_process splits long calls itself)
  return use_device(s);
}

void rvc_set_process_begin(rvc_set *s, const float *const *in, size_t len) {
  if (!s) return;
  if (!s->kids.empty()) {
    // the children's staging in ONE job of the copy crew (a job costs its wake-up whatever its size), then child by child: copy-in
    // and kernels of child k are enqueued while child k + 1 ...
    std::vector<rvc_set *> &run = s->stage_sets;
    std::vector<int> &c0 = s->stage_c0;
    run.clear(); c0.clear();
    for (size_t k = 0; k < s->kids.size(); ++k)
      if (begin_accepts(s->kids[k], in, len)) { run.push_back(s->kids[k]); c0.push_back(s->kid_c0[k]); }
    if (run.empty()) return;
    stage_rows(run.data(), c0.data(), (int)run.size(), in, nullptr, len);
    for (rvc_set *k : run) begin_core(k, len);
    return;
  }
  if (!begin_accepts(s, in, len)) return;
  const int zero = 0;
  rvc_set *one[1] = {s};
  stage_rows(one, &zero, 1, in, nullptr, len);
  begin_core(s, len);
}

// wait for the output of the pending call to be in h_out; false = deliver zeros
static bool end_wait(rvc_set *s) {
  const size_t len = s->pending_len;
  if (len == 0) return false;
  bool ok = s->pending_ok;
  if (ok && s->flag_count > 0) {
    // poll the completion flags the audio workgroups write behind their output stores
    const unsigned want = s->flag_seq;
    const int nf = s->flag_count;
    s->flag_count = 0;
    volatile unsigned *f = s->h_flags;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nf; ++i) {
      unsigned spins = 0;
      while (f[i] != want) {
        if ((++spins & 0x3ffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
          hipSetDevice(s->device);               // lost kernel? fall back to the stream and report what it says
          ok = hipStreamSynchronize(s->st_main) == hipSuccess && f[i] == want;
          if (!ok) fail(s, RVC_ERR_HIP, hipGetLastError(), "process_end (completion flag)");
          break;
        }
      }
      if (!ok) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else if (ok) {
    hipSetDevice(s->device);
    ok = hipEventSynchronize(s->ev_out) == hipSuccess;   // output copied back; later stream work may still run
    if (!ok) fail(s, RVC_ERR_HIP, hipGetLastError(), "process_end");
  }
  s->pending_ok = false;
  return ok;
}
static void end_zeros(rvc_set *s, float *const *out, size_t len) {
  for (int c = 0; c < s->nch; ++c)
    if (out[c]) std::memset(out[c], 0, len * sizeof(float));   // not initialised / empty IR / failed: zeros
}

void rvc_set_process_end(rvc_set *s, float *const *out) {
  if (!s || !out) return;
  if (!s->kids.empty()) {
    std::vector<rvc_set *> &run = s->stage_sets;
    std::vector<int> &c0 = s->stage_c0;
    run.clear(); c0.clear();
    size_t len = 0;
    for (size_t k = 0; k < s->kids.size(); ++k) {
      rvc_set *c = s->kids[k];
      const size_t l = c->pending_len;
      if (l == 0) continue;
      if (end_wait(c)) { run.push_back(c); c0.push_back(s->kid_c0[k]); len = l; }
      else end_zeros(c, out + s->kid_c0[k], l);
      c->pending_len = 0;
    }
    if (!run.empty()) stage_rows(run.data(), c0.data(), (int)run.size(), nullptr, out, len);
    return;
  }
  const size_t len = s->pending_len;
  if (len == 0) return;
  const bool ok = end_wait(s);
  s->pending_len = 0;
  const int zero = 0;
  rvc_set *one[1] = {s};
  if (ok) stage_rows(one, &zero, 1, nullptr, out, len);
  else end_zeros(s, out, len);
}

int rvc_set_host_buffers(rvc_set *s, float **in, float **out) {
  if (!s) return 0;
  if (!s->kids.empty()) {
    int ok = 1;
    for (size_t k = 0; k < s->kids.size(); ++k)
      ok &= rvc_set_host_buffers(s->kids[k], in ? in + s->kid_c0[k] : nullptr, out ? out + s->kid_c0[k] : nullptr);
    return ok;
  }
  for (int c = 0; c < s->nch; ++c) {
    if (in) in[c] = s->live ? s->h_in + (size_t)c * s->max_len : nullptr;
    if (out) out[c] = s->live ? s->h_out + (size_t)c * s->max_len : nullptr;
  }
  return s->live ? 1 : 0;
}

void rvc_set_process(rvc_set *s, const float *const *in, float *const *out, size_t len) {
  if (!s || len == 0 || !out) return;
  size_t done = 0;
  std::vector<const float *> &ins = s->in_ptrs;      // sized at create: no allocation here (FFTConvolver.h:44-47)
  std::vector<float *> &outs = s->out_ptrs;
  while (done < len) {   // calls longer than max_len are split; results are call-pattern independent
    const size_t chunk = std::min(len - done, s->max_len ? s->max_len : len);
    for (int c = 0; c < s->nch; ++c) {
      ins[c] = in ? in[c] + done : nullptr;
      outs[c] = out[c] ? out[c] + done : nullptr;
    }
    rvc_set_process_begin(s, in ? ins.data() : nullptr, chunk);
    rvc_set_process_end(s, outs.data());
    done += chunk;
  }
}

void rvc_set_clear(rvc_set *s) {
  if (!s) return;
  for (rvc_set *k : s->kids) rvc_set_clear(k);
  if (!s->live) return;
  // Outstanding tail jobs still write into rings; let them finish, then restart the clock.
  hipSetDevice(s->device);
  hipStreamSynchronize(s->st_bg);
  hipStreamSynchronize(s->st_main);
  drop_jobs(s);
  s->n = 0;
  s->tail_fft_done = 0;
  s->tail_out_done = s->T.delay;
  s->ypre_block = -1;
  s->xa_next = 0;
  s->w_next = 0;
  s->xt_valid_lo = 0;
  s->tA.restart(); s->tT.restart();
}

void rvc_set_reset(rvc_set *s) {
  if (!s) return;
  drop_kids(s);
  free_device_state(s);
  s->err = RVC_OK;
  s->errstr.clear();
}

int rvc_set_is_finished(rvc_set *s) {
  if (!s) return 1;
  for (rvc_set *k : s->kids)
    if (!rvc_set_is_finished(k)) return 0;
  if (!s->live) return 1;
  hipSetDevice(s->device);
  if (s->st_bg == s->st_main) return 1;      // no tail stream: the tail job runs inline, nothing is ever in the background
  return hipStreamQuery(s->st_bg) == hipSuccess ? 1 : 0;
}

void rvc_set_sync(rvc_set *s) {
  if (!s) return;
  for (rvc_set *k : s->kids) rvc_set_sync(k);
  if (!s->streams_ok) return;
  hipSetDevice(s->device);
  hipStreamSynchronize(s->st_bg);
  hipStreamSynchronize(s->st_main);
}

void rvc_set_fork(rvc_set *s) {
  if (s && !s->kids.empty()) (void)fence_children_in(s, true);
}
void rvc_set_join(rvc_set *s) {
  if (s && !s->kids.empty()) (void)fence_children_out(s, true);
}

int rvc_set_channels(const rvc_set *s) { return s ? s->nch : 0; }
size_t rvc_set_head_block(const rvc_set *s) { return s ? s->head : 0; }
size_t rvc_set_tail_block(const rvc_set *s) { return s ? s->tail : 0; }
size_t rvc_set_max_len(const rvc_set *s) { return s ? s->max_len : 0; }
int rvc_set_partitions(const rvc_set *s, int stage) {
  if (s && !s->kids.empty()) {          // (the largest over the children: channels may carry IRs of different lengths)
    int p = 0;
    for (const rvc_set *k : s->kids) p = std::max(p, rvc_set_partitions(k, stage));
    return p;
  }
  return !s ? 0 : (stage == 0 ? s->A.P : (stage == 1 ? s->T.P : s->W.P));
}
int rvc_set_tile_rows(const rvc_set *s, int stage) {
  if (!s) return 0;
  if (!s->kids.empty()) return rvc_set_tile_rows(s->kids[0], stage);
  const Tile &t = stage == 0 ? s->tA : s->tT;
  return (s->live && t.on) ? t.K1 : 0;
}
int rvc_set_subsets(const rvc_set *s) { return !s ? 0 : (s->kids.empty() ? 1 : (int)s->kids.size()); }
void *rvc_set_stream(rvc_set *s, int which) {
  if (!s || which < 0) return nullptr;
  if (!s->kids.empty()) {               // 0: the stream the caller orders against; child k's streams are 2 + 2 k (foreground)
    if (which < 2) return which == 0 ? rvc_set_stream(s->kids[0], 0) : nullptr;   // and 3 + 2 k (tail): diagnostics
    const size_t k = (size_t)(which - 2) / 2;
    return k < s->kids.size() ? rvc_set_stream(s->kids[k], which % 2) : nullptr;
  }
  return which == 0 ? (void *)s->st_main : (which == 1 ? (void *)s->st_bg : nullptr);
}
int rvc_last_error(const rvc_set *s) {
  if (!s) return RVC_ERR_BAD_ARG;
  for (const rvc_set *k : s->kids)
    if (k->err != RVC_OK) return k->err;
  return s->err;
}
const char *rvc_last_error_string(const rvc_set *s) {
  if (!s) return "null handle";
  for (const rvc_set *k : s->kids)
    if (k->err != RVC_OK) return k->errstr.c_str();
  return s->errstr.c_str();
}

long rvc_set_kernel_time(rvc_set *s, int kernel, double *total_ms) {
  if (total_ms) *total_ms = 0.0;
  if (!s || kernel < 0 || kernel >= kNumKernelIds) return 0;
  if (!s->kids.empty()) {
    long n = 0;
    for (rvc_set *k : s->kids) {
      double ms = 0.0;
      n += rvc_set_kernel_time(k, kernel, &ms);
      if (total_ms) *total_ms += ms;
    }
    return n;
  }
  rvc_set_sync(s);
  fold_timing(s, kernel);
  if (total_ms) *total_ms = s->timed_ms[kernel];
  return s->timed_n[kernel];
}

void rvc_set_kernel_time_reset(rvc_set *s) {
  if (!s) return;
  for (rvc_set *k : s->kids) { k->timed_parent = s; rvc_set_kernel_time_reset(k); }
  if (!s->timed_parent) {          // the clock of rvc_set_kernel_intervals starts here
    hipStream_t st = s->kids.empty() ? s->st_main : s->kids[0]->st_main;
    if (st && hipSetDevice(s->device) == hipSuccess) {
      if (!s->timed_base) hipEventCreate(&s->timed_base);
      if (s->timed_base) { hipEventRecord(s->timed_base, st); hipEventSynchronize(s->timed_base); }
    }
  }
  rvc_set_sync(s);
  drop_timing(s);
}

long rvc_set_kernel_intervals(rvc_set *s, int kernel, double *start_ms, double *end_ms, long cap) {
  if (!s || kernel < 0 || kernel >= kNumKernelIds) return 0;
  long n = 0;
  if (!s->kids.empty()) {
    for (rvc_set *k : s->kids) {
      const long got = rvc_set_kernel_intervals(k, kernel, start_ms ? start_ms + n : nullptr, end_ms ? end_ms + n : nullptr,
                                                cap > n ? cap - n : 0);
      n += got;
    }
    return n;
  }
  rvc_set_sync(s);
  fold_timing(s, kernel);
  for (const auto &iv : s->timed_iv[kernel]) {
    if (n < cap && start_ms && end_ms) { start_ms[n] = iv.first; end_ms[n] = iv.second; }
    ++n;
  }
  return n;
}

void rvc_set_timing(rvc_set *s, int enable) {
  if (!s) return;
  s->timing = enable != 0;
  for (rvc_set *k : s->kids) k->timing = s->timing;
}

rvc_set *rvc_create(int device) { return rvc_set_create(1, device, RVC_FLAG_BG_STREAM); }
int rvc_init(rvc_set *h, size_t head_block, size_t tail_block, const float *ir, size_t ir_len) {
  const float *irs[1] = {ir};
  const size_t lens[1] = {ir_len};
  return rvc_set_init(h, head_block, tail_block, irs, lens, 0);
}
void rvc_process(rvc_set *h, const float *in, float *out, size_t len) {
  const float *ins[1] = {in};
  float *outs[1] = {out};
  rvc_set_process(h, ins, outs, len);
}
void rvc_clear(rvc_set *h) { rvc_set_clear(h); }
void rvc_reset(rvc_set *h) { rvc_set_reset(h); }
int rvc_is_finished(rvc_set *h) { return rvc_set_is_finished(h); }
void rvc_destroy(rvc_set *h) { rvc_set_destroy(h); }

// Known-answer entries for the transforms themselves (AudioFFT::fft / ifft, AudioFFT.cpp:114-159): one bare
// 2B-point real transform through the SAME kernels the convolver stages launch (launch_fft_fwd / launch_fft_inv
// with the stage's twiddle tables), host buffers in and out, split-complex like the reference's facade.
static int debug_fft(int device, size_t n, int f64, bool inverse, const float *in_t, float *out_t, const float *re_in,
                     const float *im_in, float *re_out, float *im_out) {
  if (n < 2 || (n & (n - 1)) != 0) return 0;                       // power of two (AudioFFT.cpp:996)
  const size_t B = n / 2;
  const int logB = ilog2(B);
  if (logB > (f64 ? 13 : 14)) return 0;
  rvc_set *s = rvc_set_create(1, device, f64 ? RVC_FLAG_FFT_F64 : 0u);
  if (!s) return 0;
  const TuneScope tune_scope(s);
  bool ok = ensure_streams(s) && use_device(s);
  Stage g;
  g.B = B; g.logB = logB; g.set64(f64 ? 3 : 0);
  float *d_t = nullptr;
  float2 *d_f = nullptr;
  ok = ok && make_twiddles(s, g);
  ok = ok && hipMalloc(&d_t, sizeof(float) * 2 * n) == hipSuccess && hipMalloc(&d_f, sizeof(float2) * 2 * B) == hipSuccess;
  if (ok && !inverse) {
    ok = hipMemcpy(d_t, in_t, sizeof(float) * n, hipMemcpyHostToDevice) == hipSuccess;
    rvc::FwdArgs a{};
    a.src = d_t; a.src_chan_stride = (long long)n; a.src_mask = ~0ull;
    a.seg0 = 0; a.valid_len = (int)n; a.lo = 0; a.hi = (long long)n;
    a.tw = g.twp(g.f64f); a.wsplit = g.wsp(g.f64f); a.tw8 = g.t8p(g.f64f); a.tw8_half = g.t8h(g.f64f); a.tw_half = g.twh(g.f64f);
    a.dst = d_f; a.dst_chan_stride = (long long)B; a.row0 = 0; a.row_mask = ~0ull;
    ok = ok && rvc::launch_fft_fwd(logB, g.f64f, a, 1, 1, s->st_main) == hipSuccess &&
         hipStreamSynchronize(s->st_main) == hipSuccess;
    std::vector<float2> X(B);
    ok = ok && hipMemcpy(X.data(), d_f, sizeof(float2) * B, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) {   // unpack: bin 0 carries (DC, Nyquist), both real (AudioFFT.cpp:130-136)
      re_out[0] = X[0].x; im_out[0] = 0.f;
      re_out[B] = X[0].y; im_out[B] = 0.f;
      for (size_t k = 1; k < B; ++k) { re_out[k] = X[k].x; im_out[k] = X[k].y; }
    }
  } else if (ok) {
    // the stage kernel delivers samples [B, 2B) of the inverse (overlap-save); the first half is the second half
    // of the same spectrum shifted by B samples, i.e. with bins multiplied by (-1)^k
    std::vector<float2> Y(2 * B);
    for (int half = 0; half < 2; ++half) {
      float2 *y = Y.data() + (size_t)half * B;
      const float sN = (half == 0 && (B & 1)) ? -1.f : 1.f;          // Nyquist bin k = B
      y[0] = make_float2(re_in[0], sN * re_in[B]);
      for (size_t k = 1; k < B; ++k) {
        const float sg = (half == 0 && (k & 1)) ? -1.f : 1.f;
        y[k] = make_float2(sg * re_in[k], sg * im_in[k]);
      }
    }
    ok = hipMemcpy(d_f, Y.data(), sizeof(float2) * 2 * B, hipMemcpyHostToDevice) == hipSuccess;
    rvc::InvArgs v{};
    v.Y = d_f; v.y_chan_stride = (long long)(2 * B); v.tw = g.twp(g.f64i); v.wsplit = g.wsp(g.f64i); v.tw8 = g.t8p(g.f64i); v.tw8_half = g.t8h(g.f64i); v.tw_half = g.twh(g.f64i);
    v.blk0 = 0; v.dst = d_t; v.dst_chan_stride = (long long)n; v.dst_origin = 0; v.dst_mask = ~0ull;
    v.lo = 0; v.hi = (long long)n; v.add = nullptr;
    ok = ok && rvc::launch_fft_inv(logB, g.f64i, v, 2, 1, s->st_main) == hipSuccess &&
         hipStreamSynchronize(s->st_main) == hipSuccess;
    ok = ok && hipMemcpy(out_t, d_t, sizeof(float) * n, hipMemcpyDeviceToHost) == hipSuccess;
  }
  hipFree(d_t); hipFree(d_f);
  free_stage(s, g);
  rvc_set_destroy(s);
  return ok ? 1 : 0;
}

int rvc_debug_rfft(int device, size_t n, int f64, const float *data, float *re, float *im) {
  if (!data || !re || !im) return 0;
  return debug_fft(device, n, f64, false, data, nullptr, nullptr, nullptr, re, im);
}
int rvc_debug_irfft(int device, size_t n, int f64, float *data, const float *re, const float *im) {
  if (!data || !re || !im) return 0;
  return debug_fft(device, n, f64, true, nullptr, data, re, im, nullptr, nullptr);
}

// One launch of a delay-line kernel on caller-provided rows: the complex multiply-accumulate kernels in isolation
// (tests: against Utilities.cpp:62-111 applied the way FFTConvolver.cpp:176-187 applies it).
int rvc_debug_fdl(int device, int kind, int channels, int B, int P, int M, int delay, long long k0, int ring_rows,
                  const float *H, const float *X, const float *Yadd, float *Y, long long x_hi, long long x_from) {
  if (!H || !X || !Y || channels < 1 || B < 2 || (B & (B - 1)) || P < 1 || M < 1 || ring_rows < 1 || (ring_rows & (ring_rows - 1)))
    return 0;
  if (kind == 1 && M != 8 && M != 16 && M != 32) return 0;
  if (hipSetDevice(device) != hipSuccess) return 0;
  const Tuning tune_now = tune_defaults_now();         // (no set: the variants rvc_debug_set_tuning has selected)
  const TuneScope tune_scope(&tune_now.launch);
  const size_t nh = (size_t)channels * P * B, nx = (size_t)channels * ring_rows * B, ny = (size_t)channels * M * B;
  const size_t nadd = Yadd ? (kind == 1 ? ny : (size_t)channels * B) : 0;
  float2 *dH = nullptr, *dX = nullptr, *dY = nullptr, *dA = nullptr;
  bool ok = hipMalloc(&dH, nh * sizeof(float2)) == hipSuccess && hipMalloc(&dX, nx * sizeof(float2)) == hipSuccess &&
            hipMalloc(&dY, ny * sizeof(float2)) == hipSuccess && (!nadd || hipMalloc(&dA, nadd * sizeof(float2)) == hipSuccess);
  ok = ok && hipMemcpy(dH, H, nh * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(dX, X, nx * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemset(dY, 0xFF, ny * sizeof(float2)) == hipSuccess &&
       (!nadd || hipMemcpy(dA, Yadd, nadd * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess);
  if (ok) {
    rvc::FirArgs a{};
    a.H = dH; a.h_chan_stride = (long long)P * B;
    a.X = dX; a.x_chan_stride = (long long)ring_rows * B; a.x_row_mask = (unsigned long long)ring_rows - 1;
    a.Y = dY; a.y_chan_stride = (long long)M * B;
    a.k0 = k0; a.M = M; a.P = P; a.delay = delay; a.B = B; a.tag = delay ? 1 : 0;
    if (kind == 1) {             // sweep: output row j in slot (k0 + j) & (M - 1); Yadd = first-level rows (second-level form)
      a.x_hi = x_hi; a.x_from = x_from; a.y_row_mask = (unsigned)(M - 1);
      a.Ybase = dA; a.ybase_chan_stride = (long long)M * B; a.ybase_row_mask = (unsigned)(M - 1);
      ok = rvc::launch_fdl_sweep(a, channels, nullptr) == hipSuccess;
    } else {                     // launch_fir: the LDS-tiled / row / patch kernel by shape; Yadd = a sweep's row (M = 1)
      a.Yadd = (M == 1) ? dA : nullptr; a.yadd_chan_stride = B;
      ok = rvc::launch_fir(a, channels, nullptr) == hipSuccess;
    }
    ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(Y, dY, ny * sizeof(float2), hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(dH); (void)hipFree(dX); (void)hipFree(dY); (void)hipFree(dA);
  return ok ? 1 : 0;
}

long rvc_debug_guard_check(rvc_set *s) {
  if (!s) return -1;
  if (!s->kids.empty()) {
    long bad = 0;
    for (rvc_set *k : s->kids) {
      const long b = rvc_debug_guard_check(k);
      if (b < 0) return -1;
      bad += b;
    }
    return bad;
  }
  if (s->guards.empty()) return s->tune.guard ? 0 : -1;
  hipSetDevice(s->device);
  rvc_set_sync(s);
  std::vector<unsigned char> band(kGuardBytes);
  long bad = 0;
  for (const rvc_set::GuardRec &g : s->guards) {
    if (g.fenced) continue;                     // (fence mode: a stray access has faulted already)
    for (int side = 0; side < 2; ++side) {
      const char *src = side == 0 ? g.base : g.base + kGuardBytes + g.bytes;
      if (hipMemcpy(band.data(), src, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
      for (unsigned char v : band) bad += v != 0xFF;
    }
  }
  return bad;
}

int rvc_debug_fence_probe(rvc_set *s) {
  if (!s) return -1;
  if (!s->kids.empty()) return rvc_debug_fence_probe(s->kids[0]);
  hipSetDevice(s->device);
  rvc_set_sync(s);
  for (const rvc_set::GuardRec &g : s->guards)
    if (g.fenced) {
      unsigned char buf[16];
      const hipError_t in = hipMemcpy(buf, g.base + g.mapped - 16, 16, hipMemcpyDeviceToHost);    // last bytes of the mapping
      const hipError_t out = hipMemcpy(buf, g.base + g.mapped, 16, hipMemcpyDeviceToHost);        // first bytes behind it
      (void)hipGetLastError();
      return (in == hipSuccess && out != hipSuccess) ? 1 : 0;
    }
  return -1;
}

int rvc_debug_plan(int n_channels, unsigned flags, size_t head_block, size_t tail_block, size_t longest_ir,
                   size_t *head_run, size_t *tail_run, size_t *zero_latency_samples) {
  if (n_channels < 1 || head_block == 0 || tail_block == 0) return 0;
  if (head_block > tail_block) std::swap(head_block, tail_block);   // TwoStageFFTConvolver.cpp:100-104
  // (child sets plan with the whole set's channel count, rvc_set::plan_nch: this is the plan of a set of n_channels however
  //  many children serve it)
  const Tuning tn = tune_defaults_now();
  const StagePlan p = plan_stages(n_channels, flags, tn.tail_slack, tn.mix64, head_block, tail_block, true, longest_ir);
  if (head_run) *head_run = p.hb;
  if (tail_run) *tail_run = p.tb;
  if (zero_latency_samples) *zero_latency_samples = p.split;
  return p.td;
}

int rvc_set_plan(const rvc_set *s, rvc_plan *out, size_t out_size) {
  // (the struct may grow at its end: a caller compiled against an earlier, shorter one gets the fields it knows)
  if (!s || !out || out_size < offsetof(rvc_plan, head_block)) return 0;
  std::memset(out, 0, out_size);
  const rvc_set *k = s->kids.empty() ? s : s->kids[0];      // (children share one plan: rvc_set::plan_nch, longest_hint)
  rvc_plan p{};
  p.channels = s->nch;
  p.subsets = s->kids.empty() ? 1 : (int)s->kids.size();
  p.initialised = s->inited ? 1 : 0;
  p.two_stage = k->two_stage ? 1 : 0;
  p.tail_on_second_stream = (s->flags & RVC_FLAG_BG_STREAM) != 0;
  p.head_block = k->head;
  p.tail_block = k->tail;
  p.max_len = k->max_len;
  for (const rvc_set *c : (s->kids.empty() ? std::vector<rvc_set *>{const_cast<rvc_set *>(s)} : s->kids)) {
    p.head_partitions = std::max(p.head_partitions, c->A.P);
    p.tail_partitions = std::max(p.tail_partitions, c->T.P);
    p.wide_partitions = std::max(p.wide_partitions, c->W.P);
    if (c->live) k = c;                                     // (a child with empty impulses holds no stages: describe a live one)
  }
  if (k->live) {
    p.live = 1;
    p.zero_latency_samples = k->split;
    p.tail_delay = k->T.P > 0 ? k->T.delay : 0;
    p.head_f64 = (k->A.f64f ? 1 : 0) | (k->A.f64i ? 2 : 0);
    p.tail_f64 = k->T.P > 0 ? ((k->T.f64f ? 1 : 0) | (k->T.f64i ? 2 : 0)) : 0;
    p.head_tile_blocks = k->tA.on ? k->tA.K1 : 0;
    p.tail_tile_blocks = k->tT.on ? k->tT.K1 : 0;
    // (the one-launch block kernel is float only: a zero-latency stage with a transform in double -- heads of 2048 .. 8192 in sets
    //  of more than 8 channels by default, any head with RVC_FLAG_FFT_F64 -- takes the general path too)
    p.block_path = (k->block_general || !rvc::fused_supported(k->A.logB, k->A.f64())) ? 1 : 0;
    p.long_call_block = k->T.PF > 0 ? k->T.B : 0;
    p.wide_block = k->W.P > 0 ? k->W.B : 0;
    p.head_patch_in_launch = k->same_block ? 1 : 0;
    p.tail_spread = k->tT.on ? (k->tT.lag1 | (k->tT.lag2 << 1)) : 0;
    p.tail_sweep_slices = p.tail_spread ? sweep_slices(k) : 1;
    p.tail_phase_groups = k->tT.on ? k->tT.G : 1;
    p.tail_third_level = (k->tT.on && k->tT.s3) ? 1 : 0;
    p.head_third_level = (k->tA.on && k->tA.s3) ? 1 : 0;
    // the reference's structure at these sizes: head + tail0 cover IR[0, 2T) at the head block, the tail runs 2 blocks late
    p.reference_structure = (k->T.P == 0 || (k->T.delay == 2)) ? 1 : 0;
  }
  std::memcpy(out, &p, std::min(out_size, sizeof(p)));
  return 1;
}

int rvc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char *rvc_version(void) { return "reevr_amd 0.2 (gfx950)"; }
int rvc_abi_version(void) { return RVC_ABI_VERSION; }

}  // extern "C"
