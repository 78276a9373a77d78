// rvc_state.cpp -- device state of a set: allocations (with the guard-band and fence development modes), streams and events,
// twiddle tables, IR spectra (one batched double transform per stage), do_init, child sets.
//
// What it replaces in the reference: the buffer set-up of TwoStageFFTConvolver::init / reset (libs/FFTConvolver/
// TwoStageFFTConvolver.cpp:51-148) and FFTConvolver::init / reset (FFTConvolver.cpp:56-152) incl. the per-partition IR transforms
// (:129-137), and the object lifetime of Convolver (src/dsp/Convolver.cpp:56-75).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rvc_set.h"

namespace rvc_eng {

bool fail(rvc_set *s, int code, hipError_t e, const char *what) {
  if (s->err == RVC_OK) {
    s->err = code;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    s->errstr = buf;
  }
  return false;
}

bool use_device(rvc_set *s) {
  // (hipGetDevice is a thread-local read; hipSetDevice takes a runtime lock: skipped when the thread is on the set's device already)
  static const bool always = std::getenv("RVC_SETDEVICE_ALWAYS") != nullptr;     // (measurement: the pre-round-6 behaviour)
  int cur = -1;
  if (!always && hipGetDevice(&cur) == hipSuccess && cur == s->device) return true;
  hipError_t e = hipSetDevice(s->device);
  if (e != hipSuccess) return fail(s, RVC_ERR_NO_DEVICE, e, "hipSetDevice");
  return true;
}

// guard = 2, the "electric fence": [unmapped | mapping, payload END-aligned | unmapped]. An out-of-bounds READ -- also one
// whose value a select then drops, the clamped-loader class of bug -- past the end of an allocation is a GPU memory fault
// (the process aborts: run under tools/fence_fuzz.py, never inside the test-suite). Under-runs land in the 0xFF slack in
// front of the payload (or, beyond it, in the lower unmapped range).
hipError_t fence_alloc(rvc_set *s, void **p, size_t bytes) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = s->device;
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorNotSupported;
  const size_t want = (bytes + 255) & ~(size_t)255;              // (keeps the payload 256-byte aligned)
  const size_t mapped = (want + gran - 1) / gran * gran;
  rvc_set::GuardRec r{};
  r.fenced = true; r.bytes = bytes; r.mapped = mapped; r.va_bytes = mapped + 2 * gran;
  e = hipMemAddressReserve(&r.va, r.va_bytes, gran, nullptr, 0);
  if (e != hipSuccess) return e;
  e = hipMemCreate(&r.handle, mapped, &prop, 0);
  if (e != hipSuccess) { hipMemAddressFree(r.va, r.va_bytes); return e; }
  r.base = (char *)r.va + gran;
  e = hipMemMap(r.base, mapped, 0, r.handle, 0);
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (e == hipSuccess) e = hipMemSetAccess(r.base, mapped, &acc, 1);
  if (e == hipSuccess) e = hipMemsetAsync(r.base, 0xFF, mapped, s->st_main);
  if (e == hipSuccess) e = hipStreamSynchronize(s->st_main);
  if (e != hipSuccess) { hipMemUnmap(r.base, mapped); hipMemRelease(r.handle); hipMemAddressFree(r.va, r.va_bytes); return e; }
  r.payload = r.base + (mapped - want);
  *p = r.payload;
  s->guards.push_back(r);
  return hipSuccess;
}

hipError_t dev_alloc_raw(rvc_set *s, void **p, size_t bytes) {
  if (!s->tune.guard) return hipMalloc(p, bytes);
  if (s->tune.guard == 2 && s->streams_ok) return fence_alloc(s, p, bytes);
  char *base = nullptr;
  hipError_t e = hipMalloc(&base, bytes + 2 * kGuardBytes);
  if (e != hipSuccess) return e;
  // (on the set's own stream and waited for: the set's streams are non-blocking ones, a fill on the null stream could
  //  land AFTER the first writes of the buffer's owner and poison valid data)
  e = s->streams_ok ? hipMemsetAsync(base, 0xFF, bytes + 2 * kGuardBytes, s->st_main) : hipMemset(base, 0xFF, bytes + 2 * kGuardBytes);
  if (e == hipSuccess) e = s->streams_ok ? hipStreamSynchronize(s->st_main) : hipDeviceSynchronize();
  if (e != hipSuccess) { hipFree(base); return e; }
  *p = base + kGuardBytes;
  rvc_set::GuardRec r{};
  r.base = base; r.bytes = bytes; r.payload = base + kGuardBytes;
  s->guards.push_back(r);
  return hipSuccess;
}

void dev_free(rvc_set *s, void *p) {
  if (!p) return;
  for (size_t i = 0; i < s->guards.size(); ++i)
    if (s->guards[i].payload == (char *)p) {
      const rvc_set::GuardRec r = s->guards[i];
      if (r.fenced) { hipMemUnmap(r.base, r.mapped); hipMemRelease(r.handle); hipMemAddressFree(r.va, r.va_bytes); }
      else hipFree(r.base);
      s->guards.erase(s->guards.begin() + (long)i);
      return;
    }
  hipFree(p);
}

bool ensure_streams(rvc_set *s) {
  if (s->streams_ok) return true;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= s->device)
    return fail(s, RVC_ERR_NO_DEVICE, e, "no usable HIP device (this engine has no CPU fallback)");
  if (!use_device(s)) return false;
  RVC_CK(rvc::prepare_kernels());
  // (Measured for the many-channel lock-step loop, profiles/r2_bg_overlap.txt: a high-priority foreground stream and a
  //  background stream confined to 192 / 128 / 64 CUs by a CU mask change the step time by less than 2 % either way.)
  RVC_CK(hipStreamCreateWithFlags(&s->st_main, hipStreamNonBlocking));
  // The tail stream only where it is asked for: the runtime multiplexes streams onto a few hardware queues (4 by default), and
  // two child sets whose foreground streams land on ONE queue serialise (measured: six streams for a two-child set -- the
  // children's launches did not overlap at all, 14.0 instead of 15.5 Gsamples/s). Without the flag st_bg is st_main.
  if ((s->flags & RVC_FLAG_BG_STREAM) != 0) RVC_CK(hipStreamCreateWithFlags(&s->st_bg, hipStreamNonBlocking));
  else s->st_bg = s->st_main;
  RVC_CK(hipEventCreateWithFlags(&s->ev_ingest, hipEventDisableTiming));
  RVC_CK(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
  RVC_CK(hipEventCreateWithFlags(&s->ev_fence, hipEventDisableTiming));
  for (s->ev_free = 0; s->ev_free < rvc_set::kMaxJobs; ++s->ev_free)
    RVC_CK(hipEventCreateWithFlags(&s->ev_pool[s->ev_free], hipEventDisableTiming));
  s->streams_ok = true;
  return true;
}

void free_stage(rvc_set *s, Stage &g) {
  void *all[] = {g.H, g.X, g.Y, g.d_ir, g.tw, g.wsplit, g.twd, g.wsplitd, g.tw8, g.tw8d, g.tw8dh, g.tw8fh, g.twfh};
  for (void *q : all) dev_free(s, q);
  g = Stage();
}

void drop_jobs(rvc_set *s) {   // return the events of all queued tail jobs to the pool
  for (; s->job_count > 0; --s->job_count) {
    s->ev_pool[s->ev_free++] = s->jobs[s->job_head].ev;
    s->job_head = (s->job_head + 1) % rvc_set::kMaxJobs;
  }
  s->job_head = 0;
}

// read the pending event pairs of one kernel family into its running totals and release the events
void fold_timing(rvc_set *s, int id) {
  auto &v = s->timed[id];
  if (v.empty()) return;
  if (hipEventSynchronize(v.back().b) != hipSuccess) (void)hipGetLastError();
  for (auto &t : v) {
    float ms = 0.f;
    // (a pair no launch recorded -- a launcher that found nothing to do -- reads as an error: dropped, and the runtime's sticky
    //  "last error" cleared so that it does not surface in whatever checks hipGetLastError next)
    const hipError_t te = hipEventElapsedTime(&ms, t.a, t.b);
    if (te != hipSuccess) (void)hipGetLastError();
    if (te == hipSuccess) {
      s->timed_ms[id] += ms; ++s->timed_n[id];
      const hipEvent_t base = s->timed_parent ? s->timed_parent->timed_base : s->timed_base;
      float st = 0.f;
      if (base && s->timed_iv[id].size() < ((size_t)1 << 16) && hipEventElapsedTime(&st, base, t.a) == hipSuccess)
        s->timed_iv[id].push_back({(double)st, (double)st + (double)ms});
    }
    hipEventDestroy(t.a); hipEventDestroy(t.b);
  }
  v.clear();
}

void drop_timing(rvc_set *s) {
  for (int id = 0; id < kNumKernelIds; ++id) {
    for (auto &t : s->timed[id]) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    s->timed[id].clear();
    s->timed_ms[id] = 0.0;
    s->timed_n[id] = 0;
    s->timed_iv[id].clear();
  }
}

void free_device_state(rvc_set *s) {
  if (s->streams_ok) {
    hipSetDevice(s->device);
    hipStreamSynchronize(s->st_bg);
    hipStreamSynchronize(s->st_main);
  }
  drop_jobs(s);
  drop_timing(s);
  free_stage(s, s->A);
  free_stage(s, s->T);
  free_stage(s, s->W);
  dev_free(s, s->xring); dev_free(s, s->tailring); dev_free(s, s->d_in); dev_free(s, s->d_out); dev_free(s, s->ypre);
  dev_free(s, s->tA.s1); dev_free(s, s->tA.s2); dev_free(s, s->tA.s3); dev_free(s, s->tT.s1); dev_free(s, s->tT.s2); dev_free(s, s->tT.s3);
  s->ypre = nullptr;
  s->tA = Tile(); s->tT = Tile();
  s->ypre_cur = nullptr;
  s->ypre_block = -1;
  if (s->h_in) hipHostFree(s->h_in);
  if (s->h_out) hipHostFree(s->h_out);
  if (s->h_flags) hipHostFree(s->h_flags);
  s->h_flags = nullptr;
  s->xring = s->tailring = s->d_in = s->d_out = s->h_in = s->h_out = nullptr;
  s->ring_cap = 0;
  s->live = false;
  s->inited = false;
  s->head = s->tail = s->max_len = 0;
  s->n = 0;
  s->tail_fft_done = 0;
  s->tail_out_done = 2;
  s->T.delay = 2;
  s->xa_next = 0;
  s->w_next = 0;
  s->xt_valid_lo = 0;
}

bool make_twiddles(rvc_set *s, Stage &g) {
  const size_t B = g.B;
  const size_t nws = B;   // e^{-i pi k / B}, k < B (the generic kernels use the first B/2+1)
  std::vector<float2> tw(B), ws(nws + 1);
  std::vector<double2> twd(B), wsd(nws + 1);
  for (size_t j = 0; j < B; ++j) {
    const double ang = -2.0 * kPi * (double)j / (double)B;
    twd[j] = make_double2(std::cos(ang), std::sin(ang));
    tw[j] = make_float2((float)twd[j].x, (float)twd[j].y);
  }
  for (size_t k = 0; k <= nws; ++k) {
    const double ang = -kPi * (double)k / (double)B;
    wsd[k] = make_double2(std::cos(ang), std::sin(ang));
    ws[k] = make_float2((float)wsd[k].x, (float)wsd[k].y);
  }
  RVC_CK(dev_alloc(s, &g.tw, sizeof(float2) * B));
  RVC_CK(dev_alloc(s, &g.wsplit, sizeof(float2) * (nws + 1)));
  RVC_CK(hipMemcpy(g.tw, tw.data(), sizeof(float2) * B, hipMemcpyHostToDevice));
  RVC_CK(hipMemcpy(g.wsplit, ws.data(), sizeof(float2) * (nws + 1), hipMemcpyHostToDevice));
  const bool dbl = g.logB <= 13;   // the double transform needs B * 16 bytes of LDS (+pad) <= 136 KiB
  if (dbl) {
    RVC_CK(dev_alloc(s, &g.twd, sizeof(double2) * B));
    RVC_CK(dev_alloc(s, &g.wsplitd, sizeof(double2) * (nws + 1)));
    RVC_CK(hipMemcpy(g.twd, twd.data(), sizeof(double2) * B, hipMemcpyHostToDevice));
    RVC_CK(hipMemcpy(g.wsplitd, wsd.data(), sizeof(double2) * (nws + 1), hipMemcpyHostToDevice));
  }
  // per-pass tables of the radix-8 kernels (layout documented in rvc_internal.h)
  auto pass_tables = [](int logB, std::vector<double2> &t8d) {
    const size_t Bt = (size_t)1 << logB;
    const int N8 = logB / 3;
    for (int j = 1; j < N8; ++j) {            // leg-major [r][k]: coalesced per-leg loads (Plan8::off8)
      const size_t p = (size_t)1 << (3 * j);
      for (int r = 0; r < 8; ++r)
        for (size_t k = 0; k < p; ++k) {
          const double ang = -2.0 * kPi * (double)r * (double)k / (double)(8 * p);
          t8d.push_back(make_double2(std::cos(ang), std::sin(ang)));
        }
    }
    if (logB % 3 == 2) {
      for (int r = 0; r < 4; ++r)
        for (size_t k = 0; k < Bt / 4; ++k) {
          const double ang = -2.0 * kPi * (double)r * (double)k / (double)Bt;
          t8d.push_back(make_double2(std::cos(ang), std::sin(ang)));
        }
    }
  };
  const int n8e = rvc::fft8_table_entries(g.logB);
  if (n8e > 0) {
    std::vector<double2> t8d;
    pass_tables(g.logB, t8d);
    const size_t o = t8d.size();
    if (o != (size_t)n8e) return fail(s, RVC_ERR_HIP, hipSuccess, "twiddle table layout");
    std::vector<float2> t8(o);
    for (size_t i = 0; i < o; ++i) t8[i] = make_float2((float)t8d[i].x, (float)t8d[i].y);
    RVC_CK(dev_alloc(s, &g.tw8, sizeof(float2) * o));
    RVC_CK(hipMemcpy(g.tw8, t8.data(), sizeof(float2) * o, hipMemcpyHostToDevice));
    if (dbl) {
      RVC_CK(dev_alloc(s, &g.tw8d, sizeof(double2) * o));
      RVC_CK(hipMemcpy(g.tw8d, t8d.data(), sizeof(double2) * o, hipMemcpyHostToDevice));
    }
  }
  if (g.logB == 13) {                         // the double inverse as two half-size sub-transforms
    std::vector<double2> th;
    pass_tables(12, th);
    if (th.size() != (size_t)rvc::fft8_table_entries(12)) return fail(s, RVC_ERR_HIP, hipSuccess, "twiddle table layout");
    RVC_CK(dev_alloc(s, &g.tw8dh, sizeof(double2) * th.size()));
    RVC_CK(hipMemcpy(g.tw8dh, th.data(), sizeof(double2) * th.size(), hipMemcpyHostToDevice));
  }
  if (g.logB == 14) {                         // the float inverse as two half-size (8192-point) sub-transforms: pass tables + base twiddles
    std::vector<double2> th;
    pass_tables(13, th);
    if (th.size() != (size_t)rvc::fft8_table_entries(13)) return fail(s, RVC_ERR_HIP, hipSuccess, "twiddle table layout");
    std::vector<float2> tf(th.size()), tb(B / 2);
    for (size_t i = 0; i < th.size(); ++i) tf[i] = make_float2((float)th[i].x, (float)th[i].y);
    for (size_t j = 0; j < B / 2; ++j) {
      const double ang = -2.0 * kPi * (double)j / (double)(B / 2);
      tb[j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    RVC_CK(dev_alloc(s, &g.tw8fh, sizeof(float2) * tf.size()));
    RVC_CK(hipMemcpy(g.tw8fh, tf.data(), sizeof(float2) * tf.size(), hipMemcpyHostToDevice));
    RVC_CK(dev_alloc(s, &g.twfh, sizeof(float2) * tb.size()));
    RVC_CK(hipMemcpy(g.twfh, tb.data(), sizeof(float2) * tb.size(), hipMemcpyHostToDevice));
  }
  return true;
}

// IR partitions -> spectra: one batched forward launch over all partitions of all channels
// (replaces the per-partition loop FFTConvolver.cpp:129-137).
bool upload_ir_stage(rvc_set *s, Stage &g, const float *const *irs, const std::vector<size_t> &counts, bool on_device) {
  // d_ir: [channel][hrows * B] zero-padded samples; irs[c] may be host or (rvc_set_init_impulse) device memory
  const size_t padded = (size_t)g.hrows() * g.B;
  if (!g.d_ir) RVC_CK(dev_alloc(s, &g.d_ir, sizeof(float) * (size_t)s->nch * padded));
  RVC_CK(hipMemsetAsync(g.d_ir, 0, sizeof(float) * (size_t)s->nch * padded, s->st_main));
  for (int c = 0; c < s->nch; ++c)
    if (counts[c])
      RVC_CK(hipMemcpyAsync(g.d_ir + (size_t)c * padded, irs[c], sizeof(float) * counts[c],
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st_main));
  if (!g.H) RVC_CK(dev_alloc(s, &g.H, sizeof(float2) * (size_t)s->nch * g.hrows() * g.B));
  rvc::FwdArgs a{};
  a.src = g.d_ir; a.src_chan_stride = (long long)padded; a.src_mask = ~0ull;
  a.seg0 = 0; a.valid_len = (int)g.B; a.lo = 0; a.hi = (long long)padded;
  // One-off, so always in double where the LDS allows it (B <= 8192): the IR spectra then carry
  // only the float rounding of the stored bins, like the reference's (AudioFFT.cpp:114-137).
  const bool ir64 = g.twd != nullptr;
  a.tw = ir64 ? (const void *)g.twd : (const void *)g.tw;
  a.wsplit = ir64 ? (const void *)g.wsplitd : (const void *)g.wsplit;
  a.tw8 = ir64 ? (const void *)g.tw8d : (const void *)g.tw8;
  a.dst = g.H; a.dst_chan_stride = (long long)g.hrows() * (long long)g.B; a.row0 = 0; a.row_mask = ~0ull;
  hipError_t e = rvc::launch_fft_fwd(g.logB, ir64, a, g.hrows(), s->nch, s->st_main);
  if (e == hipSuccess) e = hipStreamSynchronize(s->st_main);
  if (e != hipSuccess) return fail(s, RVC_ERR_HIP, e, "IR spectra");
  // The time-domain partitions are kept so that an IR swap with unchanged geometry -- the plug-in's hot-swap -- allocates nothing.
  // For sets of thousands of channels they are gigabytes (BASELINE config 3 at 2048 channels: 24 GB beside 126 GB of spectra and
  // delay lines) and an allocation of a few milliseconds at the next swap is nothing beside the upload: given back.
  if (sizeof(float) * (size_t)s->nch * padded >= kKeepIrBytesMax) { dev_free(s, g.d_ir); g.d_ir = nullptr; }
  return true;
}

bool do_init(rvc_set *s, size_t head_block, size_t tail_block, bool two_stage,
             const float *const *irs, const size_t *ir_lens, size_t max_len, bool on_device) {
  // The reference's init() starts with reset() (TwoStageFFTConvolver.cpp:92, FFTConvolver.cpp:95).
  // Here everything is released only when the new geometry differs; an IR swap with unchanged
  // block sizes / partition counts (the plug-in's hot-swap, src/PluginProcessor.cpp:1680-1691)
  // keeps every buffer and only re-uploads and re-transforms the IR.
  auto drop = [&]() { if (s->streams_ok || s->live) free_device_state(s); };
  const TuneScope tune_scope(s);
  s->err = RVC_OK;
  s->errstr.clear();
  if ((s->flags & RVC_FLAG_PERSISTENT) != 0) {   // the resident-kernel mode of rounds 2-3: removed, not silently ignored
    drop();
    s->err = RVC_ERR_UNSUPPORTED;
    s->errstr = "RVC_FLAG_PERSISTENT was removed (round 4): ordinary launches are faster on every metric";
    return false;
  }
  if (head_block == 0 || (two_stage && tail_block == 0)) {   // TwoStageFFTConvolver.cpp:94-97, FFTConvolver.cpp:97-100
    drop();
    s->err = RVC_ERR_BAD_ARG;
    s->errstr = "block size 0";
    return false;
  }
  if (!irs || !ir_lens) { drop(); return fail(s, RVC_ERR_BAD_ARG, hipSuccess, "irs"); }
  if (two_stage && head_block > tail_block) std::swap(head_block, tail_block);   // :100-104

  // trailing |x| < 1e-6 is ignored (TwoStageFFTConvolver.cpp:107-110, FFTConvolver.cpp:102-106)
  std::vector<size_t> len(s->nch);
  size_t longest = 0;
  for (int c = 0; c < s->nch; ++c) {
    size_t l = irs[c] ? ir_lens[c] : 0;
    if (!on_device)   // device-resident IRs arrive with the scan already done (rvc::impulse_view)
      while (l > 0 && std::fabs(irs[c][l - 1]) < 0.000001f) --l;
    len[c] = l;
    longest = std::max(longest, l);
  }
  // Requested partition sizes, rounded up to powers of two like the reference (:117-118). The sizes
  // only set the latency of the partitioned algorithm, never its output, and this engine's latency
  // is set by the call, not by the partition: requests above what one CU's LDS can transform are
  // served with the largest supported partition instead (rvc_set_head_block / _tail_block report
  // what is used). A host running 16384- or 32768-frame blocks gets the same samples.
  const size_t longest_set = std::max(longest, s->longest_hint);
  const StagePlan plan = plan_stages(s->plan_nch ? s->plan_nch : s->nch, s->flags, s->tune.tail_slack, s->tune.mix64, head_block, tail_block, two_stage, longest_set);
  const size_t hb_req = plan.hb_req, hb = plan.hb, split = plan.split;
  const bool want64 = plan.want64;
  auto stage64 = [&](size_t B) { return plan.stage64(B); };
  size_t tb = plan.tb;
  const int td = plan.td;
  if (longest == 0) {   // empty IR: success, process() gives zeros (:112-115)
    drop();
    // (a child set whose channels all carry empty impulses still needs its stream: the parent fences the other children against
    //  child 0's stream and hands it out as the set's ordering stream; its zeros are an asynchronous fill on that stream)
    if (s->is_kid && !ensure_streams(s)) return false;
    s->inited = true;
    s->head = hb; s->tail = tb; s->max_len = max_len ? max_len : hb_req;
    return true;
  }
  const size_t eff_max_len = max_len ? max_len : hb_req;
  const bool no_resize = (s->flags & (RVC_FLAG_FIXED_PARTITIONS | RVC_FLAG_BG_STREAM)) != 0;

  // partition counts (ceil(float/float) as FFTConvolver.cpp:115; exact below 2^24 samples)
  std::vector<size_t> lenA(s->nch);   // samples the zero-latency stage covers; the tail and wide
  size_t pa = 0, pt = 0;              // stages keep the WHOLE IR at their block size (see Stage::PF)
  for (int c = 0; c < s->nch; ++c) {
    const size_t la = std::min(len[c], split);
    lenA[c] = la;
    pa = std::max(pa, (la + hb - 1) / hb);
    if (len[c] > split) pt = std::max(pt, (len[c] - split + tb - 1) / tb);
  }
  // Long-call stage of a single-stage (FFTConvolver) set: the whole IR once more at block 8192, used
  // only by calls that touch several such blocks (the adaptive path of step_device). It has no
  // streaming role -- P stays 0: no tail jobs, no tail ring -- its delay line is rebuilt from the
  // time ring whenever a long call needs it.
  const size_t lb = 8192;
  const bool uni_long = !two_stage && !no_resize && hb < lb && longest > 2 * lb && eff_max_len >= 4 * lb;
  if (uni_long) tb = lb;                         // (tb is 0 for single-stage sets otherwise)
  const size_t pf = (pt > 0 || uni_long) ? (longest + tb - 1) / tb : 0;   // rows of the whole-IR table at block tb
  // wide stage: only for float transforms (136 KiB of LDS), a tail block below 16384 and an IR of
  // several wide blocks; and only if calls can be long enough to use it
  const size_t wb = (size_t)RVC_MAX_BLOCK;
  const bool wide = pf > 0 && !want64 && tb < wb && longest > 4 * wb && eff_max_len >= 4 * wb && !no_resize;
  const size_t pw = wide ? (longest + wb - 1) / wb : 0;

  // ---- IR swap with unchanged geometry: keep all device state, refresh the spectra ----
  const size_t tail_pub = two_stage ? tb : 0;    // what rvc_set_tail_block reports: 0 for single-stage sets
  if (s->live && s->two_stage == two_stage && s->head == hb && s->tail == tail_pub && s->max_len == eff_max_len &&
      s->A.P == (int)pa && s->T.P == (int)pt && s->T.PF == (int)pf && s->W.P == (int)pw && (pf == 0 || s->T.delay == td)) {
    if (!use_device(s)) return false;
    hipStreamSynchronize(s->st_bg);
    hipStreamSynchronize(s->st_main);
    drop_jobs(s);
    if (!upload_ir_stage(s, s->A, irs, lenA, on_device)) { free_device_state(s); return false; }
    if (pf > 0 && !upload_ir_stage(s, s->T, irs, len, on_device)) { free_device_state(s); return false; }
    if (pw > 0 && !upload_ir_stage(s, s->W, irs, len, on_device)) { free_device_state(s); return false; }
    s->n = 0; s->tail_fft_done = 0; s->tail_out_done = td; s->xa_next = 0; s->ypre_block = -1;
    s->w_next = 0; s->xt_valid_lo = 0; s->tA.restart(); s->tT.restart();
    return true;
  }

  drop();
  if (!ensure_streams(s)) return false;
  if (!use_device(s)) return false;

  s->head = hb;
  s->tail = tail_pub;
  s->two_stage = two_stage;
  s->split = two_stage ? split : 0;
  s->max_len = eff_max_len;
  Stage &A = s->A, &T = s->T;
  A.B = hb; A.logB = ilog2(hb); A.P = (int)pa; A.delay = 0; A.set64(stage64(hb));
  A.mcap = s->max_len / hb + 2;
  A.rows = next_pow2(pa + A.mcap + 1);
  if (!make_twiddles(s, A)) return false;
  if (!upload_ir_stage(s, A, irs, lenA, on_device)) return false;
  RVC_CK(dev_alloc(s, &A.X, sizeof(float2) * (size_t)s->nch * A.rows * A.B));
  RVC_CK(dev_alloc(s, &A.Y, sizeof(float2) * (size_t)s->nch * A.mcap * A.B));
  if (pf > 0) {
    T.B = tb; T.logB = ilog2(tb); T.P = (int)pt; T.PF = (int)pf; T.delay = td; T.set64(stage64(tb));
    T.mcap = s->max_len / tb + 3;
    T.rows = next_pow2(pf + T.mcap + 2);
    if (!make_twiddles(s, T)) return false;
    if (!upload_ir_stage(s, T, irs, len, on_device)) return false;
    RVC_CK(dev_alloc(s, &T.X, sizeof(float2) * (size_t)s->nch * T.rows * T.B));
    RVC_CK(dev_alloc(s, &T.Y, sizeof(float2) * (size_t)s->nch * T.mcap * T.B));
  }
  Stage &W = s->W;
  if (pw > 0) {
    W.B = wb; W.logB = ilog2(wb); W.P = (int)pw; W.delay = 0; W.set64(0);
    W.mcap = s->max_len / wb + 3;
    W.rows = next_pow2(pw + W.mcap + 2);
    if (!make_twiddles(s, W)) return false;
    if (!upload_ir_stage(s, W, irs, len, on_device)) return false;
    RVC_CK(dev_alloc(s, &W.X, sizeof(float2) * (size_t)s->nch * W.rows * W.B));
    RVC_CK(dev_alloc(s, &W.Y, sizeof(float2) * (size_t)s->nch * W.mcap * W.B));
  }
  // input history a long call must leave behind: 2 tail blocks for the tail transforms, P+2 head
  // blocks for a rebuild of the head delay line; with a wide stage also a whole wide / tail delay
  // line of history (their rows are rebuilt from the ring when the call pattern changes)
  const size_t span = std::max(hb, tb);
  s->keep = 2 * (long long)span + ((long long)pa + 2) * (long long)hb;
  if (pw > 0) s->keep = std::max<long long>(s->keep, (long long)(pw + 2) * (long long)wb);
  if (pf > 0 && (pw > 0 || uni_long)) s->keep = std::max<long long>(s->keep, (long long)(pf + 4) * (long long)tb);
  s->ring_cap = next_pow2(s->max_len + (size_t)s->keep + 6 * std::max(span, pw > 0 ? wb : (size_t)0) + 4 * hb);
  RVC_CK(dev_alloc(s, &s->xring, sizeof(float) * (size_t)s->nch * s->ring_cap));
  if (pt > 0) RVC_CK(dev_alloc(s, &s->tailring, sizeof(float) * (size_t)s->nch * s->ring_cap));
  RVC_CK(dev_alloc(s, &s->ypre, sizeof(float2) * 2 * (size_t)s->nch * A.B));
  RVC_CK(hipMemsetAsync(s->ypre, 0, sizeof(float2) * 2 * (size_t)s->nch * A.B, s->st_main));
  s->ypre_block = -1;
  s->fold = rvc::fused_fold_supported(A.logB) && !A.f64();
  s->block_general = A.logB >= 11 && (size_t)s->nch * A.B >= ((size_t)1 << 20);   // (measured: BASELINE config 5's geometry, 4096 channels)
  // time tiling: where a per-block sweep is long enough to be bandwidth- rather than latency-bound
  {
    const bool tiling = (s->flags & RVC_FLAG_NO_TIME_TILING) == 0;
    const bool force2 = tiling && (s->flags & RVC_FLAG_FORCE_TWO_LEVEL) != 0;   // tests: two levels whatever the size
    const bool force = force2 || (tiling && (s->flags & RVC_FLAG_FORCE_TIME_TILING) != 0);   // tests: tile whatever the size
    const size_t K = (size_t)rvc::kSweepRows;
    Tile &tA = s->tA, &tT = s->tT;
    tA = Tile(); tT = Tile();
    tA.on = tiling && s->fold && !s->block_general && A.B >= 64 &&
            (force ? pa >= 3 : (pa >= 8 && (size_t)s->nch * (pa - 2) * A.B * 16 >= ((size_t)4 << 20)));
    // round 6: also where the per-block call takes the GENERAL path (a transform in double -- RVC_FLAG_FFT_F64, heads of 2048 .. 8192
    // in larger sets --, many channels with a large head block): the delay-line launch between the two transform launches becomes
    // sweep(s) + patch over the same tile state (head_stage; Stage A with delay 0: the block's own row exists when its sweep runs)
    const bool general = s->block_general || !rvc::fused_supported(A.logB, A.f64());
    s->head_gen = tiling && general && A.B >= 64 && (force ? pa >= 3 : (pa >= 8 && (size_t)s->nch * pa * A.B * 16 >= ((size_t)4 << 20)));
    if (s->head_gen) tA.on = true;
    // (pa >= 8: a shrunk zero-latency stage of 8 partitions -- head 1024 under a tail of 8192 -- measured 9.1 Gsamples/s untiled
    //  against 14.9 tiled at 2048 channels, profiles/r4_tail_slack.txt; it was 16 while every many-channel stage had >= 16)
    tT.on = tiling && tb >= 64 && (force ? pt >= 1 : pt >= 16);
    // one level of 8 blocks, or -- long delay lines -- a first level of 16 / 32 blocks with second-level sweeps every 8
    auto first_level = [&](size_t P) -> int {
      // knob k1: 0 = by length (16 above kTwoLevelMinP partitions, 32 from kLongLineMinP on: measured, profiles/r3_tuning.txt, r5_k1.txt)
      const int tk1 = s->tune.k1;
      int k1 = (tk1 == 32 || tk1 == 16 || tk1 == 8) ? tk1 : ((int)P >= rvc::kLongLineMinP ? 32 : 16);
      if (force2 && k1 == 8) k1 = 16;
      const int minp = s->tune.two_min_p >= 0 ? s->tune.two_min_p : rvc::kTwoLevelMinP;
      return (force2 || (int)P > minp) ? k1 : (int)K;
    };
    tA.K1 = first_level(pa);
    tA.rows1 = tA.K1;
    tT.K1 = first_level(pt);
    tT.rows1 = tT.K1;
    if (tA.on) {
      RVC_CK(dev_alloc(s, &tA.s1, sizeof(float2) * (size_t)s->nch * (size_t)tA.rows1 * A.B));
      if (tA.K1 > (int)K) RVC_CK(dev_alloc(s, &tA.s2, sizeof(float2) * (size_t)s->nch * K * A.B));
    }
    if (tT.on) {
      RVC_CK(dev_alloc(s, &tT.s1, sizeof(float2) * (size_t)s->nch * (size_t)tT.rows1 * T.B));
      if (tT.K1 > (int)K) RVC_CK(dev_alloc(s, &tT.s2, sizeof(float2) * (size_t)s->nch * K * T.B));
    }
  }
  // Spread tail sweeps (rvc_schedule.cpp "uniform call cost"): sets whose tail job runs on their own stream; knob "tail_spread"
  {
    Tile &tT = s->tT;
    const int nch_all = s->plan_nch ? s->plan_nch : s->nch;
    const bool can = tT.on && (s->flags & RVC_FLAG_BG_STREAM) == 0;
    const int spread = s->tune.tail_spread >= 0 ? s->tune.tail_spread : (nch_all >= kSpreadMinChannels ? kSpreadDefault : 0);
    tT.lag1 = (can && (spread & 1)) ? 1 : 0;
    tT.lag2 = (can && (spread & 2) && tT.K1 > rvc::kSweepRows) ? 1 : 0;
    const bool stagger = s->tune.kid_stagger >= 0 ? s->tune.kid_stagger != 0 : kKidStaggerDefault;
    tT.first_len = (can && stagger && s->kid_count > 1) ? std::max(1, tT.K1 - s->kid_index * rvc::kSweepRows / s->kid_count) : 0;
    // phase groups: G contiguous channel ranges; group p's first tile is phi[p] = p + 8 (p mod K1/8) blocks short (distinct modulo
    // 8: one group per patch depth; spread over the K1 blocks of a tile: one first-level sweep every K1 / G periods)
    int G = s->tune.tail_phases >= 0 ? s->tune.tail_phases : (nch_all >= kSpreadMinChannels ? kPhasesDefault : 1);
    G = std::max(1, std::min({G, Tile::kMaxPhases, s->nch}));
    if (!tT.on) G = 1;
    if (G > 1) {
      // with phase groups only the FIRST-level sweeps are ever spread (a group's second-level sweep is an eighth of a small launch):
      // by knob, or by default where a tail period is many calls long
      const long long cpp = A.B ? (long long)tb / (long long)A.B : 1;
      tT.lag2 = 0;
      if (s->tune.tail_spread < 0) tT.lag1 = (can && cpp >= kSpreadPhasesMinCalls) ? 1 : 0;
    }
    tT.G = G;
    // third-level sweeps (Tile::s3): many-channel sets -- where a patch's rows are bandwidth, not latency; knob "tail_third"
    const bool third = tT.on && (s->tune.tail_third >= 0 ? s->tune.tail_third != 0 : (nch_all >= kSpreadMinChannels && kThirdDefault));
    if (third) RVC_CK(dev_alloc(s, &tT.s3, sizeof(float2) * (size_t)s->nch * (size_t)rvc::kThirdRows * T.B));
    for (int p = 0; p < G; ++p) {
      Tile::Phase &q = tT.ph[p];
      q = Tile::Phase();
      q.c0 = (int)((long long)s->nch * p / G);
      q.n = (int)((long long)s->nch * (p + 1) / G) - q.c0;
      // (+ 2 per child set: the children's first-level sweeps -- one group's every K1 / G periods -- fall into different periods)
      q.phi = G > 1 ? (p * rvc::kSweepRows / G + rvc::kSweepRows * (p % std::max(1, tT.K1 / rvc::kSweepRows)) + 2 * s->kid_index) % tT.K1 : 0;
    }
  }
  s->same_block = s->tA.on && !s->head_gen && rvc::fused_same_block(A.logB) && s->tune.same_block != 0;
  {
    // third-level sweeps of the zero-latency stage (Tile::s3; same-block sets only: their sweeps run when the newest row exists)
    const int nch_all = s->plan_nch ? s->plan_nch : s->nch;
    const bool third = (s->same_block || s->head_gen) && (s->tune.head_third >= 0 ? s->tune.head_third != 0
                                                                 : (kThirdDefault && (size_t)nch_all * A.B * sizeof(float2) >= kHeadThirdMinRowBytes));
    if (third) RVC_CK(dev_alloc(s, &s->tA.s3, sizeof(float2) * (size_t)s->nch * (size_t)rvc::kThirdRows * A.B));
  }
  RVC_CK(dev_alloc(s, &s->d_in, sizeof(float) * (size_t)s->nch * s->max_len));
  RVC_CK(dev_alloc(s, &s->d_out, sizeof(float) * (size_t)s->nch * s->max_len));
  RVC_CK(hipHostMalloc(&s->h_in, sizeof(float) * (size_t)s->nch * s->max_len, hipHostMallocDefault));
  RVC_CK(hipHostMalloc(&s->h_out, sizeof(float) * (size_t)s->nch * s->max_len, hipHostMallocDefault));
  RVC_CK(hipHostMalloc(&s->h_flags, sizeof(unsigned) * (size_t)s->nch, hipHostMallocDefault));   // (>= audio workgroups)
  std::memset(s->h_flags, 0, sizeof(unsigned) * (size_t)s->nch);
  s->flag_seq = 0; s->flag_count = 0;
  RVC_CK(hipStreamSynchronize(s->st_main));
  RVC_CK(hipStreamSynchronize(s->st_bg));
  s->n = 0;
  s->tail_fft_done = 0;
  s->tail_out_done = td;
  T.delay = td;                  // (also when the stage is absent: clear() restarts the tail clock from it)
  s->xa_next = 0;
  s->w_next = 0;
  s->xt_valid_lo = 0;
  s->live = true;
  s->inited = true;
  return true;
}

void drop_kids(rvc_set *s) {
  for (rvc_set *k : s->kids) rvc_set_destroy(k);
  s->kids.clear();
  s->kid_c0.clear();
}
// the set's streams and pre-created events (a set that gets children gives its own up: the runtime maps streams onto a few
// hardware queues, and an idle pair would still take two of them away from the children)
void drop_streams(rvc_set *s) {
  if (!s->streams_ok) return;
  hipSetDevice(s->device);
  for (int i = 0; i < s->ev_free; ++i) hipEventDestroy(s->ev_pool[i]);
  s->ev_free = 0;
  hipEventDestroy(s->ev_ingest);
  hipEventDestroy(s->ev_out);
  hipEventDestroy(s->ev_fence);
  if (s->st_bg != s->st_main) hipStreamDestroy(s->st_bg);
  hipStreamDestroy(s->st_main);
  s->st_main = s->st_bg = nullptr;
  s->ev_ingest = s->ev_out = s->ev_fence = nullptr;
  s->streams_ok = false;
}
// (re)build the children for this init; false: the set stays childless
bool make_kids(rvc_set *s, int n) {
  if (n <= 1) { drop_kids(s); return false; }
  if ((int)s->kids.size() == n) return true;
  drop_kids(s);
  if (s->streams_ok || s->live) free_device_state(s);
  drop_streams(s);
  const int per = s->nch / n, rem = s->nch % n;   // the first `rem` children serve one channel more
  int c0 = 0;
  for (int k = 0; k < n; ++k) {
    const int mine = per + (k < rem ? 1 : 0);
    rvc_set *c = rvc_set_create(mine, s->device, s->flags | RVC_FLAG_NO_SUBSETS);
    if (!c) { drop_kids(s); return false; }
    c->timing = s->timing;
    c->is_kid = true;
    c->tune = s->tune;
    c->plan_nch = s->nch;
    c->kid_index = k; c->kid_count = n;
    s->kids.push_back(c);
    s->kid_c0.push_back(c0);
    s->stage_sets.reserve((size_t)n); s->stage_c0.reserve((size_t)n);      // (no allocation on the process path)
    c0 += mine;
  }
  return true;
}

// the parent mirrors what its accessors report
void adopt_kid_geometry(rvc_set *s, bool ok) {
  const rvc_set *k = s->kids[0];
  s->head = k->head; s->tail = k->tail; s->max_len = k->max_len; s->two_stage = k->two_stage;
  s->inited = ok; s->live = false;
  s->err = RVC_OK; s->errstr.clear();
  for (const rvc_set *c : s->kids)
    if (c->err != RVC_OK && s->err == RVC_OK) { s->err = c->err; s->errstr = c->errstr; }
}

// A failed init must not keep the stages it had already allocated (the sticky error stays readable)
void release_after_failed_init(rvc_set *s) {
  if (!s->streams_ok && !s->live) return;
  const int err = s->err;
  const std::string msg = s->errstr;
  free_device_state(s);
  s->err = err;
  s->errstr = msg;
}

}  // namespace rvc_eng
