// rvc_schedule.cpp -- the absolute-time block scheduler of the MI355X partitioned-convolution engine: tiles, sweeps, patches,
// tail jobs, the per-block latency path and the general path of step_device; fences between a set and its child sets.
//
// What it replaces in the reference (paths relative to the reference tree):
//   TwoStageFFTConvolver::{init,process,clear,reset}   libs/FFTConvolver/TwoStageFFTConvolver.cpp:51-233
//   FFTConvolver::{init,process,clear,reset}           libs/FFTConvolver/FFTConvolver.cpp:56-212
//   Convolver's background thread + WaitableEvent      src/dsp/Convolver.cpp:21-95  -> second HIP stream + events
//   StereoConvolver's 2-4 way fan-out                  src/dsp/StereoConvolver.cpp:22-62 -> channels of one set
//
// Scheme. The reference computes y = x * ir with three overlap-add sub-convolvers: head
// (IR[0,T), block h), tail0 (IR[T,2T), block h, result delivered T samples later) and tail
// (IR[2T,..), block T, result delivered 2T later). Here the same sum is organised as two
// overlap-save stages driven by ABSOLUTE sample time n (samples since clear()):
//   stage A (zero latency): block h, partitions of IR[0,2T)  -- head and tail0 share their input
//            spectra, so they are one delay line of up to 2T/h partitions, one FFT, one IFFT;
//   stage T (tail):         block T, partitions of IR[2T,..), Y_m = sum_i H_i X_{m-2-i}: the
//            contribution to output block m needs input blocks <= m-2 only, so it is computed one
//            whole tail period ahead (exactly the slack the reference gives its background
//            thread) into a time-indexed ring that stage A's epilogue adds.
//   Lock-step sets of many channels that run the tail job on their own stream do not need the second block of that slack:
//   their stage T runs ONE block late (Stage::delay = 1) and the freed period buys a tail at block 2T (long tails) or a
//   stage A that only covers IR[0,T) (do_init, "What the tail's period of slack is spent on").
// Every buffer is a ring indexed by absolute sample / block number, so a process() call of ANY
// length (one 512-sample block, a ragged 37 samples, or 40 s at once) is the same four steps:
// ingest -> [tail: FFT new blocks, FIR, IFFT -> tail ring] -> stage A: FFT, FIR, IFFT(+tail) -> out.
// A block that a call leaves partly filled is simply transformed again (zero-padded) by the
// next call, like FFTConvolver.cpp:164-173. clear() just restarts the clock.
//
// Block-synchronous calls (one call per host block, the plug-in's pattern) have two refinements on top of that:
//   * causal time tiling of the delay lines (Tile tA / tT, rvc_internal.h): every 8th block a sweep reads a stage's IR
//     spectra and delay line once and leaves partial sums for 8 blocks, the blocks in between patch in the few
//     partitions whose input arrived since; long delay lines get two levels of it (first-level tiles of 16 / 32 blocks);
//   (the resident-kernel mode of rounds 2-3, RVC_FLAG_PERSISTENT, was removed in round 4: it lost to this launch path on
//   latency, p99 and throughput; the flag is rejected at create)
#include <cstring>

#include "rvc_set.h"

namespace rvc_eng {

// Queue a tail job's completion event. The queue holds at most kMaxJobs entries; a caller that never
// reads the tail blocks it produced (kMaxJobs tail periods without a wait) makes the oldest job's event
// be waited for here, which frees its slot.
bool push_job(rvc_set *s, long long m_lo, long long m_hi, hipStream_t st) {
  if (s->job_count == rvc_set::kMaxJobs) {
    rvc_set::Job &o = s->jobs[s->job_head];
    RVC_CK(hipStreamWaitEvent(s->st_main, o.ev, 0));
    s->ev_pool[s->ev_free++] = o.ev;
    s->job_head = (s->job_head + 1) % rvc_set::kMaxJobs;
    --s->job_count;
  }
  rvc_set::Job j{m_lo, m_hi, s->ev_pool[--s->ev_free]};
  RVC_CK(hipEventRecord(j.ev, st));
  s->jobs[(s->job_head + s->job_count) % rvc_set::kMaxJobs] = j;
  ++s->job_count;
  return true;
}

struct Timer {   // brackets one launch with events when timing is on
  rvc_set *s; int id; hipStream_t st; TimedLaunch t{}; bool on;
  Timer(rvc_set *s_, int id_, hipStream_t st_) : s(s_), id(id_), st(st_), on(s_->timing) {
    if (on) { hipEventCreate(&t.a); hipEventCreate(&t.b); rvc::set_launch_events(t.a, t.b); }
  }
  ~Timer() {
    if (!on) return;
    rvc::set_launch_events(nullptr, nullptr);
    s->timed[id].push_back(t);
    if (s->timed[id].size() >= 1024) fold_timing(s, id);   // bounded: streaming use with RVC_FLAG_TIMING does not grow
  }
};

// ---- causal time tiling: sweep launches shared by both stages -----------------------------------
// The delay line of a stage as a sweep sees it: the tail stage's partitions d.. with delay d (2, or 1 for the delay-1 forms);
// the zero-latency stage's partitions 2.. with delay 2 -- its two newest partitions belong to the per-block launch, so a
// sweep would only fetch their IR rows to multiply them with rows that have not arrived (2 of 38 rows of config 2's sweep).
// stage_lag: how far behind the block being prepared the newest input row lies that a sweep may use -- the zero-latency
// stage's two newest partitions belong to the per-block launch, the tail's newest row is `delay` blocks back.
// (Round 6: a zero-latency stage whose per-block launch patches its OWN block -- same_block -- runs every sweep for blocks kb..
//  when row kb - 1 exists: behind the launch that completed block kb - 1, or in front of block kb's. Its sweeps take that row too --
//  it meets block kb + 1 in partition 2 -- and every patch is one partition shorter: lag 1.)
// (head_gen -- the general per-block path, time-tiled: its sweeps and patches cover every partition from 0 on, and a block's sweep runs
//  behind that block's own forward transform: lag 0.)
int stage_lag(const rvc_set *s, bool tail) { return tail ? s->T.delay : (s->head_gen ? 0 : (s->same_block ? 1 : 2)); }
rvc::FirArgs stage_line(rvc_set *s, bool tail) {
  Stage &g = tail ? s->T : s->A;
  const long long B = (long long)g.B;
  rvc::FirArgs r{};
  if (tail) { r.H = g.H + (long long)g.delay * B; r.h_chan_stride = (long long)g.PF * B; r.delay = g.delay; r.tag = 1; }
  else if (s->head_gen) { r.H = g.H; r.h_chan_stride = (long long)g.P * B; r.delay = 0; r.tag = 0; }
  else { r.H = g.H + 2 * B; r.h_chan_stride = (long long)g.P * B; r.delay = 2; r.tag = 0; }
  r.X = g.X; r.x_chan_stride = (long long)g.rows * B; r.x_row_mask = g.rows - 1;
  r.P = (tail || s->head_gen) ? g.P : std::max(g.P - 2, 0); r.B = (int)B;
  return r;
}
// first level: blocks [k0, k0 + K1), every partition, the input rows <= x_hi
rvc::FirArgs sweep1_args(rvc_set *s, bool tail, long long k0, long long x_hi) {
  const Tile &t = tail ? s->tT : s->tA;
  rvc::FirArgs r = stage_line(s, tail);
  r.Y = t.s1; r.y_chan_stride = (long long)t.rows1 * r.B; r.y_row_mask = (unsigned)(t.rows1 - 1);
  r.k0 = k0; r.M = t.K1; r.x_hi = x_hi;
  return r;
}
// second level: blocks [g0, g0 + 8) of the tile that started at t0: first-level rows + the input rows the first-level sweep left
// out, t0 - L - lag1 + 1 .. g0 - L - lag2 (L = stage_lag; lag1 / lag2: Tile, spread sweeps leave the newest row to the patches)
rvc::FirArgs sweep2_args(rvc_set *s, bool tail, long long g0) {
  const Tile &t = tail ? s->tT : s->tA;
  const long long K = rvc::kSweepRows;
  rvc::FirArgs r = stage_line(s, tail);
  r.Y = t.s2; r.y_chan_stride = K * r.B; r.y_row_mask = (unsigned)(K - 1);
  r.Ybase = t.s1; r.ybase_chan_stride = (long long)t.rows1 * r.B; r.ybase_row_mask = (unsigned)(t.rows1 - 1);
  const long long L = stage_lag(s, tail);
  r.k0 = g0; r.M = (int)K; r.x_from = t.t0 - L - t.lag1 + 1; r.x_hi = g0 - L - t.lag2;
  // the oldest row that counts (x_from) meets block g0 + 7 in partition g0 + 7 - delay - x_from
  r.P = (int)std::min<long long>(r.P, g0 - t.t0 + K - 1 + L + t.lag1 - r.delay);
  return r;
}
// third level: blocks [h, h + 4), h = g0 + 4 half way through the group that starts at g0: the group's rows (first-level ones for the
// tile's first group, else second-level ones) + the input rows that group's sweep left out, up to h - L
rvc::FirArgs sweep3_args(rvc_set *s, bool tail, long long g0, long long h) {
  const Tile &t = tail ? s->tT : s->tA;
  const long long K = rvc::kSweepRows, K3 = rvc::kThirdRows;
  rvc::FirArgs r = stage_line(s, tail);
  r.Y = t.s3; r.y_chan_stride = K3 * r.B; r.y_row_mask = (unsigned)(K3 - 1);
  const bool first = g0 == t.t0 || t.K1 <= K;
  if (first) { r.Ybase = t.s1; r.ybase_chan_stride = (long long)t.rows1 * r.B; r.ybase_row_mask = (unsigned)(t.rows1 - 1); }
  else { r.Ybase = t.s2; r.ybase_chan_stride = K * r.B; r.ybase_row_mask = (unsigned)(K - 1); }
  const long long L = stage_lag(s, tail), lagp = first ? t.lag1 : t.lag2;
  r.k0 = h; r.M = (int)K3; r.x_from = g0 - L - lagp + 1; r.x_hi = h - L;
  // the oldest row that counts (x_from) meets block h + 3 in partition h + 3 - delay - x_from
  r.P = (int)std::min<long long>(r.P, h - g0 + K3 - 1 + L + lagp - r.delay);
  return r;
}
// where the partial sums of block b live -- b inside the current first-level tile, and past its first group only once
// that group's second-level sweep has run -- and the per-channel stride of those rows
const float2 *tile_row(const rvc_set *s, bool tail, long long b, long long *stride) {
  const Tile &t = tail ? s->tT : s->tA;
  const size_t B = tail ? s->T.B : s->A.B;
  if (t.third(b)) {
    *stride = (long long)rvc::kThirdRows * (long long)B;
    return t.s3 + (size_t)((unsigned long long)b & (unsigned long long)(rvc::kThirdRows - 1)) * B;
  }
  if (t.K1 > rvc::kSweepRows && t.group(b) != t.t0) {
    *stride = (long long)rvc::kSweepRows * (long long)B;
    return t.s2 + (size_t)((unsigned long long)b & (unsigned long long)(rvc::kSweepRows - 1)) * B;
  }
  *stride = (long long)t.rows1 * (long long)B;
  return t.s1 + (size_t)((unsigned long long)b & (unsigned long long)(t.rows1 - 1)) * B;
}

// ---- tail stage pieces -------------------------------------------------------------------
// Spectra of the tail blocks a call ending at n1 completed. `src2` = the call's own input when
// the ring does not hold it yet (long single-stream calls), else nullptr.
bool tail_spectra(rvc_set *s, long long n0, long long n1, const float *src2, size_t in_stride, hipStream_t st) {
  Stage &T = s->T;
  const long long tb = (long long)T.B;
  const long long mb0 = s->tail_fft_done, mb1 = n1 / tb;   // tail blocks [mb0, mb1) completed by this call
  if (mb1 <= mb0) return true;
  rvc::FwdArgs f{};
  f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
  f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = n0;
  f.seg0 = (mb0 - 1) * tb; f.valid_len = (int)(2 * tb); f.lo = 0; f.hi = n1;
  f.tw = T.twp(T.f64f); f.wsplit = T.wsp(T.f64f); f.tw8 = T.t8p(T.f64f); f.tw8_half = T.t8h(T.f64f); f.tw_half = T.twh(T.f64f);
  f.dst = T.X; f.dst_chan_stride = (long long)T.rows * tb; f.row0 = mb0; f.row_mask = T.rows - 1;
  Timer t(s, 4, st);
  RVC_CK(rvc::launch_fft_fwd(T.logB, T.f64f, f, (int)(mb1 - mb0), s->nch, st));
  s->tail_fft_done = mb1;
  return true;
}

// A wide call skips the tail transforms, so the tail delay line may have a hole below
// xt_valid_lo. Rebuild rows [lo, xt_valid_lo) from the time ring (complete blocks; the ring keeps
// a whole delay line of history when a wide stage exists).
bool ensure_tail_spectra(rvc_set *s, long long lo, hipStream_t st) {
  Stage &T = s->T;
  if (lo < 0) lo = 0;
  if (lo >= s->xt_valid_lo) return true;
  const long long tb = (long long)T.B;
  rvc::FwdArgs f{};
  f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
  f.seg0 = (lo - 1) * tb; f.valid_len = (int)(2 * tb); f.lo = 0; f.hi = s->xt_valid_lo * tb;
  f.tw = T.twp(T.f64f); f.wsplit = T.wsp(T.f64f); f.tw8 = T.t8p(T.f64f); f.tw8_half = T.t8h(T.f64f); f.tw_half = T.twh(T.f64f);
  f.dst = T.X; f.dst_chan_stride = (long long)T.rows * tb; f.row0 = lo; f.row_mask = T.rows - 1;
  Timer t(s, 4, st);
  RVC_CK(rvc::launch_fft_fwd(T.logB, T.f64f, f, (int)(s->xt_valid_lo - lo), s->nch, st));
  s->xt_valid_lo = lo;
  return true;
}

// ---- uniform call cost: spread sweeps -----------------------------------------------------------------------------------------
// The reference moves the tail convolution to a background thread so that every process() call costs the same (src/dsp/
// Convolver.cpp:84-95, TwoStageFFTConvolver.cpp:213-222). A time-tiled lock-step set without a second stream runs its tail job
// inside the call that completes a tail block -- and with it, every K1-th / 8-th tail block, a first- / second-level sweep over the
// WHOLE delay line: milliseconds in one call where the others take tens of microseconds. With Tile::lag1 / lag2 = 1 a sweep leaves
// out the newest row that exists when its first block is due (the patches add that row's partition, one more each), so everything
// it reads exists ONE TAIL PERIOD earlier: it is planned right behind the tail job of the block before (plan_next_tail_sweep) and
// issued in channel slices behind the per-block launches of the period in between (issue_tail_slices, one slice per call, spaced
// over the period); whatever is left when its first block is due is issued then (flush_tail_sweep: ragged call patterns).
bool launch_sweep_slice(rvc_set *s, Tile::Pending &q, int count, hipStream_t st) {
  while (q.on && count-- > 0) {
    const int c0 = q.base + q.next, n = std::min(q.per, q.span - q.next);
    rvc::FirArgs a = q.a;
    a.H += (long long)c0 * a.h_chan_stride; a.X += (long long)c0 * a.x_chan_stride; a.Y += (long long)c0 * a.y_chan_stride;
    if (a.Ybase) a.Ybase += (long long)c0 * a.ybase_chan_stride;
    a.stage_channels = s->nch;
    {
      Timer tm(s, q.timer_id, st);
      RVC_CK(rvc::launch_fdl_sweep(a, n, st));
    }
    q.next += n; ++q.issued;
    if (q.next >= q.span) q.on = false;
  }
  return true;
}
bool flush_tail_sweep(rvc_set *s, hipStream_t st) { return launch_sweep_slice(s, s->tT.pend, 1 << 30, st); }
// one per-block call has been enqueued: the slices that are due by now (evenly over the calls of a tail period but the last)
static bool issue_due(rvc_set *s, Tile::Pending &q) {
  if (!q.on) return true;
  ++q.calls;
  const long long due = ((long long)q.calls * q.slices + q.budget - 1) / q.budget;
  return launch_sweep_slice(s, q, (int)std::min<long long>(due, q.slices) - q.issued, s->st_main);
}
bool issue_tail_slices(rvc_set *s) {
  Tile &t = s->tT;
  if (!issue_due(s, t.pend)) return false;
  for (int p = 0; p < t.G && t.G > 1; ++p)
    if (!issue_due(s, t.ph[p].pend)) return false;
  return true;
}
// calls of a tail period that can carry a slice: all but the one that completes the tail block (it carries the tail job)
static int slice_budget(const rvc_set *s) {
  return std::max(1, (int)std::max<long long>(1, (long long)s->T.B / (long long)s->A.B) - 1);
}
static int slice_channels(const rvc_set *s, int span) {
  const int want = std::max(1, std::min({slice_budget(s), 16, span}));
  return (span + want - 1) / want;
}
int sweep_slices(const rvc_set *s) {
  const int span = s->tT.G > 1 ? std::max(1, s->tT.ph[0].n) : s->nch;
  const int per = slice_channels(s, span);
  return (span + per - 1) / per;
}
// a sweep over channels [base, base + span) to be issued in slices by the calls to come
void plan_sweep(rvc_set *s, Tile::Pending &q, const rvc::FirArgs &a, int timer_id, int base, int span) {
  q = Tile::Pending();
  q.on = true; q.a = a; q.timer_id = timer_id; q.base = base; q.span = span;
  q.budget = slice_budget(s);
  q.per = slice_channels(s, span);
  q.slices = (span + q.per - 1) / q.per;
}
// behind the tail job of block m: the sweep block m + 1 will need, if it is a spread one
void plan_next_tail_sweep(rvc_set *s, long long m) {
  Tile &t = s->tT;
  const long long next = m + 1, td = s->T.delay;
  if (t.G > 1) {                   // phase groups: a group whose tile ends with block m (first-level sweeps only)
    if (!t.lag1) return;
    for (int p = 0; p < t.G; ++p) {
      Tile::Phase &q = t.ph[p];
      if (q.n <= 0) continue;
      t.load_phase(p);
      if (!t.holds(next)) {
        t.start(next, t.K1);
        plan_sweep(s, q.pend, sweep1_args(s, true, next, next - td - t.lag1), 10, q.c0, q.n);
        t.store_phase(p);
      }
    }
    return;
  }
  if (!t.holds(next)) {
    if (!t.lag1) return;
    t.start(next, t.K1);
    plan_sweep(s, t.pend, sweep1_args(s, true, next, next - td - t.lag1), 10, 0, s->nch);
  } else if (t.K1 > rvc::kSweepRows && t.group(next) == next && t.s0 != next && t.lag2) {
    plan_sweep(s, t.pend, sweep2_args(s, true, next), 12, 0, s->nch);
    t.s0 = next;
  }
}

// Tail contributions (IR[2T,..), delivered T.delay tail blocks late) for output blocks
// [tail_out_done, m_hi) into the time-indexed tail ring. Needs spectra of blocks < m_hi - T.delay.
bool tail_rows(rvc_set *s, long long m_hi, hipStream_t st) {
  Stage &T = s->T;
  const long long tb = (long long)T.B;
  const long long m_lo = s->tail_out_done;
  if (m_hi <= m_lo) return true;
  const long long td = T.delay;
  if (!ensure_tail_spectra(s, m_lo - td - (long long)T.P + 1, st)) return false;
  rvc::FirArgs r{};
  r.H = T.H + td * tb; r.h_chan_stride = (long long)T.PF * tb;      // partitions td.. of the whole-IR table
  r.X = T.X; r.x_chan_stride = (long long)T.rows * tb; r.x_row_mask = T.rows - 1;
  r.Y = T.Y; r.y_chan_stride = (long long)T.mcap * tb;
  r.k0 = m_lo; r.M = (int)(m_hi - m_lo); r.P = T.P; r.delay = (int)td; r.B = (int)tb; r.tag = 1;
  const float2 *yrows = T.Y;                      // where the inverse transforms read the spectra
  const bool tiled_row = s->tT.on && r.M == 1;
  if (tiled_row && s->tT.G > 1) {
    // Phase groups (Tile::G): every group of channels runs the schedule below on its own clock -- exactly one group is at the start
    // of a second-level group or of a tile in any tail period, and sweeps over ITS channels only; then ONE patch launch for all
    // channels, every group at its own depth (rvc::PatchGroups; a group whose sweep row is complete has depth 0: the row is
    // copied), into T.Y, which ONE inverse launch reads.
    Tile &t = s->tT;
    rvc::PatchGroups pg{};
    for (int p = 0; p < t.G; ++p) {
      const Tile::Phase &q = t.ph[p];
      if (q.n <= 0) continue;
      auto ranged = [&](rvc::FirArgs a) {            // the launch's share of the channels
        a.H += (long long)q.c0 * a.h_chan_stride; a.X += (long long)q.c0 * a.x_chan_stride;
        if (a.Y) a.Y += (long long)q.c0 * a.y_chan_stride;
        if (a.Ybase) a.Ybase += (long long)q.c0 * a.ybase_chan_stride;
        if (a.Yadd) a.Yadd += (long long)q.c0 * a.yadd_chan_stride;
        a.stage_channels = s->nch;
        return a;
      };
      // a sweep that starts at THIS block and took every row that exists leaves block m_lo's row complete: it writes that row where
      // the patch launch would (FirArgs::Y0 = the group's rows of T.Y) and the group sits the patch launch out
      bool row_done = false;
      auto direct = [&](rvc::FirArgs a, bool complete) {
        if (complete && a.k0 == m_lo) { a.Y0 = r.Y + (long long)q.c0 * r.y_chan_stride; a.y0_chan_stride = r.y_chan_stride; row_done = true; }
        return a;
      };
      t.load_phase(p);
      if (!launch_sweep_slice(s, t.ph[p].pend, 1 << 30, st)) return false;     // (what the calls in between did not carry of a spread sweep)
      if (!t.holds(m_lo)) {
        const int len = t.fresh ? std::max(1, t.K1 - q.phi) : t.K1;
        t.start(m_lo, len);
        const rvc::FirArgs w = direct(ranged(sweep1_args(s, true, m_lo, m_lo - td - t.lag1)), t.lag1 == 0);
        Timer tm(s, 10, st);
        RVC_CK(rvc::launch_fdl_sweep(w, q.n, st));
      }
      const long long g0 = t.group(m_lo);
      if (g0 != t.t0 && t.s0 != g0) {
        const rvc::FirArgs w = direct(ranged(sweep2_args(s, true, g0)), t.lag2 == 0);
        Timer tm(s, 12, st);
        RVC_CK(rvc::launch_fdl_sweep(w, q.n, st));
        t.s0 = g0;
      }
      if (t.s3 && m_lo == g0 + rvc::kThirdRows && t.h0 != m_lo) {     // half way through the group: third-level sweep
        const rvc::FirArgs w = direct(ranged(sweep3_args(s, true, g0, m_lo)), true);
        Timer tm(s, 13, st);
        RVC_CK(rvc::launch_fdl_sweep(w, q.n, st));
        t.h0 = m_lo;
      }
      // the group's entry of the one patch launch below: its sweep row + the partitions whose input arrived since (none: a copy)
      long long stride = 0;
      const float2 *row = tile_row(s, true, m_lo, &stride);
      const int gi = pg.n_groups++;
      pg.c0[gi] = q.c0; pg.n[gi] = q.n;
      pg.P[gi] = (int)std::min<long long>(t.third(m_lo) ? m_lo - t.h0 : m_lo - g0 + (g0 == t.t0 ? t.lag1 : 0), T.P);   // (a spread first-level sweep left its newest row out)
      if (row_done) pg.P[gi] = -1;
      pg.Yadd[gi] = row; pg.yadd_chan_stride[gi] = stride;
      t.store_phase(p);
    }
    bool any_patch = false;                        // (every group's row in place already -- all of them fresh after init / clear(): no launch)
    for (int i = 0; i < pg.n_groups; ++i) any_patch = any_patch || pg.P[i] >= 0;
    if (any_patch) {
      Timer tm(s, 5, st);
      RVC_CK(rvc::launch_fdl_patch_groups(r, pg, s->nch, st));
    }
  } else if (tiled_row) {
    // block-synchronous streaming, time-tiled: output block m_lo lies in the current tile -- whose sweeps have run, were spread over
    // the calls since the block before, or run now -- and only the partitions whose input arrived after the (second-level) sweep
    // are added to that sweep's row
    Tile &t = s->tT;
    if (!flush_tail_sweep(s, st)) return false;     // (whatever the calls in between did not carry)
    if (!t.holds(m_lo)) {
      const int len = (t.fresh && t.first_len > 0) ? std::min(t.first_len, t.K1) : t.K1;
      t.start(m_lo, len);
      const rvc::FirArgs w = sweep1_args(s, true, m_lo, m_lo - td - t.lag1);   // (m_lo - td: the newest delay-line row that exists)
      Timer tm(s, 10, st);
      RVC_CK(rvc::launch_fdl_sweep(w, s->nch, st));
    }
    const long long g0 = t.group(m_lo);
    if (g0 != t.t0 && t.s0 != g0) {                // entering the next group of 8: second-level sweep
      const rvc::FirArgs w = sweep2_args(s, true, g0);
      Timer tm(s, 12, st);
      RVC_CK(rvc::launch_fdl_sweep(w, s->nch, st));
      t.s0 = g0;
    }
    if (t.s3 && m_lo == g0 + rvc::kThirdRows && t.h0 != m_lo) {       // half way through the group: third-level sweep
      const rvc::FirArgs w = sweep3_args(s, true, g0, m_lo);
      Timer tm(s, 13, st);
      RVC_CK(rvc::launch_fdl_sweep(w, s->nch, st));
      t.h0 = m_lo;
    }
    long long stride = 0;
    const float2 *row = tile_row(s, true, m_lo, &stride);
    // input rows that came after the sweep: g0 - td - lag + 1 .. m_lo - td (lag: that of the sweep the row is from; third level: none)
    const long long recent = t.third(m_lo) ? m_lo - t.h0 : m_lo - g0 + (g0 == t.t0 ? t.lag1 : t.lag2);
    if (recent > 0) {
      r.P = (int)std::min<long long>(recent, T.P);
      r.Yadd = row; r.yadd_chan_stride = stride;
      Timer tm(s, 5, st);
      RVC_CK(rvc::launch_fir(r, s->nch, st));
    } else {                                       // the group's first block of an un-spread sweep: its row is complete
      yrows = row; r.y_chan_stride = stride;
    }
  } else {
    s->tT.drop();                                  // several rows at once: plain delay line, any tile is dropped
    Timer t(s, 5, st);
    RVC_CK(rvc::launch_fir(r, s->nch, st));
  }
  rvc::InvArgs v{};
  v.Y = yrows; v.y_chan_stride = r.y_chan_stride; v.tw = T.twp(T.f64i); v.wsplit = T.wsp(T.f64i); v.tw8 = T.t8p(T.f64i); v.tw8_half = T.t8h(T.f64i); v.tw_half = T.twh(T.f64i);
  v.blk0 = m_lo;
  v.dst = s->tailring; v.dst_chan_stride = (long long)s->ring_cap; v.dst_origin = 0; v.dst_mask = s->ring_cap - 1;
  v.lo = 0; v.hi = (long long)1 << 62;
  v.add = nullptr;
  {
    Timer t(s, 6, st);
    RVC_CK(rvc::launch_fft_inv(T.logB, T.f64i, v, r.M, s->nch, st));
  }
  s->tail_out_done = m_hi;
  // the next block's sweep, if a spread one is due: planned now (everything it reads exists), issued in slices by the calls to come
  if (tiled_row && st == s->st_main && (s->tT.lag1 || s->tT.lag2)) plan_next_tail_sweep(s, m_lo);
  return true;
}

// The reference's background job (TwoStageFFTConvolver.cpp:213-222, :247-250), one tail period
// ahead: when a call completes tail block(s), transform them and compute every tail contribution
// whose inputs now exist. On the second stream when RVC_FLAG_BG_STREAM is set.
bool run_tail_job(rvc_set *s, long long n0, long long n1, const float *src2, size_t in_stride, bool bg) {
  const long long tb = (long long)s->T.B;
  const long long mb1 = n1 / tb;
  if (mb1 <= s->tail_fft_done) return true;
  hipStream_t st = bg ? s->st_bg : s->st_main;
  if (bg) {   // startBackgroundProcessing: the job may start once its input is in the ring
    RVC_CK(hipEventRecord(s->ev_ingest, s->st_main));
    RVC_CK(hipStreamWaitEvent(st, s->ev_ingest, 0));
  }
  if (!tail_spectra(s, n0, n1, src2, in_stride, st)) return false;
  const long long m_lo = s->tail_out_done;
  if (!tail_rows(s, mb1 + s->T.delay, st)) return false;
  if (bg && !push_job(s, m_lo, mb1 + s->T.delay, st)) return false;
  return true;
}

// waitForBackgroundProcessing: make the foreground stream wait for the job(s) that produced the tail blocks a call
// ending at n1 reads -- and only those: the job enqueued when tail block m-2 completed delivers output block m, a whole
// tail period later (TwoStageFFTConvolver.cpp:213-222: wait, swap, start the next job), and runs under the head-stage
// work of the period in between.
bool wait_tail_jobs(rvc_set *s, long long n1) {
  const long long m_need = (n1 - 1) / (long long)s->T.B;
  while (s->job_count > 0 && s->jobs[s->job_head].m_lo <= m_need) {   // (jobs are ordered)
    const rvc_set::Job j = s->jobs[s->job_head];
    RVC_CK(hipStreamWaitEvent(s->st_main, j.ev, 0));
    s->job_head = (s->job_head + 1) % rvc_set::kMaxJobs;
    --s->job_count;
    s->ev_pool[s->ev_free++] = j.ev;
  }
  return true;
}

// ---- head stage pieces -------------------------------------------------------------------
// Forward transforms of head blocks [k_lo, k_hi] (samples at or beyond n_hi read as zero: the
// unplayed rest of a partly filled block).
bool head_spectra(rvc_set *s, long long k_lo, long long k_hi, long long n_hi, const float *src2,
                  size_t in_stride, long long src2_from, long long ring_from = -1) {
  Stage &A = s->A;
  const long long hb = (long long)A.B;
  rvc::FwdArgs f{};
  f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
  f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = src2_from;
  f.seg0 = (k_lo - 1) * hb; f.valid_len = (int)(2 * hb); f.lo = 0; f.hi = n_hi;
  f.tw = A.twp(A.f64f); f.wsplit = A.wsp(A.f64f); f.tw8 = A.t8p(A.f64f); f.tw8_half = A.t8h(A.f64f); f.tw_half = A.twh(A.f64f);
  f.dst = A.X; f.dst_chan_stride = (long long)A.rows * hb; f.row0 = k_lo; f.row_mask = A.rows - 1;
  if (ring_from >= 0) {   // the transform kernel also appends the call's recent samples to the time ring
    f.ring_out = s->xring; f.ring_out_chan_stride = (long long)s->ring_cap; f.ring_out_mask = s->ring_cap - 1;
    f.ring_out_from = ring_from;
  }
  Timer t(s, 1, s->st_main);
  RVC_CK(rvc::launch_fft_fwd(A.logB, A.f64f, f, (int)(k_hi - k_lo + 1), s->nch, s->st_main));
  return true;
}

// First head block a stage-A run starting at block ka must transform: ka itself when the delay
// line is contiguous, else (the adaptive long-call path skipped blocks) P-1 blocks of history too.
long long head_fft_from(const rvc_set *s, long long ka) {
  if (ka <= s->xa_next) return ka;
  const long long lo = ka - (long long)s->A.P + 1;
  return lo < 0 ? 0 : lo;
}

// Zero-latency stage over samples [na, nb) of the current call (which starts at n0): FFT, delay
// line, inverse FFT + tail ring -> d_out[na - n0 ..).
bool head_stage(rvc_set *s, long long n0, long long na, long long nb, const float *src2, size_t in_stride,
                float *d_out, size_t out_stride, bool bg, long long ring_from = -1) {
  Stage &A = s->A, &T = s->T;
  const bool has_tail = T.P > 0;
  const long long hb = (long long)A.B;
  const long long ka = na / hb, kb = (nb - 1) / hb;
  const int M = (int)(kb - ka + 1);
  const long long fft_from = head_fft_from(s, ka);
  if (!head_spectra(s, fft_from, kb, nb, src2, in_stride, n0, ring_from)) return false;
  s->xa_next = (nb % hb == 0) ? kb + 1 : kb;
  rvc::FirArgs r{};
  r.H = A.H; r.h_chan_stride = (long long)A.P * hb;
  r.X = A.X; r.x_chan_stride = (long long)A.rows * hb; r.x_row_mask = A.rows - 1;
  r.Y = A.Y; r.y_chan_stride = (long long)A.mcap * hb;
  r.k0 = ka; r.M = M; r.P = A.P; r.delay = 0; r.B = (int)hb;
  const float2 *yrows = A.Y;                      // where the inverse transform reads the spectrum row(s)
  // one WHOLE block in block order, time-tiled (head_gen): the block's row of the tile -- whose sweep runs now if the block starts a
  // tile / a group / the second half of a group -- + the partitions whose input arrived since that sweep (tail_rows' schedule with
  // delay 0: the block's own row exists). Anything else drops the tile and walks the whole delay line.
  const bool tiled = s->head_gen && s->tA.on && M == 1 && na % hb == 0 && nb == na + hb && fft_from == ka;
  if (s->head_gen && !tiled) s->tA.drop();
  if (tiled) {
    Tile &t = s->tA;
    if (!t.holds(ka)) {
      const rvc::FirArgs w = sweep1_args(s, false, ka, ka);
      t.start(ka, t.K1);
      Timer tm(s, 9, s->st_main);
      RVC_CK(rvc::launch_fdl_sweep(w, s->nch, s->st_main));
    }
    const long long g0 = t.group(ka);
    if (g0 != t.t0 && t.s0 != g0) {
      const rvc::FirArgs w = sweep2_args(s, false, g0);
      Timer tm(s, 11, s->st_main);
      RVC_CK(rvc::launch_fdl_sweep(w, s->nch, s->st_main));
      t.s0 = g0;
    }
    if (t.s3 && ka == g0 + rvc::kThirdRows && t.h0 != ka) {
      const rvc::FirArgs w = sweep3_args(s, false, g0, ka);
      Timer tm(s, 14, s->st_main);
      RVC_CK(rvc::launch_fdl_sweep(w, s->nch, s->st_main));
      t.h0 = ka;
    }
    long long stride = 0;
    const float2 *row = tile_row(s, false, ka, &stride);
    const long long recent = t.third(ka) ? ka - t.h0 : ka - g0;
    if (recent > 0) {
      r.P = (int)std::min<long long>(recent, (long long)A.P);
      r.Yadd = row; r.yadd_chan_stride = stride;
      Timer tm(s, 2, s->st_main);
      RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
    } else {                                     // the first block of a sweep: its row is complete
      yrows = row; r.y_chan_stride = stride;
    }
  } else {
    Timer t(s, 2, s->st_main);
    RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
  }
  if (has_tail) {
    if (bg) { if (!wait_tail_jobs(s, nb)) return false; }
    else if (!tail_rows(s, (nb - 1) / (long long)T.B + 1, s->st_main)) return false;   // lazily, if skipped
  }
  rvc::InvArgs v{};
  v.Y = yrows; v.y_chan_stride = r.y_chan_stride; v.tw = A.twp(A.f64i); v.wsplit = A.wsp(A.f64i); v.tw8 = A.t8p(A.f64i); v.tw8_half = A.t8h(A.f64i); v.tw_half = A.twh(A.f64i);
  v.blk0 = ka;
  v.dst = d_out + (na - n0); v.dst_chan_stride = (long long)out_stride; v.dst_origin = na; v.dst_mask = ~0ull;
  v.lo = na; v.hi = nb;
  v.add = has_tail ? s->tailring : nullptr;
  v.add_chan_stride = (long long)s->ring_cap; v.add_mask = s->ring_cap - 1;
  v.add_from = has_tail ? (long long)T.delay * (long long)T.B : 0;
  Timer t(s, 3, s->st_main);
  RVC_CK(rvc::launch_fft_inv(A.logB, A.f64i, v, M, s->nch, s->st_main));
  return true;
}

// Ypre_kb = sum_{i>=1} H_i X_{kb-i}: everything of block kb's spectrum that does not depend on
// block kb's own input (FFTConvolver.cpp:176-185)
// (with s->fold the H_1 X_{kb-1} term moves into block kb's fused kernel and this is sum_{i>=2})
rvc::FirArgs premultiply_args(rvc_set *s, long long kb) {
  Stage &A = s->A;
  const int d = s->fold ? 2 : 1;
  rvc::FirArgs r{};
  r.H = A.H + (size_t)d * A.B; r.h_chan_stride = (long long)A.P * (long long)A.B;
  r.X = A.X; r.x_chan_stride = (long long)A.rows * (long long)A.B; r.x_row_mask = A.rows - 1;
  r.Y = s->ypre + (size_t)(kb & 1) * (size_t)s->nch * A.B; r.y_chan_stride = (long long)A.B;
  r.k0 = kb; r.M = 1; r.P = A.P - d; r.delay = d; r.B = (int)A.B;
  if (r.P < 0) r.P = 0;
  return r;
}

// A first-level sweep of the zero-latency stage for the tile of blocks starting at kb: partial sums of blocks
// kb .. kb+K1-1 over the input rows the sweep may use (<= kb - stage_lag); row kb is complete (= sum_{i>=2} H_i X_{kb-i}).
bool run_head_sweep1(rvc_set *s, long long kb) {
  const rvc::FirArgs r = sweep1_args(s, false, kb, kb - stage_lag(s, false));
  {
    Timer t(s, 9, s->st_main);
    RVC_CK(rvc::launch_fdl_sweep(r, s->nch, s->st_main));
  }
  s->tA.start(kb, s->tA.K1);
  s->ypre_block = kb;
  s->ypre_cur = tile_row(s, false, kb, &s->ypre_cur_stride);
  return true;
}
// The second-level sweep for the group of 8 blocks starting at g0 inside the current tile; row g0 is complete.
bool run_head_sweep2(rvc_set *s, long long g0) {
  const rvc::FirArgs r = sweep2_args(s, false, g0);
  {
    Timer t(s, 11, s->st_main);
    RVC_CK(rvc::launch_fdl_sweep(r, s->nch, s->st_main));
  }
  s->tA.s0 = g0;
  s->ypre_block = g0;
  s->ypre_cur = tile_row(s, false, g0, &s->ypre_cur_stride);
  return true;
}

// The third-level sweep for blocks [h, h + 4), h = g0 + 4 half way through the group that starts at g0 (whose rows exist).
bool run_head_sweep3(rvc_set *s, long long g0, long long h) {
  const rvc::FirArgs r = sweep3_args(s, false, g0, h);
  {
    Timer t(s, 14, s->st_main);
    RVC_CK(rvc::launch_fdl_sweep(r, s->nch, s->st_main));
  }
  s->tA.h0 = h;
  return true;
}

bool run_premultiply(rvc_set *s, long long kb) {
  if (s->tA.on) return run_head_sweep1(s, kb);    // (state was invalidated: a stand-alone sweep starts a new tile at kb)
  const rvc::FirArgs r = premultiply_args(s, kb);
  if (r.P > 0) {      // (no partitions beyond the folded ones: the accumulator stays zero, as allocated)
    Timer t(s, 8, s->st_main);
    RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
  }
  s->ypre_block = kb;
  s->ypre_cur = r.Y;
  s->ypre_cur_stride = r.y_chan_stride;
  return true;
}

// Host-pointer calls: the moment the kernel that produces the call's output is enqueued, enqueue
// the copy back to the pinned buffer and an event behind it. The host then waits for THAT event,
// not for the stream: the pre-multiplied accumulator and the tail job of the next block keep
// running after process() has returned (they were never on the reference's critical path either).
bool emit_output_copy(rvc_set *s) {
  if (s->out_copy_len == 0) return true;
  const size_t len = s->out_copy_len;
  s->out_copy_len = 0;
  if (!s->zero_copy) {             // staging rows are max_len apart (stable per-channel pointers: rvc_set_host_buffers)
    if (len == s->max_len) RVC_CK(hipMemcpyAsync(s->h_out, s->d_out, sizeof(float) * len * s->nch, hipMemcpyDeviceToHost, s->st_main));
    else RVC_CK(hipMemcpy2DAsync(s->h_out, sizeof(float) * s->max_len, s->d_out, sizeof(float) * s->max_len, sizeof(float) * len,
                                 (size_t)s->nch, hipMemcpyDeviceToHost, s->st_main));
  }
  RVC_CK(hipEventRecord(s->ev_out, s->st_main));
  return true;
}

// Single-stage sets with a long-call stage: no tail job keeps that stage's delay line current, so a
// call that did not go through it leaves a hole -- everything up to the end of the call is marked
// missing and the next long call rebuilds the rows it needs from the time ring (ensure_tail_spectra).
void mark_long_stage_stale(rvc_set *s, long long n1) {
  Stage &T = s->T;
  if (T.PF > 0 && T.P == 0) {
    s->tail_fft_done = n1 / (long long)T.B;
    s->xt_valid_lo = s->tail_fft_done;
    s->tT.drop();
  }
}


// one process() step of at most max_len samples, device buffers, asynchronous
bool step_device(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride, size_t len) {
  Stage &A = s->A, &T = s->T;
  const long long n0 = s->n, n1 = n0 + (long long)len;
  const bool has_tail = T.P > 0;
  const bool bg = has_tail && (s->flags & RVC_FLAG_BG_STREAM);
  const long long hb = (long long)A.B;
  const long long k0 = n0 / hb, k1 = (n1 - 1) / hb;

  // A call that crosses ONE head-block boundary (a host whose buffer size is not the head block: 480 frames against
  // 512) = the end of block k0 + the start of block k0 + 1: two steps of the latency path -- one launch each -- instead
  // of the general path's ingest / transform / delay line / inverse launches (measured, stereo pair, host pointers: 37.5 ->
  // ~22 us per 480-frame call). The reference does the same thing in its own terms: it runs a block's transform when its
  // input buffer fills, in the middle of the call (FFTConvolver.cpp:140-207). Host-pointer calls: only the second step
  // publishes completion flags / is followed by the copy back (in-order stream: it implies the first).
  if (k1 == k0 + 1 && !s->block_general && rvc::fused_supported(A.logB, A.f64())) {
    const size_t part1 = (size_t)((k0 + 1) * hb - n0);
    const size_t copy_len = s->out_copy_len;
    s->out_copy_len = 0;
    const bool ok = step_device(s, d_in, in_stride, d_out, out_stride, part1);
    s->out_copy_len = copy_len;
    return ok && step_device(s, d_in + part1, in_stride, d_out + part1, out_stride, len - part1);
  }

  // ---- latency path: the call stays inside one head block (the plugin's per-block call) ----
  if (k0 == k1 && !s->block_general && rvc::fused_supported(A.logB, A.f64())) {
    if (has_tail) {
      if (bg) { if (!wait_tail_jobs(s, n1)) return false; }
      else if (!tail_rows(s, (n1 - 1) / (long long)T.B + 1, s->st_main)) return false;
    }
    if (k0 > s->xa_next) {   // the long-call path skipped head blocks: rebuild the delay line's history
      if (!head_spectra(s, head_fft_from(s, k0), k0 - 1, n0, nullptr, 0, 0)) return false;
      s->xa_next = k0;
      s->ypre_block = -1;
      s->tA.drop();
    }
    // same_block: what block k0 needs from the tile -- its sweep row -- exists? (it does in block order: the sweep that starts a
    // tile / a group runs behind the launch of the block before; not after clear(), a history rebuild, a skipped block)
    if (s->same_block) {
      Tile &ta = s->tA;
      const bool in_tile = ta.t0 >= 0 && k0 >= ta.t0 && k0 < ta.t0 + ta.K1;
      const long long g0 = in_tile ? ta.group(k0) : -1;
      if (!in_tile) { if (!run_head_sweep1(s, k0)) return false; }
      else if (g0 != ta.t0 && ta.s0 != g0) {
        if (k0 == g0) { if (!run_head_sweep2(s, g0)) return false; }
        else if (!run_head_sweep1(s, k0)) return false;               // (mid-group without its rows: start over at k0)
      }
    } else if (s->ypre_block != k0 && !run_premultiply(s, k0)) return false;
    rvc::FusedArgs g{};
    g.in = d_in; g.in_chan_stride = (long long)in_stride;
    g.ring = s->xring; g.ring_chan_stride = (long long)s->ring_cap; g.ring_mask = s->ring_cap - 1;
    g.n0 = n0; g.n1 = n1; g.k = k0;
    g.tw = A.tw; g.wsplit = A.wsplit; g.tw8 = A.tw8;
    g.H0 = A.H; g.h_chan_stride = (long long)A.P * hb;
    g.H1 = (s->fold && A.P > 1) ? A.H + hb : nullptr;
    g.Ypre = s->ypre_cur; g.ypre_chan_stride = s->ypre_cur_stride;
    g.Xrow = A.X; g.x_chan_stride = (long long)A.rows * hb; g.x_row_mask = A.rows - 1;
    g.out = d_out; g.out_chan_stride = (long long)out_stride;
    g.add = has_tail ? s->tailring : nullptr;
    g.add_chan_stride = (long long)s->ring_cap; g.add_mask = s->ring_cap - 1;
    g.add_from = has_tail ? (long long)T.delay * (long long)T.B : 0;
    const bool block_done = n1 % hb == 0;
    // host-pointer call through the pinned buffers: the audio workgroups publish completion flags and
    // process_end polls them -- no event behind the kernel, no wait for the kernel's tail
    const bool flagged = s->out_copy_len != 0 && s->zero_copy && !s->timing;
    if (flagged) {
      g.done_flag = s->h_flags;
      g.seq = ++s->flag_seq;
      s->flag_count = rvc::fused_audio_workgroups(A.logB, s->nch);
      s->out_copy_len = 0;             // nothing to copy back, no event to record
    }
    if (s->same_block) {
      // the patch wave of every workgroup adds, to block k0's sweep row, the partitions whose input arrived after that sweep and
      // hands the row to the audio wave through LDS (FusedArgs::handover); the group's first block: the sweep row as it is
      Tile &ta = s->tA;
      const long long g0 = ta.group(k0);
      rvc::FirArgs f = premultiply_args(s, k0);                        // partitions 2.., delay 2, one row
      // (the group's sweep took the rows <= g0 - stage_lag; the patch's newest row is k0 - 2)
      f.P = (int)std::min<long long>((ta.third(k0) ? k0 - ta.h0 : k0 - g0) - (2 - stage_lag(s, false)), (long long)A.P - 2);
      f.Yadd = tile_row(s, false, k0, &f.yadd_chan_stride);
      f.Y = nullptr;
      if (f.P < 0) f.P = 0;
      g.Ypre = f.Yadd; g.ypre_chan_stride = f.yadd_chan_stride;       // (read when there is nothing to patch; a valid row anyway)
      {
        Timer t(s, 7, s->st_main);
        RVC_CK(rvc::launch_fused2(A.logB, g, f, s->nch, s->st_main));
      }
      if (!emit_output_copy(s)) return false;
      if (block_done) {      // behind the launch, off the call's latency path: the sweep that starts the next tile / group
        const long long kn = k0 + 1;
        if (!(kn > ta.t0 && kn < ta.t0 + ta.K1)) { if (!run_head_sweep1(s, kn)) return false; }
        else if (ta.group(kn) == kn && ta.s0 != kn) { if (!run_head_sweep2(s, kn)) return false; }
        else if (ta.s3 && kn == ta.group(kn) + rvc::kThirdRows && ta.h0 != kn && (ta.group(kn) == ta.t0 || ta.s0 == ta.group(kn))) {
          if (!run_head_sweep3(s, ta.group(kn), kn)) return false;        // half way through the group
        }
      }
      s->ypre_block = -1;
    } else if (s->fold) {
      // the workgroups appended to this launch prepare block k0+1's accumulator (other half of ypre)
      const long long kn = k0 + 1;
      rvc::FirArgs f = premultiply_args(s, kn);
      bool new_tile = false, new_group = false;
      if (s->tA.on && block_done) {
        Tile &ta = s->tA;
        if (ta.t0 >= 0 && kn > ta.t0 && kn < ta.t0 + ta.K1) {
          const long long g0 = ta.group(kn);
          if (g0 == ta.t0 || ta.s0 == g0) {
            // inside a group whose sweep rows exist: that row + the partitions whose input arrived after the sweep
            f.P = (int)std::min<long long>(kn - g0, (long long)A.P - 2);
            f.Yadd = tile_row(s, false, kn, &f.yadd_chan_stride);
            if (f.P <= 0) { f.P = 0; f.Y = const_cast<float2 *>(f.Yadd); f.y_chan_stride = f.yadd_chan_stride; }   // nothing to add
          } else if (kn == g0) {   // the next group of 8 starts: a second-level sweep behind this launch (its row kn is complete)
            f.P = 0;
            new_group = true;
          } else {                 // (cannot happen in block order; be safe: start over)
            f.P = 0;
            new_tile = true;
          }
        } else {      // the tile is used up: a sweep behind this launch starts the next one (its row kn is complete)
          f.P = 0;
          new_tile = true;
        }
      }
      if (!block_done) f.P = 0;
      {
        Timer t(s, 7, s->st_main);
        RVC_CK(rvc::launch_fused2(A.logB, g, f, s->nch, s->st_main));
      }
      if (!emit_output_copy(s)) return false;
      if (block_done) {
        if (new_tile) {
          if (!run_premultiply(s, kn)) return false;     // (sweep launch: sets the tile, ypre_block, ypre_cur)
        } else if (new_group) {
          if (!run_head_sweep2(s, kn)) return false;
        } else {
          s->ypre_block = kn;
          s->ypre_cur = f.Y;
          s->ypre_cur_stride = f.y_chan_stride;
        }
      }
    } else {
      {
        Timer t(s, 7, s->st_main);
        RVC_CK(rvc::launch_fused(A.logB, g, s->nch, s->st_main));
      }
      if (!emit_output_copy(s)) return false;
    }
    s->xa_next = block_done ? k0 + 1 : k0;
    // off the latency path: this call's share of a spread tail sweep, the tail job if a tail block just completed, and (two-launch
    // scheme) the pre-multiplied accumulator of the next block if this one is complete
    if (has_tail && block_done && !issue_tail_slices(s)) return false;
    if (has_tail && !run_tail_job(s, n0, n1, nullptr, in_stride, bg)) return false;
    if (!s->fold && block_done && !run_premultiply(s, k0 + 1)) return false;
    mark_long_stage_stale(s, n1);
    s->n = n1;
    return true;
  }

  // ---- general path: any length ----
  // 1. ingest the call's input into the time ring. For a long call on a single stream the
  // transforms read the call's buffer directly (FwdArgs::src2) and only the history later calls
  // can still need is copied: 2 tail blocks for the tail transforms, P+2 head blocks for a
  // rebuild of the head delay line. (With the tail on the second stream the job may outlive the
  // caller's buffer, so everything is copied; those calls are short.)
  const long long keep = s->keep;
  // (Also the per-block call of many channels with a large head block -- block_general, BASELINE config 5's geometry: one
  //  whole block on its boundary. The head transform reads the block from the caller's buffer and appends it to the ring
  //  itself: no separate copy launch in front of it -- 4 % of that step, a 0.49-of-peak copy feeding the transform.)
  const bool block_call = s->block_general && !bg && k0 == k1 && n0 % hb == 0 && n1 % hb == 0;
  const bool fuse_in = !bg && ((long long)len > keep || block_call);
  // (the adaptive long-call path below lets its forward transform append the history: no ingest launch)
  const bool has_long = T.PF > 0;                 // a whole-IR table at block T exists (two-stage sets; long-call stage of single-stage sets)
  const long long tbq = has_long ? (long long)T.B : 1;
  const long long wbq = s->W.P > 0 ? (long long)s->W.B : 1;
  const bool wide = s->W.P > 0 && !bg && ((n1 - 1) / wbq - n0 / wbq) >= 3;
  const bool adaptive = !wide && has_long && !bg && (s->flags & RVC_FLAG_FIXED_PARTITIONS) == 0 &&
                        ((n1 - 1) / tbq - n0 / tbq) >= 3;
  const bool fft_ingests = (wide && fuse_in) ||
                           (adaptive && fuse_in && rvc::fwd_appends_ring(T.logB) && (n0 / tbq) >= s->tail_fft_done);
  // (likewise the head stage's forward transform when the call goes through the two-stage path)
  const bool head_ingests = !adaptive && fuse_in && rvc::fwd_appends_ring(A.logB);
  if (!fft_ingests && !head_ingests) {
    rvc::IngestArgs a{};
    const long long skip = (fuse_in && (long long)len > keep) ? (long long)len - keep : 0;
    a.src = d_in + skip; a.src_chan_stride = (long long)in_stride;
    a.ring = s->xring; a.ring_chan_stride = (long long)s->ring_cap; a.ring_mask = s->ring_cap - 1;
    a.n0 = n0 + skip; a.len = (long long)len - skip;
    Timer t(s, 0, s->st_main);
    RVC_CK(rvc::launch_ingest(a, s->nch, s->st_main));
  }
  const float *src2 = fuse_in ? d_in : nullptr;

  // 2a. very long calls: the same idea one size up. A call touching >= 4 blocks of 16384 samples
  // goes through the wide stage (whole IR at block 16384: half the partitions of stage T). Neither
  // the head nor the tail stage runs; their state is rebuilt lazily by later, shorter calls.
  if (wide) {
    Stage &W = s->W;
    const long long wb = (long long)W.B;
    const long long m_first = n0 / wb, m_last = (n1 - 1) / wb;
    long long fft_lo = m_first;
    if (m_first > s->w_next) {                    // not contiguous with the last wide call: rebuild history rows
      fft_lo = m_first - (long long)W.P + 1;
      if (fft_lo < 0) fft_lo = 0;
    }
    {
      rvc::FwdArgs f{};
      f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
      f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = n0;
      f.seg0 = (fft_lo - 1) * wb; f.valid_len = (int)(2 * wb); f.lo = 0; f.hi = n1;
      f.tw = W.twp(W.f64f); f.wsplit = W.wsp(W.f64f); f.tw8 = W.t8p(W.f64f); f.tw8_half = W.t8h(W.f64f); f.tw_half = W.twh(W.f64f);
      f.dst = W.X; f.dst_chan_stride = (long long)W.rows * wb; f.row0 = fft_lo; f.row_mask = W.rows - 1;
      if (fft_ingests) {
        f.ring_out = s->xring; f.ring_out_chan_stride = (long long)s->ring_cap; f.ring_out_mask = s->ring_cap - 1;
        f.ring_out_from = n1 - keep;
      }
      Timer t(s, 4, s->st_main);
      RVC_CK(rvc::launch_fft_fwd(W.logB, false, f, (int)(m_last - fft_lo + 1), s->nch, s->st_main));
    }
    s->w_next = (n1 % wb == 0) ? m_last + 1 : m_last;
    rvc::FirArgs r{};
    r.H = W.H; r.h_chan_stride = (long long)W.P * wb;
    r.X = W.X; r.x_chan_stride = (long long)W.rows * wb; r.x_row_mask = W.rows - 1;
    r.Y = W.Y; r.y_chan_stride = (long long)W.mcap * wb;
    r.k0 = m_first; r.M = (int)(m_last - m_first + 1); r.P = W.P; r.delay = 0; r.B = (int)wb; r.tag = 2;
    {
      Timer t(s, 5, s->st_main);
      RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
    }
    rvc::InvArgs v{};
    v.Y = W.Y; v.y_chan_stride = r.y_chan_stride; v.tw = W.twp(W.f64i); v.wsplit = W.wsp(W.f64i); v.tw8 = W.t8p(W.f64i); v.tw8_half = W.t8h(W.f64i); v.tw_half = W.twh(W.f64i);
    v.blk0 = m_first;
    v.dst = d_out; v.dst_chan_stride = (long long)out_stride; v.dst_origin = n0; v.dst_mask = ~0ull;
    v.lo = n0; v.hi = n1;
    v.add = nullptr;
    {
      Timer t(s, 6, s->st_main);
      RVC_CK(rvc::launch_fft_inv(W.logB, false, v, r.M, s->nch, s->st_main));
    }
    // the tail stage saw none of this: its transforms are marked missing (rebuilt on demand) and the
    // tail-ring rows of blocks delivered directly are never needed
    const long long tb = (long long)T.B;
    s->tail_fft_done = n1 / tb;
    s->xt_valid_lo = s->tail_fft_done;
    const long long done = (n1 % tb == 0) ? (n1 - 1) / tb + 1 : (n1 - 1) / tb;
    if (s->tail_out_done < done) s->tail_out_done = done;
    s->tT.drop();
    s->n = n1;
    return true;
  }

  // 2. adaptive partitioning for long calls. The result does not depend on the partition sizes,
  // only the latency does -- and a call that hands over many tail blocks at once has no use for
  // 512-sample latency inside them. A call touching >= 4 tail blocks is therefore produced
  // entirely by ONE uniform delay line at block T over the whole IR (partitions 0..P_T+1,
  // delay 0): transform the tail blocks it completes (plus the partly filled one it ends in),
  // one FIR, one inverse transform windowed to [n0, n1). The head stage is not run at all; its
  // state (delay-line history, tail-ring rows) is rebuilt lazily by the next short call.
  if (adaptive) {
    const long long tb = (long long)T.B;
    const long long m_first = n0 / tb, m_last = (n1 - 1) / tb;
    {
      const long long ring_from = fft_ingests ? n1 - keep : -1;
      const long long mb0 = s->tail_fft_done, mb1 = n1 / tb;
      const int extra = (n1 % tb != 0) ? 1 : 0;       // the partly filled block the call ends in
      if (!ensure_tail_spectra(s, m_first - (long long)T.PF + 1, s->st_main)) return false;   // (after a wide call)
      // (A two-way pipeline over block time -- second half's transforms on the side stream under the
      // first half's delay line -- was measured and lost 35 %: each half-size launch keeps ~8 us of
      // fixed cost. One launch per stage it is.)
      auto fwd = [&](long long r0, long long r1, hipStream_t st) -> bool {   // transforms of rows [r0, r1)
        if (r1 <= r0) return true;
        rvc::FwdArgs f{};
        f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
        f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = n0;
        f.seg0 = (r0 - 1) * tb; f.valid_len = (int)(2 * tb); f.lo = 0; f.hi = n1;
        f.tw = T.twp(T.f64f); f.wsplit = T.wsp(T.f64f); f.tw8 = T.t8p(T.f64f); f.tw8_half = T.t8h(T.f64f); f.tw_half = T.twh(T.f64f);
        f.dst = T.X; f.dst_chan_stride = (long long)T.rows * tb; f.row0 = r0; f.row_mask = T.rows - 1;
        if (ring_from >= 0) {
          f.ring_out = s->xring; f.ring_out_chan_stride = (long long)s->ring_cap; f.ring_out_mask = s->ring_cap - 1;
          f.ring_out_from = ring_from;
        }
        Timer t(s, 4, st);
        RVC_CK(rvc::launch_fft_fwd(T.logB, T.f64f, f, (int)(r1 - r0), s->nch, st));
        return true;
      };
      auto fir_inv = [&](long long r0, long long r1) -> bool {               // output rows [r0, r1)
        rvc::FirArgs r{};
        r.H = T.H; r.h_chan_stride = (long long)T.PF * tb;
        r.X = T.X; r.x_chan_stride = (long long)T.rows * tb; r.x_row_mask = T.rows - 1;
        r.Y = T.Y + (r0 - m_first) * tb; r.y_chan_stride = (long long)T.mcap * tb;
        r.k0 = r0; r.M = (int)(r1 - r0); r.P = T.PF; r.delay = 0; r.B = (int)tb; r.tag = 2;
        {
          Timer t(s, 5, s->st_main);
          RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
        }
        rvc::InvArgs v{};
        v.Y = r.Y; v.y_chan_stride = r.y_chan_stride; v.tw = T.twp(T.f64i); v.wsplit = T.wsp(T.f64i); v.tw8 = T.t8p(T.f64i); v.tw8_half = T.t8h(T.f64i); v.tw_half = T.twh(T.f64i);
        v.blk0 = r0;
        v.dst = d_out; v.dst_chan_stride = (long long)out_stride; v.dst_origin = n0; v.dst_mask = ~0ull;
        v.lo = n0; v.hi = n1;
        v.add = nullptr;
        Timer t(s, 6, s->st_main);
        RVC_CK(rvc::launch_fft_inv(T.logB, T.f64i, v, r.M, s->nch, s->st_main));
        return true;
      };
      if (!fwd(mb0, mb1 + extra, s->st_main)) return false;
      if (!fir_inv(m_first, m_last + 1)) return false;
      if (mb1 > mb0) s->tail_fft_done = mb1;
      // tail-ring rows of blocks delivered directly are never needed; the one the call ends in is
      // computed lazily if a later short call continues inside it
      const long long done = (n1 % tb == 0) ? m_last + 1 : m_last;
      if (s->tail_out_done < done) s->tail_out_done = done;
      s->tT.drop();
      s->n = n1;
      return true;
    }
  }

  // 3. two-stage path: tail job one period ahead, then the zero-latency stage over the whole call
  if (block_call && head_ingests) {
    // (the per-block call of many channels: the head transform appends the block to the ring, so the tail job runs BEHIND it
    //  and reads whole rows from the ring alone -- which keeps the row-looping form of its 8192-bin transforms, 0.57 instead
    //  of 0.34 of the HBM peak at 2048 rows. Nothing of THIS call's output depends on the job: with delay d >= 1 the job that
    //  tail block m completes serves output blocks >= m + d, the earliest of which starts with the NEXT call. That holds only
    //  because both run on st_main in this order (block_call implies !bg): a tail job on another stream would have to be
    //  waited for by the next call's head stage, as the bg path does.)
    if (bg || (has_tail && T.delay < 1)) return fail(s, RVC_ERR_HIP, hipSuccess, "block_call: tail job must follow the head stage on st_main with delay >= 1");
    if (!head_stage(s, n0, n0, n1, src2, in_stride, d_out, out_stride, bg, n0)) return false;
    if (has_tail && !issue_tail_slices(s)) return false;
    if (has_tail && !run_tail_job(s, n0, n1, nullptr, in_stride, bg)) return false;
    mark_long_stage_stale(s, n1);
    s->n = n1;
    return true;
  }
  if (has_tail && !run_tail_job(s, n0, n1, src2, in_stride, bg)) return false;
  if (!head_stage(s, n0, n0, n1, src2, in_stride, d_out, out_stride, bg, head_ingests ? std::max(n0, n1 - keep) : -1)) return false;
  mark_long_stage_stale(s, n1);
  s->n = n1;
  return true;
}

// Child sets run on their own streams; the CALLER of a device-pointer entry still sees one: child 0's foreground stream is
// the set's (rvc_set_stream(s, 0)) -- no further stream, the runtime has few hardware queues to map them on. Going in, every
// other child's foreground stream waits for what the caller has ordered before that stream (the producer of d_in); going
// out, that stream waits for every other child's work of this call (so an event / a kernel behind it sees d_out
// complete). One event record + one wait per further child and direction; rvc_set_process_device_blocks fences ONCE around
// its whole loop, so inside it the children still run unsynchronised (which is where their gain comes from). A caller that makes
// ONE device-pointer call per block for thousands of channels pays the fence per block -- a barrier between the children at every
// block, measured 13.9 against 15.5 Gsamples/s on one queue and 16.2 for the fenced-once loop (examples/lockstep_instances) -- and
// either keeps one queue (RVC_FLAG_NO_SUBSETS) or takes the children UNFENCED (RVC_FLAG_CHILD_SETS: no fence anywhere, the caller
// brackets any run of calls with rvc_set_fork / rvc_set_join -- these two functions -- or orders its work against every child's
// stream, rvc_set_stream(s, 2 + 2 k)).
bool fence_children_in(rvc_set *s, bool explicit_call) {
  rvc_set *f = s->kids[0];
  if (s->tune.kid_fence == 0 || (!explicit_call && (s->flags & RVC_FLAG_CHILD_SETS) != 0) || !f->streams_ok) return true;
  if (!use_device(s)) return false;
  RVC_CK(hipEventRecord(f->ev_fence, f->st_main));
  for (size_t k = 1; k < s->kids.size(); ++k)
    if (s->kids[k]->streams_ok) RVC_CK(hipStreamWaitEvent(s->kids[k]->st_main, f->ev_fence, 0));
  return true;
}
bool fence_children_out(rvc_set *s, bool explicit_call) {
  rvc_set *f = s->kids[0];
  if (s->tune.kid_fence == 0 || (!explicit_call && (s->flags & RVC_FLAG_CHILD_SETS) != 0) || !f->streams_ok) return true;
  for (size_t k = 1; k < s->kids.size(); ++k) {
    rvc_set *c = s->kids[k];
    if (!c->streams_ok) continue;
    RVC_CK(hipEventRecord(c->ev_fence, c->st_main));
    RVC_CK(hipStreamWaitEvent(f->st_main, c->ev_fence, 0));
  }
  return true;
}
void forward_device_call(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride, size_t len) {
  for (size_t k = 0; k < s->kids.size(); ++k)
    rvc_set_process_device(s->kids[k], d_in + (size_t)s->kid_c0[k] * in_stride, in_stride,
                           d_out + (size_t)s->kid_c0[k] * out_stride, out_stride, len);
}

bool zero_device_out(rvc_set *s, float *d_out, size_t out_stride, size_t len) {
  if (len == 0 || !d_out) return true;
  if (s->streams_ok) {
    // (a set that has streams but cannot select its device any more: the caller would read stale output with last_error OK)
    const hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) return fail(s, RVC_ERR_NO_DEVICE, e, "hipSetDevice (zeroing the output of a failed / empty set)");
    RVC_CK(hipMemset2DAsync(d_out, out_stride * sizeof(float), 0, len * sizeof(float), (size_t)s->nch, s->st_main));
  } else if (hipSetDevice(s->device) == hipSuccess) {   // never initialised with a non-empty IR: no stream yet
    (void)hipMemset2D(d_out, out_stride * sizeof(float), 0, len * sizeof(float), (size_t)s->nch);
  }
  return true;
}

}  // namespace rvc_eng
