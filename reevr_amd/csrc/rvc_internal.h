// rvc_internal.h -- launch interface between the engine (rvc_plan / rvc_state / rvc_schedule / rvc_abi .cpp, rvc_set.h) and the gfx950
// kernels (rvc_kernels.hip). Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct rvc_impulse;

namespace rvc {

// Everything time-domain is addressed by ABSOLUTE sample index n (samples since the last
// clear()); rings are power-of-two sized and indexed with n & mask. Everything
// frequency-domain is addressed by absolute block index (row) and bin.
//
// Spectrum rows hold B = block complex bins (float2, interleaved). Bin 0 is PACKED:
// .x = DC, .y = Nyquist (both are real for real signals), so a row is exactly B complex
// values (a power of two, 8*B bytes, always 16-byte aligned) instead of the reference's
// B+1 split-complex values (Utilities.h:197-272).

struct FwdArgs {            // real time-domain segment(s) -> spectrum row(s)
  const float *src;         // [channel][ring] time-domain samples
  long long src_chan_stride;
  unsigned long long src_mask;   // index = n & src_mask
  const float *src2;        // optional second source: the current call's input, [channel][len]; samples
  long long src2_chan_stride;   // n >= src2_from are read from src2[n - src2_from] instead of the ring
  long long src2_from;      // (saves the ingest copy of long calls); nullptr = ring only
  float *ring_out;          // optional (radix-8 kernels only): every block's own B samples with index
  long long ring_out_chan_stride;   // >= ring_out_from (and < hi) are also appended to this time ring
  unsigned long long ring_out_mask; // -- the history later calls need -- so that a long call needs no
  long long ring_out_from;  // separate ingest launch
  long long seg0;           // absolute start of row 0's 2B-sample segment
  int valid_len;            // samples at the start of the segment that may be non-zero (2B: overlap-save
                            // input segment, B: zero-padded IR partition)
  long long lo, hi;         // global validity window: samples outside [lo, hi) read as zero
  const void *tw;           // B twiddles      e^{-2 pi i j / B}     (float2 or double2, see f64)
  const void *wsplit;       // B twiddles      e^{-2 pi i k / 2B}
  const void *tw8;          // per-pass tables of the radix-8 kernels (B >= 512), see Plan8
  float2 *dst;              // [channel][row][B]
  long long dst_chan_stride;
  long long row0;           // absolute row index of row 0
  unsigned long long row_mask;   // row slot = (row0 + r) & row_mask
  const void *tw8_half;     // optional (float2): per-pass tables and base twiddles of the HALF-size transform -- the 16384-bin float forward
  const void *tw_half;      // then runs as two 8192-point sub-transforms in two workgroups (k_fft8_fwd_dif2)
  int rows;                 // set by the launcher (several small transforms share a workgroup)
};

struct FirArgs {            // Y[k] = sum_i H[i] * X[k - delay - i]   (frequency-domain delay line)
  const float2 *H;          // [channel][P][B]
  long long h_chan_stride;
  const float2 *X;          // [channel][ring rows][B]
  long long x_chan_stride;
  unsigned long long x_row_mask;
  float2 *Y;                // [channel][M rows][B]
  long long y_chan_stride;
  long long k0;             // absolute block index of output row 0
  int M;                    // output rows
  int P;                    // partitions
  int delay;                // block delay of the line: the tail stage's 2 (1 for the delay-1 forms of many-channel sets), the
                            // zero-latency stage's 0 (its sweeps and patches start at partition 2 with delay 2)
  int B;
  int tag;                  // names the kernel instantiation for profilers: 0 head stage, 1 tail stage
                            // (delay 2), 2 whole-IR delay line of the adaptive long-call path
  // Block-synchronous time tiling (single-row kernels and the sweep kernel only):
  const float2 *Yadd;       // optional [channel][B] row added to the single output row (a sweep's partial sum)
  long long yadd_chan_stride;
  long long x_hi;           // sweep: input rows with index > x_hi have not arrived yet and read as zero
  unsigned y_row_mask;      // sweep: output row of block k0 + j is slot (k0 + j) & y_row_mask of Y
  // Second-level sweep of the two-level tiling (rvc_internal.h, "Causal time tiling"): only the input rows
  // x_from <= row <= x_hi count (the older ones are in the first-level rows already), and the first-level rows
  // Ybase (slot (k0 + j) & ybase_row_mask, [channel][rows][B]) are added to the output.
  long long x_from;         // sweep: input rows with index < x_from read as zero (0: every row since the clock started)
  const float2 *Ybase;      // sweep: optional rows added to the output rows
  long long ybase_chan_stride;
  unsigned ybase_row_mask;
  float2 *Y0;               // sweep: optional [channel][B]: the row of block k0 -- complete when the sweep took every row that exists -- goes
  long long y0_chan_stride; // HERE instead of its slot of Y (phase groups: straight into the row the inverse transform reads, no copy)
  int stream;               // sweep (set by launch_fdl_sweep): the stage's IR spectra of this launch exceed the last-level cache:
                            // accumulator rows are stored non-temporally, second-level sweeps load their IR rows non-temporally
  int stage_channels;       // sweeps and patches: this launch covers a SLICE of the channels of a stage of stage_channels of them (0: all of
                            // them are in this launch): kernel form and cache policy are chosen for the whole stage, so slices agree
};

// Patches of the phase groups of a tail stage in ONE launch (launch_fdl_patch_groups): channels [c0, c0 + n) of group g add P
// recent partitions to the row Yadd + c * yadd_chan_stride (c = the channel's index in the launch); P = 0 copies the row, P < 0: nothing
// to do (the group's sweep of this very block wrote the row where the patch would, FirArgs::Y0).
struct PatchGroups {
  static constexpr int kMax = 8;
  int n_groups;
  int c0[kMax], n[kMax], P[kMax];
  const float2 *Yadd[kMax];
  long long yadd_chan_stride[kMax];
};

struct InvArgs {            // spectrum row(s) -> last B samples of the inverse transform (overlap-save)
  const float2 *Y;          // [channel][M rows][B]
  long long y_chan_stride;
  const void *tw;
  const void *wsplit;
  const void *tw8;
  long long blk0;           // absolute block index of row 0; row r produces samples [(blk0+r)B, (blk0+r+1)B)
  float *dst;               // [channel][...]
  long long dst_chan_stride;
  long long dst_origin;     // element index = (n - dst_origin) & dst_mask
  unsigned long long dst_mask;
  long long lo, hi;         // only samples with lo <= n < hi are written
  const float *add;         // optional [channel][ring] stream added to the result (tail contribution)
  long long add_chan_stride;
  unsigned long long add_mask;
  long long add_from;       // add applies for n >= add_from
  const void *tw8_half;     // optional: the per-pass tables (double2, fft8_table_entries(logB - 1)) of the HALF-size transform: the
                            // 8192-bin double inverse then runs as two 4096-point sub-transforms (k_fft8_inv_dif2)
  const void *tw_half;      // optional (float2, B / 2 entries e^{-2 pi i j / (B/2)}): the final radix-2 pass of a half-size plan that has one
                            // (the 16384-bin float inverse as two 8192-point sub-transforms)
  int rows;                 // set by the launcher
};

// One head block per call (the plugin's per-block process()): ingest + forward transform +
// Y = Ypre + H0 * X + inverse transform + tail add, one launch, one workgroup per channel.
// Ypre = sum_{i>=1} H_i X_{k-i} is prepared off the critical path when block k-1 completes
// (the reference's _preMultiplied, FFTConvolver.cpp:176-185).
struct FusedArgs {
  const float *in;          // [channel][len] the call's input (device)
  long long in_chan_stride;
  float *ring;              // time ring (read history, append the call's input)
  long long ring_chan_stride;
  unsigned long long ring_mask;
  long long n0, n1;         // the call covers absolute samples [n0, n1), all inside block k
  long long k;              // block index
  const void *tw, *wsplit, *tw8;
  const float2 *H0;         // [channel][B] partition 0 of the IR spectra
  const float2 *H1;         // partition 1 (k_fused_block2 only; nullptr when the stage has one partition)
  long long h_chan_stride;
  const float2 *Ypre;       // [channel][B]: sum_{i>=1} H_i X_{k-i} (k_fused_block) / sum_{i>=2} (k_fused_block2)
  long long ypre_chan_stride;
  float2 *Xrow;             // [channel][rows][B]: where X_k is stored for later blocks
  long long x_chan_stride;
  unsigned long long x_row_mask;
  float *out;               // [channel][len]
  long long out_chan_stride;
  const float *add;         // tail ring or nullptr
  long long add_chan_stride;
  unsigned long long add_mask;
  long long add_from;
  int channels;             // set by the launcher
  // Host-pointer calls: when set, every audio workgroup publishes `seq` to done_flag[workgroup]
  // (pinned host memory, system scope) right behind its output stores, and the host polls that
  // instead of waiting for an event behind the whole kernel.
  // k_fused_block2w with a patch of THIS block (launch_fused2 sets it when FirArgs f describes block k's own accumulator,
  // rvc::fused_same_block): the patch wave leaves sum_{i>=2, recent} H_i X_{k-i} + the sweep row in LDS, the audio wave adds it
  // behind its forward transform -- the accumulator never travels through global memory (Ypre is not read).
  int handover;
  unsigned *done_flag;
  unsigned seq;
  unsigned long long *dbg;  // always nullptr in the shipped build (rounds 2-3 stamped workgroup 0's phases through it: docs/history.md section 5b)
};

struct IngestArgs {
  const float *src;         // [channel][len] (device)
  long long src_chan_stride;
  float *ring;
  long long ring_chan_stride;
  unsigned long long ring_mask;
  long long n0;
  long long len;
};

// All launchers return hipGetLastError() of the launch. logB = log2(block).
// f64: run the transform in double (LDS data + twiddles); tw/wsplit must then point to double2 tables.
hipError_t launch_fft_fwd(int logB, bool f64, const FwdArgs &a, int rows, int channels, hipStream_t st);
hipError_t launch_fft_inv(int logB, bool f64, const InvArgs &a, int rows, int channels, hipStream_t st);
hipError_t launch_fir(const FirArgs &a, int channels, hipStream_t st);
// one single-row patch launch over all `channels`: a.H / a.X / a.Y / a.k0 / a.delay / a.B as for launch_fir, partitions and base row per
// channel group from g (a.P / a.Yadd are ignored); every group's P <= the patch kernel's limit (kSweepRows - 1 + kSweepLagMax)
hipError_t launch_fdl_patch_groups(const FirArgs &a, const PatchGroups &g, int channels, hipStream_t st);
// whether launch_fft_fwd honours FwdArgs::ring_out for this block size / precision
bool fwd_appends_ring(int logB);
// fused single-block step; supported for 6 <= logB <= 13 (float transforms only)
bool fused_supported(int logB, bool f64);
hipError_t launch_fused(int logB, const FusedArgs &a, int channels, hipStream_t st);
// One launch per block: the audio path with H_1 X_{k-1} folded in (a.Ypre = sum_{i>=2}) plus, when
// f.P > 0, the workgroups that compute the next block's sum_{i>=2} (f: M = 1 row, any delay).
bool fused_fold_supported(int logB);
// head blocks whose folded launch is ONE workgroup of an audio wave + a patch wave per channel group (128 / 256 / 512): with a
// time-tiled delay line the patch wave then works on the SAME block as the audio wave and hands its row over through LDS
// (FusedArgs::handover): f = the patch of block a.k (Yadd = its sweep row, P recent partitions), f.Y unused
bool fused_same_block(int logB);
int fused_audio_workgroups(int logB, int channels);
hipError_t launch_fused2(int logB, const FusedArgs &a, const FirArgs &f, int channels, hipStream_t st);
hipError_t launch_ingest(const IngestArgs &a, int channels, hipStream_t st);
// arm (a, b) / disarm (nullptr, nullptr) kernel-exact timing events for the next launch on this thread
void set_launch_events(hipEvent_t a, hipEvent_t b);
void get_launch_events(hipEvent_t *a, hipEvent_t *b);   // (for launchers in other translation units)
// one-off: raise the dynamic-LDS limit of the large FFT kernels
hipError_t prepare_kernels();
// Kernel / schedule variants the launchers choose between. Every set carries its own copy (rvc_set::tune, fixed when the set
// is created); the engine announces it to the launchers of THIS thread for the duration of an entry point (TuneScope in
// rvc_set.h), so two handles used from two threads never see each other's knobs. Defaults = what ships.
struct LaunchTune {
  int fft_loop = -1;     // row-looping form of the 8192-bin transforms: -1 by size, 0 never, 1 whenever legal
  int fft_many = -1;     // many-rows form of the 4096-bin transforms (twiddles per pass, 4 workgroups per CU): -1 from 2048 rows on / 0 / 1
  int tile_rot = 1;      // sweeps / patches on long rows take channel c's bin tiles in the order rotated by c (XCD spread)
  int block_occ = 0;     // 4 = the lean 4-waves-per-SIMD per-block kernel for many-channel launches (measurement)
  int patch_nt = 1;      // 0 = ordinary loads in the stand-alone patch kernel (default: non-temporal for many channels)
  int sweep_split = -1;  // -1 auto / 0 own-tile form / 1 partition-split form of the 8-block sweeps
  int sweep_lw = 0;      // 4 = 16-byte lanes for the 16-block first-level sweeps (default by row length)
  int sweep_d = 0;       // 8 = eight row pairs requested ahead in the long-tile sweeps (default 4)
  int sweep_lds = -1;    // LDS-fed first-level sweeps, accumulators split over waves (rvc_sweep.hip): -1 default / 0 off / 1 / 2 / 3
  int sweep_nt = -1;     // sweeps of a stage stream (non-temporal accumulator-row stores, second-level IR loads): -1 by the stage's size / 0 / 1
  int inv_dif = -1;      // 8192-bin DOUBLE inverse as two 4096-point sub-transforms in two workgroups (k_fft8_inv_dif2): 0 off / else on
  int mac3 = -1;         // three-product complex multiply-accumulate in the LDS-fed 32-block sweeps: -1 default / 0 off / 1 on
  int fwd_dif14 = -1;    // 16384-bin FLOAT forward as two 8192-point sub-transforms in two workgroups (k_fft8_fwd_dif2<13>): 0 off / else on
  int inv_dif14 = -1;    // 16384-bin FLOAT inverse as two 8192-point sub-transforms in two workgroups (k_fft8_inv_dif2<13, float>): 0 off / else on
  int block_lanex = -1;  // per-block kernel of head 512: second exchange of its transforms lane-locally (v_permlane32/16_swap + DPP): -1 by size / 0 / 1
};
void set_launch_tune(const LaunchTune *t);   // thread-local; nullptr = the defaults above
const LaunchTune &launch_tune();

// radix-8 kernels (logB >= 9): number of entries of their per-pass twiddle table, laid out as
//   for j = 1 .. N8-1 (N8 = logB / 3):  p = 8^j entries [k < p][r < 8] = e^{-2 pi i r k / (8 p)}
//   then, if logB % 3 == 2:             [k < B/4][r < 4]               = e^{-2 pi i r k / B}
int fft8_table_entries(int logB);

// time-tile (output rows per thread) the FIR launcher will pick for M rows
int fir_time_tile(int M);

// Causal time tiling of the block-synchronous delay line. A "sweep" computes, for the a.M output blocks
// k0 .. k0 + M - 1 at once, the part of  Y_k = sum_i H_i X_{k - delay - i}  whose input rows have already
// arrived (index <= a.x_hi): every IR row and every delay-line row is read ONCE for M blocks instead of once
// per block. What is missing from block k0 + j -- at most j (+ delay-dependent) recent rows -- is added when that
// block is due (a single-row launch with FirArgs::Yadd = the sweep's row). Rows go to slot (k0 + j) & y_row_mask.
//
// Two levels for long delay lines (P > kTwoLevelMinP partitions): with P partitions a tile of K blocks costs
// (2 P + K) / K rows per block for the sweep and ~K + 1 for the patches -- for hundreds of partitions (BASELINE
// config 3: 350 tail partitions) no single K is good. So a FIRST-level sweep covers K1 = 16 or 32 blocks (all
// partitions, rows <= t0 - 2), every kSweepRows = 8 blocks inside that tile a SECOND-level sweep adds the rows that
// arrived since the first-level sweep (at most K1 + 1 partitions: FirArgs::x_from / Ybase) for the next 8 blocks,
// and the per-block patches still add at most 7 (+ 2 + lag) partitions: 2 P / K1 + K1 / 8 + 12 rows per block.
constexpr int kSweepRows = 8;     // tile of the level the patches work on (second level, or the only one)
constexpr int kSweepRowsMax = 32; // largest first-level tile
constexpr int kThirdRows = 4;     // blocks a third-level sweep (half way through a group of kSweepRows) prepares
constexpr int kTwoLevelMinP = 24; // up to this many partitions one level of 8 blocks is as good (P / 4 + 10 rows per block against
                                  // P / 8 + 12); measured at 32 partitions (config 2's zero-latency stage): 375 -> 365 B/sample, +2-4 %
constexpr int kLongLineMinP = 48; // from this many partitions on the first level covers 32 blocks. Round 3: 80 (the one-wave 32-block sweep ran at
                                  // 0.50 of the HBM peak); with the LDS-fed three-product form (0.62-0.73) the longer tile pays earlier -- measured on
                                  // MI355X (profiles/r5_k1.txt): 58 tail partitions of 8192 (config 2) 16.81 -> 17.33 Gsamples/s, 64 head partitions of
                                  // 256 (config 3) 11.56 -> 11.87, 29 tail partitions (config 5's geometry) 22.25 -> 21.79: stays at 16
constexpr int kSweepLagMax = 3;   // a sweep may run up to this many blocks before its first row is due (x_hi that much older)
// M = 8, 16 or 32 output rows
hipError_t launch_fdl_sweep(const FirArgs &a, int channels, hipStream_t st);
// A prepared impulse as rvc_set_init_impulse sees it (rvc_impulse.hip): device pointers of the
// prepared channels and their lengths with trailing |x| < 1e-6 dropped (TwoStageFFTConvolver.cpp:107-110).
struct ImpulseView {
  int device;
  int channels;
  size_t size;
  const float *ch[4];
  size_t trimmed[4];
};
bool impulse_view(rvc_impulse *m, ImpulseView *v);   // synchronises the impulse's stream

}  // namespace rvc
