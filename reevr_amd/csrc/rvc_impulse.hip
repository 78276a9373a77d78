// rvc_impulse.hip -- impulse preparation on the device: the deterministic
// array stages of the reference's Impulse::recalcImpulse (src/dsp/Impulse.cpp:307-360), so that an
// IR goes raw samples -> prepared IR -> partition spectra without leaving HBM.
//
//   k_imp_reduce        calculateAutoGain's energy sum (Impulse.cpp:691-708) + the peak scan (:343-349)
//   k_imp_stage_a       auto gain * reverse * trim * gain in one pass (:319-338, :436-486)
//   k_imp_dectab        the per-frame cumulative decay of applyDecay (:625-631), all frames at once
//   k_imp_stft          one 4096-point STFT frame per workgroup: window, forward transform, per-bin
//                       decay, inverse transform (:611-635). The transform runs in double like the
//                       reference's (AudioFFT.cpp:114-159) with the spectrum rounded to float between
//                       the two, so the frames match the CPU path to the last float bit or two.
//   k_imp_ola           overlap-add of the frames in the reference's order, window normalisation
//                       (:637-648), then clip (:488-499) and the attack/decay envelope (:651-680)
//   k_imp_last_nz       the trailing-silence scan of TwoStageFFTConvolver::init (:107-110) for
//                       rvc_set_init_impulse
//
// All of it is bandwidth-trivial (an IR is a few MB); the point is residency and latency: a
// parameter tweak re-runs recalc + init in well under a millisecond of device time instead of the
// reference's tens of milliseconds of CPU STFT per channel.
#include "rvc_internal.h"
#include "rvc_fft_lds.hpp"
#include "../../include/reevr_amd/rvc.h"

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#pragma clang fp contract(off)   // keep the reference's separate float roundings (mul, then mul / add)

namespace rvc {

constexpr int IMP_N = RVC_IMPULSE_FFT_SIZE;      // 4096 real points
constexpr int IMP_B = IMP_N / 2;                 // as a 2048-point complex transform
constexpr int IMP_LOGB = 11;
constexpr int IMP_HOP = IMP_N / 4;               // Impulse.h:23
constexpr int IMP_LUT = RVC_IMPULSE_LUT_SIZE;
constexpr int IMP_RED = 4096;                    // samples per workgroup of the reductions

struct ImpPtrs { const float *src[4]; float *dst[4]; };

__global__ void __launch_bounds__(256) k_imp_reduce(const ImpPtrs p, const int nc, const size_t n,
                                                    double *__restrict__ part_energy, float *__restrict__ part_max) {
  __shared__ double se[256];
  __shared__ float sm[256];
  const size_t base = (size_t)blockIdx.x * IMP_RED;
  double e = 0.0;
  float mx = 0.f;
  for (int j = (int)threadIdx.x; j < IMP_RED; j += 256) {
    const size_t i = base + j;
    if (i < n) {
      const double vl = (double)p.src[0][i], vr = (double)p.src[1][i];
      e += vl * vl + vr * vr;
      for (int c = 0; c < nc; ++c) mx = fmaxf(mx, fabsf(p.src[c][i]));
    }
  }
  se[threadIdx.x] = e;
  sm[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      se[threadIdx.x] += se[threadIdx.x + w];
      sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part_energy[blockIdx.x] = se[0];
    part_max[blockIdx.x] = sm[0];
  }
}

// out[c][j] = (raw[c][src(j)] * autoGain) * gain for the kept range [start, start + m)
__global__ void __launch_bounds__(256) k_imp_stage_a(const ImpPtrs p, const size_t n, const size_t start, const size_t m,
                                                     const int reverse, const float auto_gain, const float gain) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (j >= m) return;
  const size_t t = start + j;
  const size_t src = reverse ? (n - 1 - t) : t;
  float v = p.src[c][src] * auto_gain;
  v = v * gain;
  p.dst[c][j] = v;
}

// tab[b][k] = (float) prod_{b' in (skip, b]} lut[k]   (decayACC, Impulse.cpp:608, :626-628)
__global__ void __launch_bounds__(256) k_imp_dectab(const double *__restrict__ lut, float *__restrict__ tab,
                                                    const int nframes, const int skip) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= IMP_LUT) return;
  const double f = lut[k];
  double acc = 1.0;
  for (int b = 0; b < nframes; ++b) {
    if (b > skip) acc = acc * f;
    tab[(size_t)b * IMP_LUT + k] = (float)acc;
  }
}

__global__ void __launch_bounds__(256) k_imp_stft(const ImpPtrs p, float *__restrict__ frames, const size_t frames_chan_stride,
                                                  const float *__restrict__ window, const float *__restrict__ tab,
                                                  const cx<double> *__restrict__ tw, const cx<double> *__restrict__ wsplit,
                                                  const size_t n, const int skip) {
  typedef cx<double> C;
  __shared__ C s[IMP_B];
  const int tid = threadIdx.x;
  const int b = blockIdx.x, c = blockIdx.y;
  const float *buf = p.src[c];
  const size_t start = (size_t)b * IMP_HOP;

  for (int m = tid; m < IMP_B; m += 256) {            // block[i] = buf[start + i] * window[i], zero padded
    const int q = 2 * m;
    float v0 = 0.f, v1 = 0.f;
    if (start + q < n) v0 = buf[start + q] * window[q];
    if (start + q + 1 < n) v1 = buf[start + q + 1] * window[q + 1];
    s[m] = mk<double>((double)v0, (double)v1);
  }
  __syncthreads();
  cfft_lds<IMP_LOGB, false, double>(s, tw, tid);

  // forward real split -> float spectrum (re/im are float vectors in the reference) -> decay ->
  // inverse real split, pair (k, B-k) at a time, in place
  const bool dec_on = b > skip;
  const float *dec = tab + (size_t)b * IMP_LUT;
  const double sc = 0.5 / (double)IMP_B;              // 1/N of the inverse (AudioFFT.cpp:158: 2/N on N/2 points)
  for (int k = tid; k <= IMP_B / 2; k += 256) {
    if (k == 0) {
      const C z = s[0];
      float dc = (float)(z.x + z.y), ny = (float)(z.x - z.y);
      if (dec_on) ny *= dec[IMP_B];                   // bin N/2; the DC bin is never scaled (:625 starts at k = 1)
      s[0] = mk<double>(sc * ((double)dc + (double)ny), sc * ((double)dc - (double)ny));
    } else {
      const C A = s[k], Bc = cconj(s[IMP_B - k]);
      const C E = mk<double>(0.5 * (A.x + Bc.x), 0.5 * (A.y + Bc.y));
      const C D = mk<double>(0.5 * (A.x - Bc.x), 0.5 * (A.y - Bc.y));
      const C O = mk<double>(D.y, -D.x);
      const C wO = cmul(wsplit[k], O);
      const C X0 = cadd(E, wO), X1 = csub(E, wO);
      float r0 = (float)X0.x, i0 = (float)X0.y;        // X[k]
      float r1 = (float)X1.x, i1 = (float)(-X1.y);     // X[B-k]
      if (dec_on) {
        const float d0 = dec[k], d1 = dec[IMP_B - k];
        r0 *= d0; i0 *= d0; r1 *= d1; i1 *= d1;
      }
      const C Yk = mk<double>((double)r0, (double)i0), Yc = mk<double>((double)r1, -(double)i1);
      const C E2 = mk<double>(sc * (Yk.x + Yc.x), sc * (Yk.y + Yc.y));
      const C D2 = mk<double>(sc * (Yk.x - Yc.x), sc * (Yk.y - Yc.y));
      const C O2 = cmul(cconj(wsplit[k]), D2);
      s[k] = mk<double>(E2.x - O2.y, E2.y + O2.x);
      if (k != IMP_B - k) s[IMP_B - k] = mk<double>(E2.x + O2.y, -E2.y + O2.x);
    }
  }
  __syncthreads();
  cfft_lds<IMP_LOGB, true, double>(s, tw, tid);

  float *dst = frames + (size_t)c * frames_chan_stride + (size_t)b * IMP_N;
  for (int m = tid; m < IMP_B; m += 256) {
    const C z = s[m];
    reinterpret_cast<float2 *>(dst)[m] = make_float2((float)z.x, (float)z.y);
  }
}

// output[o] = sum over the (up to 4) frames covering o, ascending frame index like the
// reference's loop (:637-644); / norm; clip; attack; decay
__global__ void __launch_bounds__(256) k_imp_ola(const ImpPtrs p, const float *__restrict__ frames, const size_t frames_chan_stride,
                                                 const float *__restrict__ window, const size_t n, const int nframes,
                                                 const int has_decay, const int attack_size, const int decay_size,
                                                 int *__restrict__ last) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (o >= n) return;
  float v;
  if (has_decay) {
    const float *f = frames + (size_t)c * frames_chan_stride;
    const int h = (int)(o / IMP_HOP);
    float acc = 0.f, norm = 0.f;
    for (int b = h - 3; b <= h; ++b) {
      if (b < 0 || b >= nframes) continue;
      const int i = (int)(o - (size_t)b * IMP_HOP);     // < 4096; o < n so i < blockSize of frame b
      acc += f[(size_t)b * IMP_N + i];
      norm += window[i];
    }
    v = norm > 0.0f ? acc / norm : 0.f;
  } else {
    v = p.src[c][o];
  }
  v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
  const int i = (int)o, size = (int)n;
  if (i < attack_size) v *= (float)i / (float)attack_size;
  if (i >= size - decay_size) {
    const float t = (float)(i - (size - decay_size)) / (float)decay_size;
    v *= 1.0f - (float)sqrt((double)t);                 // pow(t, 0.5), :669
  }
  p.dst[c][o] = v;
  // trailing-silence scan for the convolver's init (TwoStageFFTConvolver.cpp:107-110), for free here
  const bool nz = !(fabsf(v) < 0.000001f);
  const unsigned long long ball = __ballot(nz);
  if (ball && (threadIdx.x & 63) == 63 - __builtin_clzll(ball)) atomicMax(&last[c], i + 1);
}

__global__ void __launch_bounds__(256) k_imp_last_nz(const ImpPtrs p, const size_t n, int *__restrict__ last) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= n) return;
  if (!(fabsf(p.src[c][i]) < 0.000001f)) atomicMax(&last[c], (int)i + 1);
}

}  // namespace rvc

// ------------------------------------------------------------------------------------------------

struct rvc_impulse {
  int device = 0;
  int err = RVC_OK;
  std::string errstr;
  hipStream_t st = nullptr;
  int nc = 0;
  size_t raw_len = 0;      // samples per raw channel
  size_t cap = 0;          // allocated floats per channel
  size_t size = 0;         // current prepared length (bufferLL.size())
  bool staged = false;     // stage A has run since set_raw
  float *d_raw[4] = {nullptr, nullptr, nullptr, nullptr};
  float *d_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  double *d_part_e = nullptr;
  float *d_part_m = nullptr;
  size_t part_cap = 0;
  float *d_frames = nullptr;
  size_t frames_cap = 0;   // floats
  float *d_tab = nullptr;
  size_t tab_cap = 0;
  double *d_lut = nullptr;
  float *d_window = nullptr;
  rvc::cx<double> *d_tw = nullptr, *d_wsplit = nullptr;
  int *d_last = nullptr;
  float peak = 0.f;
  int trim_l = 0, trim_r = 0;
  int last_nz[4] = {0, 0, 0, 0};   // per channel: 1 + index of the last sample with |x| >= 1e-6
  bool last_valid = false;
};

namespace {

bool ifail(rvc_impulse *m, int code, hipError_t e, const char *what) {
  if (m->err == RVC_OK) {
    m->err = code;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    m->errstr = buf;
  }
  return false;
}

#define IMP_CK(expr)                                                 \
  do {                                                               \
    hipError_t e__ = (expr);                                         \
    if (e__ != hipSuccess) return ifail(m, RVC_ERR_HIP, e__, #expr); \
  } while (0)

bool imp_ready(rvc_impulse *m) {
  if (m->st) {
    const hipError_t e0 = hipSetDevice(m->device);
    return e0 == hipSuccess || ifail(m, RVC_ERR_NO_DEVICE, e0, "hipSetDevice");
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= m->device)
    return ifail(m, RVC_ERR_NO_DEVICE, e, "no usable HIP device (this engine has no CPU fallback)");
  IMP_CK(hipSetDevice(m->device));
  IMP_CK(hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking));
  // Blackman window exactly as Impulse::Impulse builds it (Impulse.cpp:64-70, float arithmetic)
  std::vector<float> window(rvc::IMP_N);
  const float w = 2.0f * 3.14159265358979323846f / rvc::IMP_N;
  for (int i = 0; i < rvc::IMP_N / 2; ++i) window[i] = 0.42f - 0.50f * std::cos(i * w) + 0.08f * std::cos(2.0f * i * w);
  for (int i = rvc::IMP_N / 2; i < rvc::IMP_N; ++i) window[i] = window[rvc::IMP_N - 1 - i];
  const double kPi = 3.14159265358979323846;
  std::vector<rvc::cx<double>> tw(rvc::IMP_B), ws(rvc::IMP_B + 1);
  for (int j = 0; j < rvc::IMP_B; ++j) {
    const double a = -2.0 * kPi * j / rvc::IMP_B;
    tw[j].x = std::cos(a); tw[j].y = std::sin(a);
  }
  for (int k = 0; k <= rvc::IMP_B; ++k) {
    const double a = -kPi * k / rvc::IMP_B;
    ws[k].x = std::cos(a); ws[k].y = std::sin(a);
  }
  IMP_CK(hipMalloc(&m->d_window, sizeof(float) * window.size()));
  IMP_CK(hipMalloc(&m->d_tw, sizeof(tw[0]) * tw.size()));
  IMP_CK(hipMalloc(&m->d_wsplit, sizeof(ws[0]) * ws.size()));
  IMP_CK(hipMalloc(&m->d_lut, sizeof(double) * rvc::IMP_LUT));
  IMP_CK(hipMalloc(&m->d_last, sizeof(int) * 4));
  IMP_CK(hipMemcpy(m->d_window, window.data(), sizeof(float) * window.size(), hipMemcpyHostToDevice));
  IMP_CK(hipMemcpy(m->d_tw, tw.data(), sizeof(tw[0]) * tw.size(), hipMemcpyHostToDevice));
  IMP_CK(hipMemcpy(m->d_wsplit, ws.data(), sizeof(ws[0]) * ws.size(), hipMemcpyHostToDevice));
  return true;
}

inline void imp_dev_free(void *p) {
  if (p) hipFree(p);
}

void imp_free(rvc_impulse *m) {
  if (m->st) hipStreamSynchronize(m->st);
  for (int c = 0; c < 4; ++c) {
    if (m->d_raw[c]) hipFree(m->d_raw[c]);
    if (m->d_buf[c]) hipFree(m->d_buf[c]);
    m->d_raw[c] = m->d_buf[c] = nullptr;
  }
  void *ptrs[] = {m->d_part_e, m->d_part_m, m->d_frames, m->d_tab, m->d_lut, m->d_window, m->d_tw, m->d_wsplit, m->d_last};
  for (void *q : ptrs)
    if (q) hipFree(q);
  m->d_part_e = nullptr; m->d_part_m = nullptr; m->d_frames = nullptr; m->d_tab = nullptr; m->d_lut = nullptr;
  m->d_window = nullptr; m->d_tw = nullptr; m->d_wsplit = nullptr; m->d_last = nullptr;
  m->cap = m->part_cap = m->frames_cap = m->tab_cap = 0;
  if (m->st) hipStreamDestroy(m->st);
  m->st = nullptr;
}

rvc::ImpPtrs ptrs_of(const rvc_impulse *m, bool from_raw) {
  rvc::ImpPtrs p{};
  for (int c = 0; c < 4; ++c) {
    p.src[c] = from_raw ? m->d_raw[c] : m->d_buf[c];
    p.dst[c] = m->d_buf[c];
  }
  return p;
}

bool stage_a(rvc_impulse *m, const rvc_impulse_params *p) {
  m->peak = 0.f;
  m->trim_l = m->trim_r = 0;
  m->size = 0;
  m->staged = true;
  m->last_valid = false;
  const size_t n = m->raw_len;
  if (n == 0 || m->nc == 0) return true;
  if (!imp_ready(m)) return false;
  const int nblk = (int)((n + rvc::IMP_RED - 1) / rvc::IMP_RED);
  if ((size_t)nblk > m->part_cap) {
    imp_dev_free(m->d_part_e);
    imp_dev_free(m->d_part_m);
    m->d_part_e = nullptr; m->d_part_m = nullptr; m->part_cap = 0;
    IMP_CK(hipMalloc(&m->d_part_e, sizeof(double) * nblk));
    IMP_CK(hipMalloc(&m->d_part_m, sizeof(float) * nblk));
    m->part_cap = (size_t)nblk;
  }
  const rvc::ImpPtrs pr = ptrs_of(m, true);
  hipLaunchKernelGGL(rvc::k_imp_reduce, dim3(nblk), dim3(256), 0, m->st, pr, m->nc, n, m->d_part_e, m->d_part_m);
  std::vector<double> pe(nblk);
  std::vector<float> pm(nblk);
  IMP_CK(hipMemcpyAsync(pe.data(), m->d_part_e, sizeof(double) * nblk, hipMemcpyDeviceToHost, m->st));
  IMP_CK(hipMemcpyAsync(pm.data(), m->d_part_m, sizeof(float) * nblk, hipMemcpyDeviceToHost, m->st));
  IMP_CK(hipStreamSynchronize(m->st));
  double energy = 0.0;
  float maxabs = 0.f;
  for (int i = 0; i < nblk; ++i) {
    energy += pe[i];
    maxabs = std::fmax(maxabs, pm[i]);
  }
  float auto_gain = 1.0f;                                 // Impulse.cpp:699-707
  if (energy > 0.0) {
    double g = 1.0 / std::sqrt(energy);
    if (g > 1.0) g = 1.0;
    auto_gain = (float)g;
  }
  // max_i |x_i * g| = |max_i |x_i| * g| : float multiplication by a constant is monotonic
  m->peak = maxabs * auto_gain;
  // applyTrim, Impulse.cpp:436-470 (float * size_t -> float, truncated)
  const size_t total = n;
  const size_t start = (size_t)(p->trim_left * total);
  const size_t end = total - (size_t)(p->trim_right * total);
  if (start >= end || start >= total || end > total) return true;   // everything trimmed: empty buffers
  m->trim_l = (int)start;
  m->trim_r = (int)(total - end);
  const size_t keep = end - start;
  hipLaunchKernelGGL(rvc::k_imp_stage_a, dim3((unsigned)((keep + 255) / 256), m->nc), dim3(256), 0, m->st, pr, n, start,
                     keep, p->reverse ? 1 : 0, auto_gain, p->gain);
  IMP_CK(hipGetLastError());
  m->size = keep;
  return true;
}

bool stage_b(rvc_impulse *m, const rvc_impulse_params *p) {
  const size_t n = m->size;
  if (n == 0) return true;
  if (!imp_ready(m)) return false;
  const rvc::ImpPtrs pb = ptrs_of(m, false);
  const int nframes = (int)((n + rvc::IMP_HOP - 1) / rvc::IMP_HOP);
  const size_t fstride = (size_t)nframes * rvc::IMP_N;
  const bool has_decay = p->decay_lut != nullptr;
  if (has_decay) {
    if (fstride * m->nc > m->frames_cap) {
      imp_dev_free(m->d_frames);
      m->d_frames = nullptr; m->frames_cap = 0;
      IMP_CK(hipMalloc(&m->d_frames, sizeof(float) * fstride * m->nc));
      m->frames_cap = fstride * m->nc;
    }
    if ((size_t)nframes * rvc::IMP_LUT > m->tab_cap) {
      imp_dev_free(m->d_tab);
      m->d_tab = nullptr; m->tab_cap = 0;
      IMP_CK(hipMalloc(&m->d_tab, sizeof(float) * (size_t)nframes * rvc::IMP_LUT));
      m->tab_cap = (size_t)nframes * rvc::IMP_LUT;
    }
    const int skip = (int)std::ceil(100 /* EARLY_REFLECTIONS_MS, src/Globals.h:34 */ * p->srate / (1000.0 * rvc::IMP_N));
    IMP_CK(hipMemcpyAsync(m->d_lut, p->decay_lut, sizeof(double) * rvc::IMP_LUT, hipMemcpyHostToDevice, m->st));
    hipLaunchKernelGGL(rvc::k_imp_dectab, dim3((rvc::IMP_LUT + 255) / 256), dim3(256), 0, m->st, m->d_lut, m->d_tab, nframes, skip);
    hipLaunchKernelGGL(rvc::k_imp_stft, dim3(nframes, m->nc), dim3(256), 0, m->st, pb, m->d_frames, fstride, m->d_window,
                       m->d_tab, m->d_tw, m->d_wsplit, n, skip);
  }
  const int size = (int)n;
  const int attack_size = (int)(p->attack * size);          // Impulse.cpp:656-657
  const int decay_size = (int)(p->decay * size);
  IMP_CK(hipMemsetAsync(m->d_last, 0, sizeof(int) * 4, m->st));
  hipLaunchKernelGGL(rvc::k_imp_ola, dim3((unsigned)((n + 255) / 256), m->nc), dim3(256), 0, m->st, pb, m->d_frames, fstride,
                     m->d_window, n, nframes, has_decay ? 1 : 0, attack_size, decay_size, m->d_last);
  IMP_CK(hipGetLastError());
  IMP_CK(hipMemcpyAsync(m->last_nz, m->d_last, sizeof(int) * 4, hipMemcpyDeviceToHost, m->st));
  IMP_CK(hipStreamSynchronize(m->st));   // decay_lut is caller memory: consumed before return
  m->last_valid = true;
  return true;
}

}  // namespace

namespace rvc {

bool impulse_view(rvc_impulse *m, ImpulseView *v) {
  if (!m || !v) return false;
  v->device = m->device;
  v->channels = m->nc;
  v->size = m->size;
  for (int c = 0; c < 4; ++c) { v->ch[c] = nullptr; v->trimmed[c] = 0; }
  if (m->size == 0 || m->nc == 0) return true;
  if (!imp_ready(m)) return false;
  if (!m->last_valid) {   // stage B has not run since the buffers changed (stage A only, or rvc_impulse_write)
    IMP_CK(hipMemsetAsync(m->d_last, 0, sizeof(int) * 4, m->st));
    hipLaunchKernelGGL(k_imp_last_nz, dim3((unsigned)((m->size + 255) / 256), m->nc), dim3(256), 0, m->st, ptrs_of(m, false),
                       m->size, m->d_last);
    IMP_CK(hipMemcpyAsync(m->last_nz, m->d_last, sizeof(int) * 4, hipMemcpyDeviceToHost, m->st));
    IMP_CK(hipStreamSynchronize(m->st));
    m->last_valid = true;
  }
  for (int c = 0; c < m->nc; ++c) { v->ch[c] = m->d_buf[c]; v->trimmed[c] = (size_t)m->last_nz[c]; }
  return true;
}

}  // namespace rvc

extern "C" {

rvc_impulse *rvc_impulse_create(int device) {
  rvc_impulse *m = new (std::nothrow) rvc_impulse();
  if (m) m->device = device < 0 ? 0 : device;
  return m;
}

void rvc_impulse_destroy(rvc_impulse *m) {
  if (!m) return;
  if (m->st) hipSetDevice(m->device);
  imp_free(m);
  delete m;
}

int rvc_impulse_set_raw(rvc_impulse *m, int n_channels, const float *const *raw, size_t len) {
  if (!m) return 0;
  m->err = RVC_OK;
  m->errstr.clear();
  m->size = 0; m->staged = false; m->last_valid = false; m->peak = 0.f; m->trim_l = m->trim_r = 0;
  if ((n_channels != 2 && n_channels != 4) || (!raw && len)) { ifail(m, RVC_ERR_BAD_ARG, hipSuccess, "n_channels / raw"); return 0; }
  if (len >= ((size_t)1 << 31) - rvc::IMP_N) { ifail(m, RVC_ERR_UNSUPPORTED, hipSuccess, "impulse longer than 2^31 samples"); return 0; }
  m->nc = n_channels;
  m->raw_len = len;
  if (len == 0) return 1;
  if (!imp_ready(m)) return 0;
  if (len > m->cap) {
    for (int c = 0; c < 4; ++c) {
      imp_dev_free(m->d_raw[c]);
      imp_dev_free(m->d_buf[c]);
      m->d_raw[c] = m->d_buf[c] = nullptr;
    }
    m->cap = 0;
    for (int c = 0; c < 4; ++c) {
      if (hipMalloc(&m->d_raw[c], sizeof(float) * len) != hipSuccess || hipMalloc(&m->d_buf[c], sizeof(float) * len) != hipSuccess) {
        ifail(m, RVC_ERR_HIP, hipErrorOutOfMemory, "hipMalloc(impulse)");
        return 0;
      }
    }
    m->cap = len;
  }
  for (int c = 0; c < n_channels; ++c) {
    if (!raw[c]) { ifail(m, RVC_ERR_BAD_ARG, hipSuccess, "raw[c]"); return 0; }
    hipError_t e = hipMemcpyAsync(m->d_raw[c], raw[c], sizeof(float) * len, hipMemcpyHostToDevice, m->st);
    if (e != hipSuccess) { ifail(m, RVC_ERR_HIP, e, "hipMemcpy(raw)"); return 0; }
  }
  hipError_t e = hipStreamSynchronize(m->st);
  if (e != hipSuccess) { ifail(m, RVC_ERR_HIP, e, "hipMemcpy(raw)"); return 0; }
  return 1;
}

int rvc_impulse_stage_a(rvc_impulse *m, const rvc_impulse_params *p) {
  if (!m || !p) return 0;
  if (m->err != RVC_OK) return 0;
  return stage_a(m, p) ? 1 : 0;
}

int rvc_impulse_stage_b(rvc_impulse *m, const rvc_impulse_params *p) {
  if (!m || !p) return 0;
  if (m->err != RVC_OK) return 0;
  if (!m->staged) { ifail(m, RVC_ERR_NOT_INIT, hipSuccess, "stage B before stage A"); return 0; }
  return stage_b(m, p) ? 1 : 0;
}

int rvc_impulse_recalc(rvc_impulse *m, const rvc_impulse_params *p) {
  return rvc_impulse_stage_a(m, p) && rvc_impulse_stage_b(m, p);
}

void rvc_impulse_decay_lut(const float *mag, double srate, float decay_rate, double *lut) {
  // Impulse.cpp:561-590; constants src/Globals.h:36-38
  const float kMaxGain = 24.f, kRatePos = 2.f, kRateNeg = 0.9f;
  const double decayPerSecond = 1.0 - kRateNeg, growPerSecond = 1.0 + kRatePos;
  const double decayPerBlock = std::pow(decayPerSecond, (RVC_IMPULSE_FFT_SIZE / srate) * decay_rate);
  const double growPerBlock = std::pow(growPerSecond, (RVC_IMPULSE_FFT_SIZE / srate) * decay_rate);
  const double lnDecay = std::log(decayPerBlock), lnGrow = std::log(growPerBlock);
  for (int i = 0; i < RVC_IMPULSE_LUT_SIZE; ++i) {
    const float dB = 20.0f * std::log10(mag[i]);
    float norm = (kMaxGain - dB) / (2.f * kMaxGain);
    norm = norm < 0.f ? 0.f : (norm > 1.f ? 1.f : norm);
    norm = (norm * 2.f - 1.f) * -1.f;
    double d = 1.0;
    if (norm > 0.f) d = std::exp(norm * lnGrow);
    else if (norm < 0.f) d = std::exp(-norm * lnDecay);
    lut[i] = d;
  }
}

int rvc_impulse_channels(const rvc_impulse *m) { return m ? m->nc : 0; }
size_t rvc_impulse_size(const rvc_impulse *m) { return m ? m->size : 0; }
float rvc_impulse_peak(const rvc_impulse *m) { return m ? m->peak : 0.f; }
int rvc_impulse_trim_left_samples(const rvc_impulse *m) { return m ? m->trim_l : 0; }
int rvc_impulse_trim_right_samples(const rvc_impulse *m) { return m ? m->trim_r : 0; }

int rvc_impulse_read(rvc_impulse *m, int channel, float *dst, size_t n) {
  if (!m || channel < 0 || channel >= m->nc || n > m->size || (!dst && n)) return 0;
  if (n == 0) return 1;
  if (!imp_ready(m)) return 0;
  hipError_t e = hipMemcpyAsync(dst, m->d_buf[channel], sizeof(float) * n, hipMemcpyDeviceToHost, m->st);
  if (e == hipSuccess) e = hipStreamSynchronize(m->st);
  if (e != hipSuccess) { ifail(m, RVC_ERR_HIP, e, "hipMemcpy(read)"); return 0; }
  return 1;
}

int rvc_impulse_write(rvc_impulse *m, int channel, const float *src, size_t n) {
  if (!m || channel < 0 || channel >= m->nc || n > m->size || (!src && n)) return 0;
  if (n == 0) return 1;
  if (!imp_ready(m)) return 0;
  m->last_valid = false;
  hipError_t e = hipMemcpyAsync(m->d_buf[channel], src, sizeof(float) * n, hipMemcpyHostToDevice, m->st);
  if (e == hipSuccess) e = hipStreamSynchronize(m->st);
  if (e != hipSuccess) { ifail(m, RVC_ERR_HIP, e, "hipMemcpy(write)"); return 0; }
  return 1;
}

const float *rvc_impulse_device_ptr(rvc_impulse *m, int channel) {
  if (!m || channel < 0 || channel >= m->nc || m->size == 0) return nullptr;
  return m->d_buf[channel];
}

int rvc_impulse_last_error(const rvc_impulse *m) { return m ? m->err : RVC_ERR_BAD_ARG; }
const char *rvc_impulse_last_error_string(const rvc_impulse *m) { return m ? m->errstr.c_str() : "null handle"; }

}  // extern "C"

// ================================================================================================
// Wet-bus epilogue on the device: what processBlock does with the
// convolvers' output buffers, src/PluginProcessor.cpp:1800-1876, in one pass over device-resident
// blocks: crossfade of the fading-in convolver (per-sample alpha, :1808-1821), true-stereo sum
// LL + RL / RR + LR (:1828-1838), reverb envelope and mid/side width (:1840-1857), dry/wet mix
// (:1860-1876). Same float operations in the same order as the reference loop.
// ================================================================================================
namespace rvc {

struct WetArgs {
  const float *cur[4];    // LL, RR, LR, RL of the current convolver (LR / RL nullptr: not quad or true stereo off)
  const float *load[2];   // LL, RR of the fading-in convolver (nullptr: not fading)
  long long xfade, xfadelen;
  const float *yrev;      // reverb envelope per sample, nullptr = 1
  float width, drygain, wetgain;
  const float *dry[2];    // nullptr: wet only
  float *out[2];
  size_t n;
};

__global__ void __launch_bounds__(256) k_wet_mix(const WetArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  float ll = a.cur[0][i], rr = a.cur[1][i];
  float lr = a.cur[2] ? a.cur[2][i] : 0.f, rl = a.cur[3] ? a.cur[3][i] : 0.f;
  float wl = 0.f, wr = 0.f;                                  // wetBuffer starts cleared
  // (this file is compiled with fp contraction off -- #pragma at the top, -ffp-contract=fast-honor-pragmas --
  //  so every product and sum rounds on its own like the reference's scalar loop)
  if (a.load[0]) {
    // alpha = clamp(1 - xfade / xfadelen, 0, 1) with xfade counting down one per sample (:1809, :1820)
    float alpha = 1.f - (float)(a.xfade - (long long)i) / (float)a.xfadelen;
    alpha = alpha < 0.f ? 0.f : (alpha > 1.f ? 1.f : alpha);
    const float keep = 1.f - alpha;
    ll *= keep; rr *= keep; lr *= keep; rl *= keep;
    const float fl = a.load[0][i] * alpha, fr = a.load[1][i] * alpha;   // :1812-1813
    wl += fl;                                                // :1829-1830
    wr += fr;
  }
  wl += ll; wr += rr;                                        // :1834-1835
  if (a.cur[2]) { wl += rl; wr += lr; }                      // :1836-1838 (L gets RL, R gets LR)
  const float env = a.yrev ? a.yrev[i] : 1.f;
  const float lin = wl * env, rin = wr * env;                // :1844-1845
  const float mid = (lin + rin) * 0.5f, side = (lin - rin) * 0.5f;
  const float normalization = 1.0f / (1.0f + a.width);
  const float sw = side * a.width;
  float lout = (mid + sw) * normalization;
  float rout = (mid - sw) * normalization;
  if (a.dry[0]) {                                            // :1860-1876
    const float dl = a.dry[0][i] * a.drygain, dr = a.dry[1][i] * a.drygain;
    const float gl = lout * a.wetgain, gr = rout * a.wetgain;
    lout = dl + gl;
    rout = dr + gr;
  }
  a.out[0][i] = lout;
  a.out[1][i] = rout;
}

}  // namespace rvc

extern "C" int rvc_wet_mix_device(int device, void *stream, const rvc_wet_params *p) {
  if (!p || !p->cur[0] || !p->cur[1] || !p->out[0] || !p->out[1]) return 0;
  if ((p->cur[2] == nullptr) != (p->cur[3] == nullptr)) return 0;
  if ((p->load[0] == nullptr) != (p->load[1] == nullptr)) return 0;
  if ((p->dry[0] == nullptr) != (p->dry[1] == nullptr)) return 0;
  if (p->load[0] && p->xfadelen <= 0) return 0;
  if (p->n == 0) return 1;
  if (hipSetDevice(device < 0 ? 0 : device) != hipSuccess) return 0;
  rvc::WetArgs a{};
  for (int c = 0; c < 4; ++c) a.cur[c] = p->cur[c];
  for (int c = 0; c < 2; ++c) { a.load[c] = p->load[c]; a.dry[c] = p->dry[c]; a.out[c] = p->out[c]; }
  a.xfade = p->xfade; a.xfadelen = p->xfadelen; a.yrev = p->yrev;
  a.width = p->width; a.drygain = p->drygain; a.wetgain = p->wetgain; a.n = p->n;
  hipLaunchKernelGGL(rvc::k_wet_mix, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 1 : 0;
}

// ================================================================================================
// Send pre-stage on the device: what processBlock does in FRONT of the
// convolvers for blocks that stay in HBM -- send envelope (src/PluginProcessor.cpp:1640-1653, without
// the serial IIR send filters :1643-1650, which stay with the host), the warm-up ring write (:1655-1668)
// and the pre-delay ring write + read (:1766-1790). One pass; the reference's "write the whole block,
// then read it back predelay samples late" becomes: a sample whose read position was written by this
// very block is taken from the block itself, every other one from the ring as it was.
// ================================================================================================
namespace rvc {

struct SendArgs {
  const float *in[2];
  const float *ysend;
  float *send[2];
  float *delay_ring[2];
  long long delay_size, delaypos, predelay;
  float *delayed[2];
  float *warm_ring[2];
  long long warm_size, warmwritepos;
  long long n;
};

__global__ void __launch_bounds__(256) k_send_pre(const SendArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= a.n) return;
  const float *in = a.in[c];
  auto send_at = [&](long long j) -> float { return a.ysend ? in[j] * a.ysend[j] : in[j]; };   // :1641-1642
  const float s = send_at(i);
  if (a.send[c]) a.send[c][i] = s;
  // rings: sample i lands on slot (pos + i) % size; when the block is longer than the ring the LAST writer wins
  if (a.warm_ring[c] && i + a.warm_size >= a.n) a.warm_ring[c][(a.warmwritepos + i) % a.warm_size] = s;
  if (a.delay_ring[c] && i + a.delay_size >= a.n) a.delay_ring[c][(a.delaypos + i) % a.delay_size] = s;
  if (a.delayed[c]) {
    // read position (delaypos + size - predelay + i) % size = the slot block sample j = i - predelay (mod size)
    // was written to, if that sample exists in this block (the last such j, as above)
    long long j = (i - a.predelay) % a.delay_size;
    if (j < 0) j += a.delay_size;
    float v;
    if (j < a.n) {
      j += ((a.n - 1 - j) / a.delay_size) * a.delay_size;
      v = send_at(j);
    } else {
      v = a.delay_ring[c][(a.delaypos + a.delay_size - a.predelay % a.delay_size + i) % a.delay_size];
    }
    a.delayed[c][i] = v;
  }
}

}  // namespace rvc

extern "C" int rvc_send_pre_device(int device, void *stream, const rvc_send_params *p) {
  if (!p || !p->in[0] || !p->in[1]) return 0;
  if ((p->send[0] == nullptr) != (p->send[1] == nullptr)) return 0;
  if ((p->delayed[0] == nullptr) != (p->delayed[1] == nullptr)) return 0;
  if ((p->delay_ring[0] == nullptr) != (p->delay_ring[1] == nullptr)) return 0;
  if ((p->warm_ring[0] == nullptr) != (p->warm_ring[1] == nullptr)) return 0;
  if (p->delayed[0] && (!p->delay_ring[0] || p->delay_size <= 0 || p->predelay < 0 || p->delaypos < 0)) return 0;
  if (p->delay_ring[0] && (p->delay_size <= 0 || p->delaypos < 0 || p->delaypos >= p->delay_size)) return 0;
  if (p->warm_ring[0] && (p->warm_size <= 0 || p->warmwritepos < 0 || p->warmwritepos >= p->warm_size)) return 0;
  if (p->n == 0) return 1;
  if (hipSetDevice(device < 0 ? 0 : device) != hipSuccess) return 0;
  rvc::SendArgs a{};
  for (int c = 0; c < 2; ++c) {
    a.in[c] = p->in[c]; a.send[c] = p->send[c]; a.delay_ring[c] = p->delay_ring[c];
    a.delayed[c] = p->delayed[c]; a.warm_ring[c] = p->warm_ring[c];
  }
  a.ysend = p->ysend;
  a.delay_size = p->delay_size; a.delaypos = p->delaypos; a.predelay = p->predelay;
  a.warm_size = p->warm_size; a.warmwritepos = p->warmwritepos; a.n = (long long)p->n;
  hipLaunchKernelGGL(rvc::k_send_pre, dim3((unsigned)((p->n + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 1 : 0;
}
