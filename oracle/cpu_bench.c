/*
 * cpu_bench.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 *
 * The CPU path timed the way the plug-in drives it (src/PluginProcessor.cpp:1793-1797): one
 * convolver instance per thread, `block`-frame process() calls back to back, tail inline. No Python
 * in the loop: the convolver entry points are passed in as plain C function pointers, so the same
 * driver times either the untouched reference (oracle/_ref, ref_twostage_*) or this repo's C
 * restatement (orc_twostage_*). SURVEY.md 8d "CPU baseline, same run": (i) one thread,
 * (ii) all host cores, one instance (channel) per thread.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void *(*cb_create_fn)(void);
typedef void (*cb_destroy_fn)(void *);
typedef int (*cb_init_fn)(void *, size_t, size_t, const float *, size_t);
typedef int (*cb_init1_fn)(void *, size_t, const float *, size_t);   /* FFTConvolver::init: one block size (tail == 0) */
typedef void (*cb_process_fn)(void *, const float *, float *, size_t);

typedef struct {
  cb_create_fn create;
  cb_destroy_fn destroy;
  cb_init_fn init;
  cb_process_fn process;
  size_t head, tail, block, ir_len, frames;
  const float *ir, *in;
  double seconds;
  pthread_barrier_t *bar;
  /* results */
  unsigned long long samples;
  double elapsed;
  int ok;
} cb_thread;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *cb_worker(void *arg) {
  cb_thread *t = (cb_thread *)arg;
  void *conv = t->create();
  float *out = (float *)malloc(sizeof(float) * t->block);
  t->ok = conv && out && (t->tail ? t->init(conv, t->head, t->tail, t->ir, t->ir_len)
                                  : ((cb_init1_fn)(void (*)(void))t->init)(conv, t->head, t->ir, t->ir_len));
  /* warm-up: one pass over a tail period so that every buffer is touched */
  if (t->ok)
    for (size_t i = 0; i + t->block <= t->frames && i < 2 * (t->tail ? t->tail : 8192); i += t->block) t->process(conv, t->in + i, out, t->block);
  pthread_barrier_wait(t->bar);
  const double t0 = now_s();
  unsigned long long done = 0;
  if (t->ok) {
    for (;;) {
      /* one pass = the whole input in block-sized calls; the clock is read every 64 calls */
      size_t i = 0;
      int stop = 0;
      while (i + t->block <= t->frames) {
        for (int k = 0; k < 64 && i + t->block <= t->frames; ++k, i += t->block) t->process(conv, t->in + i, out, t->block);
        if (now_s() - t0 >= t->seconds) { stop = 1; break; }
      }
      done += i;
      if (stop) break;
    }
  }
  t->elapsed = now_s() - t0;
  t->samples = done;
  free(out);
  if (conv) t->destroy(conv);
  return NULL;
}

/* Runs n_threads independent convolvers for ~`seconds`. Thread t uses irs[t % n_irs] and
 * ins[t % n_ins]. Returns the wall time between the common start and the last thread's end;
 * samples[t] = frames thread t convolved. 0.0 on failure. */
double orc_cpu_bench(void *create, void *destroy, void *init, void *process, int n_threads, size_t head, size_t tail,
                     size_t block, const float *const *irs, int n_irs, size_t ir_len, const float *const *ins, int n_ins,
                     size_t frames, double seconds, unsigned long long *samples) {
  if (n_threads < 1 || n_irs < 1 || n_ins < 1 || block == 0 || frames < block) return 0.0;
  cb_thread *th = (cb_thread *)calloc((size_t)n_threads, sizeof(cb_thread));
  pthread_t *ids = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  pthread_barrier_t bar;
  if (!th || !ids || pthread_barrier_init(&bar, NULL, (unsigned)n_threads + 1) != 0) { free(th); free(ids); return 0.0; }
  for (int t = 0; t < n_threads; ++t) {
    th[t].create = (cb_create_fn)create; th[t].destroy = (cb_destroy_fn)destroy;
    th[t].init = (cb_init_fn)init; th[t].process = (cb_process_fn)process;
    th[t].head = head; th[t].tail = tail; th[t].block = block; th[t].ir_len = ir_len; th[t].frames = frames;
    th[t].ir = irs[t % n_irs]; th[t].in = ins[t % n_ins];
    th[t].seconds = seconds; th[t].bar = &bar;
    pthread_create(&ids[t], NULL, cb_worker, &th[t]);
  }
  pthread_barrier_wait(&bar);          /* every instance is initialised and warm */
  const double t0 = now_s();
  int ok = 1;
  for (int t = 0; t < n_threads; ++t) {
    pthread_join(ids[t], NULL);
    ok = ok && th[t].ok;
    if (samples) samples[t] = th[t].samples;
  }
  const double wall = now_s() - t0;
  pthread_barrier_destroy(&bar);
  free(th); free(ids);
  return ok ? wall : 0.0;
}
