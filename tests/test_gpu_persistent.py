"""RVC_FLAG_PERSISTENT: the plug-in's per-block calls served by ONE resident kernel fed through a doorbell in pinned
host memory (reference loop: TwoStageFFTConvolver.cpp:151-233 with len <= head block). Parity against the oracle for
the block-synchronous pattern and for every way of leaving and re-entering it; lifecycle: clear(), re-init (IR swap and
new geometry), parking after idle and relaunch, destroy while resident, many create / destroy rounds.
Every test is bounded by pytest-timeout: a protocol bug shows as a hang, never as a stuck GPU box."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import reevr_amd  # noqa: E402
from reevr_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(120)]
TOL = 1e-5


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def oracle(irs, x, head, tail, clear_at=0):
    out = []
    for c in range(x.shape[0]):
        o = O.TwoStageFFTConvolver("orc")
        assert o.init(head, tail, irs[c])
        y = np.empty(x.shape[1], np.float32)
        if clear_at:
            y[:clear_at] = o.process(x[c, :clear_at])
            o.clear()
        y[clear_at:] = o.process(x[c, clear_at:])
        out.append(y)
    return np.stack(out)


@pytest.mark.parametrize("head,tail,ir_len,nch", [(512, 8192, 60000, 2), (512, 8192, 9000, 1), (1024, 8192, 40000, 3),
                                                  (2048, 8192, 30000, 2), (4096, 8192, 50000, 2), (512, 1024, 7000, 4)])
def test_block_calls_host_and_device(head, tail, ir_len, nch):
    import torch
    nblk = max(40, 5 * tail // head + 24)
    irs = [synth.synth_ir(ir_len - 100 * c, 1, 700 + c)[0] for c in range(nch)]
    x = np.stack([synth.synth_input(head * nblk, 20 + c) for c in range(nch)])
    want = oracle(irs, x, head, tail)
    s = reevr_amd.ConvolverSet(nch, persistent=True)
    assert s.init(head, tail, irs, max_len=head), s.last_error_string
    got = np.concatenate([s.process(x[:, i * head:(i + 1) * head]) for i in range(nblk)], axis=1)
    assert s.last_error == 0, s.last_error_string
    for c in range(nch):
        assert rel_rms(got[c], want[c]) <= TOL, c
    s.clear()                                   # resident kernel stays; the clock restarts
    dx = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    y = s.process_device_blocks(dx, head).cpu().numpy()      # commands queued ahead of the device
    assert s.last_error == 0, s.last_error_string
    for c in range(nch):
        assert rel_rms(y[c], want[c]) <= TOL, c
    s.close()


@pytest.mark.parametrize("seed", list(range(12)))
def test_fuzz_call_patterns_persistent(seed):
    """Block calls through the resident kernel mixed with everything that leaves it: several calls inside one block,
    ragged calls across blocks, multi-block and long calls (ordinary launches), block-aligned clear()."""
    rng = np.random.RandomState(9100 + seed)
    head = int(rng.choice([512, 1024, 2048]))
    tail = int(rng.choice([4 * head, 8192, 16 * head]))
    tail = max(tail, 2 * head)
    nch = int(rng.randint(1, 4))
    base = int(2 * tail + rng.choice([1, 4, 12]) * tail - rng.randint(0, tail // 2))
    irs = [synth.synth_ir(max(1, base - c * 333), 1, 800 + 3 * seed + c)[0] for c in range(nch)]
    total = int(min(50 * tail, 400000))
    total -= total % head
    sched, done = [], 0
    while done < total:
        r = rng.randint(0, 30)
        if r == 0:
            n = int(rng.randint(1, head))
        elif r == 1 and done % head:
            n = head - done % head
        elif r == 2:
            n = int(rng.randint(2, 6)) * head
        elif r == 3:
            n = int(rng.randint(5, 8)) * tail
        else:
            n = head if done % head == 0 else head - done % head
        n = max(1, min(n, total - done))
        sched.append(n)
        done += n
    x = np.stack([synth.synth_input(total, 13 * seed + c) for c in range(nch)])
    s = reevr_amd.ConvolverSet(nch, persistent=True)
    assert s.init(head, tail, irs, max_len=max(sched)), s.last_error_string
    clear_at = int(rng.randint(len(sched) // 4, len(sched))) if rng.randint(0, 3) == 0 else -1
    got = np.empty_like(x)
    pos = start = 0
    for i, n in enumerate(sched):
        if i >= clear_at >= 0 and pos % head == 0 and start == 0:
            s.clear()
            start = pos
        got[:, pos:pos + n] = s.process(x[:, pos:pos + n])
        pos += n
    assert s.last_error == 0, s.last_error_string
    want = oracle(irs, x, head, tail, clear_at=start)
    for c in range(nch):
        assert rel_rms(got[c], want[c]) <= TOL, f"seed {seed} head {head} tail {tail} nch {nch} clear@{start} ch {c}"
    s.close()


def test_reinit_while_resident_and_parking():
    head, tail, nblk = 512, 8192, 60
    x = np.stack([synth.synth_input(head * nblk, c) for c in range(2)])
    os.environ["RVC_PERSIST_IDLE_MS"] = "40"          # park after 40 ms without a call
    try:
        s = reevr_amd.ConvolverSet(2, persistent=True)
        for rnd, (ir_len, hb) in enumerate([(30000, 512), (30000, 512), (50000, 512), (20000, 1024)]):
            irs = list(synth.synth_ir(ir_len, 2, 40 + rnd))      # same geometry twice (IR swap), then new ones
            assert s.init(hb, tail, irs, max_len=hb), s.last_error_string
            out = []
            for i in range(nblk * head // hb):
                out.append(s.process(x[:, i * hb:(i + 1) * hb]))
                if i % 17 == 16:
                    time.sleep(0.12)                              # the kernel parks itself; the next call relaunches it
            got = np.concatenate(out, axis=1)
            want = oracle(irs, x[:, :got.shape[1]], hb, tail)
            for c in range(2):
                assert rel_rms(got[c], want[c]) <= TOL, (rnd, c)
            assert s.last_error == 0, s.last_error_string
        s.reset()                                                 # reset() while resident, then use again
        irs = list(synth.synth_ir(9000, 2, 77))
        assert s.init(head, tail, irs, max_len=head)
        got = np.concatenate([s.process(x[:, i * head:(i + 1) * head]) for i in range(20)], axis=1)
        want = oracle(irs, x[:, :got.shape[1]], head, tail)
        assert rel_rms(got[0], want[0]) <= TOL
        s.close()
    finally:
        del os.environ["RVC_PERSIST_IDLE_MS"]


def test_create_destroy_rounds_and_slot_limit():
    """Destroy while resident; many rounds (leak / stuck-kernel check); more persistent sets than resident slots:
    the surplus ones run on ordinary launches with the same results."""
    head, tail = 512, 8192
    irs = list(synth.synth_ir(20000, 2, 5))
    x = np.stack([synth.synth_input(head * 12, c) for c in range(2)])
    want = oracle(irs, x, head, tail)
    for _ in range(25):
        s = reevr_amd.ConvolverSet(2, persistent=True)
        assert s.init(head, tail, irs, max_len=head)
        got = np.concatenate([s.process(x[:, i * head:(i + 1) * head]) for i in range(12)], axis=1)
        assert rel_rms(got[1], want[1]) <= TOL
        s.close()                                                 # resident kernel told to quit
    sets = [reevr_amd.ConvolverSet(2, persistent=True) for _ in range(4)]
    for s in sets:
        assert s.init(head, tail, irs, max_len=head)
    outs = [[] for _ in sets]
    for i in range(12):
        for j, s in enumerate(sets):
            outs[j].append(s.process(x[:, i * head:(i + 1) * head]))
    for j, s in enumerate(sets):
        assert rel_rms(np.concatenate(outs[j], axis=1)[0], want[0]) <= TOL, j
        assert s.last_error == 0
        s.close()


def test_other_threads_free_device_memory_while_resident():
    """hipFree waits for every stream of the device -- also for a resident kernel that is being fed. While the audio
    thread streams blocks through a persistent set, another thread creates / re-initialises / destroys sets and impulse
    objects (the plug-in's IR hot-swap with a new geometry): the resident kernel stands down for the frees (FreeGuard),
    the other thread is never held for long, and the audio stays correct."""
    import threading
    head, tail, nblk = 512, 8192, 3000
    irs = list(synth.synth_ir(30000, 2, 9))
    x = np.stack([synth.synth_input(head * nblk, c) for c in range(2)])
    want = oracle(irs, x, head, tail)
    s = reevr_amd.ConvolverSet(2, persistent=True)
    assert s.init(head, tail, irs, max_len=head)
    got = []
    stop = threading.Event()
    worst = [0.0]
    rounds = [0]
    errors = []

    def worker():
        try:
            k = 0
            while not stop.is_set():
                t0 = time.perf_counter()
                b = reevr_amd.ConvolverSet(2)
                assert b.init(256 << (k % 3), 8192, list(synth.synth_ir(5000 + 3000 * (k % 4), 2, 50 + k)), max_len=4096)
                b.process(x[:, :3000])
                b.close()                                   # hipFree while the other set's kernel is resident
                imp = reevr_amd.Impulse()
                imp.prepare(48000.0)
                imp.setRaw(0.5 * x[0, :4000 + 1000 * (k % 3)], 0.5 * x[1, :4000 + 1000 * (k % 3)])
                imp.recalcImpulse()
                del imp
                if not stop.is_set():                       # (a round cut short by the end of the stream does not count)
                    worst[0] = max(worst[0], time.perf_counter() - t0)
                    rounds[0] += 1
                k += 1
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=worker)
    th.start()
    for i in range(nblk):
        got.append(s.process(x[:, i * head:(i + 1) * head]))
        time.sleep(0.0003)                              # a (fast) real-time host: a block every 0.3 ms for ~1 s
    stop.set()
    th.join(timeout=30)
    assert not th.is_alive() and not errors, errors
    got = np.concatenate(got, axis=1)
    assert s.last_error == 0, s.last_error_string
    for c in range(2):
        assert rel_rms(got[c], want[c]) <= TOL, c
    assert rounds[0] >= 3, rounds[0]
    assert worst[0] < 1.5, f"a round of create/init/destroy on the other thread took {worst[0]:.2f} s"
    s.close()


def test_stereo_convolver_shim_persistent_cpp():
    """examples/host_block_loop.cpp with the persistent flag: the C++ drop-in classes through the C ABI, quad (two
    resident kernels), bounded run; prints the call latency it measured."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "host_block_loop")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", ROOT, "example"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "512", "600", "1", "100", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["persistent"] == 1 and rec["channels"] == 4 and abs(rec["checksum"]) > 0
    r0 = subprocess.run([exe, "512", "600", "1", "100", "0"], capture_output=True, text=True, timeout=60)
    rec0 = json.loads(r0.stdout.strip().splitlines()[-1])
    assert abs(rec["checksum"] - rec0["checksum"]) <= 1e-3 * max(1.0, abs(rec0["checksum"]))    # same audio either way


@pytest.mark.parametrize("head,tail", [(512, 1024), (512, 512), (4096, 8192)])
def test_fresh_set_pipelined_device_blocks_tail_hand_off(head, tail):
    """Pipelined device-pointer block calls on a FRESH persistent set (no host pass before, every buffer NaN-poisoned by
    the guard mode) with tail <= 2 x head: the step that completes a tail block can still be in flight when its output is
    first needed; the fallback that then produces the tail block on the second stream must be waited for before the
    resident kernel adds it (a stale or half-written tail ring shows as NaN / a parity miss here)."""
    import torch
    nch = 2
    irs = [synth.synth_ir(2 * tail + 3 * tail - 50 * c, 1, 760 + c)[0] for c in range(nch)]
    nblk = max(48, 12 * tail // head)
    x = np.stack([synth.synth_input(head * nblk, 33 + c) for c in range(nch)])
    want = oracle(irs, x, head, tail)
    for rep in range(2):
        reevr_amd.set_tuning("guard", 1)
        try:
            s = reevr_amd.ConvolverSet(nch, persistent=True, fft_f32=True)
            assert s.init(head, tail, irs, max_len=head), s.last_error_string
        finally:
            reevr_amd.set_tuning("guard", 0)
        got = s.process_device_blocks(torch.from_numpy(x).cuda(), head).cpu().numpy()
        assert s.last_error == 0, s.last_error_string
        assert s.guard_check() == 0
        s.close()
        assert np.isfinite(got).all()
        for c in range(nch):
            assert rel_rms(got[c], want[c]) <= TOL, (rep, c)


def test_freeing_while_another_sets_kernel_is_parked_does_not_stall():
    """A resident kernel that has PARKED itself (idle owner) must not make other threads' frees wait out the guard's
    timeout: destroying an unrelated set and an impulse object takes milliseconds, not seconds."""
    head, tail = 512, 8192
    irs = [synth.synth_ir(30000, 1, 5)[0]]
    os.environ["RVC_PERSIST_IDLE_MS"] = "50"
    try:
        s = reevr_amd.ConvolverSet(1, persistent=True)
        assert s.init(head, tail, irs, max_len=head)
        x = synth.synth_input(head * 4, 0)[None, :]
        for i in range(4):
            s.process(x[:, i * head:(i + 1) * head])
        time.sleep(0.5)                               # the resident kernel parks itself; its owner does not call again
        t0 = time.perf_counter()
        for _ in range(3):
            o = reevr_amd.ConvolverSet(2)
            assert o.init(64, 256, [irs[0][:2000], irs[0][:1500]])
            o.close()
        dt = time.perf_counter() - t0
        assert dt < 2.0, f"three create/init/destroy rounds took {dt:.2f} s beside a parked kernel"
        y = s.process(x[:, :head])                    # the owner comes back: relaunch, still correct
        assert s.last_error == 0 and np.isfinite(y).all()
        s.close()
    finally:
        os.environ.pop("RVC_PERSIST_IDLE_MS", None)
