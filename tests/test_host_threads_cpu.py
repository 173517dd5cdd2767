"""Several handles on several host threads (include/film_hip.h: "Handles are not thread-safe; several handles may coexist"): every
thread owns ONE plan-only handle and packs weights, plans, exports plans / layouts / tune tables concurrently with the others - what the
threads share is process-global state of the library (the multi-threaded weight packer's work queue, static tables, the launchers'
per-device attribute flags), and that is what this test is for: run it on the ThreadSanitizer build of the host side,

    tools/sanitize/build_host.sh thread
    FILM_NO_TORCH=1 FILM_HIP_LIB=/tmp/film_san_thread/libfilm_hip_thread.so TSAN_OPTIONS=halt_on_error=1 \
        LD_PRELOAD="$(g++ -print-file-name=libtsan.so) $(g++ -print-file-name=libstdc++.so.6)" python -m pytest tests/test_host_threads_cpu.py -q

(round-6 record: profiles/r06_tsan_host_threads.log).  Without the sanitizer it still checks that concurrent handles give the results of
sequential ones."""
import json
import threading

import numpy as np


def _work(opt, weights, shapes, out, idx, barrier):
    from film_hip.engine import FilmEngine
    try:
        eng = FilmEngine(opt, device=-1)
        barrier.wait(timeout=60)                 # start the packers / planners of all threads together
        eng.set_weights(weights)                 # film_set_weight x N + film_finalize: the multi-threaded packer
        plans = [json.dumps(eng.plan(*s), sort_keys=True) for s in shapes]
        eng.set_option('fold2x2', 2)             # drops the plans, re-plans with another family
        plans += [json.dumps(eng.plan(*s), sort_keys=True) for s in shapes[:1]]
        eng.set_option('fold2x2', 1)
        out[idx] = (plans, eng.export_packed().copy(), eng.export_layouts().copy(), eng.export_tune())
        eng.close()
    except BaseException as e:   # noqa: BLE001 - reported by the main thread
        out[idx] = e


def test_handles_on_concurrent_threads_match_sequential_ones(tiny_weights):
    from film_hip.options import TINY
    shapes = [(1, 32, 32), (2, 64, 48), (1, 128, 64)]
    nthreads = 4
    ref = [None]
    _work(TINY, tiny_weights, shapes, ref, 0, threading.Barrier(1))
    assert not isinstance(ref[0], BaseException), ref[0]
    out = [None] * nthreads
    barrier = threading.Barrier(nthreads)
    threads = [threading.Thread(target=_work, args=(TINY, tiny_weights, shapes, out, i, barrier)) for i in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for got in out:
        assert not isinstance(got, BaseException), got
        # (workspace offsets, op lists, tile defaults: a plan is a pure function of the handle's options and the shape)
        assert got[0] == ref[0][0]
        assert np.array_equal(got[1], ref[0][1]) and np.array_equal(got[2], ref[0][2]) and got[3] == ref[0][3]
