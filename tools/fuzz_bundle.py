#!/usr/bin/env python
"""Mutation fuzzer for film_load_bundle (csrc/film_bundle.cpp), the native reader of SavedModel variables bundles - files that come
from somewhere else.  Plan-only handles: runs without a GPU.  Test infrastructure, not part of the product.

  python tools/fuzz_bundle.py SEED CASES [VERIFY=1]      mutated copies (flipped bytes, truncations, 0xff runs, insertions; 80 % in the
                                                         index table) of the bundles of tests/bundle_writer.py and tests/tf_like_writer.py;
                                                         every case must end in FILM_OK or a FilmError; prints "done <loaded> <refused>"

Under AddressSanitizer (the host translation units compile with g++; the kernel launchers are never reached from a plan-only handle):

  cd frame-interpolation_amd/csrc && for f in film_bundle film_engine film_layers film_planner; do \\
      g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \\
          -DFILM_SRC_ID='"asan"' -c $f.cpp -o /tmp/asan/$f.o; done
  (stubs for the 12 undefined film_launch_* symbols: `int s(void) __asm__("<mangled name>"); int s(void) { abort(); }` each)
  g++ -shared -fPIC -fsanitize=address -o /tmp/asan/libfilm_hip_asan.so /tmp/asan/*.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
  FILM_NO_TORCH=1 FILM_HIP_LIB=/tmp/asan/libfilm_hip_asan.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \\
      LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" python tools/fuzz_bundle.py 7 150 0
  (libstdc++ in LD_PRELOAD: ASan's __cxa_throw interceptor must find the real one when the process starts; python does not link it)

Round 5: 9 000 cases with the product library + 10 800 under ASan, crc verification on and off: no finding in the native reader;
two UnicodeDecodeErrors in film_hip/engine.py (checkpoint keys are bytes of the file) - fixed, tests/test_tf_bundle_cpu.py keeps 240 cases."""
import os
import random
import shutil
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'frame-interpolation_amd'), os.path.join(R, 'tests')]


def mutate(rng, b):
    mode = rng.random()
    if mode < 0.5:
        for _ in range(rng.choice([1, 1, 2, 4, 8])):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif mode < 0.7:
        b = b[:rng.randrange(len(b))]
    elif mode < 0.85:
        p = rng.randrange(len(b))
        b[p:p + rng.choice([1, 4, 8])] = bytes([0xff] * rng.choice([1, 4, 8]))
    else:
        p = rng.randrange(len(b))
        b[p:p] = bytes(rng.randrange(256) for _ in range(rng.choice([1, 3, 16])))
    return b


def main():
    from film_hip import options as O, tf_bundle as tb, weights as W
    from film_hip.engine import FilmEngine, FilmError
    import bundle_writer as bw
    import tf_like_writer as tw
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    verify = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    rng = random.Random(seed)
    work = tempfile.mkdtemp(prefix='film_fuzz_')
    wts = W.make_synthetic_weights(O.TINY, seed=0)
    bw.save_film_bundle(os.path.join(work, 'a'), wts, O.TINY)
    full = {tb.checkpoint_key(k, O.TINY)[:-len(tb.VAR_SUFFIX)]: k for k in wts}
    paths = {tb.checkpoint_key(k, O.TINY)[:-len(tb.VAR_SUFFIX)]: v for k, v in wts.items()}
    tw.write_tf_like_bundle(os.path.join(work, 'b', 'variables', 'variables'), paths, full_names=full, num_shards=2, block_size=384)
    loaded = refused = 0
    for i in range(n):
        dst = os.path.join(work, 'case')
        shutil.rmtree(dst, ignore_errors=True)
        shutil.copytree(os.path.join(work, 'a' if rng.random() < 0.5 else 'b'), dst)
        files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(dst) for f in fs)
        idx = [f for f in files if f.endswith('.index')]
        f = rng.choice(idx) if rng.random() < 0.8 else rng.choice(files)
        data = mutate(rng, bytearray(open(f, 'rb').read()))
        with open(f, 'wb') as fh:
            fh.write(bytes(data))
        sys.stdout.write(f'case {seed}:{i} {os.path.relpath(f, dst)}\n')   # (the last line names the case if the process dies)
        sys.stdout.flush()
        eng = FilmEngine(O.TINY, device=-1)
        try:
            eng.load_bundle(dst, verify=verify)
            loaded += 1
        except FilmError:
            refused += 1
        eng.close()
    shutil.rmtree(work, ignore_errors=True)
    print('done', loaded, refused)


if __name__ == '__main__':
    main()
