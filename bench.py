#!/usr/bin/env python
"""Benchmark of the FILM hot path on MI355X:  interpolated 1080p frames / second.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one mid-frame of a 1920x1080 frame pair through the tiled path of the reference CLI
(`--align 64 --block_height 2 --block_width 2`, eval/interpolator_test.py:57-70 -> four 960x540
patches, each padded to 960x576, eval/interpolator.py:192-206) - BASELINE.json configs[2], the
configuration the metric is quoted on.  Frames are resident in HBM before the timed region; pad /
patch / crop / stitch on the device are inside it.  Each rank (= one GPU) interpolates its own
frame pairs: no collective on the data path, weights are broadcast once over RCCL before timing
("scaling": "weak").

Extra objects in the JSON line:
  roofline     fp32-MFMA roofline of the dominant kernel class (the MFMA convolutions): FLOPs the matrix
               pipe EXECUTES in one forward (the Winograd F(4,3) / F(2,3) kernels and the sub-pixel fold
               execute 1/2, 2/3, 9/16 of the direct convolution's multiplies) / summed duration of its
               launches, measured with hipEvents around every launch on the launch stream (engine
               profile mode), vs 157.3 TFLOP/s -> `achieved`, `frac` (a utilisation, <= 1).
               `direct_equivalent` prices the same time with SURVEY 8d's direct-convolution FLOP count
               (can exceed the peak: that is the algorithmic saving, not utilisation).
  cpu_baseline the CPU oracle (PyTorch-CPU/oneDNN + numpy restatement of the TF graph, kind "port")
               timed on this host's cores on ONE real 960x576 tile of the workload (a frame is four such
               tiles run one after the other, as the reference's tile loop does): frames/s = 1 / (4 t).

Workloads (`--workload`): the default `1080p_2x2` is BASELINE configs[2], the configuration the metric is quoted on.
`4k_4x4_T6` is configs[4]: one 3840x2160 pair, 4x4 tiles, `--times_to_interpolate 6` - a step is the WHOLE recursion
(63 generated frames x 16 tiles = 1008 tile-forwards of 960x576, breadth first on the device); `value` = generated
frames / s and `ms_per_depth` lists the six depths.  `--scaling strong` (tiled workloads) shards the TILES of the one
pair over the ranks for the whole recursion tree (film_hip.sharding.TileShardedRecursion: no collective until the
final gather of the generated tiles to rank 0, which is inside the timed region): total work fixed, "scaling": "strong".

`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N
ranks (one per GPU, 127.0.0.1 rendezvous) and refuses to run when fewer than N GPUs are visible; `n_gpus` in
the line is the number of ranks that actually reported.  `--plan-only` drives the same launcher / broadcast /
sharding / timing-reduction code on CPU (gloo, plan-only engine handles, no compute) for the CPU tests.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'frame-interpolation_amd')
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONV_FLOP_PER_PIXEL = 4246240.6875  # sum over the 171 convs, per padded input pixel
WARP_BYTES_PER_PIXEL = 5201.58      # 22 warps, read once + flow + write once
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA; one fp32 product costs 6 (bf16x6) / 3 (bf16x3) bf16 products
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (H, W, align, block_shape, padded tile (h, w), tiles)
    '1080p_2x2': (1080, 1920, 64, [2, 2], (576, 960), 4),
    '256': (256, 256, 64, None, (256, 256), 1),
    'vimeo': (256, 448, 64, None, (256, 448), 1),
    'photos': (768, 1024, 64, None, (768, 1024), 1),
    # BASELINE configs[3]: a batch of Vimeo-90K sized pairs per GPU (8 pairs per step per GPU)
    'vimeo_b8': (256, 448, 64, None, (256, 448), 8),
    # BASELINE configs[4]: 3840x2160 pair, 4x4 tiles of 960x540 (padded to 960x576), times_to_interpolate = 6
    '4k_4x4_T6': (2160, 3840, 64, [4, 4], (576, 960), 16),
    # the same recursion at a depth that fits a quick run (tests, smoke runs of the recursion driver)
    '4k_4x4_T2': (2160, 3840, 64, [4, 4], (576, 960), 16),
    '1080p_2x2_T3': (1080, 1920, 64, [2, 2], (576, 960), 4),
}
RECURSIONS = {'4k_4x4_T6': 6, '4k_4x4_T2': 2, '1080p_2x2_T3': 3}   # times_to_interpolate (1 everywhere else)
METRIC = {'4k_4x4_T6': 'interpolated frames/sec @4K (4x4 tiles, times_to_interpolate 6)',
          '4k_4x4_T2': 'interpolated frames/sec @4K (4x4 tiles, times_to_interpolate 2)'}


# Parity of the EXACT timed callable (same DeviceInterpolator, graph, lanes) against the committed reference-graph golden vectors
# (tests/golden/ref_*.npz: the reference's own create_model / Interpolator code executed by tools/make_ref_golden.py; tf_*.npz from a
# real TensorFlow install are preferred when present).  workload -> (golden case, batch, H, W, frame_pair kwargs): the inputs the
# goldens were made with (tests/inputs.py; their checksums are stored in the file and re-checked here).
PARITY_CASES = {
    '1080p_2x2': ('1080p', 1, 1080, 1920, dict(seed=2, shift=(11, -17), fg_shift=(-9, 21))),
    'vimeo_b8': ('vimeo', 8, 256, 448, dict(seed=3)),
}
PARITY_TOL = 1e-3   # north_star: |delta| < 1e-3 per pixel, fp32


def _tests_on_path():
    if os.path.join(ROOT, 'tests') not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))


def parity_inputs(workload):
    _tests_on_path()
    import golden_util as G
    import inputs as TI
    case, b, h, w, kw = PARITY_CASES[workload]
    g, prov = G.load(case)
    x0, x1 = TI.frame_pair(b, h, w, **kw)
    G.check_inputs(g, x0, x1)
    return g, prov, case, x0, x1


def parity_check(workload, fn, dev):
    """Runs the golden's frame pair through `fn` (the callable the timed loop calls) and returns (output tensor, report dict)."""
    _tests_on_path()
    import golden_util as G
    g, prov, case, x0, x1 = parity_inputs(workload)
    got = fn(torch.from_numpy(x0).to(dev), torch.from_numpy(x1).to(dev))
    torch.cuda.synchronize()
    d = G.diff(g, 'image', got.cpu().numpy())
    return got.clone(), {'max_abs_vs_ref_graph_golden' if prov != 'tf' else 'max_abs_vs_tensorflow_golden': d,
                         'golden': f'tests/golden/{"tf" if prov == "tf" else "ref"}_{case}.npz', 'tolerance': PARITY_TOL,
                         'through': 'the timed callable itself (DeviceInterpolator, same engine handle / executor / lanes), device-resident frames'}


def synth_pair(h, w, seed):
    """smooth random image (uniform noise, 9x9 box filter), second frame = first
    shifted by (3,-2) px + sigma 0.01 noise, so that the flows are non-trivial."""
    rng = np.random.default_rng(seed)
    x = rng.random((h + 8, w + 8, 3), dtype=np.float32)
    c = np.cumsum(np.cumsum(np.pad(x, ((1, 0), (1, 0), (0, 0))), axis=0, dtype=np.float64), axis=1)
    k = 9
    box = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
    x0 = box[:h, :w].astype(np.float32)
    x0 = (x0 - x0.min()) / max(float(x0.max() - x0.min()), 1e-6)
    x1 = np.roll(x0, (3, -2), axis=(0, 1)) + rng.normal(0, 0.01, x0.shape).astype(np.float32)
    return x0[None], x1[None].astype(np.float32)


def cli_files_workload(eng, args):
    """`--workload cli_1080p_T3` (SURVEY 8(d) "end to end", reference eval/interpolator_cli.py:152-177): two synthetic 1080p PNGs in a
    temporary directory -> `--times_to_interpolate 3 --align 64 --block_height 2 --block_width 2` -> nine PNGs, files to files,
    decode / H2D / recursion / quantise / D2H / encode all inside the timed region.  Timed twice: the device pipeline the CLI uses
    (eval.util.interpolate_pairs_to_files) and the round-3 path (float32 frames back in one blocking copy, host to_uint8, PNGs one
    after the other on one thread); the files of the two runs are compared byte for byte."""
    import filecmp
    import shutil
    import tempfile
    from eval import interpolator as interpolator_lib
    from eval import interpolator_cli as cli
    from eval import util
    T, H, Wd = 3, 1080, 1920
    tmp = tempfile.mkdtemp(prefix='film_cli_bench_')
    try:
        x0, x1 = synth_pair(H, Wd, 5)
        for name, x in (('in_000.png', x0[0]), ('in_001.png', x1[0])):
            util.write_image(os.path.join(tmp, name), np.clip(x, 0, 1))
        it = interpolator_lib.Interpolator('', 64, [2, 2], engine=eng)
        inputs = cli.list_input_frames(tmp)
        new_dir, old_dir = os.path.join(tmp, 'new'), os.path.join(tmp, 'old')
        ms_new, ms_old = [], []
        for rep in range(args.warmup + args.steps):
            cli.output_frames([], new_dir)
            t0 = time.perf_counter()
            n_new, _ = util.interpolate_pairs_to_files(inputs, 0, 1, 1, T, it, new_dir)
            if rep >= args.warmup:
                ms_new.append((time.perf_counter() - t0) * 1e3)
        for rep in range(1 + min(args.steps, 3)):
            t0 = time.perf_counter()
            frames = list(util.interpolate_recursively_from_files(inputs, T, it))
            cli.output_frames(frames, old_dir)
            if rep >= 1:
                ms_old.append((time.perf_counter() - t0) * 1e3)
        same = all(filecmp.cmp(os.path.join(new_dir, f), os.path.join(old_dir, f), shallow=False) for f in sorted(os.listdir(old_dir)))
        n_files = len(os.listdir(new_dir))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    gen = 2 ** T - 1
    ms = float(np.median(ms_new))
    return {
        'metric': 'interpolated frames/sec @1080p, files to files (interpolator_cli, times_to_interpolate 3)', 'value': round(gen / (ms * 1e-3), 4),
        'unit': 'frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'cli_1080p_T3: two 1920x1080 PNGs -> align 64, block_shape [2, 2], times_to_interpolate 3 -> 9 PNGs (7 generated); '
                               'PNG decode, H2D, 7 tiled forwards, device to_uint8, D2H, PNG encode all timed',
                   'files_written': n_files, 'frames_generated': gen, 'host_threads': os.cpu_count()},
        'round3_path': {'ms_per_step': round(float(np.median(ms_old)), 3), 'frames_per_s': round(gen / (float(np.median(ms_old)) * 1e-3), 4),
                        'what': 'float32 frames back in one blocking copy, host to_uint8, serial PIL PNG encode'},
        'speedup_vs_round3_path': round(float(np.median(ms_old)) / ms, 3),
        'files_byte_identical_to_round3_path': bool(same),
    }


def cpu_baseline(weights):
    """Times the oracle on the host cores on ONE real tile of the headline workload: a 960x576 pair of the
    published net (2.348 TFLOP of convolutions + the 22 gather warps), no scaling by FLOP ratios.  The
    reference's tile loop runs the four tiles of a 1080p frame one after the other (eval/interpolator.py:
    199-202), so a frame costs four tile times."""
    from oracle import film_oracle as fo
    ncores = os.cpu_count() or 1
    # oneDNN with one thread per hardware thread thrashes on many-core hosts; pick the best of a few
    # thread counts on one mid-size layer of the tile (level 2: 240x144, 256 -> 256 channels).
    probe_x = torch.randn(1, 256, 144, 240)
    probe_w = torch.randn(256, 256, 3, 3)
    best_t, best_dt = 1, float('inf')
    for nt in sorted({min(ncores, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        d = time.perf_counter() - t0
        if d < best_dt:
            best_t, best_dt = nt, d
    ncores = best_t
    torch.set_num_threads(ncores)
    xs, _ = synth_pair(64, 64, 1)
    fo.film_forward(xs, xs, weights, fo.Options())    # warm-up of thread pools / primitive caches (64x64)
    x0, x1 = synth_pair(576, 960, 2)
    reps, total = 0, 0.0
    while total < 10.0 and reps < 8:                  # bounded sample: one tile is about 10-20 s of CPU work
        t0 = time.perf_counter()
        fo.film_forward(x0, x1, weights, fo.Options())
        total += time.perf_counter() - t0
        reps += 1
    dt = total / reps
    return {
        'value': round(1.0 / (4 * dt), 6), 'unit': 'frames/s (1080p 2x2-tiled: four tiles per frame, one after the other)',
        'cores': ncores, 'kind': 'port',
        'kind_detail': 'PyTorch-CPU (oneDNN) + numpy restatement of the TF graph (oracle/film_oracle.py) - NOT the TF2 reference, '
                       'which is not installable here',
        'sample': f'{reps} x one 960x576 tile pair (one of the four tiles of a 1080p 2x2-tiled frame), published film_net, '
                  f'{dt:.2f} s per tile ({total:.1f} s in all) on {ncores} threads (PyTorch-CPU oneDNN convs + numpy '
                  f'warp/resize restatement, oracle/film_oracle.py); frame time = 4 tile times, nothing scaled; '
                  f'the TF2 reference itself is not installable here',
        'seconds_per_tile': round(dt, 3), 'seconds': round(total, 3),
    }


def autotune_once(eng, dist, rank, warm, sync=None):
    """ONE autotune for the job: rank 0 builds + measures its plans with `warm()` while the others wait, then every rank imports
    rank 0's tile choices (text, film_export_tune) - eight ranks would otherwise each time every candidate, with 8 x 32
    weight-packing threads on the same host cores, and could end up with different tiles (same results, different speed).
    `warm` MUST NOT enter a collective: only rank 0 calls it (round-4 ADVICE: the strong-scaling step does, and deadlocked here)."""
    from film_hip.sharding import share_tune
    if rank == 0:
        warm()
        if sync is not None:
            sync()
    share_tune(eng, dist, src=0)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: one rank per GPU via
    torch.distributed.run on 127.0.0.1.  Refuses (exit code 2) when fewer than N GPUs are visible."""
    import subprocess
    if not args.plan_only:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f'bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible on this '
                             f'node; refusing to print a {have}-GPU number as an {args.gpus}-GPU one.\n')
            raise SystemExit(2)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def plan_only_run(args, world, rank):
    """CPU rehearsal of the N-rank path (tests/test_dist_cpu.py): gloo rendezvous, rank 0 packs the weights,
    broadcast of the weight set, every rank builds the plan of its shard of `--pairs` frame pairs, barrier +
    max-over-ranks timing, rank 0 prints the line.  No compute: `value` is null and `plan_only` true."""
    import torch.distributed as dist
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED, TINY
    from film_hip.sharding import broadcast_weights, shard_range
    opt = TINY if args.tiny_net else PUBLISHED
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    eng = FilmEngine(opt, device=-1)
    if rank == 0:
        eng.set_weights(W.make_synthetic_weights(opt, seed=0))
    if world > 1:
        broadcast_weights(eng, dist, src=0)
    H, Wd, align, block, tile_hw, ntiles = WORKLOADS[args.workload]
    T = RECURSIONS.get(args.workload, 1)
    b, e = shard_range(args.pairs, world, rank)
    stitched_ok = None
    if args.scaling == 'strong':
        # tile-sharded single pair: the driver, the tile ownership and the final gather on gloo, with a stand-in for the
        # model (mean of the two tiles: tile-local like the real one) on a frame 1/8 of the workload's size
        from film_hip.sharding import TileShardedRecursion, tiles_of_rank
        if block is None:
            raise SystemExit('bench.py: --scaling strong needs a tiled workload')
        mean = lambda a, c: (a + c) * 0.5   # noqa: E731
        g = torch.Generator().manual_seed(5)
        f1 = torch.rand((H // 8 // block[0] * block[0], Wd // 8 // block[1] * block[1], 3), generator=g)
        f2 = torch.rand(f1.shape, generator=g)
        drv = TileShardedRecursion(mean, block, dist if world > 1 else None)
        b, e = 0, len(tiles_of_rank(block, world, rank))
    if world > 1:
        # the same once-per-job autotune sequence as the GPU path below (plan-only handles measure nothing: the header line travels);
        # in strong mode rank 0 alone warms up - on its own tiles, WITHOUT entering the gather's collectives
        warm = (lambda: drv.local(f1, f2, T)) if args.scaling == 'strong' else (lambda: eng.plan(max(1, e - b), tile_hw[0], tile_hw[1]))
        autotune_once(eng, dist, rank, warm)
        dist.barrier()
    t0 = time.perf_counter()
    nops = 0
    for _ in range(args.steps):
        if args.scaling == 'strong':
            seq = drv.run(f1, f2, T)
            if e > b:
                nops = len(eng.plan(e - b, tile_hw[0], tile_hw[1])['ops'])
            if rank == 0:
                want = TileShardedRecursion(mean, block, None).run(f1, f2, T)
                stitched_ok = bool(torch.equal(seq, want)) and seq.shape[0] == 2 ** T + 1
        elif e > b:
            nops = len(eng.plan(ntiles if block else (e - b), tile_hw[0], tile_hw[1])['ops'])
    dt = time.perf_counter() - t0
    reported, units = 1, e - b
    digest = float(np.abs(eng.export_packed()[::1013]).sum())
    same = True
    owned, comm_size = [list(range(b, e))], 1
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt, float(units), 1.0], dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, units, reported = float(tmax[0]), int(t[1]), int(t[2])
        d = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(d, torch.tensor([digest], dtype=torch.float64))
        same = all(float(x) == float(d[0]) for x in d)
        owned = [None] * world      # which units (tiles when strong, pairs when weak) every rank took
        dist.all_gather_object(owned, list(drv.tiles) if args.scaling == 'strong' else list(range(*shard_range(args.pairs, world, rank))))
        comm_size = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'interpolated frames/sec @1080p', 'value': None, 'unit': 'frames/s', 'plan_only': True,
                          'n_gpus': reported, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(dt / max(1, args.steps) * 1e3, 3), 'higher_is_better': True,
                          'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': args.workload, 'pairs_sharded': units if args.scaling == 'weak' else None,
                                     'tiles_sharded': units if args.scaling == 'strong' else None,
                                     'stitched_identical_to_one_rank': stitched_ok, 'plan_ops': nops,
                                     'weights_identical_on_all_ranks': same},
                          'ranks': {'communicator_size': comm_size, 'units_by_rank': owned}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='default 10 (2 for 4k_4x4_T6: a step is 1008 tile-forwards)')
    ap.add_argument('--warmup', type=int, default=None, help='default 3 (1 for 4k_4x4_T6)')
    ap.add_argument('--workload', default='1080p_2x2', choices=sorted(WORKLOADS) + ['cli_1080p_T3'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak (default): every rank interpolates its own frame pairs; strong: ONE pair, its tiles sharded over '
                         'the ranks for the whole recursion tree, generated tiles gathered to rank 0 inside the timed region')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='one stream, plan order (option graph = 0)')
    ap.add_argument('--graph', action='store_true', help='hipGraph replay of the two lanes (option graph = 1) instead of the default direct two-lane launches')
    ap.add_argument('--lanes', type=int, default=1, choices=[0, 1, 2],
                    help='0: every op on ONE stream (serialised kernels: what the committed rocprofv3 kernel trace uses, '
                         'so that per-kernel durations are not inflated by overlap); 1: two lanes (side stream for the small / HBM-bound work), decoder behind the flow '
                         'estimator (default); 2: coarse decoder levels on the side stream beside the estimator (measured slower)')
    ap.add_argument('--precision', type=int, default=0, choices=[0, 1, 2],
                    help='engine precision mode of the MAIN measurement: 0 = fp32 MFMA (default, the headline), 1 = bf16x6, 2 = bf16x3')
    ap.add_argument('--flow-scale', type=float, default=1.0,
                    help='experiment: multiply the last 1x1 layer of every flow predictor (kernel and bias) by this (0 = zero flows: every warp '
                         'is the identity, the smoothest possible gather; 1 = the seeded synthetic weights)')
    ap.add_argument('--fuse', type=int, default=None, help='engine option "fuse" (bit mask, default 31)')
    ap.add_argument('--opt', action='append', default=[], metavar='KEY=VALUE', help='extra engine option (film_set_option), repeatable - A/B experiments')
    ap.add_argument('--wino2d', type=int, default=None, choices=[0, 1, 2], help='engine option "wino2d" (nested Winograd kernel); default: the engine default')
    ap.add_argument('--no-split', action='store_true', help='skip the extra bf16x6 / bf16x3 precision-mode measurements')
    ap.add_argument('--profile-out', default='', help='write the per-op profile JSON here')
    ap.add_argument('--plan-only', action='store_true',
                    help='CPU rehearsal of the N-rank path: gloo + plan-only engine handles, no compute (tests)')
    ap.add_argument('--tiny-net', action='store_true', help='with --plan-only: the small test architecture')
    ap.add_argument('--pairs', type=int, default=8, help='with --plan-only: frame pairs sharded over the ranks')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2 if args.workload == '4k_4x4_T6' else 10
    if args.warmup is None:
        args.warmup = 1 if args.workload == '4k_4x4_T6' else 3

    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args)          # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n')
        raise SystemExit(2)
    if args.plan_only:
        return plan_only_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU fallback')
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f'bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible\n')
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev)

    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED
    from film_hip.torch_io import DeviceInterpolator

    eng = FilmEngine(PUBLISHED, device=local_rank)
    weights = None
    if rank == 0:
        weights = W.make_synthetic_weights(PUBLISHED, seed=0)
        if args.flow_scale != 1.0:
            for k in list(weights):
                if k.startswith('predict_flow') and '/conv_4/' in k:
                    weights[k] = (weights[k] * np.float32(args.flow_scale)).astype(np.float32)
        eng.set_weights(weights)
    bcast_ms = None
    if world > 1:
        # one-time RCCL broadcast of the packed weight blob (137.7 MB) from rank 0
        from film_hip.sharding import broadcast_weights
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        broadcast_weights(eng, dist, src=0, device=dev)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb0) * 1e3
    if os.environ.get('FILM_TUNE_MS'):
        eng.set_option('tune_ms', int(os.environ['FILM_TUNE_MS']))
    if args.no_graph:
        eng.set_option('graph', 0)
    elif args.graph:
        eng.set_option('graph', 1)
    if args.lanes != 1:
        eng.set_option('lanes', args.lanes)
    if args.wino2d is not None:
        eng.set_option('wino2d', args.wino2d)
    if args.fuse is not None:
        eng.set_option('fuse', args.fuse)
    for kv in args.opt:
        k, v = kv.split('=')
        eng.set_option(k, int(v))
    if args.precision:
        eng.set_option('precision', args.precision)
        args.no_split = True

    if args.workload == 'cli_1080p_T3':
        if world != 1:
            raise SystemExit('bench.py: cli_1080p_T3 is a one-GPU workload')
        print(json.dumps(cli_files_workload(eng, args)))
        return
    H, Wd, align, block, tile_hw, ntiles = WORKLOADS[args.workload]
    T = RECURSIONS.get(args.workload, 1)
    strong = args.scaling == 'strong'
    if strong and block is None:
        raise SystemExit('bench.py: --scaling strong shards the tiles of one pair: it needs a tiled workload')
    pairs = ntiles if block is None else 1   # frame pairs per step (batched workloads have no tiling)
    frames_per_step = pairs * (2 ** T - 1)   # generated frames per step (per GPU when weak, per job when strong)
    # strong scaling: every rank holds the SAME pair and owns some of its tiles; weak: every rank has its own pairs
    x0n, x1n = zip(*[synth_pair(H, Wd, 2 + (0 if strong else rank) + 17 * k) for k in range(pairs)])
    x0 = torch.from_numpy(np.concatenate(x0n)).to(dev)
    x1 = torch.from_numpy(np.concatenate(x1n)).to(dev)
    dev_it = DeviceInterpolator(eng, align=align, block_shape=block)
    it = dev_it.batch if pairs > 1 else dev_it
    if strong:
        from film_hip.sharding import TileShardedRecursion
        drv = TileShardedRecursion(DeviceInterpolator(eng, align=align).batch, block, dist)
        step = lambda: drv.run(x0[0], x1[0], T)                                      # noqa: E731
    elif T > 1:
        from film_hip.recursive import interpolate_pair_recursively
        step = lambda: interpolate_pair_recursively(x0[0], x1[0], T, dev_it)             # noqa: E731
    else:
        step = lambda: it(x0, x1)                                                    # noqa: E731

    out = None
    if world > 1:
        # strong mode: step() ends in collectives (ok-flag all-reduce + gather), which ranks 1.. would not enter while rank 0
        # tunes - rank 0 warms up on its own share of the tiles instead (TileShardedRecursion.local: no collective)
        autotune_once(eng, dist, rank, (lambda: drv.local(x0[0], x1[0], T)) if strong else step, torch.cuda.synchronize)
    parity, parity_ref = None, None
    if args.workload in PARITY_CASES and T == 1 and not strong and args.flow_scale == 1.0:
        parity_ref, parity = parity_check(args.workload, it, dev)
        if not parity[next(iter(parity))] < PARITY_TOL:
            sys.stderr.write(f'bench.py: rank {rank}: PARITY FAILED before the timed loop: {json.dumps(parity)}\n')
            raise SystemExit(3)
    first = step()
    first = None if first is None else first.clone()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    reported = 1
    rank_ms = None
    if dist is not None:
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [round(float(v.item()) / args.steps * 1e3, 3) for v in every]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        reported = int(ones.item())
    assert (out is not None or (strong and rank != 0)) and (out is None or bool(torch.isfinite(out).all()))
    # the timed output itself: the last timed step must reproduce the first call on the same inputs bit for bit (no atomics,
    # fixed summation orders: a side lane or a replayed graph that read stale data would show here), and the golden pair pushed through the
    # timed callable AFTER the loop must reproduce its pre-loop result bit for bit
    timed_same = None if (out is None or first is None) else bool(torch.equal(out, first))
    if parity is not None:
        again, _ = parity_check(args.workload, it, dev)
        parity['bit_identical_before_and_after_the_timed_loop'] = bool(torch.equal(again, parity_ref))
        del again, parity_ref
        if not parity['bit_identical_before_and_after_the_timed_loop']:
            sys.stderr.write(f'bench.py: rank {rank}: the golden pair changed its result across the timed loop\n')
            raise SystemExit(3)
    if timed_same is False:
        sys.stderr.write(f'bench.py: rank {rank}: the last timed step differs from the first call on the same inputs\n')
        raise SystemExit(3)
    del first

    result = None
    if rank == 0:
        ms_per_depth = None
        if T > 1 and world == 1:
            # one more recursion with a host synchronisation after every depth (depth d = 2^(d-1) pairs x the tiles)
            frames = torch.stack([x0[0], x1[0]]).contiguous()
            ms_per_depth = []
            for _ in range(T):
                torch.cuda.synchronize()
                td = time.perf_counter()
                mids = dev_it.batch(frames[:-1].contiguous(), frames[1:].contiguous())
                torch.cuda.synchronize()
                ms_per_depth.append(round((time.perf_counter() - td) * 1e3, 2))
                nxt = torch.empty((2 * frames.shape[0] - 1,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=dev)
                nxt[0::2] = frames
                nxt[1::2] = mids
                frames = nxt
            del frames, mids, nxt
        # ---- roofline of the dominant kernel class: hipEvents around every launch, on the launch stream
        # steady state: a normal forward is queued right in front of the profiled one, with no host
        # synchronisation in between, so the first kernels are not timed on a GPU that is ramping up from idle
        it(x0, x1)
        eng.set_option('profile', 1)
        it(x0, x1)
        torch.cuda.synchronize()
        prof = eng.profile()      # the LAST model invocation of that call (one chunk of tiles when the frame was chunked)
        eng.set_option('profile', 0)
        if args.profile_out:
            with open(args.profile_out, 'w') as f:
                json.dump(prof, f)
        cls = prof['classes']
        conv = cls['conv_mfma']
        npix = prof['B'] * prof['H'] * prof['W']      # padded input pixels of the profiled invocation
        alg_flops = CONV_FLOP_PER_PIXEL * npix
        # conv_mfma launches carry all conv FLOPs except the Cin=3 first layer and the tiny 1x1 heads
        conv_tflops = conv['flops'] / (conv['ms'] * 1e-3) / 1e12
        total_ms = sum(c['ms'] for c in cls.values())
        # FLOPs the matrix pipe really executes: the Winograd F(2,3) kernel (tile id & 256) does 2/3 of the direct
        # convolution's multiplies, the sub-pixel-folded upsample + 2x2 conv (tag ':phases') 9/16
        exec_flops = 0.0
        per_kernel = {k: {'launches': 0, 'ms': 0.0, 'executed_flops': 0.0} for k in ('conv_wino2d_kernel', 'conv_wino43_kernel', 'conv_fold4_kernel')}
        for o in prof['ops']:
            if o['kind'] != 'conv_mfma':
                continue
            f = o['flops']
            if o['tile'] & 8192:
                f *= 1.0 / 3.0                                   # nested Winograd F(4,3)x x F(2,3)y: 3 multiplies per output of 9
            elif o['tile'] & 256:
                f *= 0.5 if o['tile'] & 2048 else 2.0 / 3.0    # Winograd F(4,3) / F(2,3) along x
            elif o['tile'] & 16384:
                f *= 4.0 / 16.0                                  # conv_fold4_kernel: upsample + 2x2 in its difference form, 4 multiplies of 16
            elif o['tag'].endswith(':phases'):
                f *= 9.0 / 16.0
            exec_flops += f
            kname = 'conv_wino2d_kernel' if (o['tile'] & 8192) else 'conv_fold4_kernel' if (o['tile'] & 16384) else 'conv_wino43_kernel' if ((o['tile'] & 256) and (o['tile'] & 2048)) else None
            if kname:
                per_kernel[kname]['launches'] += 1
                per_kernel[kname]['ms'] += o['ms']
                per_kernel[kname]['executed_flops'] += f
        exec_tflops = exec_flops / (conv['ms'] * 1e-3) / 1e12
        traffic, traffic_src, warp_traffic, traffic_why = None, None, None, None
        try:  # fabric-side bytes per launch REPLAYED from the committed PMC passes (profiles/; PMC needs its own rocprofv3
            # processes) - only from a file made with THIS build of the kernels (film_version() carries a hash of csrc/ + the header)
            import glob
            pmc_files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_conv.json')))
            if pmc_files and args.workload == '1080p_2x2':
                pmc = json.load(open(pmc_files[-1]))
                traffic_src = os.path.relpath(pmc_files[-1], ROOT)
                if pmc.get('build') != eng.version():
                    traffic_why = (f'{traffic_src} was collected with build "{pmc.get("build")}", this run is "{eng.version()}": not replayed '
                                   f'(tools/gpu_pmc.sh + tools/pmc_summary.py refresh it)')
                else:
                    traffic = round(pmc['hbm_bytes_per_launch'])
                    wcls = next((v for k, v in pmc.get('classes', {}).items() if k.startswith('warp')), None)
                    if wcls and wcls.get('hbm_bytes_per_launch'):
                        warp_traffic = round(wcls['hbm_bytes_per_launch'])
        except Exception as e:   # noqa: BLE001
            traffic, traffic_why = None, repr(e)
        # the roofline the conv class is priced against: the fp32 MFMA peak in the default mode; in the opt-in
        # split modes the dense bf16 MFMA peak divided by the bf16 products one fp32 product costs
        peak = {0: PEAK_FP32_MFMA_TFLOPS, 1: PEAK_BF16_MFMA_TFLOPS / 6, 2: PEAK_BF16_MFMA_TFLOPS / 3}[args.precision]
        roofline = {
            'bound': 'mfma',
            'kernel': ('conv class = conv_wino2d_kernel (every 3x3 layer of the levels >= 1536 pixels: 91 % of the FLOPs) / conv_buf_kernel / conv_c3_kernel (fp32 v_mfma_f32_32x32x2_f32); '
                       '`achieved` / `frac` count the FLOPs the matrix pipe executes (nested Winograd F(4,3)x x F(2,3)y layers x1/3, F(4,3) x1/2, F(2,3) x2/3, folded 2x2 layers x9/16 '
                       'of the direct convolution)')
                      if not args.precision else
                      ('conv class in the opt-in split mode = conv_winox3_kernel / conv_halo_split_kernel '
                       '(v_mfma_f32_32x32x16_bf16, fp32 accumulate) + the fp32 kernels on the small / 2x2 layers; '
                       'peak = dense bf16 MFMA peak / bf16 products per fp32 product'),
            'achieved': round(exec_tflops, 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': round(exec_tflops / peak, 4),
            'frac_note': 'utilisation of the peak by the multiplies the matrix pipe EXECUTES; the kernels that execute fewer of them per output '
                         '(conv_wino2d_kernel: 1/3 of the direct count) lower it while the step gets faster - see direct_equivalent and kernels[]',
            'traffic': traffic if not args.precision else None,
            'traffic_note': (f'REPLAYED, not measured in this run: bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) of the rocprofv3 PMC passes '
                             f'committed in {traffic_src}, collected with the same build of the kernels ("{eng.version()}"; PMC needs its own '
                             f'rocprofv3 processes)') if traffic else traffic_why,
            # the two Winograd kernels on their own (executed FLOPs: nested F(4,3)x x F(2,3)y = 1/3, 1-D F(4,3) = 1/2 of the direct
            # count); `dominant_kernel` = the one with the larger share of the step
            'kernels': None if args.precision else [
                {'name': k, 'launches': v['launches'], 'ms': round(v['ms'], 3),
                 'executed_tflops': round(v['executed_flops'] / max(v['ms'], 1e-9) / 1e9, 3),
                 'frac': round(v['executed_flops'] / max(v['ms'], 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4),
                 'direct_equivalent_tflops': round(v['executed_flops'] * (3.0 if k == 'conv_wino2d_kernel' else 4.0 if k == 'conv_fold4_kernel' else 2.0) / max(v['ms'], 1e-9) / 1e9, 3),
                 'avg_launch_ms': round(v['ms'] / max(1, v['launches']), 5)}
                for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]['ms']) if v['launches']],
            'launches_per_step': conv['launches'],
            'avg_launch_ms': round(conv['ms'] / conv['launches'], 5),
            'class_ms_per_step': round(conv['ms'], 3),
            'executed_flops_per_step': exec_flops,
            'algorithmic_flops_per_step': conv['flops'],
            'all_conv_flops_per_step': alg_flops,
            'share_of_kernel_time': round(conv['ms'] / total_ms, 4),
            'direct_equivalent': {'achieved': round(conv_tflops, 3), 'ratio_to_peak': round(conv_tflops / peak, 4),
                                  'note': 'same launches priced with the direct convolution\'s FLOPs (SURVEY 8d formula); exceeds the fp32 MFMA '
                                          'peak because Winograd / the sub-pixel fold execute fewer multiplies - an algorithmic saving, not a utilisation'},
        }
        roofline['dominant_kernel'] = roofline['kernels'][0] if roofline.get('kernels') else None
        if roofline['dominant_kernel'] is not None and not args.precision:
            # the dominant kernel by K depth: its prologue / epilogue are a fixed cost per workgroup (s_memtime: 10-19k + 7-9k cycles
            # against 1.5k per K chunk of eight channels), so the short-K layers pull the kernel's average down
            ktot = {o['tag']: o['Ctot'] for o in eng.plan(prof['B'], prof['H'], prof['W'])['ops'] if o['kind'] == 'conv_mfma'}
            classes = []
            for lo, hi in ((0, 64), (65, 128), (129, 256), (257, 528), (529, 100000)):
                sel = [o for o in prof['ops'] if o['kind'] == 'conv_mfma' and (o['tile'] & 8192) and lo <= ktot.get(o['tag'], -1) <= hi]
                if sel:
                    ms_c = sum(o['ms'] for o in sel)
                    classes.append({'K': f'{lo}..{hi}' if hi < 100000 else f'{lo}..', 'launches': len(sel), 'ms': round(ms_c, 3),
                                    'frac': round(sum(o['flops'] for o in sel) / 3.0 / max(ms_c, 1e-9) / 1e9 / PEAK_FP32_MFMA_TFLOPS, 4)})
            roofline['dominant_kernel']['by_input_channels'] = classes
        extra = {}
        if 'warp' in cls:
            wgbs = cls['warp']['bytes'] / (cls['warp']['ms'] * 1e-3) / 1e9
            extra['roofline_warp'] = {
                'bound': 'hbm', 'kernel': 'warp_vec_kernel (bilinear gather)', 'achieved': round(wgbs, 1),
                'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(wgbs / PEAK_HBM_GBS, 4),
                'traffic': warp_traffic if not args.precision else None,
                'traffic_note': (f'REPLAYED like roofline.traffic ({traffic_src}): (2*FETCH_SIZE + WRITE_SIZE) of the warp launches / their number; '
                                 f'algorithmic bytes per launch = {cls["warp"]["bytes"] / max(cls["warp"]["launches"], 1):.0f}') if warp_traffic else None,
                'launches_per_step': cls['warp']['launches'], 'class_ms_per_step': round(cls['warp']['ms'], 3),
                'algorithmic_bytes_per_step': cls['warp']['bytes'],
            }
        extra['kernel_ms_per_step'] = {k: round(v['ms'], 3) for k, v in cls.items()}
        if world == 1 and T == 1 and not strong:
            # the same step through the drop-in boundary with HOST buffers (numpy in, numpy out: film_interpolate with
            # FILM_MEM_HOST copies the two frames up and the result down inside the call) - never `value`
            x0h, x1h = x0.cpu().numpy(), x1.cpu().numpy()
            host_call = lambda: eng.interpolate_frames(x0h, x1h, align=align, block_shape=block) if pairs == 1 else eng.interpolate_frames(x0h, x1h, align=align)   # noqa: E731
            host_call()
            t_h = []
            for _ in range(5):
                th0 = time.perf_counter()
                host_call()
                t_h.append((time.perf_counter() - th0) * 1e3)
            pin_in, pin_out = torch.empty(x0.shape, dtype=torch.float32, pin_memory=True), torch.empty(x0.shape, dtype=torch.float32, pin_memory=True)
            torch.cuda.synchronize()
            pin_in.copy_(x0.cpu())
            scratch0, scratch1 = torch.empty_like(x0), torch.empty_like(x1)     # never into x0 / x1: later measurements read them
            torch.cuda.synchronize()
            th0 = time.perf_counter(); scratch0.copy_(pin_in); scratch1.copy_(pin_in); torch.cuda.synchronize(); h2d = (time.perf_counter() - th0) * 1e3   # noqa: E702
            del scratch0, scratch1
            th0 = time.perf_counter(); pin_out.copy_(out if out is not None else x0); torch.cuda.synchronize(); d2h = (time.perf_counter() - th0) * 1e3   # noqa: E702
            hm = float(np.median(t_h))
            extra['host_buffers'] = {'ms_per_step': round(hm, 3), 'frames_per_s': round(frames_per_step / (hm * 1e-3), 4),
                                     'h2d_ms': round(h2d, 3), 'd2h_ms': round(d2h, 3), 'bytes_in': int(2 * x0h.nbytes), 'bytes_out': int(x0h.nbytes),
                                     'over_device_resident_ms': round(hm - dt / args.steps * 1e3, 3),
                                     'note': 'numpy -> numpy through film_interpolate(FILM_MEM_HOST), pageable host memory, the reference interface '
                                             '(eval/interpolator.py:152-209); the call pipelines its copies: second frame uploaded behind the first '
                                             'frame\'s first layers, first half of the result downloaded behind the second half\'s last layer (option '
                                             'host_overlap); h2d / d2h: the same bytes from / to pinned memory on their own; the headline `value` '
                                             'keeps the frames resident in HBM'}
        value = (1 if strong else reported) * args.steps * frames_per_step / dt
        result = {
            'metric': METRIC.get(args.workload, 'interpolated frames/sec @1080p'), 'value': round(value, 4), 'unit': 'frames/s',
            'n_gpus': reported, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f32' if not args.precision else ('f32 via bf16x6 exact-split MFMA (opt-in mode)' if args.precision == 1 else 'bf16x3 split MFMA, f32 accumulate (opt-in mode)'),
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {Wd}x{H} pair, align {align}, block_shape {block} -> '
                                   f'{ntiles} tile(s) of {tile_hw[1]}x{tile_hw[0]} in one batch, film_net published '
                                   f'config, seeded synthetic weights, t=0.5',
                       'times_to_interpolate': T, 'generated_frames_per_step': frames_per_step,
                       'tile_forwards_per_step': frames_per_step * (ntiles if block else 1),
                       'frames_per_step_per_gpu': None if strong else frames_per_step,
                       'parallelism': (f'{world} GPU(s): the {ntiles} tiles of ONE pair sharded over the ranks for the whole recursion tree, '
                                       f'one gather of the generated tiles to rank 0 per step (timed); weights RCCL-broadcast once') if strong
                                      else f'{world} independent GPU(s), weights RCCL-broadcast once',
                       'roofline_profiled_on': f'one model invocation of {prof["B"]} tile(s) / pair(s) of {prof["W"]}x{prof["H"]}',
                       'exec': 'one stream' if args.no_graph else 'hipGraph replay' if args.graph else 'direct launches, two lanes', 'lanes': args.lanes},
            'parity': parity,
            'timed_output_bit_identical_to_first_call': timed_same,
            # crc32 of rank 0's last timed output (weak: its own pairs, seeds independent of the world size; strong: the gathered frames of
            # the one pair): equal across --gpus N for the same workload - tests/test_gpu_multi.py compares 2 ranks with 1
            'output_crc32': None if out is None else int(__import__('zlib').crc32(out.detach().cpu().numpy().tobytes())),
            'build': eng.version(),
            'roofline': roofline,
        }
        result.update(extra)
        if ms_per_depth is not None:
            result['ms_per_depth'] = ms_per_depth
        if dist is not None:   # first contact with N > 1 ranks: what the job looked like from the inside
            result['ranks'] = {'communicator_size': dist.get_world_size(), 'backend': dist.get_backend(), 'ms_per_step_by_rank': rank_ms,
                               'ms_per_step_min': min(rank_ms), 'ms_per_step_max': max(rank_ms),
                               'weight_broadcast_ms': None if bcast_ms is None else round(bcast_ms, 2),
                               'gather_ms_per_step': (round(drv.gather_ms / max(1, drv.runs), 3) if strong and hasattr(drv, 'gather_ms') else None),
                               'tune': 'rank 0 measured, every rank imported its choices (film_export_tune / film_import_tune)'}
        if world == 1 and not args.no_split and T == 1 and '+extra' in eng.version():   # the opt-in modes exist in FILM_EXTRA_FAMILIES=1 builds only
            # Extra, NOT the headline value: the opt-in precision mode "bf16x6" (exact 3-way bf16 split of every fp32
            # operand, six partial products, fp32 accumulate) on the same workload, with its distance from the
            # default fp32-MFMA result.
            ref_out = out.clone()
            for mode, name, what in ((1, 'bf16x6', 'exact 3-way bf16 split, six partial products'),
                                     (2, 'bf16x3', '2-way nearest bf16 split, three partial products')):
                try:   # an extra must never cost the headline line
                    eng.set_option('precision', mode)
                    for _ in range(max(1, args.warmup)):
                        out2 = it(x0, x1)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        out2 = it(x0, x1)
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t1
                    result['precision_mode_' + name] = {
                        'value': round(args.steps * frames_per_step / dt2, 4), 'unit': 'frames/s', 'ms_per_step': round(dt2 / args.steps * 1e3, 3),
                        'max_abs_diff_vs_f32_mode': float((out2 - ref_out).abs().max()),
                        'note': f'opt-in (film_set_option precision={mode}: {what}, fp32 accumulate); '
                                'the headline value above is the fp32-MFMA default',
                    }
                except Exception as e:   # noqa: BLE001
                    result['precision_mode_' + name] = {'error': repr(e)}
            eng.set_option('precision', 0)
        if world == 1 and not args.no_cpu_baseline:
            try:
                result['cpu_baseline'] = cpu_baseline(weights)
            except Exception as e:   # noqa: BLE001
                result['cpu_baseline'] = {'error': repr(e)}
        else:
            result['cpu_baseline'] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
