"""Builds libfilm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

  python -m film_hip.build          (from frame-interpolation_amd/)
  make -C frame-interpolation_amd/csrc   (equivalent)
"""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), 'csrc')
OUT = os.path.join(_HERE, 'libfilm_hip.so')
SOURCES = ['conv_igemm.hip', 'misc_kernels.hip', 'film_engine.cpp', 'film_planner.cpp', 'film_layers.cpp']
HEADERS = ['film_kernels.h', 'film_internal.h', 'conv_buf_impl.h', 'conv_halo_impl.h', 'conv_split_impl.h', 'conv_wino_impl.h', 'conv_wino43_impl.h', 'conv_wino2d_impl.h', 'conv_winox3_impl.h', 'conv_foldx3_impl.h', 'conv_c3_impl.h', os.path.join('..', '..', 'include', 'film_hip.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']


def _stale(target: str, deps) -> bool:
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get('HIPCC', 'hipcc')
    bdir = os.path.join(CSRC, 'build')
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [hipcc] + FLAGS + ['-c', sp, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    if force or procs or _stale(OUT, objs + [os.path.join(CSRC, 'film_hip.map')]):
        # ONE HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (no SONAME,
        # its libraries NEED the unversioned name "libamdhip64.so"); linking against
        # /opt/rocm/lib/libamdhip64.so would record the SONAME "libamdhip64.so.7", which glibc does not
        # match with torch's copy -> two runtimes in one process (the second one sees no GPU, and
        # streams / pointers cross runtimes).  So the final link is done by g++ against an empty,
        # SONAME-less stub: DT_NEEDED becomes "libamdhip64.so", which resolves to whichever runtime is
        # already loaded (torch's, when torch was imported first) or to /opt/rocm/lib via RUNPATH.
        stub_dir = os.path.join(bdir, 'stub')
        os.makedirs(stub_dir, exist_ok=True)
        stub_c = os.path.join(stub_dir, 'empty.c')
        with open(stub_c, 'w') as f:
            f.write('/* link-time stub: gives libfilm_hip.so a DT_NEEDED of "libamdhip64.so" */\n')
        subprocess.check_call(['gcc', '-shared', '-fPIC', '-o', os.path.join(stub_dir, 'libamdhip64.so'), stub_c])
        rocm_lib = os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'lib')
        cmd = ['g++', '-shared', '-fPIC', '-o', OUT] + objs + [
            '-Wl,--version-script,' + os.path.join(CSRC, 'film_hip.map'),      # exports = the C-ABI of include/film_hip.h, nothing else
            '-L' + stub_dir, '-Wl,--no-as-needed', '-lamdhip64', '-Wl,--as-needed',
            '-Wl,-rpath,' + rocm_lib, '-Wl,--enable-new-dtags']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
