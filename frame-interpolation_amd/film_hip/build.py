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
PUBLIC_HEADER = os.path.join('..', '..', 'include', 'film_hip.h')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']


def sources():
    """Every translation unit under csrc/ (sorted): nothing to keep in sync by hand (a stale list once shipped a stale kernel)."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def headers():
    """Every header a translation unit can include: csrc/*.h (sorted) + the public C-ABI header."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [PUBLIC_HEADER]


def source_id() -> str:
    """12 hex digits of the sha1 over the bytes of csrc/{*.cpp, *.h, *.hip, film_hip.map} in sorted order followed by
    include/film_hip.h - what `make -C csrc print-src-id` prints too.  Compiled into film_version() so that a bench line, a PMC
    summary and a tune cache can be tied to the kernel sources they were produced with."""
    import hashlib
    h = hashlib.sha1()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.h')) or f == 'film_hip.map')
    for n in names + [PUBLIC_HEADER]:
        with open(os.path.join(CSRC, n), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def extra_families() -> bool:
    """FILM_EXTRA_FAMILIES=1 at build time also instantiates the kernel families no default plan selects (the opt-in bf16
    precision modes, the F(2,3) / halo kernels behind test options); the default library holds only what a default plan can run."""
    return os.environ.get('FILM_EXTRA_FAMILIES', '0') not in ('', '0')


def _stale(target: str, deps) -> bool:
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def lib_path() -> str:
    """film_hip/libfilm_hip.so, or libfilm_hip_extra.so for the FILM_EXTRA_FAMILIES=1 flavour (both can sit side by side)."""
    return os.path.join(_HERE, 'libfilm_hip_extra.so' if extra_families() else 'libfilm_hip.so')


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get('HIPCC', 'hipcc')
    out = lib_path()
    bdir = os.path.join(CSRC, 'build_extra' if extra_families() else 'build')
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in headers()]
    sid = source_id()
    sid_file = os.path.join(bdir, 'src_id.txt')
    sid_changed = not os.path.isfile(sid_file) or open(sid_file).read().strip() != sid
    flags = FLAGS + (['-DFILM_EXTRA_FAMILIES=1'] if extra_families() else [])
    objs = []
    procs = []
    for src in sources():
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        # film_version() lives in film_engine.cpp and carries the source id: that unit is rebuilt whenever any source changed
        is_id_unit = src == 'film_engine.cpp'
        if force or _stale(obj, [sp] + hdrs) or (is_id_unit and sid_changed):
            cmd = [hipcc] + flags + ([f'-DFILM_SRC_ID="{sid}"'] if is_id_unit else []) + ['-c', sp, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    with open(sid_file, 'w') as f:
        f.write(sid + '\n')
    if force or procs or _stale(out, objs + [os.path.join(CSRC, 'film_hip.map')]):
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
        cmd = ['g++', '-shared', '-fPIC', '-o', out] + objs + [
            '-Wl,--version-script,' + os.path.join(CSRC, 'film_hip.map'),      # exports = the C-ABI of include/film_hip.h, nothing else
            '-L' + stub_dir, '-Wl,--no-as-needed', '-lamdhip64', '-Wl,--as-needed',
            '-Wl,-rpath,' + rocm_lib, '-Wl,--enable-new-dtags']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
