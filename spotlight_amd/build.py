"""Builds spotlight_amd/csrc/libspotlight_hip.so for gfx950 with hipcc (in-tree).

    python -m spotlight_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['slk_api.hip', 'slk_sort.hip', 'slk_rng.hip', 'slk_mtjump.hip', 'slk_bilinear.hip', 'slk_shard.hip', 'slk_seq.hip', 'slk_eval.hip', 'slk_shuffle.hip', 'slk_seqprep.hip', 'slk_embed.hip', 'slk_probe.hip', 'slk_epoch.hip']
HEADERS = ['slk_common.h', 'slk_kernels.h', os.path.join('..', '..', 'include', 'spotlight_hip.h')]
LIB = os.path.join(CSRC, 'libspotlight_hip.so')
ARCH = 'gfx950'
# -ffp-contract=off: torch's eager ops round every product before adding; with FMA contraction
# gp*v_pos + gn*v_neg does not cancel to an exact 0.0 when an interaction's sampled negative
# equals its positive (gn == -gp), and Adagrad turns that 1e-11 residue into an O(lr) update.
# The path is HBM-bound, so the extra VALU op is free.


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libspotlight_hip.so')
    return exe


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc(), '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-ffp-contract=off',
                   '-Wall', '-Wno-unused-function', '-Wno-pass-failed', '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        if verbose and out:
            print(out.decode())
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
